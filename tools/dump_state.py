"""State dumps in the reference's JSON formats (src/render_mcpg/render_mcpg.cpp:322-416), so that the reference's
analysis notebooks (scripts/duckdb queries.md, scripts/evaluate_locking_fast.py) run on this build's learning state:

    mc_dump.json             one object per ADAPTIVE Markov-chain state: id, N, hash, w_cos, sum_w, w_tgt "x y z", last_update_count
                             (grid.h:25; the notebooks read it from this file), tgt_change / w_change / cos_change (the three debug
                             floats of grid.h:8-10, which the fork's update kernel never writes: 0)
    lc_dump.json             one object per light-cache cell: hash, irr, N, update_succeeded, update_canceled (grid.h:44-45)
    update_buffer_dump.json  one object per slot that received updates in the last frame: update_count (0: applied),
                             last_update_count, ids, weights, targets, positions -- the reference dumps its whole 17 GB
                             slot array (:382-416); this build's queue is compact and empty between frames, so the records
                             come from the last frame's learning-write log, with their arrival ranks

    python tools/dump_state.py [--frames 64] [--scene synth_sepulcher] [--width W --height H] [--out DIR] [--limit N]

The statistics need the property "debug: LC lock statistics" (the light cache then takes the reference's per-cell try-lock,
light_cache.glsl:59-64, instead of this build's lock-free publish; slower).  `--limit` keeps the first N entries per file."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
import mqhip


def mc_records(mc, last, limit=None):
    n = min(len(mc), limit) if limit else len(mc)
    for v, l in zip(mc[:n], last[:n]):
        yield {"id": int(v["id"]), "N": int(v["n_hash"] & 0xffff), "hash": int(v["n_hash"] >> 16), "w_cos": float(v["w_cos"]),
               "sum_w": float(v["sum_w"]), "w_tgt": "%r %r %r" % tuple(float(x) for x in v["w_tgt"]), "last_update_count": int(l),
               "tgt_change": 0.0, "w_change": 0.0, "cos_change": 0.0}


def lc_records(lc, stats, limit=None):
    n = min(len(lc), limit) if limit else len(lc)
    irr = lc["irr"][:n].view(np.float16).astype(np.float32).reshape(-1, 3)
    for v, i, s in zip(lc[:n], irr, stats[:n]):
        yield {"hash": int(v["hash"]), "irr": [float(x) for x in i], "N": int(v["N"]), "update_succeeded": int(s["update_succeeded"]), "update_canceled": int(s["update_canceled"])}


def update_buffer_records(log, limit=None):
    """kind-1 records of a learning-write log -> the reference's MCUpdate objects of the touched slots"""
    upd = log[log[:, 15] == 1]
    order = np.lexsort((upd[:, 13] >> 16, upd[:, 14]))
    upd = upd[order]
    out = []
    for slot in np.unique(upd[:, 14])[: limit or None]:
        r = upd[(upd[:, 14] == slot) & ((upd[:, 13] >> 16) < 10)]
        f = r.view(np.float32)
        fmt = lambda a: "[%s]" % ", ".join("(%.3f, %.3f, %.3f)" % tuple(x) for x in a)
        out.append({"slot": int(slot), "update_count": 0, "last_update_count": len(r), "ids": [int(x) for x in r[:, 7]] + [0] * (10 - len(r)),
                    "weights": [float(x) for x in f[:, 3]] + [0.0] * (10 - len(r)), "targets": fmt(f[:, 4:7]), "positions": fmt(f[:, 0:3])})
    return out


def dump(ctx, out_dir, limit):
    n_adaptive = int(ctx.get_property("adaptive grid buf size")); n_static = int(ctx.get_property("static grid buf size")); n_lc = int(ctx.get_property("LC buf size"))
    mc = ctx.state_read(0, n_adaptive + n_static)[:n_adaptive]  # the reference dumps the adaptive part (render_mcpg.cpp:324)
    lc = ctx.state_read(1, n_lc)
    stats = ctx.state_read(3, n_lc); last = ctx.state_read(4, n_adaptive + n_static)
    log = ctx.learn_log()
    files = {"mc_dump.json": list(mc_records(mc, last, limit)), "lc_dump.json": list(lc_records(lc, stats, limit)), "update_buffer_dump.json": update_buffer_records(log, limit)}
    for name, recs in files.items():
        with open(os.path.join(out_dir, name), "w") as f:
            json.dump(recs, f, indent=2)
    ok, cancel = int(stats["update_succeeded"].sum(dtype=np.uint64)), int(stats["update_canceled"].sum(dtype=np.uint64))
    hist = np.bincount(last[last > 0])
    return {"valid_mc_states": int((mc["sum_w"] > 0).sum()), "used_lc_cells": int((lc["N"] > 0).sum()), "lc_update_succeeded": ok, "lc_update_canceled": cancel,
            "lc_succeeded_percent": round(100.0 * ok / max(1, ok + cancel), 2), "slots_by_last_update_count": {int(k): int(v) for k, v in enumerate(hist) if k and v}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64); ap.add_argument("--scene", default="synth_sepulcher"); ap.add_argument("--scene-seed", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--out", default="."); ap.add_argument("--limit", type=int, default=100000)
    a = ap.parse_args()
    ctx = mqhip.Context(0)
    ctx.json_defaults()
    for k, v in {"randomize seed": 0, "seed": 0x5EED, "spp": 1, "max path length": 3, "volume spp": 0, "debug: LC lock statistics": 1}.items():
        ctx.set_property(k, v)
    ctx.synth_scene(a.scene, a.scene_seed); ctx.commit(); ctx.connect(a.width, a.height)
    for f in range(a.frames - 1):
        ctx.process(ctx.synth_camera(f))
    ctx.set_property("debug: log learning writes", 1)
    ctx.process(ctx.synth_camera(a.frames - 1))
    print(json.dumps(dump(ctx, a.out, a.limit)))


if __name__ == "__main__":
    main()
