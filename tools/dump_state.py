"""State dumps in the reference's JSON formats (src/render_mcpg/render_mcpg.cpp:322-380), so that the reference's
analysis notebooks (scripts/duckdb queries.md) run on this build's learning state:

    mc_dump.json   one object per ADAPTIVE Markov-chain state: id, N, hash, w_cos, sum_w, w_tgt "x y z",
                   tgt_change / w_change / cos_change (the reference's three debug floats: not kept here, written as 0)
    lc_dump.json   one object per light-cache cell: hash, irr [f16 bits x3 as floats], N, update_succeeded / update_canceled
                   (this build keeps the two statistics as global counters, not per cell: written as 0)

    python tools/dump_state.py [--frames 64] [--scene synth_sepulcher] [--width W --height H] [--out DIR] [--limit N]

`--limit` keeps the first N entries (the full adaptive table has 32.8 M states).  update_buffer_dump.json has no
counterpart: the reference's 512-byte-per-state update array is replaced by a compact queue that is empty between frames."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
import mqhip


def mc_records(mc, limit=None):
    mc = mc[:limit] if limit else mc
    for v in mc:
        yield {"id": int(v["id"]), "N": int(v["n_hash"] & 0xffff), "hash": int(v["n_hash"] >> 16), "w_cos": float(v["w_cos"]),
               "sum_w": float(v["sum_w"]), "w_tgt": "%r %r %r" % tuple(float(x) for x in v["w_tgt"]),
               "tgt_change": 0.0, "w_change": 0.0, "cos_change": 0.0}


def lc_records(lc, limit=None):
    lc = lc[:limit] if limit else lc
    irr = lc["irr"].view(np.float16).astype(np.float32).reshape(-1, 3)
    for v, i in zip(lc, irr):
        yield {"hash": int(v["hash"]), "irr": [float(x) for x in i], "N": int(v["N"]), "update_succeeded": 0, "update_canceled": 0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64); ap.add_argument("--scene", default="synth_sepulcher"); ap.add_argument("--scene-seed", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--out", default="."); ap.add_argument("--limit", type=int, default=100000)
    a = ap.parse_args()
    ctx = mqhip.Context(0)
    ctx.json_defaults()
    for k, v in {"randomize seed": 0, "seed": 0x5EED, "spp": 1, "max path length": 3, "volume spp": 0}.items():
        ctx.set_property(k, v)
    ctx.synth_scene(a.scene, a.scene_seed); ctx.commit(); ctx.connect(a.width, a.height)
    for f in range(a.frames):
        ctx.process(ctx.synth_camera(f))
    n_adaptive = int(ctx.get_property("adaptive grid buf size")); n_static = int(ctx.get_property("static grid buf size"))
    mc = ctx.state_read(0, n_adaptive + n_static)[:n_adaptive]  # the reference dumps the adaptive part (render_mcpg.cpp:324)
    lc = ctx.state_read(1, int(ctx.get_property("LC buf size")))
    with open(os.path.join(a.out, "mc_dump.json"), "w") as f:
        json.dump(list(mc_records(mc, a.limit)), f, indent=2)
    with open(os.path.join(a.out, "lc_dump.json"), "w") as f:
        json.dump(list(lc_records(lc, a.limit)), f, indent=2)
    print("valid MC states: %d of %d, used LC cells: %d of %d" % ((mc["sum_w"] > 0).sum(), len(mc), (lc["N"] > 0).sum(), len(lc)))


if __name__ == "__main__":
    main()
