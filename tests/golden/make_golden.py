#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference ships no golden vectors and cannot run here
(SURVEY.md 8c), so these pin the ORACLE's outputs (regression) and give the GPU tests fixed
expected values that travel to the GPU box.  Inputs: the seeded synthetic scene `synth_tiny(3)`
produced by the product's host-side generator (scene data, not algorithm)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mqhip  # noqa: E402
import orc  # noqa: E402

PROPS = {"reference mode": 1, "randomize seed": 0, "seed": 0x5EED, "spp": 2, "max path length": 3,
         "adaptive grid buf size": 1 << 16, "static grid buf size": 1 << 12, "LC buf size": 1 << 14}
W, H = 64, 48


def build():
    ctx = mqhip.Context(-1)
    ctx.synth_scene("synth_tiny", 3)
    for k, v in PROPS.items():
        ctx.set_property(k, v)
    o = orc.Oracle(orc.params_from_ctx(ctx, ctx.get_constants()))
    orc.mirror_scene(ctx, o)
    o.commit(1)
    return ctx, o


def ray_set(ctx, n=4096):
    g = ctx.get_geometry(0)
    lo, hi = g["vtx"].min(0), g["vtx"].max(0)
    rng = np.random.default_rng(1234)
    org = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return org, d


def main():
    ctx, o = build()
    out = {}
    o.connect(W, H)
    for f in (0, 5):
        u = ctx.synth_camera(f)
        o.process(u, threads=4)
        out["irr_f%d" % f] = o.irradiance().copy()
        out["hits_f%d" % f] = o.output(orc.OUT_HITS).copy()
        out["gbuffer_f%d" % f] = o.output(orc.OUT_GBUFFER).copy()
        out["albedo_f%d" % f] = o.output(orc.OUT_GB_ALBEDO).copy()
    org, d = ray_set(ctx)
    p, t, uv = o.trace_rays(org, d)
    out["ray_prim"], out["ray_t"], out["ray_uv"] = p, t, uv
    rng = np.random.default_rng(99)
    for op in (orc.OP_EXP2, orc.OP_LOG2, orc.OP_SINCOS2PI, orc.OP_XORSHIFT, orc.OP_PCG4D16):
        ni = orc.OP_ARITY[op][0]
        if op in (orc.OP_XORSHIFT, orc.OP_PCG4D16):
            x = rng.integers(1, 2 ** 32, (256, ni), dtype=np.uint64).astype(np.uint32).view(np.float32)
        elif op == orc.OP_LOG2:
            x = np.exp(rng.random((256, ni)) * 40 - 20).astype(np.float32)
        else:
            x = (rng.random((256, ni)) * 20 - 10).astype(np.float32)
        out["op%d_in" % op] = x
        out["op%d_out" % op] = o.math_eval(op, x)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "synth_tiny_ref.npz"), **out)
    print("wrote", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
