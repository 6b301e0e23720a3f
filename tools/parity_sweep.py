"""Parity sweep beyond the pytest suite (MI355X, a minute of host time): large frames of every stand-in scene, odd sizes, long
paths, late camera frames -- radiance and every g-buffer output of the GPU path against the oracle, bit for bit --
and a guided 1920x1080 frame from a given state with the JSON-default table sizes (32.8 M + 0.8 M Markov-chain states,
4 M light-cache cells) that bench.py runs with.  Usage (GPU box): python tools/parity_sweep.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mqhip, orc
from test_gpu_parity import make_pair, _copy_learned_state
ctx = mqhip.Context(0)
TH = os.cpu_count() or 8
for name, scene, seed, W, H, props, frames in (
    ("materials 1080p spp2 len4", "synth_materials", 11, 1920, 1080, {"spp": 2, "max path length": 4}, (0, 7)),
    ("sepulcher odd size", "synth_sepulcher", 2, 1918, 1079, {"spp": 1}, (0, 100)),
    ("sepulcher frames", "synth_sepulcher", 2, 1920, 1080, {"spp": 1, "max path length": 5}, (150, 300)),
    ("start 720p (config 2)", "synth_start", 1, 1280, 720, {"spp": 1}, (0, 33)),
):
    o = make_pair(ctx, scene, seed, {"reference mode": 1, **props}, W, H)
    for f in frames:
        u = ctx.synth_camera(f)
        ctx.process(u); o.process(u, threads=TH)
        bad = (ctx.irradiance().view(np.uint32) != o.irradiance().view(np.uint32)).any(-1)
        same = all(np.array_equal(ctx.read_output(g), o.output(r)) for g, r in ((mqhip.OUT_HITS, orc.OUT_HITS), (mqhip.OUT_GB_ALBEDO, orc.OUT_GB_ALBEDO), (mqhip.OUT_GB_IRRADIANCE, orc.OUT_GB_IRRADIANCE), (mqhip.OUT_GB_MV, orc.OUT_GB_MV), (mqhip.OUT_GBUFFER, orc.OUT_GBUFFER)))
        print(name, "frame", f, ": irradiance", int(bad.sum()), "of", bad.size, "differ; g-buffer outputs equal:", same, "; lit", float(o.irradiance()[..., :3].sum()) > 0)
    o.close()

# guided frame from a given state, table sizes of the JSON defaults (make_pair's small tables overridden)
BIG = {"adaptive grid buf size": 32777259, "static grid buf size": 800009, "LC buf size": 4000037}
o = make_pair(ctx, "synth_sepulcher", 2, {"reference mode": 0, "spp": 1, "max path length": 3, **BIG}, 128, 72)
for f in range(4):
    o.process(ctx.synth_camera(36 + f), threads=1)
omc, olc = o.state(0).copy(), o.state(1).copy()
ctx.connect(1920, 1080); o.connect(1920, 1080)
ctx.set_property("debug: freeze learning", 1)
o.set_params(orc.params_from_ctx(ctx, ctx.get_constants()))
u = ctx.synth_camera(39); ctx.process(u); o.process(u, threads=TH)
o.state(0)[:] = omc; o.state(1)[:] = olc
_copy_learned_state(ctx, o)
u = ctx.synth_camera(40); ctx.process(u); o.process(u, threads=TH)
bad = (ctx.irradiance().view(np.uint32) != o.irradiance().view(np.uint32)).any(-1)
print("guided from state, JSON-default table sizes, 1920x1080: irradiance", int(bad.sum()), "of", bad.size, "differ; states learned", int((omc["sum_w"] > 0).sum()), "; lit", float((o.irradiance()[..., :3].sum(-1) > 0).mean()))
