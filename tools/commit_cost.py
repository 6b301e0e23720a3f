"""Cost of a per-frame geometry commit (only the non-static slots change) next to a full commit, and of the frame
that follows, on the bench scene.  Usage: python tools/commit_cost.py [scene] [n_boxes ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mqhip
from test_gpu_parity import _boxes

scene = sys.argv[1] if len(sys.argv) > 1 else "synth_sepulcher"
counts = [int(a) for a in sys.argv[2:]] or [100, 1000, 4000]
ctx = mqhip.Context(0)
ctx.json_defaults()
for k, v in {"randomize seed": 0, "seed": 0x5EED, "spp": 1, "max path length": 3}.items():
    ctx.set_property(k, v)
ctx.synth_scene(scene, 2)
t0 = time.perf_counter(); ctx.commit(); t_full = time.perf_counter() - t0
print("scene %s: %r, full commit %.1f ms" % (scene, ctx.scene_stats(), 1e3 * t_full))
ctx.connect(1920, 1080)
ext0 = ctx.get_geometry(0)["ext"][:1]


def frames(first, n):
    for f in range(first, first + n):
        ctx.process(ctx.synth_camera(f))
    ctx.sync()
    ctx.timing_reset()
    for f in range(first + n, first + 2 * n):
        ctx.process(ctx.synth_camera(f))
    ctx.sync()
    k, render, update = ctx.timing_get()
    return "%.3f ms per frame (%d frames)" % ((render + update) / max(k, 1), k)


print("frames without extra per-frame geometry:", frames(0, 20))
rng = np.random.default_rng(1)
for n in counts:
    u = ctx.synth_camera(40)
    cam = np.array(u.cam_x[:3], np.float32)
    centres = cam + (rng.random((n, 3), dtype=np.float32) * 2 - 1) * np.array([900, 900, 200], np.float32)
    base, idx = _boxes(centres, 8.0)
    ext = np.repeat(ext0, len(idx))
    dts = []
    for it in range(6):
        vtx = base + np.float32(it)
        ctx.set_geometry(5, vtx, vtx - 1.0, idx, ext, mqhip.MQ_GEO_OPAQUE)
        t0 = time.perf_counter(); ctx.commit(); dts.append(time.perf_counter() - t0)
        ctx.process(ctx.synth_camera(40 + it)); ctx.sync()
    print("%6d boxes = %7d triangles: per-frame commit %.2f ms (first %.2f ms), counts %r" % (n, len(idx), 1e3 * np.median(dts[1:]), 1e3 * dts[0], ctx.commit_counts()))
    print("   frames with them:", frames(40, 20))
