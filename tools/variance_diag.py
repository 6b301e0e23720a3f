"""Per-pixel variance of the guided against the unguided estimator on a static view (diagnostic for the convergence curve)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
import mqhip
W, H, N = 192, 128, 512
scene, seed = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("synth_start", 11)
extra = dict(kv.split("=") for kv in sys.argv[3:])
ctx = mqhip.Context(0)
res = {}
for name, mode, warm in (("unguided", 1, 0), ("guided", 0, 128)):
    ctx.header_defaults(); ctx.synth_scene(scene, seed)
    for k, v in {"randomize seed": 0, "seed": 0x5EED, "adaptive grid buf size": 1 << 18, "static grid buf size": 1 << 14, "LC buf size": 1 << 16, "reference mode": mode, "spp": 1, "max path length": 3}.items():
        ctx.set_property(k, v)
    for k, v in extra.items():
        ctx.set_property(k, float(v))
    ctx.commit(); ctx.connect(W, H)
    u = ctx.synth_camera(0)
    for f in range(warm):
        u.frame = 50000 + f; ctx.process(u)
    s1 = np.zeros((H, W)); s2 = np.zeros((H, W)); lit = 0.0
    for f in range(N):
        u.frame = 2000 + f; ctx.process(u)
        l = ctx.irradiance()[..., :3].astype(np.float64).mean(-1)
        s1 += l; s2 += l * l; lit += (l > 0).mean()
    mean = s1 / N; var = s2 / N - mean * mean
    res[name] = (mean, var, lit / N)
    print("%-9s mean %.4f  mean variance %.3f  median variance %.5f  lit pixels per frame %.3f  max pixel variance %.1f" % (name, mean.mean(), var.mean(), np.median(var), lit / N, var.max()))
vu, vg = res["unguided"][1], res["guided"][1]
ok = vu > 0
ratio = vg[ok] / vu[ok]
print("pixels where guided variance < unguided: %.3f; median ratio %.3f; ratio of mean variances %.3f" % ((ratio < 1).mean(), np.median(ratio), vg.mean() / vu.mean()))
top = np.sort(vg.ravel())[::-1]
print("guided: share of total variance in the top 1%% of pixels: %.3f; unguided: %.3f" % (top[: len(top) // 100].sum() / top.sum(), np.sort(vu.ravel())[::-1][: vu.size // 100].sum() / vu.sum()))
