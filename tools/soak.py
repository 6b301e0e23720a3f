"""Soak run: 6000 free-running guided frames of the bench scene (fly-through, wrapping), checking every 1000th frame for
finite radiance and the queue-overflow flag.  python tools/soak.py [frames [world]]  (world > 1: rank 0 of a tile partition)"""
import sys, time, numpy as np
sys.path.insert(0, "merian-quake_amd")
import mqhip
ctx = mqhip.Context(0)
ctx.json_defaults()
for k, v in {"randomize seed": 0, "seed": 0x5EED, "spp": 1, "max path length": 3}.items():
    ctx.set_property(k, v)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx.synth_scene("synth_sepulcher", 2); ctx.commit(); ctx.set_partition(0, world); ctx.connect(1920, 1080)
t0 = time.time()
for f in range(frames):
    ctx.process(ctx.synth_camera(f % 900))
    if f % 1000 == 999:
        img = ctx.read_output(mqhip.OUT_TILES).view(np.float32).reshape(-1, 4)
        c = ctx.counters()
        print("frame", f + 1, "mean", float(img[..., :3].mean()), "finite", bool(np.isfinite(img).all()), "overflow", c["queue_overflow"], "%.1f s" % (time.time() - t0), flush=True)
ctx.close()
