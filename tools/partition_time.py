"""Device time per frame of ONE rank of an N-way tile partition of the 1920x1080 bench frame, on one GPU (no exchange):
what `bench.py --gpus N` can reach at best, per kernel class.  Usage: python tools/partition_time.py [N ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
import mqhip

worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
base = None
for world in worlds:
    ctx = mqhip.Context(0)
    ctx.json_defaults()
    for k, v in {"randomize seed": 0, "seed": 0x5EED, "spp": 1, "max path length": 3}.items():
        ctx.set_property(k, v)
    ctx.synth_scene("synth_sepulcher", 2); ctx.commit()
    ctx.set_partition(0, world); ctx.connect(1920, 1080)
    for f in range(64):
        ctx.process(ctx.synth_camera(f))
    ctx.sync(); ctx.timing_reset()
    for f in range(64, 164):
        ctx.process(ctx.synth_camera(f))
    ctx.sync()
    n, render, update = ctx.timing_get()
    rounds = ctx.timing_rounds()
    ms = (render + update) / n
    base = base or ms
    print("world %d: %.3f ms per frame (x%.2f of world 1); primary %.3f+%.3f, trace %s, bounce %s, update %.3f" % (
        world, ms, base / ms, rounds[0][0] / n, rounds[0][1] / n, "+".join("%.3f" % (a / n) for a, b in rounds[1:3]), "+".join("%.3f" % (b / n) for a, b in rounds[1:3]), update / n))
    ctx.close()
