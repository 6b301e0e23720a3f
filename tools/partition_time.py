"""Device time per frame of ONE rank of an N-way tile partition of the 1920x1080 bench frame, on one GPU (no exchange):
what `bench.py --gpus N` can reach at best, per kernel class.
Usage: python tools/partition_time.py [--pipelines S[,S...]] [N ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
import mqhip

args = sys.argv[1:]
pipes = [None]
if args and args[0] == "--pipelines":
    pipes = [int(x) for x in args[1].split(",")]; args = args[2:]
worlds = [int(a) for a in args] or [1, 2, 4, 8]
for S in pipes:
    base = None
    for world in worlds:
        ctx = mqhip.Context(0)
        ctx.json_defaults()
        for k, v in {"randomize seed": 0, "seed": 0x5EED, "spp": 1, "max path length": 3}.items():
            ctx.set_property(k, v)
        if S is not None:
            ctx.set_property("pipelines", S)
        if os.environ.get("MQ_OVERLAP") is not None:
            ctx.set_property("overlap camera rays", int(os.environ["MQ_OVERLAP"]))
        ctx.synth_scene("synth_sepulcher", 2); ctx.commit()
        ctx.set_partition(0, world); ctx.connect(1920, 1080)
        for f in range(64):
            ctx.process(ctx.synth_camera(f))
        ctx.sync(); ctx.timing_set_interval(1000); ctx.timing_reset()
        t0 = time.perf_counter()
        for f in range(64, 164):
            ctx.process(ctx.synth_camera(f))
        ctx.sync()
        wall = (time.perf_counter() - t0) * 10.0  # ms per frame over 100 frames
        n, render, update = ctx.timing_get()
        ms = (render + update) / n
        ctx.timing_set_interval(1); ctx.timing_reset()
        for f in range(164, 196):
            ctx.process(ctx.synth_camera(f))
        ctx.sync()
        n2, _, update2 = ctx.timing_get()
        rounds = ctx.timing_rounds()
        base = base or ms
        print("pipelines %s world %d: %.3f ms per frame device (x%.2f of world 1), %.3f ms wall; with per-launch events: primary %.3f+%.3f, trace %s, bounce %s, update %.3f" % (
            ctx.get_property("pipelines"), world, ms, base / ms, wall, rounds[0][0] / n2, rounds[0][1] / n2, "+".join("%.3f" % (a / n2) for a, b in rounds[1:3]), "+".join("%.3f" % (b / n2) for a, b in rounds[1:3]), update2 / n2), flush=True)
        ctx.close()
