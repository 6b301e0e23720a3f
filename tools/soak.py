"""Soak run: free-running guided frames of the bench scene (fly-through, wrapping), checking every 1000th frame for finite
radiance and the overflow flags.   python tools/soak.py [frames [world [restir]]]
world > 1: rank world // 2 of a partition (interleaved tiles for the MCPG node; with restir = 1 the ReSTIR node and the post
chain run on the rank's row band -- without a halo exchange, i.e. on stale halo rows: finite and unflagged all the same, except
for bit 3 where the fly-through wraps around and every pixel jumps)."""
import sys, time, numpy as np
sys.path.insert(0, "merian-quake_amd")
import mqhip
ctx = mqhip.Context(0)
ctx.json_defaults()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
restir = len(sys.argv) > 3 and sys.argv[3] == "1"
for k, v in {"randomize seed": 0, "seed": 0x5EED, "spp": 1, "max path length": 3, "restir: randomize seed": 0, "restir: spp": 1, "restir: enable temporal reuse": 1,
             "restir: spatial reuse iterations": 1, "add: restir irradiance": 1, "band: reprojection halo": 96}.items():
    ctx.set_property(k, v)
ctx.synth_scene("synth_sepulcher", 2); ctx.commit(); ctx.set_partition(world // 2 if world > 1 else 0, world); ctx.connect(1920, 1080)
t0 = time.time()
for f in range(frames):
    u = ctx.synth_camera(f % 900)
    ctx.process(u)
    if restir:
        ctx.restir_process(u); ctx.post_process()
    if f % 1000 == 999:
        img = ctx.read_output(mqhip.OUT_TILES).view(np.float32).reshape(-1, 4)
        ok = bool(np.isfinite(img).all())
        extra = ""
        if restir:
            b = ctx.band_layout(1920, 1080, world // 2 if world > 1 else 0, world)
            fin = ctx.image(mqhip.OUT_FINAL)[b.row_begin:b.row_end]
            ok = ok and bool(np.isfinite(fin).all())
            extra = " final mean %.4f" % float(fin[..., :3].mean())
        c = ctx.counters()
        print("frame", f + 1, "mean", float(img[..., :3].mean()), "finite", ok, "overflow", c["queue_overflow"], extra, "%.1f s" % (time.time() - t0), flush=True)
ctx.close()
