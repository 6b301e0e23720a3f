"""BASELINE config 1 as stated -- "start.bsp 320x240 1 spp, naive CPU path tracer over the loaded triangles (baseline harness,
no GPU)" -- on the stand-in level synth_start(seed=1): one unguided frame traced by brute force over all 34 k triangles and
by the plain binary BVH, which must give the same image bit for bit (closest hits do not depend on the traversal), and a
guided frame; the rates are printed (pytest -s) as the CPU figures of this container.  The oracle is the harness here."""
import os
import time

import numpy as np

import orc


def test_config1_naive_cpu_path_tracer(mqlib):
    import mqhip as mq
    ctx = mq.Context(-1)  # host-only: scene and parameters, no device
    ctx.header_defaults()
    ctx.synth_scene("synth_start", 1)
    for k, v in {"randomize seed": 0, "seed": 0x5EED, "reference mode": 1, "spp": 1, "max path length": 3}.items():
        ctx.set_property(k, v)
    ctx.commit()
    W, H, threads = 320, 240, os.cpu_count() or 1
    imgs, rate = [], {}
    for accel, name in ((1, "binary BVH"), (0, "brute force")):
        o = orc.Oracle(orc.params_from_ctx(ctx, ctx.get_constants()))
        orc.mirror_scene(ctx, o)
        o.commit(accel)
        o.connect(W, H)
        t0 = time.perf_counter()
        o.process(ctx.synth_camera(3), threads=threads)
        rate[name] = W * H / (time.perf_counter() - t0) / 1e6
        imgs.append(o.irradiance().copy())
        if accel == 1:  # a guided frame too (the tables start empty: frame 0 of the learning)
            ctx.set_property("reference mode", 0)
            o.set_params(orc.params_from_ctx(ctx, ctx.get_constants()))
            t0 = time.perf_counter()
            o.process(ctx.synth_camera(4), threads=threads)
            rate["binary BVH, guided"] = W * H / (time.perf_counter() - t0) / 1e6
            assert np.isfinite(o.irradiance()).all()
            ctx.set_property("reference mode", 1)
        o.close()
    assert imgs[0][..., :3].sum() > 0
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
    print("config 1 (synth_start 320x240 1 spp, %d threads): " % threads + ", ".join("%s %.4f Msamples/s" % kv for kv in rate.items()))
    ctx.close()
