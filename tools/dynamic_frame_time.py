"""Frame time with PER-FRAME geometry: the bench frame (1920x1080, guided, JSON defaults) with P particles rebuilt, committed
and rendered every frame (QuakeNode::update_dynamic_geo + the TLAS build of the reference's graph, quake_node.cpp:896-983),
against the same frames without them.  Wall clock over the whole loop: producer + BVH build + upload + render.
Usage: python tools/dynamic_frame_time.py [P ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
import mqhip

counts = [int(a) for a in sys.argv[1:]] or [0, 256, 2048, 16384]
rng = np.random.default_rng(5)
for P in counts:
    ctx = mqhip.Context(0)
    ctx.json_defaults()
    for k, v in {"randomize seed": 0, "seed": 0x5EED, "spp": 1, "max path length": 3}.items():
        ctx.set_property(k, v)
    if os.environ.get("MQ_DYN_BVH"):  # who builds the per-frame tree: host (default) | device | auto
        ctx.set_property("per-frame BVH", os.environ["MQ_DYN_BVH"])
    ctx.synth_scene("synth_sepulcher", 2); ctx.commit(); ctx.connect(1920, 1080)
    u0 = ctx.synth_camera(0)
    parts = np.zeros(P, mqhip.PARTICLE_DTYPE)
    if P:
        parts["org"] = np.array([u0.cam_x[0], u0.cam_x[1], u0.cam_x[2]]) + rng.uniform(-200, 200, (P, 3))
        parts["vel"] = rng.uniform(-30, 30, (P, 3)); parts["seed"] = rng.integers(1, 2 ** 32, P)
        parts["color_rgba"] = rng.choice([0x0000003c, 0x00ffffff, 0x0040a0ff], P); parts["type"] = rng.choice([0, 3, 5], P)
    view = mqhip.View()
    for k in range(3):
        view.origin[k] = u0.cam_x[k]; view.forward[k] = u0.cam_w[k]; view.up[k] = u0.cam_u[k]
    view.right[0], view.right[1], view.right[2] = 0.0, -1.0, 0.0

    host = [0.0, 0.0, 0.0, 0.0]  # seconds the host spent in the producer, in the commit, in mq_process, and moving the particles (numpy, the "game")

    def frame(f):
        u = ctx.synth_camera(f)
        tg = time.perf_counter()
        if P:
            parts["prev_org"] = parts["org"]; parts["org"] = parts["org"] + parts["vel"] / 60.0
        t0 = time.perf_counter()
        host[3] += t0 - tg
        if P:
            ctx.dyn_begin(); ctx.dyn_add_particles(parts, view, 1, 2, u.cl_time, u.cl_time - 1 / 60.0); ctx.dyn_end(2)
            t1 = time.perf_counter()
            ctx.commit()
        else:
            t1 = t0
        t2 = time.perf_counter()
        ctx.process(u)
        t3 = time.perf_counter()
        host[0] += t1 - t0; host[1] += t2 - t1; host[2] += t3 - t2

    for f in range(64):
        frame(f)
    ctx.sync(); ctx.timing_set_interval(1000); ctx.timing_reset()
    host[:] = [0.0, 0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for f in range(64, 164):
        frame(f)
    ctx.sync()
    ms = (time.perf_counter() - t0) * 10.0
    tn, t_render, t_update = ctx.timing_get()
    dev_ms = (t_render + t_update) / max(tn, 1)  # the frame's own kernels, first launch to last (the device-side tree build runs beside them on another stream)
    n_tris = ctx.scene_stats()["n_tris"]
    print("particles %6d: %.3f ms per frame (wall, 100 frames), %.3f ms of it the frame's kernels on the device; host: moving the particles (numpy) %.3f + producer %.3f + commit %.3f + mq_process %.3f ms; scene triangles %d; commits that did not wait %d of %d, trees built on the device %d"
          % (P, ms, dev_ms, host[3] * 10, host[0] * 10, host[1] * 10, host[2] * 10, n_tris, ctx.commit_async_count(), ctx.commit_counts()[1], ctx.commit_device_count()), flush=True)
    ctx.close()
