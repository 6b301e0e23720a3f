"""Frame time with PER-FRAME geometry: the bench frame (1920x1080, guided, JSON defaults) with P particles rebuilt, committed
and rendered every frame (QuakeNode::update_dynamic_geo + the TLAS build of the reference's graph, quake_node.cpp:896-983),
against the same frames without them.  Wall clock over the whole loop: producer + BVH build + upload + render.
Usage: python tools/dynamic_frame_time.py [P ...]
MQ_ENTITIES=E adds E alias-model entities (a generated 600-vertex / 1200-triangle model, moving and animating) to every frame, through
mq_dyn_add_alias_batch (MQ_ENTITIES_SINGLE=1: one mq_dyn_add_alias call per entity)."""
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
    E = int(os.environ.get("MQ_ENTITIES", "0"))
    if E:
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import quake_files as Q
        mdl = os.path.join(tempfile.mkdtemp(), "m.mdl")
        Q.write_mdl(mdl, np.random.default_rng(3), numverts=600, numtris=1200, numframes=3, skinw=64, skinh=64)
        alias, _ = ctx.load_mdl(mdl, 300)
        ctx.commit()
        ents = []
        for e in range(E):
            ai = mqhip.AliasInstance()
            for k in range(3):
                ai.origin[k] = float(u0.cam_x[k] + rng.uniform(-400, 400)); ai.prev_origin[k] = ai.origin[k]; ai.angles[k] = float(rng.uniform(0, 360)); ai.prev_angles[k] = ai.angles[k]
            ai.pose1, ai.pose2, ai.blend, ai.prev_blend, ai.skin, ai.fovscale = 0, 1, 0.0, 0.0, 0, 1.0
            ents.append(ai)

    host = [0.0, 0.0, 0.0, 0.0]  # seconds the host spent in the producer, in the commit, in mq_process, and moving the particles (numpy, the "game")

    def frame(f):
        u = ctx.synth_camera(f)
        tg = time.perf_counter()
        if P:
            parts["prev_org"] = parts["org"]; parts["org"] = parts["org"] + parts["vel"] / 60.0
        t0 = time.perf_counter()
        host[3] += t0 - tg
        if P or E:
            ctx.dyn_begin()
            if P:
                ctx.dyn_add_particles(parts, view, 1, 2, u.cl_time, u.cl_time - 1 / 60.0)
            if E:
                for ai in ents:  # (the "game": every entity turns and animates)
                    ai.prev_angles[1] = ai.angles[1]; ai.angles[1] += 2.0; ai.prev_blend = ai.blend; ai.blend = (ai.blend + 0.1) % 1.0
                if os.environ.get("MQ_ENTITIES_SINGLE"):
                    for ai in ents:
                        ctx.dyn_add_alias(alias, ai)
                else:
                    ctx.dyn_add_alias_batch([alias] * E, ents)
            ctx.dyn_end(2)
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
