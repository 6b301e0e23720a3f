"""Soak run of the per-frame geometry path: reference-mode frames (deterministic) of synth_start at 640x400, a particle cloud of
changing size committed before every frame WITHOUT waiting for the device (mq_scene_commit's asynchronous path), with, at random:
clouds that outgrow their device region (full re-upload), a change of static geometry (full commit), no particles at all, a ReSTIR +
post pass, a device synchronisation.  Every 50th frame the same frame is rendered by a second context that synchronises before and
after everything (the slow, obviously ordered way) from the same inputs: first hits and radiance must be bit-identical.
   python tools/soak_dynamic.py [frames]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
import mqhip

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
W, H = 640, 400
PROPS = {"randomize seed": 0, "seed": 0x5EED, "reference mode": 1, "spp": 1, "max path length": 3, "adaptive grid buf size": 1 << 18, "static grid buf size": 1 << 14, "LC buf size": 1 << 16,
         "restir: randomize seed": 0, "restir: spp": 1}


def make(dyn_bvh=None):
    c = mqhip.Context(0)
    c.header_defaults()
    c.synth_scene("synth_start", 4)
    for k, v in PROPS.items():
        c.set_property(k, v)
    if dyn_bvh:
        c.set_property("per-frame BVH", dyn_bvh)
    c.commit(); c.connect(W, H)
    return c


a, b = make(os.environ.get("MQ_DYN_BVH")), make("host")  # MQ_DYN_BVH=device: the pipelined context's per-frame trees are built on the device, the reference context's on the host
rng = np.random.default_rng(11)
u0 = a.synth_camera(0)
eye = np.array([u0.cam_x[0], u0.cam_x[1], u0.cam_x[2]]); fwd = np.array([u0.cam_w[0], u0.cam_w[1], u0.cam_w[2]])
view = mqhip.View()
for k in range(3):
    view.origin[k] = u0.cam_x[k]; view.forward[k] = u0.cam_w[k]; view.up[k] = u0.cam_u[k]
view.right[0], view.right[1], view.right[2] = [float(x) for x in np.cross(fwd, np.array([u0.cam_u[0], u0.cam_u[1], u0.cam_u[2]]))]
g0 = a.get_geometry(0)
checked = bad = big = statics = 0
t0 = time.time()
for f in range(frames):
    r = rng.random()
    n = 0 if r < 0.05 else (int(rng.integers(6000, 9000)) if r < 0.08 else int(rng.integers(50, 1500)))  # > 4096 particles = 16 k triangles outgrow the region now and then
    big += n > 4096
    parts = np.zeros(n, mqhip.PARTICLE_DTYPE)
    if n:
        parts["org"] = eye + fwd * 60.0 + rng.uniform(-50, 50, (n, 3)); parts["prev_org"] = parts["org"] - rng.uniform(-2, 2, (n, 3))
        parts["seed"] = rng.integers(1, 2 ** 32, n); parts["color_rgba"] = rng.choice([0x0000003c, 0x00ffffff, 0x0040a0ff], n); parts["type"] = rng.choice([0, 3, 5], n)
    static_change = rng.random() < 0.01
    if static_change:  # nudge one static vertex: the static tree is rebuilt, everything is uploaded again
        statics += 1
        g0["vtx"][int(rng.integers(0, len(g0["vtx"])))] += np.float32(0.01)
    u = a.synth_camera(int(rng.integers(0, 200)))
    check = f % 50 == 49
    for ctx, careful in ((a, False), (b, True)):
        if careful and not check and not static_change:
            continue  # the reference context only follows what changes the scene for good, and renders the frames that are compared
        if careful:
            ctx.sync()
        if static_change:
            ctx.set_geometry(0, g0["vtx"], g0["vtx"], g0["idx"], g0["ext"], g0["flags"])
        if check or not careful:
            ctx.dyn_begin()
            if n:
                ctx.dyn_add_particles(parts, view, 1, 2, f / 60.0, (f - 1) / 60.0)
            ctx.dyn_end(2)
        ctx.commit()
        if check or not careful:
            ctx.process(u)
            if rng.random() < 0.1 and not careful:
                ctx.restir_process(u); ctx.post_process()
        if careful:
            ctx.sync()
    if rng.random() < 0.02:
        a.sync()
    if check:
        a.sync()
        same = all(np.array_equal(a.read_output(o), b.read_output(o)) for o in (mqhip.OUT_HITS, mqhip.OUT_IRRADIANCE, mqhip.OUT_GB_MV))
        checked += 1; bad += not same
        if not same or checked % 10 == 0:
            print("frame %d: %d compared, %d differ; commits full / per-frame / without waiting / trees built on the device: %s / %d / %d; clouds beyond the region %d, static changes %d; %.0f s"
                  % (f + 1, checked, bad, a.commit_counts(), a.commit_async_count(), a.commit_device_count(), big, statics, time.time() - t0), flush=True)
print("done: %d frames, %d compared, %d differ" % (frames, checked, bad))
a.close(); b.close()
sys.exit(1 if bad else 0)
