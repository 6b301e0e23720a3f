"""Scheduling options of this build (no reference counterpart: "pipelines", "overlap camera rays", the rank partition)
decide WHEN launches run, never what they compute: the node outputs are bit-identical for every setting, in reference
mode and for guided frames rendered from a given (frozen) learning state."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

SMALL = {"adaptive grid buf size": 1 << 18, "static grid buf size": 1 << 14, "LC buf size": 1 << 16}
SETTINGS = [(1, "off"), (2, "off"), (4, "off"), (1, "always"), (3, "always"), (1, "update pass"), (2, "auto"), (1, "last round"), (1, "last bounce")]


def _outputs(ctx):
    import mqhip
    names = ("OUT_IRRADIANCE", "OUT_GB_ALBEDO", "OUT_GB_IRRADIANCE", "OUT_GB_MV", "OUT_GBUFFER", "OUT_HITS")
    return {n: ctx.read_output(getattr(mqhip, n)).copy() for n in names}


def _render(mqlib, props, pipelines, overlap, frames, sync_every_frame, learned=None, partition=None):
    import mqhip
    ctx = mqhip.Context(0)
    ctx.header_defaults()
    ctx.synth_scene("synth_start", 4)
    for k, v in {"randomize seed": 0, "seed": 0x5EED, **SMALL, **props, "pipelines": pipelines, "overlap camera rays": overlap}.items():
        ctx.set_property(k, v)
    ctx.commit()
    if partition:
        ctx.set_partition(*partition)
    ctx.connect(328, 200)
    ctx.process(ctx.synth_camera(0))
    if learned is not None:
        ctx.state_write(0, learned[0]); ctx.state_write(1, learned[1])
        ctx.set_property("debug: freeze learning", 1)
    for f in range(1, frames):
        ctx.process(ctx.synth_camera(f))   # the host runs ahead of the device unless something reads
        if sync_every_frame:
            ctx.sync()
    out = _outputs(ctx)
    assert ctx.counters()["queue_overflow"] == 0
    ctx.close()
    return out


def _same(a, b, what):
    assert a.keys() == b.keys() and len(a) >= 2
    for k in a:
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), (what, k)


@pytest.mark.parametrize("sync_every_frame", [False, True])
def test_reference_mode_outputs_do_not_depend_on_scheduling(mqlib, sync_every_frame):
    props = {"reference mode": 1, "spp": 2, "max path length": 3}
    base = _render(mqlib, props, 1, "off", 7, True)
    assert base["OUT_IRRADIANCE"].view(np.float32).sum() > 0
    for pipelines, overlap in SETTINGS[1:]:
        _same(base, _render(mqlib, props, pipelines, overlap, 7, sync_every_frame), (pipelines, overlap))


def test_partitioned_rank_outputs_do_not_depend_on_camera_ray_overlap(mqlib):
    """"auto" (the default) overlaps from the start of the previous frame for a rank of a tile partition."""
    props = {"reference mode": 1, "spp": 1, "max path length": 3}
    a = _render(mqlib, props, 1, "off", 6, False, partition=(1, 3))
    b = _render(mqlib, props, 1, "auto", 6, False, partition=(1, 3))
    _same(a, b, "rank 1 of 3")


def test_guided_frames_from_a_frozen_state_do_not_depend_on_scheduling(mqlib):
    import mqhip
    props = {"reference mode": 0, "spp": 1, "max path length": 3}
    # learn something with the default scheduling, then replay it everywhere
    ctx = mqhip.Context(0)
    ctx.header_defaults()
    ctx.synth_scene("synth_start", 4)
    for k, v in {"randomize seed": 0, "seed": 0x5EED, **SMALL, **props}.items():
        ctx.set_property(k, v)
    ctx.commit(); ctx.connect(328, 200)
    for f in range(12):
        ctx.process(ctx.synth_camera(f))
    learned = (ctx.state_read(0, SMALL["adaptive grid buf size"] + SMALL["static grid buf size"]), ctx.state_read(1, SMALL["LC buf size"]))
    ctx.close()
    assert (learned[0]["sum_w"] > 0).sum() > 1000
    base = _render(mqlib, props, 1, "off", 5, True, learned=learned)
    assert base["OUT_IRRADIANCE"].view(np.float32).sum() > 0
    for pipelines, overlap in SETTINGS[1:]:
        _same(base, _render(mqlib, props, pipelines, overlap, 5, False, learned=learned), (pipelines, overlap))


def _frames_with_restir(mqlib, props, with_restir, learned=None, scene="synth_start"):
    import mqhip
    ctx = mqhip.Context(0)
    ctx.header_defaults()
    ctx.synth_scene(scene, 4)
    for k, v in {"randomize seed": 0, "seed": 0x5EED, **SMALL, **props, "restir: randomize seed": 0, "restir: spp": 2,
                 "restir: enable temporal reuse": 1, "restir: spatial reuse iterations": 1}.items():
        ctx.set_property(k, v)
    ctx.commit(); ctx.connect(200, 120)
    u = ctx.synth_camera(0)
    ctx.process(u)
    if with_restir:
        ctx.restir_process(u)
    if learned is not None:
        ctx.state_write(0, learned[0]); ctx.state_write(1, learned[1])
        ctx.set_property("debug: freeze learning", 1)
    for f in range(1, 6):
        u = ctx.synth_camera(f)
        ctx.process(u)
        if with_restir:
            ctx.restir_process(u)
    out = (ctx.irradiance().copy(), ctx.volume().copy(), ctx.counters()["queue_overflow"],
           float(ctx.read_output(mqhip.OUT_RESTIR_IRRADIANCE).view(np.float32).sum()) if with_restir else 0.0)
    ctx.close()
    return out


def test_restir_node_between_frames_does_not_disturb_the_mcpg_node(mqlib):
    """The ReSTIR node borrows the MCPG node's ray queues, hit buffers, queue counters and path records between its frames
    (its generate and shade rays run through mq_trace_queue_kernel).  Deterministic MCPG frames -- guided from a frozen
    state, and unguided with the volume pass -- are bit-identical with and without it."""
    import mqhip
    props = {"reference mode": 0, "spp": 1, "max path length": 3}
    ctx = mqhip.Context(0)
    ctx.header_defaults()
    ctx.synth_scene("synth_start", 4)
    for k, v in {"randomize seed": 0, "seed": 0x5EED, **SMALL, **props}.items():
        ctx.set_property(k, v)
    ctx.commit(); ctx.connect(200, 120)
    for f in range(12):
        ctx.process(ctx.synth_camera(f))
    learned = (ctx.state_read(0, SMALL["adaptive grid buf size"] + SMALL["static grid buf size"]), ctx.state_read(1, SMALL["LC buf size"]))
    ctx.close()
    a = _frames_with_restir(mqlib, props, 0, learned)
    b = _frames_with_restir(mqlib, props, 1, learned)
    assert a[2] == 0 and b[2] == 0 and b[3] > 0 and a[0][..., :3].sum() > 0
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    vol = {"reference mode": 1, "spp": 1, "max path length": 3, "volume spp": 2, "particle size": 7.0, "volume: use LC": 1, "dist guide p": 0.9,
           "Phase Prob": 0.1, "mc samples": 0, "dist mc samples": 0, "volume forward project": 0}
    a = _frames_with_restir(mqlib, vol, 0, scene="synth_start_fog")
    b = _frames_with_restir(mqlib, vol, 1, scene="synth_start_fog")
    assert a[2] == 0 and b[2] == 0 and a[1][..., :3].sum() > 0
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def test_reset_reconnect_and_repartition_with_frames_in_flight(mqlib):
    """State changes between frames while the host runs ahead of the device (camera rays of the next frame in flight on the
    side stream): mq_reset_state, a reconnect at another size, a new partition -- the frames after each change equal those
    of a fresh context that starts there (reference mode)."""
    import mqhip
    props = {"randomize seed": 0, "seed": 0x5EED, **SMALL, "reference mode": 1, "spp": 1, "max path length": 3}

    def fresh(W, H, partition, frames):
        c = mqhip.Context(0)
        c.header_defaults(); c.synth_scene("synth_start", 4)
        for k, v in props.items():
            c.set_property(k, v)
        c.commit()
        if partition:
            c.set_partition(*partition)
        c.connect(W, H)
        for f in frames:
            c.process(c.synth_camera(f))
        out = (c.irradiance().copy(), c.read_output(mqhip.OUT_TILES).copy())
        c.close()
        return out

    c = mqhip.Context(0)
    c.header_defaults(); c.synth_scene("synth_start", 4)
    for k, v in props.items():
        c.set_property(k, v)
    c.commit(); c.connect(328, 200)
    for f in range(5):
        c.process(c.synth_camera(f))
    c.reset_state()
    for f in (7, 8):
        c.process(c.synth_camera(f))
    a = c.irradiance().copy()
    assert np.array_equal(a.view(np.uint32), fresh(328, 200, None, (7, 8))[0].view(np.uint32))
    c.connect(200, 136)
    for f in (9, 10, 11):
        c.process(c.synth_camera(f))
    assert np.array_equal(c.irradiance().view(np.uint32), fresh(200, 136, None, (9, 10, 11))[0].view(np.uint32))
    c.set_partition(1, 3); c.connect(200, 136)
    for f in (12, 13):
        c.process(c.synth_camera(f))
    assert np.array_equal(c.read_output(mqhip.OUT_TILES), fresh(200, 136, (1, 3), (12, 13))[1])
    assert c.counters()["queue_overflow"] == 0
    c.close()


def test_ray_queue_overflow_is_flagged_not_fatal(mqlib):
    """A ray queue that is too small (cannot happen with the 2x + 1024 sizing; forced here by MQ_DEBUG_RAY_CAP_DIV, which tells the
    kernels of a quarter of the positions that are allocated) drops rays and raises overflow bit 1: no access past the
    queue, finite output, and the frame is recognisable as invalid.  Runs in a child process (the variable is read once)."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import mqhip
c = mqhip.Context(0)
c.header_defaults(); c.synth_scene("synth_start", 4)
for k, v in {"randomize seed": 0, "seed": 0x5EED, "adaptive grid buf size": 1 << 18, "static grid buf size": 1 << 14, "LC buf size": 1 << 16,
             "reference mode": 0, "spp": 2, "max path length": 4, "volume spp": 1, "particle size": 7.0}.items():
    c.set_property(k, v)
c.commit(); c.connect(640, 360)
for f in range(4):
    c.process(c.synth_camera(f))
img = c.irradiance()
print("RESULT", int(c.counters()["queue_overflow"]), bool(np.isfinite(img).all()), float(img[..., :3].sum()) > 0)
c.close()
""" % os.path.join(ROOT, "merian-quake_amd")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MQ_DEBUG_RAY_CAP_DIV="4"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    assert int(line[1]) & 1 and line[2] == "True" and line[3] == "True", line


def _chain_uniform(frame, mu_t):
    """Camera at the near end of tests/deep_scene.py's chain, looking down the axis of the frames; a little fog."""
    import mqhip
    u = mqhip.Uniform()
    x = 0.25 + 0.002 * frame
    for k, v in enumerate((x, 0.0, 0.0, mu_t)):
        u.cam_x[k] = v
    for k, v in enumerate((1.0, 0.0, 0.0, 1.0 / 60.0)):
        u.cam_w[k] = v
    for k, v in enumerate((0.0, 0.0, 1.0, 0.0)):
        u.cam_u[k] = v
    px = 0.25 + 0.002 * max(frame - 1, 0)
    for k, v in enumerate((px, 0.0, 0.0, 0.5 * mu_t)):
        u.prev_cam_x[k] = v
    for k, v in enumerate((1.0, 0.0, 0.0, 0.5 * mu_t)):
        u.prev_cam_w[k] = v
    for k, v in enumerate((0.0, 0.0, 1.0, 0.5 * mu_t)):
        u.prev_cam_u[k] = v
    u.sky_rt_bk = 0xffffffff; u.sky_lf_ft = 0xffffffff; u.sky_up_dn = 0xffffffff
    u.cl_time = frame / 60.0; u.frame = frame
    return u


def test_deep_traversal_stacks_with_overlapped_camera_rays(mqlib):
    """the camera rays of frame n + 1 run on their own stream BESIDE frame n's bounce, volume and ReSTIR
    kernels; every one of those spills the part of a traversal stack beyond its 12 LDS entries to a global area indexed by
    block and thread.  The launches on the side stream now have their own area.  A scene whose central rays need 15 - 16
    stack entries (tests/deep_scene.py proves it on the host), with the per-frame tree, volume passes and the ReSTIR node
    on: outputs with the overlap off == outputs with the overlap from the start of the previous frame."""
    import mqhip
    import deep_scene

    def run(overlap):
        c = mqhip.Context(0)
        c.header_defaults()
        vtx, idx, ext = deep_scene.chain_scene(80, 1.6)
        c.set_geometry(0, vtx, None, idx, ext, mqhip.MQ_GEO_OPAQUE | mqhip.MQ_GEO_STATIC)
        tex = np.full((8, 8, 4), 200, np.uint8)
        c.set_texture(1, tex, mqhip.MQ_TEX_SRGB)
        c.set_constants((4.0, 4.0, 4.0), (0.57735026919, 0.57735026919, 0.57735026919))
        for k, v in {"randomize seed": 0, "seed": 0x5EED, **SMALL, "reference mode": 1, "spp": 2, "max path length": 3, "volume spp": 2, "particle size": 7.0,
                     "volume: use LC": 0, "mc samples": 0, "dist mc samples": 0, "volume forward project": 0, "overlap camera rays": overlap,
                     "restir: randomize seed": 0, "restir: spp": 1, "restir: enable temporal reuse": 1, "restir: spatial reuse iterations": 2,
                     "restir: temporal bias correction": "raytraced", "restir: spatial bias correction": "raytraced", "inline restir rays": 1}.items():
            c.set_property(k, v)
        outs = []
        # a per-frame triangle in front of the camera: the per-frame tree's root waits at the bottom of every stack (+1 entry).
        # Committed ONCE: a commit waits for the frames in flight, and the point here is that the host runs ahead.
        tri = np.array([[3.0, -0.2, -0.2], [3.0, 0.2, -0.2], [3.0, 0.0, 0.2]], np.float32)
        c.set_geometry(5, tri, tri - np.float32(0.01), np.array([[0, 2, 1]], np.uint32), ext[:1], mqhip.MQ_GEO_OPAQUE)
        c.commit()
        nodes, _ = c.get_bvh()
        depth, _ = deep_scene.emulate_stack_depth(nodes, [0.25, 0, 0], [1, 0, 0], start_sp=1)
        assert depth >= 14, "the scene no longer overflows the 12 LDS stack entries (%d)" % depth
        c.connect(256, 160)
        for f in range(6):
            u = _chain_uniform(f, 1e-3)
            c.process(u)
            c.restir_process(u)
        for which in (mqhip.OUT_IRRADIANCE, mqhip.OUT_VOLUME, mqhip.OUT_HITS, mqhip.OUT_RESTIR_IRRADIANCE, mqhip.OUT_RESTIR_RESERVOIRS):
            outs.append(c.read_output(which).copy())
        assert c.counters()["queue_overflow"] == 0
        c.close()
        return outs

    a, b = run("off"), run("always")
    assert a[0].view(np.float32).sum() > 0 and np.isfinite(a[0].view(np.float32)).all()
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))


def _particle_frames(mqlib, frames, sync_every_frame, overlap, W=480, H=300):
    """`frames` frames with PER-FRAME geometry: a different cloud of particles (count and positions) committed before each."""
    import mqhip
    ctx = mqhip.Context(0)
    ctx.header_defaults()
    ctx.synth_scene("synth_start", 4)
    for k, v in {"randomize seed": 0, "seed": 0x5EED, **SMALL, "reference mode": 1, "spp": 2, "max path length": 3, "overlap camera rays": overlap, "accum: alpha": 0.9}.items():
        ctx.set_property(k, v)
    ctx.commit(); ctx.connect(W, H)
    rng = np.random.default_rng(77)
    u0 = ctx.synth_camera(0)
    eye = np.array([u0.cam_x[0], u0.cam_x[1], u0.cam_x[2]]); fwd = np.array([u0.cam_w[0], u0.cam_w[1], u0.cam_w[2]])
    view = mqhip.View()
    for k in range(3):
        view.origin[k] = u0.cam_x[k]; view.forward[k] = u0.cam_w[k]; view.up[k] = u0.cam_u[k]
    right = np.cross(fwd, np.array([u0.cam_u[0], u0.cam_u[1], u0.cam_u[2]]))
    for k in range(3):
        view.right[k] = right[k]
    imgs = []
    for f in range(frames):
        n = 300 + 170 * (f % 4)  # the per-frame tree changes size from frame to frame
        parts = np.zeros(n, mqhip.PARTICLE_DTYPE)
        parts["org"] = eye + fwd * 40.0 + rng.uniform(-30, 30, (n, 3)); parts["prev_org"] = parts["org"] - rng.uniform(-2, 2, (n, 3))
        parts["seed"] = rng.integers(1, 2 ** 32, n); parts["color_rgba"] = rng.choice([0x0000003c, 0x00ffffff, 0x0040a0ff], n); parts["type"] = rng.choice([0, 3, 5], n)
        ctx.dyn_begin(); ctx.dyn_add_particles(parts, view, 1, 2, f / 60.0, (f - 1) / 60.0); ctx.dyn_end(2)
        ctx.commit()           # (asynchronous from the second per-frame commit on: the frames before it are still rendering)
        ctx.process(u0); ctx.post_process()
        if sync_every_frame:
            ctx.sync()
            imgs.append(ctx.read_output(mqhip.OUT_HITS).copy())
    out = _outputs(ctx)
    out["OUT_ACCUM"] = ctx.read_output(mqhip.OUT_ACCUM).copy()  # depends on EVERY frame of the sequence
    stats = {"async": ctx.commit_async_count(), "per_frame": ctx.commit_counts()[1]}
    ctx.close()
    return out, imgs, stats


@pytest.mark.parametrize("overlap", ["off", "auto"])
def test_per_frame_geometry_commits_do_not_wait_and_do_not_race(mqlib, overlap):
    """A commit of per-frame geometry goes to the region of the device arrays that the frames in flight do not read
    (mq_api.cpp, mq_scene_commit) and does not wait for them: eight frames issued back to back, each with its own particles,
    against the same eight frames with a device synchronisation behind every one -- every output of the last frame and the
    image accumulated over all eight are bit-identical; and the particles do change the first hits from frame to frame."""
    piped, _, stats = _particle_frames(mqlib, 8, False, overlap)
    synced, hits, _ = _particle_frames(mqlib, 8, True, overlap)
    _same(piped, synced, overlap)
    assert not np.array_equal(hits[-1], hits[-2]) and not np.array_equal(hits[-2], hits[-3])
    assert stats["async"] >= 6 and stats["per_frame"] == 8, stats
