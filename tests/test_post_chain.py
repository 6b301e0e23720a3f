"""The step after the path: temporal accumulation with motion-vector reprojection, albedo re-modulation
and composition -- the graph's `accum` / `volume accum` / `add` nodes (res/default_config.json:21-133,404-435,473-497).
merian's node sources are absent from the reference tree, so the arithmetic is a DEFINITION of this build (DESIGN.md
section 3, "post chain"): PARITY UNPINNED against the reference; what is pinned is kernel == oracle, bit for bit, and
the properties the definition must have (running mean, history reset on disocclusion, 1/sqrt(N) convergence)."""
import json
import os

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = {"adaptive grid buf size": 1 << 18, "static grid buf size": 1 << 14, "LC buf size": 1 << 16}


def oracle_for(ctx, W, H):
    o = orc.Oracle(orc.params_from_ctx(ctx, ctx.get_constants()))
    orc.mirror_scene(ctx, o)
    o.commit(1)
    o.connect(W, H)
    o.post_params_from_ctx(ctx)
    return o


def host_ctx(scene, seed, props):
    import mqhip
    ctx = mqhip.Context(-1)
    ctx.header_defaults()
    ctx.synth_scene(scene, seed)
    for k, v in {"randomize seed": 0, "seed": 0x5EED, **SMALL, **props}.items():
        ctx.set_property(k, v)
    ctx.commit()
    return ctx


def test_oracle_accumulation_is_a_running_mean_and_resets_on_disocclusion(built):
    """alpha = 1: out = mean of the frames seen, history = their number; a camera jump invalidates the pixels whose
    reprojected normal / depth no longer match, and only those."""
    ctx = host_ctx("synth_tiny", 3, {"reference mode": 1, "spp": 1, "accum: alpha": 1.0})
    W, H = 64, 48
    o = oracle_for(ctx, W, H)
    u = ctx.synth_camera(10)
    for k in range(3):  # a camera at rest: previous pose = pose (the fourth components carry the medium, not the pose)
        u.prev_cam_x[k] = u.cam_x[k]; u.prev_cam_w[k] = u.cam_w[k]; u.prev_cam_u[k] = u.cam_u[k]
    frames = []
    for f in range(5):
        u.frame = 100 + f  # static camera, new random numbers
        o.process(u); o.post_process()
        frames.append(o.irradiance().astype(np.float64))
        mean = np.mean(frames, axis=0)
        acc = o.post_output(o.POST_ACCUM)
        assert np.allclose(acc, mean, rtol=1e-5, atol=1e-6)
        assert (o.post_output(o.POST_ACCUM_HISTORY) == f + 1).all()
    final = o.post_output(o.POST_FINAL)
    albedo = o.output(orc.OUT_GB_ALBEDO).view(np.float16).reshape(H, W, 4).astype(np.float32)
    emis = o.output(orc.OUT_GB_IRRADIANCE).view(np.float16).reshape(H, W, 4).astype(np.float32)
    assert np.array_equal(final[..., :3], (acc[..., :3] * albedo[..., :3] + o.post_output(o.POST_VOLUME_ACCUM)[..., :3]) + emis[..., :3])
    assert (final[..., 3] == 1).all() and final[..., :3].sum() > 0
    # a jump along the fly-through: most pixels see other surfaces -> their history restarts at 1, the rest keeps counting
    o.process(ctx.synth_camera(60)); o.post_process()
    h = o.post_output(o.POST_ACCUM_HISTORY)
    assert (h == 1).mean() > 0.3 and set(np.unique(h)) <= {1.0, 6.0}
    o.post_clear()
    o.process(ctx.synth_camera(61)); o.post_process()
    assert (o.post_output(o.POST_ACCUM_HISTORY) == 1).all()


def test_post_properties_load_from_the_reference_graph_file(built):
    """mq_load_properties_json maps the graph's node names to the post chain's property prefixes."""
    import mqhip
    ctx = mqhip.Context(-1)
    text = json.dumps({"nodes": {"accum": {"properties": {"alpha": 0.5, "max history": "inf", "depth threshold": 0.125, "enable motion vectors": False, "firefly filter enable": True}},
                                 "volume accum": {"properties": {"alpha": 0.25, "max history": 32, "normal threshold": 1.5}}}})
    ctx.load_properties_json(text, "accum"); ctx.load_properties_json(text, "volume accum")
    assert ctx.get_property("accum: alpha") == 0.5 and ctx.get_property("accum: depth threshold") == 0.125 and ctx.get_property("accum: enable motion vectors") == 0
    assert np.isinf(ctx.get_property("accum: max history")) and ctx.get_property("volume accum: max history") == 32
    assert ctx.get_property("volume accum: alpha") == 0.25 and ctx.get_property("volume accum: normal threshold") == 1.5
    ref = os.path.join("/root/reference/res/default_config.json")
    if os.path.exists(ref):  # the shipped configuration itself (absent on the GPU box)
        ctx2 = mqhip.Context(-1)
        for node in ("accum", "volume accum"):
            ctx2.load_properties_json(open(ref).read(), node)
        assert abs(ctx2.get_property("accum: alpha") - 0.951) < 1e-6 and abs(ctx2.get_property("volume accum: depth threshold") - 0.284027) < 1e-5
        ctx3 = mqhip.Context(-1)  # and they are this build's defaults
        for k in ("alpha", "max history", "normal threshold", "depth threshold", "enable motion vectors", "reuse border"):
            for p in ("accum: ", "volume accum: "):
                assert ctx2.get_property(p + k) == ctx3.get_property(p + k), p + k


@pytest.fixture(scope="module")
def gpu_ctx(mqlib):
    import mqhip
    ctx = mqhip.Context(0)
    yield ctx
    ctx.close()


def gpu_pair(ctx, scene, seed, props, W, H):
    ctx.header_defaults()
    ctx.synth_scene(scene, seed)
    for k, v in {"randomize seed": 0, "seed": 0x5EED, **SMALL, **props}.items():
        ctx.set_property(k, v)
    ctx.commit(); ctx.set_partition(0, 1); ctx.connect(W, H)
    return oracle_for(ctx, W, H)


# (forward projection on: its scatter is resolved deterministically since round 3, `volume accum` follows the projected vectors on both sides)
VOLDET = {"volume spp": 2, "particle size": 7.0, "volume: use LC": 1, "dist guide p": 0.9, "Phase Prob": 0.1, "mc samples": 0, "dist mc samples": 0, "volume forward project": 1}


@pytest.mark.gpu
@pytest.mark.parametrize("scene,extra", [("synth_tiny_fog", {**VOLDET}),                                    # surface + volume images
                                         ("synth_start", {"accum: alpha": 1.0, "accum: max history": 4}),   # capped running mean
                                         ("synth_materials", {"accum: enable motion vectors": 0, "accum: reuse border": 0, "accum: depth threshold": 0.5})])
def test_post_chain_matches_oracle(gpu_ctx, scene, extra):
    """mq_accumulate_kernel / mq_compose_kernel against the oracle over a moving camera: accumulated images, histories
    and the final composition bit-identical in every frame (the renderer underneath is in reference mode, so its outputs
    are bit-identical too)."""
    import mqhip
    ctx = gpu_ctx
    W, H = 160, 96
    o = gpu_pair(ctx, scene, 5, {"reference mode": 1, "spp": 1, **extra}, W, H)
    pairs = ((mqhip.OUT_ACCUM, o.POST_ACCUM), (mqhip.OUT_ACCUM_HISTORY, o.POST_ACCUM_HISTORY), (mqhip.OUT_VOLUME_ACCUM, o.POST_VOLUME_ACCUM),
             (mqhip.OUT_VOLUME_ACCUM_HISTORY, o.POST_VOLUME_ACCUM_HISTORY), (mqhip.OUT_FINAL, o.POST_FINAL))
    seen_reset = seen_keep = False
    for f in (0, 1, 2, 3, 30, 31):  # small steps along the fly-through, then a jump
        u = ctx.synth_camera(f * 3)
        ctx.process(u); ctx.post_process()
        o.process(u, threads=8); o.post_process()
        for g, r in pairs:
            a = ctx.read_output(g).view(np.uint32); b = o.post_output(r).view(np.uint32).reshape(-1)
            assert np.array_equal(a, b), "frame %d output %d: %d values differ" % (f, g, (a != b).sum())
        h = o.post_output(o.POST_ACCUM_HISTORY)
        seen_reset |= bool(f > 0 and (h == 1).any()); seen_keep |= bool((h > 2).any())
    assert seen_reset and seen_keep and o.post_output(o.POST_FINAL)[..., :3].sum() > 0
    if "volume spp" in extra:
        assert o.post_output(o.POST_VOLUME_ACCUM)[..., :3].sum() > 0
    ctx.post_clear(); o.post_clear()
    u = ctx.synth_camera(95)
    ctx.process(u); ctx.post_process(); o.process(u, threads=8); o.post_process()
    assert (ctx.read_output(mqhip.OUT_ACCUM_HISTORY).view(np.float32) == 1).all()
    assert np.array_equal(ctx.read_output(mqhip.OUT_FINAL).view(np.uint32), o.post_output(o.POST_FINAL).view(np.uint32).reshape(-1))


@pytest.mark.gpu
def test_convergence_curve_guided_vs_unguided(gpu_ctx):
    """The reference's own evaluation method (scripts/error_plot.py:22-56: error against a long reference at power-of-two
    iteration counts, log-log): the accumulated image (alpha = 1: running mean) of a static view converges to the long
    unguided mean like 1/sqrt(N), and Markov-chain guiding gets there with far fewer samples in the lighting it is
    made for -- `synth_lamps`: indoor rooms, one small ceiling light each (BSDF sampling finds a light for 0.3 % of the
    pixels per frame, guiding for more than half).  Two error measures per N: the RMSE (error_plot.py's; dominated by
    the rare bright outliers of either estimator) and the median absolute error per pixel.  On the many-light stand-ins
    (synth_start: 6 % of all ceiling tiles emit) guiding does NOT pay: at most five lobes per vertex cannot cover dozens
    of emitters and one-sample MIS charges the BSDF branch 1 / "BSDF Prob" -- tools/variance_diag.py, DESIGN.md.
    The curves are written to gpurun_out/r02_rmse_curve.json (committed under profiles/)."""
    import mqhip
    ctx = gpu_ctx
    W, H = 192, 128
    counts = [1, 2, 4, 8, 16, 32, 64, 128, 256]

    def run(mode, n_frames, first_frame, warm=0):
        ctx.header_defaults()
        ctx.synth_scene("synth_lamps", 3)
        for k, v in {"randomize seed": 0, "seed": 0x5EED, **SMALL, "reference mode": mode, "spp": 1, "max path length": 3, "accum: alpha": 1.0}.items():
            ctx.set_property(k, v)
        ctx.commit(); ctx.set_partition(0, 1); ctx.connect(W, H)
        u = ctx.synth_camera(0)
        for f in range(warm):  # guiding learns before the accumulation starts (the reference evaluates a converged cache)
            u.frame = 50000 + f
            ctx.process(u)
        ctx.post_clear()
        out = {}
        for f in range(n_frames):
            u.frame = first_frame + f
            ctx.process(u); ctx.post_process()
            if f + 1 in counts or f + 1 == n_frames:
                out[f + 1] = ctx.image(mqhip.OUT_ACCUM)[..., :3].astype(np.float64)
        return out
    ref = run(1, 16384, 1000000)[16384]
    assert ref.mean() > 0
    rmse, med = {}, {}
    for name, mode, warm in (("unguided", 1, 0), ("guided", 0, 128)):
        imgs = run(mode, counts[-1], 2000)
        rmse[name] = [float(np.sqrt(((imgs[n] - ref) ** 2).mean())) for n in counts]
        med[name] = [float(np.median(np.abs(imgs[n] - ref).mean(-1))) for n in counts]
    res = {"scene": "synth_lamps(seed=3) %dx%d, static view, 1 spp, max path length 3" % (W, H), "reference": "unguided running mean of 16384 frames",
           "N": counts, "rmse": rmse, "median_abs_error": med, "reference_mean": float(ref.mean())}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r02_rmse_curve.json"), "w"), indent=1)
    slope = np.polyfit(np.log2(counts), np.log2(rmse["unguided"]), 1)[0]
    assert -0.65 < slope < -0.35, (slope, rmse)  # 1 / sqrt(N)
    # the typical pixel is well ahead with guiding (measured 0.51x at N = 256; per-pixel variance 25x lower in the median,
    # tools/variance_diag.py); the RMSE is decided by the estimators' rare bright outliers -- the guided one's are brighter
    # (a BSDF-branch sample that finds the lamp is weighted 1 / "BSDF Prob") -- and is reported, not asserted
    assert med["guided"][-1] < 0.65 * med["unguided"][-1], med
    assert all(g < u for g, u in zip(med["guided"][3:], med["unguided"][3:])), med


@pytest.mark.gpu
def test_add_node_takes_the_restir_irradiance(gpu_ctx):
    """BASELINE config 5, "ReSTIR DI + MCPG GI combined": with `"add: restir irradiance"` the composition adds the ReSTIR node's
    irradiance, re-modulated with the albedo like the MCPG irradiance (one more input of the graph's `add` node,
    res/default_config.json:429-435; DEFINED by this build like the rest of the post chain).  MCPG + ReSTIR node + post chain
    over a moving camera: the final image bit-identical to the oracle, and different from the one without the input."""
    import mqhip
    ctx = gpu_ctx
    W, H = 160, 96
    o = gpu_pair(ctx, "synth_start", 5, {"reference mode": 1, "spp": 1, "restir: randomize seed": 0, "restir: seed": 77, "restir: spp": 2,
                                         "restir: enable temporal reuse": 1, "restir: spatial reuse iterations": 1, "add: restir irradiance": 1}, W, H)
    rp = orc.restir_params_from_ctx(ctx)
    for f in range(4):
        u = ctx.synth_camera(f * 3)
        ctx.process(u); ctx.restir_process(u); ctx.post_process()
        o.process(u, threads=8); o.restir_process(rp, u, threads=8); o.post_process()
        a = ctx.read_output(mqhip.OUT_FINAL).view(np.uint32); b = o.post_output(o.POST_FINAL).view(np.uint32).reshape(-1)
        assert np.array_equal(a, b), "frame %d: %d values differ" % (f, (a != b).sum())
    with_direct = ctx.image(mqhip.OUT_FINAL).copy()
    accum = ctx.image(mqhip.OUT_ACCUM); direct = ctx.image(mqhip.OUT_RESTIR_IRRADIANCE)
    albedo = ctx.read_output(mqhip.OUT_GB_ALBEDO).view(np.float16).reshape(H, W, 4).astype(np.float32)
    emis = ctx.read_output(mqhip.OUT_GB_IRRADIANCE).view(np.float16).reshape(H, W, 4).astype(np.float32)
    base = (accum[..., :3] * albedo[..., :3] + ctx.image(mqhip.OUT_VOLUME_ACCUM)[..., :3]) + emis[..., :3]
    assert np.array_equal(with_direct[..., :3], base + direct[..., :3] * albedo[..., :3]) and (direct[..., :3] * albedo[..., :3]).sum() > 0
    ctx.set_property("add: restir irradiance", 0)
    o.close()
