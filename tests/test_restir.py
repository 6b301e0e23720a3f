"""ReSTIR DI render node (src/render_restir/renderer_restir.cpp, res/shader/render_restir/*): the HIP
kernels through the C ABI against the oracle's restatement, bit for bit -- every pass is one thread per pixel with no
cross-pixel writes, so the node is deterministic.  Reference behaviour is restated from the shaders; the helpers they
take from absent headers are defined in DESIGN.md section 3 (PARITY UNPINNED against the reference, as everywhere)."""
import os

import numpy as np
import pytest

import orc

SMALL = {"adaptive grid buf size": 1 << 16, "static grid buf size": 1 << 12, "LC buf size": 1 << 14}
TH = os.cpu_count() or 8


def setup(ctx, scene, seed, props, W, H, device=True):
    ctx.header_defaults()
    ctx.synth_scene(scene, seed)
    # the MCPG pass is not wanted here: "spp" 0 leaves the g-buffer node's outputs (hits, gbuffer, mv), which the ReSTIR node reads
    for k, v in {"randomize seed": 0, "seed": 0x5EED, **SMALL, "reference mode": 1, "spp": 0, "restir: randomize seed": 0, "restir: seed": 77, **props}.items():
        ctx.set_property(k, v)
    ctx.commit()
    if device:
        ctx.set_partition(0, 1); ctx.connect(W, H)
    o = orc.Oracle(orc.params_from_ctx(ctx, ctx.get_constants()))
    orc.mirror_scene(ctx, o)
    o.commit(1)
    o.connect(W, H)
    return o


def test_oracle_restir_estimates_direct_light(built):
    """Known answer for the estimator itself: averaged over many frames, the ReSTIR DI irradiance (1 candidate per
    pixel, no reuse: plain importance sampling with the RIS weight) equals the one-bounce irradiance the path tracer
    finds with max path length 2 -- both estimate direct light at the first hit, BSDF included, albedo excluded."""
    import mqhip
    ctx = mqhip.Context(-1)
    W, H, N = 48, 32, 160
    o = setup(ctx, "synth_tiny", 3, {"restir: spp": 1}, W, H, device=False)
    rp = orc.restir_params_from_ctx(ctx)
    u = ctx.synth_camera(5)
    acc = np.zeros((H, W, 3))
    for f in range(N):
        u.frame = f
        o.process(u, threads=8); o.restir_process(rp, u, threads=8)
        acc += o.restir_output(0)[..., :3]
    ctx.set_property("spp", 1); ctx.set_property("max path length", 2)
    o.set_params(orc.params_from_ctx(ctx, ctx.get_constants()))
    ref = np.zeros((H, W, 3))
    for f in range(N):
        u.frame = 1000 + f
        o.process(u, threads=8)
        ref += o.irradiance()[..., :3]
    a, b = acc.mean() / N, ref.mean() / N
    assert b > 0 and abs(a - b) / b < 0.1, (a, b)
    res = o.restir_output(2)
    assert (res["M"] <= 1).all() and (res["flags"] <= 1).all()


@pytest.fixture(scope="module")
def gpu_ctx(mqlib):
    import mqhip
    ctx = mqhip.Context(0)
    yield ctx
    ctx.close()


def compare(ctx, o, tag):
    import mqhip
    a = ctx.read_output(mqhip.OUT_RESTIR_RESERVOIRS).view(np.uint32).reshape(-1, 16)
    b = o.restir_output(2).view(np.uint32).reshape(-1, 16)
    a, b = a.copy(), b.copy()
    a[:, 13] &= 0xffff; b[:, 13] &= 0xffff  # two padding bytes behind the half-precision radiance
    bad = (a != b).any(1)
    assert not bad.any(), "%s: %d reservoirs differ, first pixel %d: device %r oracle %r" % (tag, bad.sum(), np.argmax(bad), a[bad][0], b[bad][0])
    for which_g, which_o, name in ((mqhip.OUT_RESTIR_IRRADIANCE, 0, "irradiance"), (mqhip.OUT_RESTIR_MOMENTS, 1, "moments")):
        x = ctx.read_output(which_g).view(np.uint32); y = o.restir_output(which_o).view(np.uint32).reshape(-1)
        assert np.array_equal(x, y), "%s %s: %d values differ" % (tag, name, (x != y).sum())


CASES = [
    ("candidates only", "synth_start", {"restir: spp": 4}),
    ("temporal", "synth_start", {"restir: spp": 2, "restir: enable temporal reuse": 1, "restir: temporal bias correction": "basic", "restir: boiling filter strength": 0.4}),
    ("temporal raytraced + apply mv", "synth_materials", {"restir: spp": 1, "restir: enable temporal reuse": 1, "restir: temporal bias correction": "raytraced", "restir: apply mv": 1, "restir: temporal clamp m": 8}),
    ("spatial", "synth_start", {"restir: spp": 1, "restir: spatial reuse iterations": 3, "restir: spatital radius": 12, "restir: spatial bias correction": "basic"}),
    ("everything", "synth_sepulcher", {"restir: spp": 2, "restir: enable temporal reuse": 1, "restir: temporal bias correction": "basic", "restir: spatial reuse iterations": 2,
                                     "restir: spatial bias correction": "raytraced", "restir: shade visibility": 1, "restir: boiling filter strength": 0.2, "restir: temporal normal threshold": 0.5}),
]


@pytest.mark.gpu
@pytest.mark.parametrize("inline_rays", [0, 1], ids=["wavefront", "inline"])
@pytest.mark.parametrize("name,scene,props", CASES, ids=[c[0] for c in CASES])
def test_restir_matches_oracle(gpu_ctx, name, scene, props, inline_rays):
    """generate / temporal reuse / spatial reuse / shade over a moving camera: reservoirs (all 64 bytes), irradiance and
    moments bit-identical to the oracle in every frame; then the clear pass.  The generate and shade rays either go through
    the MCPG node's queues and traversal kernel (the default) or are traced inside the pass kernels ("inline restir rays")."""
    import mqhip
    ctx = gpu_ctx
    W, H = 150, 90  # partial tiles on both edges
    o = setup(ctx, scene, 2 if scene == "synth_sepulcher" else 5, {**props, "inline restir rays": inline_rays}, W, H)
    rp = orc.restir_params_from_ctx(ctx)
    lit = 0.0
    for f in (0, 1, 2, 3, 20):
        u = ctx.synth_camera(40 + f * 2)
        ctx.process(u); ctx.restir_process(u)
        o.process(u, threads=TH); o.restir_process(rp, u, threads=TH)
        compare(ctx, o, "%s frame %d" % (name, f))
        lit += o.restir_output(0)[..., :3].sum()
        res = o.restir_output(2)
        if "temporal" in name and f == 3:
            assert res["M"].max() > rp.spp  # history was merged
    assert lit > 0
    u = ctx.synth_camera(90)
    ctx.process(u, render=False); ctx.restir_process(u, render=False)
    o.process(u, render=False); o.restir_process(rp, u, render=False)
    compare(ctx, o, name + " clear")
    assert ctx.read_output(mqhip.OUT_RESTIR_IRRADIANCE).view(np.float32).sum() == 0


@pytest.mark.gpu
def test_config5_azad_4k_restir_plus_mcpg(gpu_ctx):
    """BASELINE config 5's frame: synth_azad(seed=4) 3840x2160, ReSTIR DI (temporal + spatial reuse) next to the MCPG
    pass (reference mode here, so that both are deterministic): both nodes' radiance images bit-identical to the oracle
    over two frames, and their sum -- what the graph's `add` node forms of the two -- is finite and lit."""
    import mqhip
    ctx = gpu_ctx
    W, H = 3840, 2160
    o = setup(ctx, "synth_azad", 4, {"spp": 1, "max path length": 3, "restir: spp": 1, "restir: enable temporal reuse": 1, "restir: spatial reuse iterations": 1}, W, H)
    rp = orc.restir_params_from_ctx(ctx)
    for f in (40, 41):
        u = ctx.synth_camera(f)
        ctx.process(u); ctx.restir_process(u)
        o.process(u, threads=TH); o.restir_process(rp, u, threads=TH)
        assert np.array_equal(ctx.irradiance().view(np.uint32), o.irradiance().view(np.uint32))
        compare(ctx, o, "config 5 frame %d" % f)
    total = ctx.irradiance()[..., :3] + ctx.image(mqhip.OUT_RESTIR_IRRADIANCE)[..., :3]
    assert np.isfinite(total).all() and total.sum() > 0


@pytest.mark.gpu
def test_config5_azad_4k_restir_plus_guided_mcpg(gpu_ctx):
    """BASELINE config 5 AS STATED: synth_azad(seed=4) 3840x2160, ReSTIR DI (temporal + spatial reuse) combined with the
    GUIDED MCPG pass.  The guided estimator is deterministic from a given learning state ("debug: freeze learning"): the oracle
    learns at 160x90 (its tables are addressed by world-space hash grids), both sides get that state at full size, and the
    next frames must be bit-identical in the MCPG radiance, the ReSTIR radiance, moments and all 64 bytes of every reservoir."""
    import mqhip
    from test_gpu_parity import _copy_learned_state
    ctx = gpu_ctx
    W, H = 3840, 2160
    props = {"reference mode": 0, "spp": 1, "max path length": 3, "restir: spp": 1, "restir: enable temporal reuse": 1, "restir: spatial reuse iterations": 1,
             "adaptive grid buf size": 1 << 18, "static grid buf size": 1 << 14, "LC buf size": 1 << 16}
    o = setup(ctx, "synth_azad", 4, props, 160, 90)
    for f in range(4):
        o.process(ctx.synth_camera(36 + f), threads=1)
    omc, olc = o.state(0).copy(), o.state(1).copy()
    assert (omc["sum_w"] > 0).sum() > 1000 and (olc["N"] > 0).sum() > 1000
    ctx.set_property("debug: freeze learning", 1)
    try:
        ctx.connect(W, H); o.connect(W, H)
        o.set_params(orc.params_from_ctx(ctx, ctx.get_constants()))
        rp = orc.restir_params_from_ctx(ctx)
        u = ctx.synth_camera(39)
        ctx.process(u); ctx.restir_process(u)            # the first frame after a connect zeroes the tables on both sides
        o.process(u, threads=TH); o.restir_process(rp, u, threads=TH)
        o.state(0)[:] = omc; o.state(1)[:] = olc
        _copy_learned_state(ctx, o)
        ctx.enable_counters(True)
        for f in (40, 41):
            u = ctx.synth_camera(f)
            o.counters(reset=True)
            ctx.process(u); ctx.restir_process(u)
            o.process(u, threads=TH); o.restir_process(rp, u, threads=TH)
            cg, co = ctx.counters(), o.counters()
            assert cg["guided_segments"] == co["guided_segments"] > 1000000 and cg["mc_state_reads"] == co["mc_state_reads"]
            bad = (ctx.irradiance().view(np.uint32) != o.irradiance().view(np.uint32)).any(-1)
            assert not bad.any(), "frame %d: %d MCPG pixels differ, first %r" % (f, bad.sum(), np.argwhere(bad)[0])
            compare(ctx, o, "config 5 guided, frame %d" % f)
        assert o.restir_output(2)["M"].max() > 1  # temporal + spatial reuse merged reservoirs
        total = ctx.irradiance()[..., :3] + ctx.image(mqhip.OUT_RESTIR_IRRADIANCE)[..., :3]
        assert np.isfinite(total).all() and total.sum() > 0
    finally:
        ctx.enable_counters(False)
        ctx.set_property("debug: freeze learning", 0)
    o.close()
