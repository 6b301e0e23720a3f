"""Known-answer tests that PIN THE ORACLE (CPU, no GPU).

The reference ships no tests or golden vectors ("parity unpinned"), so the oracle is
pinned against analytic answers and independent re-derivations instead: elementary functions
against float64 numpy, codecs against numpy.float16, the RNG against a Python big-int
implementation, pdfs against numerical quadrature, the closest-hit search against brute force, and
the whole estimator against an analytic furnace-style expected value.
"""
import numpy as np
import pytest

import orc


@pytest.fixture(scope="module")
def o(built):
    return orc.Oracle()


def test_exp2_log2_pow_sincos_accuracy(o):
    rng = np.random.default_rng(1)
    x = (rng.random((50000, 1)) * 250 - 125).astype(np.float32)
    got = o.math_eval(orc.OP_EXP2, x)[:, 0].astype(np.float64)
    assert np.max(np.abs(got / np.exp2(x[:, 0].astype(np.float64)) - 1)) < 4e-7
    x = np.exp(rng.random((50000, 1)) * 160 - 80).astype(np.float32)
    got = o.math_eval(orc.OP_LOG2, x)[:, 0].astype(np.float64)
    assert np.max(np.abs(got - np.log2(x[:, 0].astype(np.float64)))) < 2e-5
    near1 = (1 + (rng.random((20000, 1)) - 0.5) * 1e-3).astype(np.float32)
    got = o.math_eval(orc.OP_LOG2, near1)[:, 0].astype(np.float64)
    ref = np.log2(near1[:, 0].astype(np.float64))
    assert np.max(np.abs(got - ref)) < 1e-9 + 4e-7 * np.max(np.abs(ref))
    xy = np.stack([rng.random(30000) * 3, rng.random(30000) * 3 - 1], 1).astype(np.float32)
    got = o.math_eval(orc.OP_POW, xy)[:, 0].astype(np.float64)
    ref = xy[:, 0].astype(np.float64) ** xy[:, 1].astype(np.float64)
    assert np.max(np.abs(got - ref) / np.maximum(ref, 1e-30)) < 3e-6
    assert o.math_eval(orc.OP_POW, np.array([[0.0, 0.83]], np.float32))[0, 0] == 0.0
    u = (rng.random((50000, 1)) * 6 - 3).astype(np.float32)
    cs = o.math_eval(orc.OP_SINCOS2PI, u).astype(np.float64)
    a = 2 * np.pi * u[:, 0].astype(np.float64)
    assert np.max(np.abs(cs[:, 0] - np.cos(a))) < 2e-6 and np.max(np.abs(cs[:, 1] - np.sin(a))) < 2e-6
    exact = o.math_eval(orc.OP_SINCOS2PI, np.array([[0.0], [0.25], [0.5], [0.75]], np.float32))
    assert np.array_equal(exact, np.array([[1, 0], [0, 1], [-1, 0], [0, -1]], np.float32))


def test_half_conversion_matches_ieee(o):
    rng = np.random.default_rng(2)
    bits = rng.integers(0, 2 ** 32, 400000, dtype=np.uint64).astype(np.uint32)
    special = np.array([0, 0x80000000, 0x33000000, 0x33000001, 0x33800000, 0x387fc000, 0x387fe000, 0x38800000, 0x477fe000, 0x477fefff,
                        0x477ff000, 0x47800000, 0x7f800000, 0xff800000, 0x3f800000, 0x3f801000, 0x3f803000, 0x00000001, 0x007fffff], np.uint32)
    x = np.concatenate([bits, special]).view(np.float32)
    scaled = np.concatenate([x, (rng.random(200000).astype(np.float32) * 2 - 1) * np.float32(70000), (rng.random(200000).astype(np.float32) * 2 - 1) * np.float32(1e-5)])
    scaled = scaled[~np.isnan(scaled)]
    got = o.math_eval(orc.OP_F2H2F, scaled.reshape(-1, 1))[:, 0]
    with np.errstate(over="ignore"):
        ref = scaled.astype(np.float16).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def _pcg4d16(v):
    M = 0xFFFFFFFF
    v = [(x * 1664525 + 1013904223) & M for x in v]
    def mix(v):
        v[0] = (v[0] + v[1] * v[3]) & M; v[1] = (v[1] + v[2] * v[0]) & M; v[2] = (v[2] + v[0] * v[1]) & M; v[3] = (v[3] + v[1] * v[2]) & M
    mix(v)
    v = [x ^ (x >> 16) for x in v]
    mix(v)
    return v[0] or 0x9E3779B9


def test_rng_streams_against_bigint_implementation(o):
    rng = np.random.default_rng(3)
    seeds = rng.integers(0, 2 ** 32, (200, 4), dtype=np.uint64).astype(np.uint32)
    got = o.math_eval(orc.OP_PCG4D16, seeds.view(np.float32))[:, 0].view(np.uint32)
    ref = np.array([_pcg4d16([int(a) for a in row]) for row in seeds], np.uint32)
    assert np.array_equal(got, ref)
    st = rng.integers(1, 2 ** 32, (200, 1), dtype=np.uint64).astype(np.uint32)
    got = o.math_eval(orc.OP_XORSHIFT, st.view(np.float32))
    for row, s in zip(got, st[:, 0]):
        s = int(s)
        for k in range(4):
            s ^= (s << 13) & 0xFFFFFFFF; s ^= s >> 17; s ^= (s << 5) & 0xFFFFFFFF
            assert row[k] == np.float32((s >> 8) / 16777216.0)
    assert got.min() >= 0.0 and got.max() < 1.0


def test_octahedral_normal_codec_round_trip(o):
    rng = np.random.default_rng(4)
    v = rng.normal(size=(100000, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], float)
    v = np.concatenate([v, axes]).astype(np.float32)
    out = o.math_eval(orc.OP_ENC_DEC_NORMAL, v)
    d = out[:, :3].astype(np.float64)
    assert np.max(np.abs(np.linalg.norm(d, axis=1) - 1)) < 1e-6
    assert np.max(np.linalg.norm(np.cross(d, v.astype(np.float64)), axis=1)) < 1e-4  # < 0.1 mrad for 2x16 bits
    assert np.array_equal(out[-6:, :3], axes.astype(np.float32))  # axis directions are exact
    again = o.math_eval(orc.OP_ENC_DEC_NORMAL, out[:, :3].copy())
    assert np.array_equal(again[:, 3].view(np.uint32), out[:, 3].view(np.uint32))  # decode -> encode is idempotent


def _sphere_quadrature(n_theta=400, n_phi=800):
    ct = (np.arange(n_theta) + 0.5) / n_theta * 2 - 1
    ph = (np.arange(n_phi) + 0.5) / n_phi * 2 * np.pi
    ct, ph = np.meshgrid(ct, ph, indexing="ij")
    st = np.sqrt(1 - ct ** 2)
    w = np.stack([st * np.cos(ph), st * np.sin(ph), ct], -1).reshape(-1, 3)
    return w, 4 * np.pi / len(w)


@pytest.mark.parametrize("kappa", [0.0, 0.5, 5.0, 80.0])
def test_vmf_pdf_normalised_and_sampler_consistent(o, kappa):
    mu = np.array([0.3, -0.5, 0.81]); mu /= np.linalg.norm(mu)
    w, dw = _sphere_quadrature()
    # pdf via the sampler entry point: feed w as "sample" by evaluating pdf of returned dir is awkward;
    # use VMF_SAMPLE's 4th output (pdf of the drawn sample) for sampler consistency and quadrature below
    n = 200000
    rng = np.random.default_rng(5)
    inp = np.concatenate([np.tile(mu, (n, 1)), np.full((n, 1), kappa), rng.random((n, 2))], 1).astype(np.float32)
    out = o.math_eval(orc.OP_VMF_SAMPLE, inp).astype(np.float64)
    s, pdf = out[:, :3], out[:, 3]
    assert np.max(np.abs(np.linalg.norm(s, axis=1) - 1)) < 1e-5
    mean_cos = (s @ mu).mean()
    expect = 0.0 if kappa == 0 else 1 / np.tanh(kappa) - 1 / kappa
    assert abs(mean_cos - expect) < 4e-3
    # E[1/pdf] over samples = area of the support = 4 pi (importance-sampling identity)
    if kappa <= 0.5:
        assert abs((1 / pdf).mean() / (4 * np.pi) - 1) < 0.02
    # closed form pdf at the sample
    c = 1 / (4 * np.pi) if kappa == 0 else kappa / (2 * np.pi * (1 - np.exp(-2 * kappa)))
    ref = c * np.exp(kappa * ((s @ mu) - 1))
    assert np.max(np.abs(pdf / ref - 1)) < 2e-4
    assert abs((c * np.exp(kappa * ((w @ mu) - 1))).sum() * dw - 1) < 2e-3  # the closed form integrates to 1


@pytest.mark.parametrize("rough", [0.1, 0.4, 0.6, 1.0])
def test_bsdf_sampling_pdf_and_energy(o, rough):
    """sample <-> pdf consistency, pdf integrates to <= 1, and the albedo-free BSDF conserves energy."""
    n_vec = np.array([0.0, 0.0, 1.0])
    for cos_i in (0.95, 0.5, 0.15):
        wi = np.array([np.sqrt(1 - cos_i ** 2), 0, -cos_i])  # travelling into the surface
        n = 400000
        rng = np.random.default_rng(6)
        inp = np.concatenate([np.tile(wi, (n, 1)), np.tile(n_vec, (n, 1)), np.full((n, 1), rough), rng.random((n, 3))], 1).astype(np.float32)
        out = o.math_eval(orc.OP_BSDF_SAMPLE, inp).astype(np.float64)
        wo, pdf, val = out[:, :3], out[:, 3], out[:, 4]
        up = wo[:, 2] > 1e-3
        assert up.mean() > 0.5
        # importance-sampling estimate of the albedo-free directional reflectance (white furnace): <= 1
        refl = np.where(up, val / np.maximum(pdf, 1e-30), 0.0).mean()
        assert 0.2 < refl <= 1.02, refl
        if rough >= 0.4:  # the same integral by quadrature of the closed form (independent of the sampler)
            wq, dw = _sphere_quadrature(500, 1000)
            upq = wq[wq[:, 2] > 1e-3]
            assert abs(refl / (_bsdf_value_f64(wi, upq, rough).sum() * dw) - 1) < 0.02
        # E[1_A / pdf] = area(A) for A = the cap z > 0.5 (area pi): pins pdf against the sampler
        cap = up & (wo[:, 2] > 0.5)
        area = np.where(cap, 1 / np.maximum(pdf, 1e-30), 0.0).mean()
        assert abs(area / np.pi - 1) < 0.05, (rough, cos_i, area)


def _bsdf_value_f64(wi, wo, rough):
    a = rough ** 2
    vv = -wi
    h = vv + wo; h /= np.linalg.norm(h, axis=1, keepdims=True)
    ndoth, vdoth, ndoto, ndotv = h[:, 2], h @ vv, wo[:, 2], vv[2]
    D = a * a / np.pi / (ndoth ** 2 * (a * a - 1) + 1) ** 2
    G1 = lambda c: 2 * c / (c + np.sqrt(a * a + (1 - a * a) * c * c))
    F = 0.02 + 0.98 * (1 - vdoth) ** 5
    Fv = 0.02 + 0.98 * (1 - ndotv) ** 5
    return ((1 - Fv) / np.pi + F * D * G1(ndotv) * G1(ndoto) / (4 * ndotv * ndoto)) * ndoto


def test_draine_phase_and_distance_samplers(o):
    """Draine phase function integrates to 1, its sampler follows it (mean cosine by quadrature), and
    the truncated-exponential / Gaussian distance samplers match their pdfs."""
    g, a = 0.532, 14.6   # particle size 7 um (render_mcpg.cpp:134-135)
    w, dw = _sphere_quadrature()
    wi = np.array([0.0, 0.0, 1.0])
    cos_t = w @ wi
    g2 = g * g
    p = (1 / (4 * np.pi)) * (1 - g2) / (1 + g2 - 2 * g * cos_t) ** 1.5 * (1 + a * cos_t ** 2) / (1 + a * (1 + 2 * g2) / 3)
    assert abs(p.sum() * dw - 1) < 2e-3
    n = 200000
    rng = np.random.default_rng(21)
    x = np.concatenate([np.tile(wi, (n, 1)), np.full((n, 1), g), np.full((n, 1), a), rng.random((n, 2))], 1).astype(np.float32)
    out = o.math_eval(orc.OP_DRAINE, x).astype(np.float64)
    assert np.max(np.abs(np.linalg.norm(out[:, :3], axis=1) - 1)) < 1e-5
    assert abs(out[:, 2].mean() - (p * cos_t).sum() * dw) < 5e-3            # sampler mean cosine == pdf mean cosine
    ct = out[:, 2]
    ref = (1 / (4 * np.pi)) * (1 - g2) / (1 + g2 - 2 * g * ct) ** 1.5 * (1 + a * ct ** 2) / (1 + a * (1 + 2 * g2) / 3)
    assert np.max(np.abs(out[:, 3] / ref - 1)) < 1e-4                        # returned pdf == closed form at the sample
    hist, edges = np.histogram(ct, bins=20, range=(-1, 1))
    centers = 0.5 * (edges[1:] + edges[:-1])
    mass = np.array([p[(cos_t >= lo) & (cos_t < hi)].sum() * dw for lo, hi in zip(edges[:-1], edges[1:])])
    assert np.max(np.abs(hist / n - mass)) < 4e-3                              # histogram follows the pdf
    # distance: truncated exponential and Gaussian
    mu_t, tmax = 2e-3, 800.0
    y = np.zeros((n, 7), np.float32); y[:, 0] = mu_t; y[:, 1] = tmax; y[:, 2] = rng.random(n); y[:, 3] = 300; y[:, 4] = 40; y[:, 5:7] = rng.random((n, 2))
    d = o.math_eval(orc.OP_DISTANCE, y).astype(np.float64)
    assert d[:, 0].min() >= 0 and d[:, 0].max() <= tmax * (1 + 1e-5)
    xi_max = 1 - np.exp(-mu_t * tmax)
    expect_mean = (1 / mu_t - (tmax + 1 / mu_t) * np.exp(-mu_t * tmax)) / xi_max
    assert abs(d[:, 0].mean() / expect_mean - 1) < 0.01
    assert np.max(np.abs(d[:, 1] / (mu_t * np.exp(-mu_t * d[:, 0]) / xi_max) - 1)) < 1e-4
    assert abs(d[:, 2].mean() - 300) < 0.5 and abs(d[:, 2].std() - 40) < 0.5
    assert np.max(np.abs(d[:, 3] / (np.exp(-0.5 * ((d[:, 2] - 300) / 40) ** 2) / (40 * np.sqrt(2 * np.pi))) - 1)) < 1e-3


def test_hash_grid_level_width_inverse(o):
    """mc.glsl:65,73: width(level(d)) stays within one level step of the target width."""
    p = orc.json_params()
    steps, power, minw, tan = p.mc_adaptive_grid_steps_per_unit_size, p.mc_adaptive_grid_power, p.mc_adaptive_grid_min_width, p.mc_adaptive_grid_tan_alpha_half
    for dist in (1.0, 10.0, 100.0, 1000.0, 5000.0):
        w = 2 * tan * dist
        level = round(steps * np.log(max(w, minw) / minw) / np.log(power))
        width = minw * power ** (level / steps)
        assert power ** (-0.5 / steps) * 0.999 <= width / max(w, minw) <= power ** (0.5 / steps) * 1.001
    # index is deterministic, in range, sensitive to normal face and level; checksum independent of the index
    rng = np.random.default_rng(7)
    n = 20000
    size = 32777259
    x = np.zeros((n, 9), np.float32)
    x[:, :3] = rng.random((n, 3)) * 4000 - 2000
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    x[:, 3:6] = nrm; x[:, 6] = rng.integers(0, 20, n); x[:, 7] = 2.5
    x[:, 8] = np.full(n, size, np.uint32).view(np.float32)
    a = o.math_eval(orc.OP_HASHGRID, x).view(np.uint32)
    assert a[:, 0].max() < size
    assert np.array_equal(a, o.math_eval(orc.OP_HASHGRID, x).view(np.uint32))
    y = x.copy(); y[:, 6] += 1
    b = o.math_eval(orc.OP_HASHGRID, y).view(np.uint32)
    assert (a[:, 0] != b[:, 0]).mean() > 0.999 and (a[:, 1] != b[:, 1]).mean() > 0.999
    # occupancy is uniform: chi-square over 64 buckets
    hist = np.bincount(a[:, 0] % 64, minlength=64)
    assert ((hist - n / 64) ** 2 / (n / 64)).sum() < 130


def test_camera_round_trip(o):
    rng = np.random.default_rng(8)
    n = 5000
    fwd = rng.normal(size=(n, 3)); fwd /= np.linalg.norm(fwd, axis=1, keepdims=True)
    tmp = rng.normal(size=(n, 3)); up = np.cross(np.cross(fwd, tmp), fwd); up /= np.linalg.norm(up, axis=1, keepdims=True)
    x = np.zeros((n, 11), np.float32)
    x[:, 0] = rng.integers(0, 1920, n); x[:, 1] = rng.integers(0, 1080, n); x[:, 2] = 1920; x[:, 3] = 1080
    x[:, 4:7] = fwd; x[:, 7:10] = up; x[:, 10] = 1.0
    out = o.math_eval(orc.OP_CAMERA, x)
    assert np.max(np.abs(out[:, 3] - x[:, 0])) < 2e-2 and np.max(np.abs(out[:, 4] - x[:, 1])) < 2e-2
    centre = x[:1].copy(); centre[0, 0] = 959.5; centre[0, 1] = 539.5
    d = o.math_eval(orc.OP_CAMERA, centre)[0, :3]
    assert np.allclose(d, centre[0, 4:7], atol=1e-6)          # image centre looks along `forward`
    edge = x[:1].copy(); edge[0, 0] = 1919.5; edge[0, 1] = 539.5  # right edge: tan(fov_x/2) = 1 -> 45 degrees
    d = o.math_eval(orc.OP_CAMERA, edge)[0, :3].astype(np.float64)
    assert abs(np.dot(d, edge[0, 4:7]) - np.cos(np.pi / 4)) < 1e-6


def _host_ctx():
    import mqhip
    return mqhip.Context(-1)


def test_oracle_bvh_equals_brute_force(built):
    ctx = _host_ctx()
    ctx.synth_scene("synth_tiny", 7)
    orc_ = orc.Oracle(orc.params_from_ctx(ctx, ctx.get_constants()))
    orc.mirror_scene(ctx, orc_)
    g = ctx.get_geometry(0)
    lo, hi = g["vtx"].min(0), g["vtx"].max(0)
    rng = np.random.default_rng(9)
    org = (lo + (hi - lo) * rng.random((20000, 3))).astype(np.float32)
    d = rng.normal(size=(20000, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    orc_.commit(0); a = orc_.trace_rays(org, d)
    orc_.commit(1); b = orc_.trace_rays(org, d)
    assert (a[0] != 0xFFFFFFFF).mean() > 0.5
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    # back-face culling: the hit triangle always faces the ray (raytrace.glsl:73,85)
    hit = a[0] != 0xFFFFFFFF
    prim = a[0][hit]
    geos = {s: ctx.get_geometry(s) for s in range(3)}
    for s in range(3):
        m = (prim >> 28) == s
        if not m.any():
            continue
        tri = geos[s]["vtx"][geos[s]["idx"][prim[m] & 0x0FFFFFFF]].astype(np.float64)
        nrm = np.cross(tri[:, 2] - tri[:, 0], tri[:, 1] - tri[:, 0])
        assert ((nrm * d[hit][m]).sum(1) < 0).all()


def test_estimator_matches_analytic_expected_value(built):
    """One diffuse-ish floor quad under a uniform emissive sky.  The surface estimator's expected
    irradiance is L_sky * R(theta_view) with R the albedo-free directional reflectance of the BSDF;
    R is integrated here by quadrature of the oracle's own BSDF value (independent of its sampler)."""
    import mqhip
    p = orc.header_params()
    p.reference_mode = 1; p.spp = 8; p.max_path_length = 2; p.seed = 77
    p.mc_adaptive_buffer_size = 1 << 12; p.mc_static_buffer_size = 1 << 10; p.lc_buffer_size = 1 << 10
    for k in range(3):
        p.sun_color[k] = 0.0
    orc_ = orc.Oracle(p)
    vtx = np.array([[-1000, -1000, 0], [1000, -1000, 0], [1000, 1000, 0], [-1000, 1000, 0]], np.float32)
    idx = np.array([[0, 2, 1], [0, 3, 2]], np.uint32)  # cross(v2-v0, v1-v0) = +z
    ext = np.zeros(2, mqhip.EXT_DTYPE)
    ext["texnum_alpha"] = 1 | (15 << 12); ext["n1_brush"] = 0xFFFFFFFF
    orc_.set_geometry(0, vtx, None, idx, ext, 1)
    orc_.set_texture(1, np.full((4, 4, 4), 255, np.uint8), 0)            # white albedo
    sky_v = 0.5
    orc_.set_texture(2, np.full((4, 4, 4), int(sky_v * 255), np.uint8), 0)  # uniform classic sky (back layer)
    sky_front = np.zeros((4, 4, 4), np.uint8)                               # fully transparent front layer
    orc_.set_texture(3, sky_front, 0)
    orc_.commit(0)
    W, H = 32, 24
    orc_.connect(W, H)
    u = mqhip.Uniform()
    u.cam_x[:] = [0, 0, 100, 0]; u.cam_w[:] = [0.6, 0, -0.8, 1 / 60]; u.cam_u[:] = [0.8, 0, 0.6, 0]
    for k in range(4):
        u.prev_cam_x[k], u.prev_cam_w[k], u.prev_cam_u[k] = u.cam_x[k], u.cam_w[k], u.cam_u[k]
    u.prev_cam_x[3] = u.prev_cam_w[3] = u.prev_cam_u[3] = 0
    u.sky_rt_bk = 2 | (3 << 16); u.sky_lf_ft = 0xFFFF; u.sky_up_dn = 0xFFFFFFFF
    acc = np.zeros((H, W, 3))
    N = 40
    for f in range(N):
        u.frame = f
        orc_.process(u, threads=4)
        acc += orc_.irradiance()[..., :3]
    acc /= N
    v = np.float32(int(sky_v * 255)) / np.float32(255)
    L = float(np.float16(10 * (2.0 ** (3.5 * float(np.float16(v))) - 1)))
    # expected value at the centre pixel: view direction = cam_w
    wi = np.array([0.6, 0, -0.8])
    wq, dw = _sphere_quadrature(200, 400)
    upq = wq[wq[:, 2] > 1e-3]
    R = _bsdf_value_f64(wi, upq, 0.6).sum() * dw
    centre = acc[H // 2 - 2:H // 2 + 2, W // 2 - 2:W // 2 + 2].mean()
    assert abs(centre / (L * R) - 1) < 0.03, (centre, L * R)
