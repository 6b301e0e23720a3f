"""ctypes binding of the CPU oracle (oracle/libmqoracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("MQORACLE_LIB") or os.path.join(ROOT, "oracle", "libmqoracle.so")  # MQORACLE_LIB: the sanitizer build (tools/run_sanitized.sh)

(OUT_IRRADIANCE, OUT_GB_ALBEDO, OUT_GB_IRRADIANCE, OUT_GB_MV, OUT_GBUFFER, OUT_HITS, OUT_VOLUME, OUT_VOLUME_DEPTH, OUT_VOLUME_MV,
 OUT_DEBUG) = range(10)
(OP_EXP2, OP_LOG2, OP_SINCOS2PI, OP_POW, OP_F2H2F, OP_ENC_DEC_NORMAL, OP_BSDF_SAMPLE, OP_VMF_SAMPLE, OP_XORSHIFT,
 OP_PCG4D16, OP_SKY, OP_HASHGRID, OP_LDR_TO_HDR, OP_CAMERA, OP_DRAINE, OP_DISTANCE, OP_TEX_SAMPLE, OP_SKY_TEX, OP_TEX_GRAD) = range(19)
OP_ARITY = {0: (1, 1), 1: (1, 1), 2: (1, 2), 3: (2, 1), 4: (1, 1), 5: (3, 4), 6: (10, 5), 7: (6, 4), 8: (1, 4), 9: (4, 1),
            10: (3, 3), 11: (9, 2), 12: (3, 3), 13: (11, 5), 14: (7, 4), 15: (7, 4), 16: (3, 4), 17: (7, 3), 18: (7, 4)}


class Params(C.Structure):
    _fields_ = [("reference_mode", C.c_int32), ("adaptive_grid_type", C.c_int32), ("spp", C.c_int32),
                ("max_path_length", C.c_int32), ("use_light_cache_tail", C.c_int32), ("fov_tan_alpha_half", C.c_float),
                ("sun_w", C.c_float * 3), ("sun_color", C.c_float * 3), ("volume_spp", C.c_int32),
                ("volume_use_light_cache", C.c_int32), ("draine_g", C.c_float), ("draine_a", C.c_float),
                ("mc_samples", C.c_int32), ("mc_samples_adaptive_prob", C.c_float), ("distance_mc_samples", C.c_int32),
                ("mc_fast_recovery", C.c_int32), ("lc_grid_type", C.c_int32), ("lc_buffer_size", C.c_uint32),
                ("lc_grid_steps_per_unit_size", C.c_float), ("lc_grid_tan_alpha_half", C.c_float),
                ("lc_grid_min_width", C.c_float), ("lc_grid_power", C.c_float), ("mc_adaptive_buffer_size", C.c_uint32),
                ("mc_adaptive_grid_tan_alpha_half", C.c_float), ("mc_adaptive_grid_min_width", C.c_float),
                ("mc_adaptive_grid_power", C.c_float), ("mc_adaptive_grid_steps_per_unit_size", C.c_float),
                ("mc_static_buffer_size", C.c_uint32), ("mc_static_grid_width", C.c_float),
                ("distance_mc_grid_width", C.c_int32), ("volume_max_t", C.c_float), ("surf_bsdf_p", C.c_float),
                ("volume_phase_p", C.c_float), ("dir_guide_prior", C.c_float), ("dist_guide_p", C.c_float),
                ("distance_mc_vertex_state_count", C.c_uint32), ("seed", C.c_uint32), ("gbuffer_hide_sun", C.c_int32),
                ("quirk_lc_max_wo_p", C.c_int32), ("quirk_n16_wrap", C.c_int32), ("volume_forward_project", C.c_int32),
                ("enable_albedo_mipmap", C.c_int32), ("enable_emission_mipmap", C.c_int32),
                ("debug_output_connected", C.c_int32), ("debug_output_selector", C.c_int32), ("freeze_learning", C.c_int32),
                ("log_learning", C.c_int32)]


class RestirParams(C.Structure):
    _fields_ = [("spp", C.c_int32), ("seed", C.c_uint32), ("visibility_shade", C.c_int32), ("temporal_normal_reject_cos", C.c_float),
                ("temporal_depth_reject", C.c_float), ("spatial_normal_reject_cos", C.c_float), ("spatial_depth_reject", C.c_float),
                ("temporal_clamp_m", C.c_int32), ("spatial_radius", C.c_int32), ("temporal_bias_correction", C.c_int32),
                ("spatial_bias_correction", C.c_int32), ("boiling_filter_strength", C.c_float), ("spatial_reuse_iterations", C.c_int32),
                ("apply_mv", C.c_int32), ("temporal_reuse_enable", C.c_int32)]


RESERVOIR_DTYPE = np.dtype([("M", "<u4"), ("w", "<f4"), ("p_target", "<f4"), ("pos", "<f4", 3), ("normal", "<f4", 3), ("mv", "<f4", 3), ("T", "<f4"),
                            ("rad", "<u2", 3), ("pad", "<u2"), ("flags", "<u4")])
assert RESERVOIR_DTYPE.itemsize == 64


def restir_params_from_ctx(ctx):
    """the ReSTIR node's properties of a product context ("restir: ...") as the oracle's parameter block"""
    import math
    g = lambda k: ctx.get_property("restir: " + k)
    r = RestirParams()
    r.spp = int(g("spp")); r.seed = int(g("seed")); r.visibility_shade = int(g("shade visibility"))
    r.temporal_normal_reject_cos = float(np.float32(math.cos(g("temporal normal threshold")))); r.temporal_depth_reject = g("temporal depth threshold")
    r.spatial_normal_reject_cos = float(np.float32(math.cos(g("spatial normal threshold")))); r.spatial_depth_reject = g("spatial depth threshold")
    r.temporal_clamp_m = int(g("temporal clamp m")); r.spatial_radius = int(g("spatital radius"))
    r.temporal_bias_correction = int(g("temporal bias correction")); r.spatial_bias_correction = int(g("spatial bias correction"))
    r.boiling_filter_strength = g("boiling filter strength"); r.spatial_reuse_iterations = int(g("spatial reuse iterations"))
    r.apply_mv = int(g("apply mv")); r.temporal_reuse_enable = int(g("enable temporal reuse"))
    return r


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "nodes", "tris", "segments", "guided_segments", "lc_touches",
                                          "mc_updates_accepted", "mc_updates_dropped", "mc_state_reads")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(LIB)
        P = C.c_void_p
        l.orc_create.restype = P
        l.orc_create.argtypes = [C.POINTER(Params)]
        l.orc_destroy.argtypes = [P]
        l.orc_set_params.argtypes = [P, C.POINTER(Params)]
        l.orc_set_geometry.argtypes = [P, C.c_int, P, P, C.c_uint32, P, P, C.c_uint32, C.c_uint32]
        l.orc_set_texture.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32, P, C.c_uint32]
        l.orc_commit.argtypes = [P, C.c_int]
        l.orc_connect.argtypes = [P, C.c_uint32, C.c_uint32]
        l.orc_process.argtypes = [P, P, C.c_int, C.c_int]
        l.orc_process_mt.argtypes = [P, P, C.c_int, C.c_int, C.c_int]
        l.orc_output.restype = P
        l.orc_output.argtypes = [P, C.c_int, C.POINTER(C.c_size_t)]
        l.orc_get_counters.argtypes = [P, C.POINTER(Counters), C.c_int]
        l.orc_trace_rays.argtypes = [P, P, P, C.c_uint32, P, P, P]
        l.orc_math_eval.argtypes = [P, C.c_int, P, P, C.c_uint32]
        l.orc_debug_state.restype = C.c_void_p
        l.orc_debug_state.argtypes = [P, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        l.orc_learn_log.restype = C.c_void_p
        l.orc_learn_log.argtypes = [P, C.POINTER(C.c_size_t)]
        l.orc_learn_log_reset.argtypes = [P, C.c_size_t]
        l.orc_debug_apply_updates.argtypes = [P, P, C.c_size_t, P, P, C.c_size_t, C.POINTER(C.c_size_t)]
        l.orc_restir_process.argtypes = [P, C.POINTER(RestirParams), P, C.c_int, C.c_int]
        l.orc_restir_output.restype = C.c_void_p
        l.orc_restir_output.argtypes = [P, C.c_int, C.POINTER(C.c_size_t)]
        l.orc_post_set_params.argtypes = [P, C.c_int, P]
        l.orc_post_process.argtypes = [P]
        l.orc_post_set_add_restir.argtypes = [P, C.c_int]
        l.orc_post_clear.argtypes = [P]
        l.orc_post_output.restype = C.c_void_p
        l.orc_post_output.argtypes = [P, C.c_int, C.POINTER(C.c_size_t)]
        l.orc_params_header_defaults.argtypes = [C.POINTER(Params)]
        l.orc_params_json_defaults.argtypes = [C.POINTER(Params)]
        _lib = l
    return _lib


def header_params():
    p = Params()
    lib().orc_params_header_defaults(C.byref(p))
    return p


def json_params():
    p = Params()
    lib().orc_params_json_defaults(C.byref(p))
    return p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    def __init__(self, params=None):
        self.l = lib()
        self.params = params or header_params()
        self.h = C.c_void_p(self.l.orc_create(C.byref(self.params)))

    def close(self):
        if self.h:
            self.l.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, p):
        self.params = p
        self.l.orc_set_params(self.h, C.byref(p))

    def set_geometry(self, slot, vtx, prev_vtx, idx, ext, flags):
        vtx = np.ascontiguousarray(vtx, np.float32)
        prev = None if prev_vtx is None else np.ascontiguousarray(prev_vtx, np.float32)
        idx = np.ascontiguousarray(idx, np.uint32)
        ext = np.ascontiguousarray(ext)
        assert ext.dtype.itemsize == 28
        r = self.l.orc_set_geometry(self.h, slot, _ptr(vtx), _ptr(prev), vtx.size // 3, _ptr(idx), _ptr(ext), idx.size // 3, flags)
        assert r == 0, r

    def set_texture(self, texnum, rgba8, flags):
        rgba8 = np.ascontiguousarray(rgba8, np.uint8)
        h, w = rgba8.shape[:2]
        assert self.l.orc_set_texture(self.h, texnum, w, h, _ptr(rgba8), flags) == 0

    def commit(self, accel=1):
        assert self.l.orc_commit(self.h, accel) == 0

    def connect(self, w, h):
        assert self.l.orc_connect(self.h, w, h) == 0
        self.W, self.H = w, h

    def process(self, uniform, render=True, threads=1, parallel_update=False):
        """parallel_update: the update pass over the worker threads too (unordered, like the reference's dispatch): the CPU baseline only"""
        if parallel_update:
            assert self.l.orc_process_mt(self.h, C.addressof(uniform), 1 if render else 0, threads, 1) == 0
        else:
            assert self.l.orc_process(self.h, C.addressof(uniform), 1 if render else 0, threads) == 0

    def output(self, which):
        n = C.c_size_t()
        p = self.l.orc_output(self.h, which, C.byref(n))
        return np.frombuffer((C.c_char * n.value).from_address(p), dtype=np.uint8).copy()

    def irradiance(self):
        return self.output(OUT_IRRADIANCE).view(np.float32).reshape(self.H, self.W, 4)

    def volume(self):
        return self.output(OUT_VOLUME).view(np.float32).reshape(self.H, self.W, 4)

    def counters(self, reset=False):
        c = Counters()
        self.l.orc_get_counters(self.h, C.byref(c), 1 if reset else 0)
        return {n: int(getattr(c, n)) for n, _ in c._fields_}

    def trace_rays(self, org, direction):
        org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
        direction = np.ascontiguousarray(direction, np.float32).reshape(-1, 3)
        n = len(org)
        prim, t, uv = np.empty(n, np.uint32), np.empty(n, np.float32), np.empty((n, 2), np.float32)
        self.l.orc_trace_rays(self.h, _ptr(org), _ptr(direction), n, _ptr(prim), _ptr(t), _ptr(uv))
        return prim, t, uv

    MC_DTYPE = np.dtype({"names": ["id", "unused", "w_tgt", "sum_w", "w_cos", "mv", "T", "N", "hash"],
                         "formats": ["<u4", ("<f4", 3), ("<f4", 3), "<f4", "<f4", ("<u2", 3), "<f4", "<u2", "<u2"],
                         "offsets": [0, 4, 16, 28, 32, 36, 44, 48, 50], "itemsize": 52})
    LC_DTYPE = np.dtype([("hash", "<u4"), ("lock", "<u4"), ("irr", "<u2", 3), ("N", "<u2"), ("ok", "<u4"), ("cancel", "<u4")])

    def state(self, which):
        """Writable numpy view of the oracle's Markov-chain table (which = 0) or light cache (1)."""
        n, eb = C.c_size_t(), C.c_size_t()
        p = self.l.orc_debug_state(self.h, which, C.byref(n), C.byref(eb))
        dt = (self.MC_DTYPE, self.LC_DTYPE, np.dtype([("sum_w", "<f4"), ("N", "<u4"), ("m0", "<f4"), ("m1", "<f4")]))[which]
        assert p and eb.value == dt.itemsize, (eb.value, dt.itemsize)
        return np.frombuffer((C.c_char * (n.value * eb.value)).from_address(p), dtype=dt)

    (POST_ACCUM, POST_ACCUM_HISTORY, POST_VOLUME_ACCUM, POST_VOLUME_ACCUM_HISTORY, POST_FINAL) = range(5)

    def post_params_from_ctx(self, ctx):
        """the post chain's properties of a product context ("accum: ..." / "volume accum: ...")"""
        for k, prefix in enumerate(("accum: ", "volume accum: ")):
            six = np.array([ctx.get_property(prefix + n) for n in ("alpha", "max history", "normal threshold", "depth threshold", "enable motion vectors", "reuse border")], np.float32)
            assert self.l.orc_post_set_params(self.h, k, _ptr(six)) == 0
        self.l.orc_post_set_add_restir(self.h, int(ctx.get_property("add: restir irradiance")))

    def post_process(self):
        assert self.l.orc_post_process(self.h) == 0

    def post_clear(self):
        self.l.orc_post_clear(self.h)

    def post_output(self, which):
        n = C.c_size_t()
        p = self.l.orc_post_output(self.h, which, C.byref(n))
        a = np.frombuffer((C.c_char * n.value).from_address(p), dtype=np.float32).copy()
        return a.reshape(self.H, self.W, 4) if which in (0, 2, 4) else a.reshape(self.H, self.W)

    def restir_process(self, rparams, uniform, render=True, threads=1):
        assert self.l.orc_restir_process(self.h, C.byref(rparams), C.addressof(uniform), 1 if render else 0, threads) == 0

    def restir_output(self, which):
        """0 irradiance (H, W, 4) f32, 1 moments (H, W, 2) f32, 2 reservoirs (H * W) records"""
        n = C.c_size_t()
        p = self.l.orc_restir_output(self.h, which, C.byref(n))
        raw = np.frombuffer((C.c_char * n.value).from_address(p), dtype=np.uint8).copy()
        if which == 2:
            return raw.view(RESERVOIR_DTYPE)
        return raw.view(np.float32).reshape(self.H, self.W, 4 if which == 0 else 2)

    def learn_log_reset(self, capacity):
        assert self.l.orc_learn_log_reset(self.h, capacity) == 0

    def learn_log(self):
        """(n, 16) uint32 records of the learning-write log since the last reset (params.log_learning)."""
        n = C.c_size_t()
        p = self.l.orc_learn_log(self.h, C.byref(n))
        if not n.value:
            return np.zeros((0, 16), np.uint32)
        return np.frombuffer((C.c_char * (n.value * 64)).from_address(p), dtype=np.uint32).copy().reshape(-1, 16)

    def apply_updates(self, records, uniform, want_touches=False):
        """compute_updates.comp alone on the given (n, 16) uint32 records; returns the (slot, cell) touch pairs if asked."""
        r = np.ascontiguousarray(records, np.uint32).reshape(-1, 16)
        cap = 32 * len(r) + 16 if want_touches else 0
        t = np.zeros((cap, 2), np.uint32) if want_touches else None
        nt = C.c_size_t()
        rc = self.l.orc_debug_apply_updates(self.h, _ptr(r), len(r), C.addressof(uniform), _ptr(t), cap, C.byref(nt))
        assert rc == 0, rc
        if want_touches:
            assert nt.value <= cap
            return t[: nt.value]

    def math_eval(self, op, inp):
        inp = np.ascontiguousarray(inp, np.float32)
        n = inp.shape[0]
        out = np.empty((n, OP_ARITY[op][1]), np.float32)
        assert self.l.orc_math_eval(self.h, op, _ptr(inp), _ptr(out), n) == 0
        return out


def params_from_ctx(ctx, constants=None):
    """Build oracle params from a product context's property table (same keys, same values)."""
    p = header_params()
    g = ctx.get_property
    p.reference_mode = int(g("reference mode") or g("BSDF Prob") == 1.0)
    p.adaptive_grid_type = int(g("adaptive grid type")); p.spp = int(g("spp")); p.max_path_length = int(g("max path length"))
    p.use_light_cache_tail = int(g("surf: use LC")); p.mc_samples = int(g("mc samples"))
    p.mc_samples_adaptive_prob = g("adaptive grid prob"); p.mc_fast_recovery = int(g("mc fast recovery"))
    p.lc_grid_type = int(g("LC grid type")); p.lc_buffer_size = int(g("LC buf size"))
    p.lc_grid_steps_per_unit_size = g("LC grid steps per unit"); p.lc_grid_tan_alpha_half = g("LC grid tan(alpha/2)")
    p.lc_grid_min_width = g("LC grid min width"); p.lc_grid_power = g("LC grid power")
    p.mc_adaptive_buffer_size = int(g("adaptive grid buf size")); p.mc_adaptive_grid_tan_alpha_half = g("adaptive grid tan(alpha/2)")
    p.mc_adaptive_grid_min_width = g("adaptive grid min width"); p.mc_adaptive_grid_power = g("adaptive grid power")
    p.mc_adaptive_grid_steps_per_unit_size = g("adaptive grid steps per unit")
    p.mc_static_buffer_size = int(g("static grid buf size")); p.mc_static_grid_width = g("mc static width")
    p.surf_bsdf_p = g("BSDF Prob"); p.dir_guide_prior = g("ML Prior"); p.seed = int(g("seed"))
    p.volume_spp = int(g("volume spp")); p.volume_use_light_cache = int(g("volume: use LC")); p.distance_mc_samples = int(g("dist mc samples"))
    p.distance_mc_grid_width = int(g("dist mc grid width")); p.distance_mc_vertex_state_count = int(g("dist mc states per vertex"))
    p.volume_phase_p = g("Phase Prob"); p.dist_guide_p = g("dist guide p"); p.volume_forward_project = int(g("volume forward project"))
    p.enable_albedo_mipmap = int(g("enable albedo mipmap")); p.enable_emission_mipmap = int(g("enable emission mipmap"))
    p.freeze_learning = int(g("debug: freeze learning")); p.log_learning = int(g("debug: log learning writes"))
    p.debug_output_connected = int(g("debug output connected")); p.debug_output_selector = int(g("debug output"))
    import math
    d = g("particle size")
    import numpy as _np
    p.draine_g = float(_np.float32(math.exp(-2.20679 / (d + 3.91029) - 0.428934))); p.draine_a = float(_np.float32(math.exp(3.62489 - 8.29288 / (d + 5.52825))))
    p.gbuffer_hide_sun = int(g("hide sun")); p.quirk_lc_max_wo_p = int(g("quirk: LC max(wo_p,10)")); p.quirk_n16_wrap = int(g("quirk: 16-bit N*N"))
    if constants is not None:
        for k in range(3):
            p.sun_w[k] = constants["sun_direction"][k]
            p.sun_color[k] = constants["sun_color"][k]
        p.fov_tan_alpha_half = constants["fov_tan_alpha_half"]
        p.volume_max_t = constants["volume_max_t"]
    return p


def mirror_scene(ctx, oracle):
    """Copy geometry + textures held by a product context into the oracle (scene data is input, not algorithm)."""
    for slot in range(16):
        g = ctx.get_geometry(slot)
        if g is None:
            continue
        oracle.set_geometry(slot, g["vtx"], g["prev_vtx"], g["idx"], g["ext"], 1 if (g["flags"] & 1) else 0)
    for t in range(4096):
        tx = ctx.get_texture(t)
        if tx is not None:
            oracle.set_texture(t, tx[0], tx[1])
