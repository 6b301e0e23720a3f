"""ctypes binding of libmqhip.so (include/mq.h) -- plumbing for tests and bench.py.

The product is the C ABI; this file adds nothing but argument marshalling.  It mirrors the
reference's node lifecycle (describe -> connect -> process -> properties,
src/render_mcpg/render_mcpg.hpp:36-49) one call per method.  There is no CPU rendering path: if
the HIP extension cannot be loaded, import fails loudly.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmqhip.so")

MQ_MAX_GLTEXTURES = 4096
MQ_MAX_GEOMETRIES = 16
MQ_GEO_OPAQUE, MQ_GEO_STATIC = 1, 2
MQ_TEX_SRGB, MQ_TEX_LINEAR = 1, 2
(OUT_IRRADIANCE, OUT_GB_ALBEDO, OUT_GB_IRRADIANCE, OUT_GB_MV, OUT_GBUFFER, OUT_HITS, OUT_TILES, OUT_VOLUME, OUT_VOLUME_DEPTH,
 OUT_VOLUME_MV, OUT_VOLUME_TILES, OUT_DEBUG, OUT_ACCUM, OUT_ACCUM_HISTORY, OUT_VOLUME_ACCUM, OUT_VOLUME_ACCUM_HISTORY, OUT_FINAL,
 OUT_RESTIR_IRRADIANCE, OUT_RESTIR_MOMENTS, OUT_RESTIR_RESERVOIRS, OUT_VOLUME_DEPTH_TILES, OUT_COUNT) = range(22)
MQ_ENODEVICE = -2
HALO_RESTIR_RESERVOIRS, HALO_ACCUM, HALO_ACCUM_HISTORY, HALO_VOLUME_ACCUM, HALO_VOLUME_ACCUM_HISTORY, HALO_COUNT = range(6)

EXT_DTYPE = np.dtype([("texnum_alpha", "<u2"), ("texnum_fb_flags", "<u2"), ("n0_gloss_norm", "<u4"),
                      ("n1_brush", "<u4"), ("n2", "<u4"), ("st", "<u2", (6,))])
assert EXT_DTYPE.itemsize == 28
NODE_DTYPE = np.dtype([("p", "<f4", (3,)), ("e", "u1", (3,)), ("imask", "u1"), ("child_base", "<u4"), ("tri_base", "<u4"),
                       ("meta", "u1", (8,)), ("qlo", "u1", (3, 8)), ("qhi", "u1", (3, 8))])
TRI_DTYPE = np.dtype([("v", "<f4", (3, 3)), ("key", "<u4"), ("flags", "<u4"), ("pad", "<u4")])
LEAF_DTYPE = np.dtype([("v", "<f4", (4, 3)), ("key0", "<u4"), ("key1", "<u4"), ("tri0", "<u4"), ("sel", "<u4")])  # MqLeafRec: one or two triangles sharing an edge
assert NODE_DTYPE.itemsize == 80 and TRI_DTYPE.itemsize == 48 and LEAF_DTYPE.itemsize == 64


class Uniform(C.Structure):
    _fields_ = [("cam_x", C.c_float * 4), ("cam_w", C.c_float * 4), ("cam_u", C.c_float * 4),
                ("prev_cam_x", C.c_float * 4), ("prev_cam_w", C.c_float * 4), ("prev_cam_u", C.c_float * 4),
                ("sky_rt_bk", C.c_uint32), ("sky_lf_ft", C.c_uint32), ("sky_up_dn", C.c_uint32),
                ("cl_time", C.c_float), ("frame", C.c_uint32), ("player", C.c_uint32), ("rt_config", C.c_uint32)]


assert C.sizeof(Uniform) == 124


class Band(C.Structure):
    """mq_band: the pixel rows of one rank of a row partition (owned / + spatial radius / + reprojection halo)"""
    _fields_ = [(n, C.c_uint32) for n in ("row_begin", "row_end", "reuse_begin", "reuse_end", "need_begin", "need_end")]


class Constants(C.Structure):
    _fields_ = [("sun_color", C.c_float * 3), ("sun_direction", C.c_float * 3), ("fov", C.c_float),
                ("fov_tan_alpha_half", C.c_float), ("volume_max_t", C.c_float)]


class IoDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("bytes", C.c_size_t * OUT_COUNT),
                ("bytes_per_pixel", C.c_uint32 * OUT_COUNT), ("state_bytes_markovchain", C.c_size_t),
                ("state_bytes_lightcache", C.c_size_t), ("state_bytes_update_queue", C.c_size_t), ("state_bytes_volume_distancemc", C.c_size_t)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "nodes", "tris", "segments", "guided_segments", "lc_touches",
                                          "mc_updates_accepted", "mc_updates_dropped", "mc_state_reads", "pixels", "queue_rays", "queue_nodes", "queue_tris", "queue_overflow")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class View(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("forward", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3)]


PARTICLE_DTYPE = np.dtype([("org", "<f4", 3), ("prev_org", "<f4", 3), ("vel", "<f4", 3), ("color_rgba", "<u4"), ("type", "<i4"), ("seed", "<u4")])


class AliasInstance(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("angles", C.c_float * 3), ("prev_origin", C.c_float * 3), ("prev_angles", C.c_float * 3),
                ("pose1", C.c_int32), ("pose2", C.c_int32), ("blend", C.c_float), ("prev_blend", C.c_float), ("skin", C.c_int32), ("fovscale", C.c_float)]


class SpriteInstance(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("prev_origin", C.c_float * 3), ("angles", C.c_float * 3), ("scale", C.c_float), ("frame", C.c_int32)]


class FrameState(C.Structure):
    _fields_ = [("vieworg", C.c_float * 3), ("viewangles", C.c_float * 3), ("cl_time", C.c_double), ("frame", C.c_uint32), ("render", C.c_int32),
                ("has_player", C.c_int32), ("weapon", C.c_int32), ("waterlevel", C.c_int32), ("sky_mode", C.c_int32), ("sky", C.c_uint16 * 6), ("notexture", C.c_uint16),
                ("mu_overwrite", C.c_int32), ("mu_t", C.c_float), ("mu_s_div_mu_t", C.c_float * 3), ("fog_density", C.c_float), ("fog_color", C.c_float * 3)]


class MqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mq error %d: %s" % (code, msg))
        self.code = code


def load_library(path=None):
    path = path or os.environ.get("MQHIP_LIB") or LIB_PATH  # MQHIP_LIB: A/B builds of the same library
    if not os.path.exists(path):
        raise ImportError("libmqhip.so not built (%s): run __graft_entry__.build() or `make -C merian-quake_amd`" % path)
    # PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  A process must end up with
    # ONE HIP/HSA runtime, so when torch is part of the process (bench.py: RCCL via torch.distributed)
    # it has to be loaded first; libmqhip then binds to the already-loaded runtime.
    if "torch" not in sys.modules and os.environ.get("MQHIP_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(path)
    P, u32, i32, f32p, u32p, vp, sz = C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t
    sigs = {
        "mq_abi_version": (i32, []),
        "mq_create": (i32, [C.POINTER(P), i32]),
        "mq_destroy": (None, [P]),
        "mq_last_error": (C.c_char_p, [P]),
        "mq_set_property": (i32, [P, C.c_char_p, C.c_double]),
        "mq_set_property_str": (i32, [P, C.c_char_p, C.c_char_p]),
        "mq_get_property": (i32, [P, C.c_char_p, C.POINTER(C.c_double)]),
        "mq_property_count": (i32, []),
        "mq_property_name": (C.c_char_p, [i32]),
        "mq_property_type": (i32, [i32]),
        "mq_property_option": (C.c_char_p, [i32, i32]),
        "mq_load_properties_json": (i32, [P, C.c_char_p, C.c_char_p]),
        "mq_properties_header_defaults": (None, [P]),
        "mq_properties_json_defaults": (None, [P]),
        "mq_scene_set_geometry": (i32, [P, i32, vp, vp, u32, vp, vp, u32, u32]),
        "mq_scene_set_texture": (i32, [P, u32, u32, u32, vp, u32]),
        "mq_scene_commit": (i32, [P]),
        "mq_set_constants": (i32, [P, C.POINTER(Constants)]),
        "mq_get_constants": (i32, [P, C.POINTER(Constants)]),
        "mq_scene_get_geometry": (i32, [P, i32, C.POINTER(vp), C.POINTER(vp), u32p, C.POINTER(vp), C.POINTER(vp), u32p, u32p]),
        "mq_scene_get_texture": (i32, [P, u32, u32p, u32p, C.POINTER(vp), u32p]),
        "mq_scene_get_bvh": (i32, [P, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(vp), C.POINTER(C.c_uint64)]),
        "mq_timing_set_interval": (i32, [P, C.c_uint32]),
        "mq_timing_detail_frames": (i32, [P, C.POINTER(C.c_uint32)]),
        "mq_scene_layout": (i32, [P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "mq_scene_commit_counts": (i32, [P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "mq_scene_commit_async_count": (i32, [P, C.POINTER(C.c_uint32)]),
        "mq_scene_commit_device_count": (i32, [P, C.POINTER(C.c_uint32)]),
        "mq_scene_get_leaves": (i32, [P, C.POINTER(vp), C.POINTER(C.c_uint64)]),
        "mq_scene_stats": (i32, [P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), f32p]),
        "mq_describe": (i32, [P, u32, u32, C.POINTER(IoDesc)]),
        "mq_connect": (i32, [P, u32, u32]),
        "mq_process": (i32, [P, C.POINTER(Uniform), i32, vp]),
        "mq_sync": (i32, [P]),
        "mq_map_output": (i32, [P, i32, C.POINTER(vp), C.POINTER(sz)]),
        "mq_read_output": (i32, [P, i32, vp, sz]),
        "mq_last_frame_ms": (i32, [P, f32p, f32p, f32p]),
        "mq_timing_reset": (i32, [P]),
        "mq_timing_get": (i32, [P, u32p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "mq_timing_get_detail": (i32, [P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "mq_timing_get_rounds": (i32, [P, C.POINTER(C.c_double), C.POINTER(C.c_double), i32]),
        "mq_measure_stream_read": (i32, [P, sz, i32, C.POINTER(C.c_double)]),
        "mq_enable_counters": (i32, [P, i32]),
        "mq_get_counters": (i32, [P, C.POINTER(Counters)]),
        "mq_reset_state": (i32, [P]),
        "mq_debug_state_read": (i32, [P, i32, vp, sz]),
        "mq_debug_state_write": (i32, [P, i32, vp, sz]),
        "mq_debug_section_clocks": (i32, [P, C.POINTER(C.c_uint64), i32, i32]),
        "mq_dyn_begin": (i32, [P]),
        "mq_dyn_add_particles": (i32, [P, vp, u32, C.POINTER(View), u32, u32, C.c_double, C.c_double]),
        "mq_dyn_add_alias": (i32, [P, i32, C.POINTER(AliasInstance)]),
        "mq_dyn_add_alias_batch": (i32, [P, C.POINTER(C.c_int), C.POINTER(AliasInstance), C.c_uint32]),
        "mq_dyn_add_sprite": (i32, [P, i32, C.POINTER(SpriteInstance), C.POINTER(View)]),
        "mq_dyn_add_brush_model": (i32, [P, i32, f32p, f32p, f32p, f32p]),
        "mq_dyn_end": (i32, [P, i32]),
        "mq_bsp_model_count": (i32, [P]),
        "mq_load_mdl": (i32, [P, C.c_char_p, C.c_char_p, u32, C.POINTER(i32), u32p]),
        "mq_load_spr": (i32, [P, C.c_char_p, C.c_char_p, u32, C.POINTER(i32), u32p]),
        "mq_uniform_update": (i32, [C.POINTER(Uniform), C.POINTER(FrameState)]),
        "mq_constants_fov": (i32, [C.POINTER(Constants), C.c_float]),
        "mq_post_process": (i32, [P, vp]),
        "mq_restir_process": (i32, [P, C.POINTER(Uniform), i32, vp]),
        "mq_post_clear": (i32, [P]),
        "mq_debug_learn_log_read": (i32, [P, vp, sz, C.POINTER(sz)]),
        "mq_debug_apply_updates": (i32, [P, vp, u32, C.POINTER(Uniform)]),
        "mq_set_partition": (i32, [P, i32, i32]),
        "mq_tiles_per_rank": (i32, [P, u32p, C.POINTER(sz)]),
        "mq_untile": (i32, [P, vp, vp]),
        "mq_untile_volume": (i32, [P, vp, vp]),
        "mq_untile_volume_depth": (i32, [P, vp, vp]),
        "mq_untile_to": (i32, [P, vp, vp, vp]),
        "mq_band_layout": (i32, [P, u32, u32, i32, i32, C.POINTER(Band)]),
        "mq_band_gbuffer": (i32, [P, C.POINTER(Uniform), vp]),
        "mq_map_halo": (i32, [P, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz)]),
        "mq_trace_rays": (i32, [P, vp, vp, u32, vp, vp, vp]),
        "mq_math_eval": (i32, [P, i32, vp, vp, u32]),
        "mq_synth_scene": (i32, [P, C.c_char_p, u32]),
        "mq_synth_camera": (i32, [P, u32, C.POINTER(Uniform)]),
        "mq_load_bsp": (i32, [P, C.c_char_p, C.c_char_p]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    lib._mq_symbols = sorted(sigs)
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One render node instance (GBuffer + Renderer (MCPG)) on one HIP device, or host-only (device=-1)."""

    def __init__(self, device=0, lib=None):
        self.lib = lib or load_library()
        self.h = C.c_void_p()
        r = self.lib.mq_create(C.byref(self.h), device)
        if r != 0:
            raise MqError(r, "mq_create(device=%d) failed%s" % (device, " (no HIP device)" if r == MQ_ENODEVICE else ""))
        self.device = device

    def close(self):
        if self.h:
            self.lib.mq_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r):
        if r < 0:
            raise MqError(r, (self.lib.mq_last_error(self.h) or b"").decode())
        return r

    # -- properties()
    def set_property(self, key, value):
        if isinstance(value, str):
            return self._chk(self.lib.mq_set_property_str(self.h, key.encode(), value.encode()))
        return self._chk(self.lib.mq_set_property(self.h, key.encode(), float(value)))

    def get_property(self, key):
        v = C.c_double()
        self._chk(self.lib.mq_get_property(self.h, key.encode(), C.byref(v)))
        return v.value

    def property_names(self):
        return [self.lib.mq_property_name(i).decode() for i in range(self.lib.mq_property_count())]

    def load_properties_json(self, text, node):
        return self._chk(self.lib.mq_load_properties_json(self.h, text.encode(), node.encode()))

    def header_defaults(self):
        self.lib.mq_properties_header_defaults(self.h)

    def json_defaults(self):
        self.lib.mq_properties_json_defaults(self.h)

    # -- scene
    def set_geometry(self, slot, vtx, prev_vtx, idx, ext, flags):
        vtx = np.ascontiguousarray(vtx, np.float32).reshape(-1, 3)
        prev = None if prev_vtx is None else np.ascontiguousarray(prev_vtx, np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
        ext = np.ascontiguousarray(ext, EXT_DTYPE)
        assert len(ext) == len(idx)
        self._chk(self.lib.mq_scene_set_geometry(self.h, slot, _ptr(vtx), _ptr(prev), len(vtx), _ptr(idx), _ptr(ext), len(idx), flags))

    def set_texture(self, texnum, rgba8, flags):
        rgba8 = np.ascontiguousarray(rgba8, np.uint8)
        h, w = rgba8.shape[:2]
        self._chk(self.lib.mq_scene_set_texture(self.h, texnum, w, h, _ptr(rgba8), flags))

    def commit(self):
        self._chk(self.lib.mq_scene_commit(self.h))

    def scene_layout(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.mq_scene_layout(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def commit_counts(self):
        a, b = C.c_uint32(), C.c_uint32()
        self._chk(self.lib.mq_scene_commit_counts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def commit_async_count(self):
        a = C.c_uint32()
        self._chk(self.lib.mq_scene_commit_async_count(self.h, C.byref(a)))
        return a.value

    def commit_device_count(self):
        a = C.c_uint32()
        self._chk(self.lib.mq_scene_commit_device_count(self.h, C.byref(a)))
        return a.value

    def set_constants(self, sun_color, sun_direction, fov=90.0, fov_tan_alpha_half=1.0, volume_max_t=1000.0):
        c = Constants((C.c_float * 3)(*sun_color), (C.c_float * 3)(*sun_direction), fov, fov_tan_alpha_half, volume_max_t)
        self._chk(self.lib.mq_set_constants(self.h, C.byref(c)))

    def get_constants(self):
        c = Constants()
        self._chk(self.lib.mq_get_constants(self.h, C.byref(c)))
        return dict(sun_color=list(c.sun_color), sun_direction=list(c.sun_direction), fov=c.fov,
                    fov_tan_alpha_half=c.fov_tan_alpha_half, volume_max_t=c.volume_max_t)

    def get_geometry(self, slot):
        vtx, prev, idx, ext = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        nv, nt, fl = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self.lib.mq_scene_get_geometry(self.h, slot, C.byref(vtx), C.byref(prev), C.byref(nv), C.byref(idx), C.byref(ext), C.byref(nt), C.byref(fl)))
        if nt.value == 0:
            return None

        def arr(p, n, dt):
            return np.frombuffer((C.c_char * (n * np.dtype(dt).itemsize)).from_address(p.value), dtype=dt).copy()
        return dict(vtx=arr(vtx, nv.value * 3, np.float32).reshape(-1, 3), prev_vtx=arr(prev, nv.value * 3, np.float32).reshape(-1, 3),
                    idx=arr(idx, nt.value * 3, np.uint32).reshape(-1, 3), ext=arr(ext, nt.value, EXT_DTYPE), flags=fl.value)

    def get_texture(self, texnum):
        w, h, fl, p = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_void_p()
        self._chk(self.lib.mq_scene_get_texture(self.h, texnum, C.byref(w), C.byref(h), C.byref(p), C.byref(fl)))
        if not p.value:
            return None
        px = np.frombuffer((C.c_char * (w.value * h.value * 4)).from_address(p.value), dtype=np.uint8).copy().reshape(h.value, w.value, 4)
        return px, fl.value

    def get_bvh(self):
        n, t = C.c_void_p(), C.c_void_p()
        nn, nt = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.mq_scene_get_bvh(self.h, C.byref(n), C.byref(nn), C.byref(t), C.byref(nt)))
        nodes = np.frombuffer((C.c_char * (nn.value * 80)).from_address(n.value), dtype=NODE_DTYPE).copy() if nn.value else np.empty(0, NODE_DTYPE)
        tris = np.frombuffer((C.c_char * (nt.value * 48)).from_address(t.value), dtype=TRI_DTYPE).copy() if nt.value else np.empty(0, TRI_DTYPE)
        return nodes, tris

    def get_leaves(self):
        """the traversal's 64-byte leaf records (a node's tri_base / meta offsets count these)"""
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.lib.mq_scene_get_leaves(self.h, C.byref(p), C.byref(n)))
        return np.frombuffer((C.c_char * (n.value * 64)).from_address(p.value), dtype=LEAF_DTYPE).copy() if n.value else np.empty(0, LEAF_DTYPE)

    def scene_stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        s = C.c_float()
        self._chk(self.lib.mq_scene_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(s)))
        return dict(n_tris=a.value, n_nodes=b.value, bvh_bytes=c.value, sah_cost=s.value)

    def synth_scene(self, name, seed=1):
        self._chk(self.lib.mq_synth_scene(self.h, name.encode(), seed))

    def synth_camera(self, frame):
        u = Uniform()
        self._chk(self.lib.mq_synth_camera(self.h, frame, C.byref(u)))
        return u

    def load_bsp(self, path, palette=None):
        self._chk(self.lib.mq_load_bsp(self.h, path.encode(), palette.encode() if palette else None))

    # -- describe / connect / process
    def describe(self, w, h):
        d = IoDesc()
        self._chk(self.lib.mq_describe(self.h, w, h, C.byref(d)))
        return d

    def connect(self, w, h):
        self._chk(self.lib.mq_connect(self.h, w, h))
        self.W, self.H = w, h

    def process(self, uniform, render=True, stream=None):
        self._chk(self.lib.mq_process(self.h, C.byref(uniform), 1 if render else 0, stream))

    def sync(self):
        self._chk(self.lib.mq_sync(self.h))

    def map_output(self, which):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self.lib.mq_map_output(self.h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def read_output(self, which):
        _, n = self.map_output(which)
        buf = np.empty(n, np.uint8)
        self._chk(self.lib.mq_read_output(self.h, which, _ptr(buf), n))
        return buf

    def irradiance(self):
        return self.read_output(OUT_IRRADIANCE).view(np.float32).reshape(self.H, self.W, 4)

    def volume(self):
        return self.read_output(OUT_VOLUME).view(np.float32).reshape(self.H, self.W, 4)

    def last_frame_ms(self):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        self._chk(self.lib.mq_last_frame_ms(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def timing_reset(self):
        self._chk(self.lib.mq_timing_reset(self.h))

    def timing_get(self):
        n, a, b = C.c_uint32(), C.c_double(), C.c_double()
        self._chk(self.lib.mq_timing_get(self.h, C.byref(n), C.byref(a), C.byref(b)))
        return n.value, a.value, b.value

    def timing_set_interval(self, every):
        self._chk(self.lib.mq_timing_set_interval(self.h, every))

    def timing_detail_frames(self):
        n = C.c_uint32()
        self._chk(self.lib.mq_timing_detail_frames(self.h, C.byref(n)))
        return n.value

    def timing_detail(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._chk(self.lib.mq_timing_get_detail(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(primary_ms=a.value, trace_ms=b.value, bounce_ms=c.value)

    def timing_rounds(self):
        """[(trace_ms_sum, shade_ms_sum)]: entry 0 = primary rays / first-hit shading, entry 1 + r = round r."""
        a, b = (C.c_double * 9)(), (C.c_double * 9)()
        self._chk(self.lib.mq_timing_get_rounds(self.h, a, b, 9))
        return list(zip(a, b))

    def measure_stream_read(self, nbytes=2 << 30, reps=5):
        v = C.c_double()
        self._chk(self.lib.mq_measure_stream_read(self.h, nbytes, reps, C.byref(v)))
        return v.value

    def enable_counters(self, on):
        self._chk(self.lib.mq_enable_counters(self.h, 1 if on else 0))

    def counters(self):
        c = Counters()
        self._chk(self.lib.mq_get_counters(self.h, C.byref(c)))
        return c.as_dict()

    MC_DTYPE = np.dtype([("w_tgt", "<f4", 3), ("sum_w", "<f4"), ("w_cos", "<f4"), ("T", "<f4"), ("id", "<u4"), ("n_hash", "<u4"),
                         ("mv", "<u2", 3), ("pad0", "<u2"), ("pad", "<u4", 6)])
    LC_DTYPE = np.dtype([("hash", "<u4"), ("lock", "<u4"), ("irr", "<u2", 3), ("N", "<u2")])
    DIST_DTYPE = np.dtype([("sum_w", "<f4"), ("N", "<u4"), ("m0", "<f4"), ("m1", "<f4")])

    STAT_DTYPES = (np.dtype([("update_succeeded", "<u4"), ("update_canceled", "<u4")]), np.dtype("<u4"))

    def state_read(self, which, count):
        a = np.zeros(count, (self.MC_DTYPE, self.LC_DTYPE, self.DIST_DTYPE) [which] if which < 3 else self.STAT_DTYPES[which - 3])
        self._chk(self.lib.mq_debug_state_read(self.h, which, a.ctypes.data, a.nbytes))
        return a

    def state_write(self, which, a):
        a = np.ascontiguousarray(a, (self.MC_DTYPE, self.LC_DTYPE, self.DIST_DTYPE)[which])
        self._chk(self.lib.mq_debug_state_write(self.h, which, a.ctypes.data, a.nbytes))

    # -- per-frame geometry producers (QuakeNode::update_dynamic_geo)
    def dyn_begin(self):
        self._chk(self.lib.mq_dyn_begin(self.h))

    def dyn_end(self, slot):
        self._chk(self.lib.mq_dyn_end(self.h, slot))

    def dyn_add_particles(self, particles, view, texnum_blood, texnum_explosion, cl_time, prev_cl_time):
        p = np.ascontiguousarray(particles, PARTICLE_DTYPE)
        self._chk(self.lib.mq_dyn_add_particles(self.h, p.ctypes.data, len(p), C.byref(view), texnum_blood, texnum_explosion, cl_time, prev_cl_time))

    def dyn_add_alias(self, model, inst):
        self._chk(self.lib.mq_dyn_add_alias(self.h, model, C.byref(inst)))

    def dyn_add_alias_batch(self, models, insts):
        """n entities at once (on the library's worker pool): the same triangles as n calls of dyn_add_alias"""
        n = len(insts)
        ms = (C.c_int * n)(*[int(m) for m in models]); arr = (AliasInstance * n)(*insts)
        self._chk(self.lib.mq_dyn_add_alias_batch(self.h, ms, arr, n))

    def dyn_add_sprite(self, model, inst, view):
        self._chk(self.lib.mq_dyn_add_sprite(self.h, model, C.byref(inst), C.byref(view)))

    def dyn_add_brush_model(self, model, origin, angles, prev_origin, prev_angles):
        f3 = lambda v: (C.c_float * 3)(*v)
        self._chk(self.lib.mq_dyn_add_brush_model(self.h, model, f3(origin), f3(angles), f3(prev_origin), f3(prev_angles)))

    def bsp_model_count(self):
        return self.lib.mq_bsp_model_count(self.h)

    def load_mdl(self, path, first_texnum, palette=None):
        m, nt = C.c_int(), C.c_uint32()
        self._chk(self.lib.mq_load_mdl(self.h, path.encode(), palette.encode() if palette else None, first_texnum, C.byref(m), C.byref(nt)))
        return m.value, nt.value

    def load_spr(self, path, first_texnum, palette=None):
        m, nt = C.c_int(), C.c_uint32()
        self._chk(self.lib.mq_load_spr(self.h, path.encode(), palette.encode() if palette else None, first_texnum, C.byref(m), C.byref(nt)))
        return m.value, nt.value

    def restir_process(self, uniform, render=True, stream=None):
        self._chk(self.lib.mq_restir_process(self.h, C.byref(uniform), 1 if render else 0, stream))

    def post_process(self, stream=None):
        self._chk(self.lib.mq_post_process(self.h, stream))

    def post_clear(self):
        self._chk(self.lib.mq_post_clear(self.h))

    def image(self, which):
        """an RGBA32F output as (H, W, 4) float32"""
        return self.read_output(which).view(np.float32).reshape(self.H, self.W, 4)

    def learn_log(self):
        """Records of the last frame's learning-write log (property "debug: log learning writes"): (n, 16) uint32."""
        n = C.c_size_t()
        self._chk(self.lib.mq_debug_learn_log_read(self.h, None, 0, C.byref(n)))
        a = np.zeros((n.value, 16), np.uint32)
        if n.value:
            self._chk(self.lib.mq_debug_learn_log_read(self.h, a.ctypes.data, n.value, C.byref(n)))
        return a

    def apply_updates(self, records, uniform):
        """The update pass alone on caller-given queue contents: (n, 16) uint32 records with slot and rank filled in."""
        r = np.ascontiguousarray(records, np.uint32).reshape(-1, 16)
        self._chk(self.lib.mq_debug_apply_updates(self.h, r.ctypes.data, len(r), C.byref(uniform)))

    def section_clocks(self, reset=True):
        """-DMQ_PROF builds: shader clocks per code section (tools/prof_sections.py); zeros otherwise."""
        a = (C.c_uint64 * 104)()  # 40 sections + 64 histogram bins
        self._chk(self.lib.mq_debug_section_clocks(self.h, a, 104, 1 if reset else 0))
        return list(a)

    def reset_state(self):
        self._chk(self.lib.mq_reset_state(self.h))

    def set_partition(self, rank, world):
        self._chk(self.lib.mq_set_partition(self.h, rank, world))

    def tiles_per_rank(self):
        t, b = C.c_uint32(), C.c_size_t()
        self._chk(self.lib.mq_tiles_per_rank(self.h, C.byref(t), C.byref(b)))
        return t.value, b.value

    # -- row partition of the ReSTIR node / post chain
    def band_layout(self, w, h, rank, world):
        b = Band()
        self._chk(self.lib.mq_band_layout(self.h, w, h, rank, world, C.byref(b)))
        return b

    def band_gbuffer(self, uniform, stream=None):
        self._chk(self.lib.mq_band_gbuffer(self.h, C.byref(uniform), stream))

    def map_halo(self, which):
        """(send base pointer, receive base pointer, bytes per pixel row) of halo buffer `which` (HALO_*)"""
        a, b, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        self._chk(self.lib.mq_map_halo(self.h, which, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def untile(self, gathered_dev_ptr, stream=None):
        self._chk(self.lib.mq_untile(self.h, gathered_dev_ptr, stream))

    def untile_to(self, gathered_dev_ptr, image_dev_ptr, stream=None):
        self._chk(self.lib.mq_untile_to(self.h, gathered_dev_ptr, image_dev_ptr, stream))

    def untile_volume(self, gathered_dev_ptr, stream=None):
        self._chk(self.lib.mq_untile_volume(self.h, gathered_dev_ptr, stream))

    def untile_volume_depth(self, gathered_dev_ptr, stream=None):
        self._chk(self.lib.mq_untile_volume_depth(self.h, gathered_dev_ptr, stream))

    def trace_rays(self, org, direction):
        org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
        direction = np.ascontiguousarray(direction, np.float32).reshape(-1, 3)
        n = len(org)
        prim, t, uv = np.empty(n, np.uint32), np.empty(n, np.float32), np.empty((n, 2), np.float32)
        self._chk(self.lib.mq_trace_rays(self.h, _ptr(org), _ptr(direction), n, _ptr(prim), _ptr(t), _ptr(uv)))
        return prim, t, uv

    def math_eval(self, op, inp, n_out):
        inp = np.ascontiguousarray(inp, np.float32)
        n = inp.shape[0]
        out = np.empty((n, n_out), np.float32)
        self._chk(self.lib.mq_math_eval(self.h, op, _ptr(inp), _ptr(out), n))
        return out


def uniform_update(lib, uniform, state):
    """QuakeNode::process's per-frame uniform (quake_node.cpp:768-824): `uniform` holds the previous frame's on entry"""
    r = lib.mq_uniform_update(C.byref(uniform), C.byref(state))
    if r < 0:
        raise MqError(r, "mq_uniform_update")
    return uniform
