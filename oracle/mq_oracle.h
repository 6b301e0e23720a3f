/*
 * mq_oracle.h -- CPU ORACLE for the merian-quake hot path (g-buffer first hit -> MCPG surface
 * estimator -> Markov-chain update application).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load or call it.  The shipped library (libmqhip.so) never
 * links, includes or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference (UnleqitDEV/merian-quake) ships no tests, no golden vectors and
 * cannot be built here (its `merian` and `quakespasm` submodules are empty, Vulkan ray-query GLSL
 * cannot run on a CPU).  This file restates the reference's algorithm from its shader / host
 * sources (cited per function as file:line relative to the reference root).  Every helper that the
 * reference takes from the absent `merian-shaders` headers (RNG, BSDF, vMF, hash grid, normal
 * codec, camera, texture sampling) is DEFINED here; those definitions are listed in DESIGN.md
 * ("Definitions for absent symbols").  The oracle is pinned only by analytic known-answer tests
 * (tests/test_oracle_kat.py) and by brute-force cross checks, never by reference outputs.
 *
 * Plain C11, no dependencies beyond libm/pthreads.
 */
#ifndef MQ_ORACLE_H
#define MQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- input layouts: identical to the reference's scene contract ------------------------------ */

/* src/game/quake_helpers.hpp:10-34 / res/shader/scene_info.glsl.h:7-16 (28 bytes) */
typedef struct {
    uint16_t texnum_alpha;    /* low 12: texnum, high 4: alpha (0 = use texture alpha, 15 = opaque) */
    uint16_t texnum_fb_flags; /* low 12: fullbright texnum, high 4: MAT_FLAGS_* */
    uint32_t n0_gloss_norm;
    uint32_t n1_brush;
    uint32_t n2;
    uint16_t st[6]; /* s0 t0 s1 t1 s2 t2 as IEEE half */
} orc_ext_t;

/* res/shader/scene_info.glsl.h:18-32 (124 bytes) */
typedef struct {
    float cam_x[4];      /* xyz position, w = mu_t */
    float cam_w[4];      /* xyz forward,  w = time diff */
    float cam_u[4];      /* xyz up */
    float prev_cam_x[4]; /* w = mu_s.r */
    float prev_cam_w[4]; /* w = mu_s.g */
    float prev_cam_u[4]; /* w = mu_s.b */
    uint32_t sky_rt_bk, sky_lf_ft, sky_up_dn;
    float cl_time;
    uint32_t frame;
    uint32_t player;
    uint32_t rt_config;
} orc_uniform_t;

/* The macro table of src/render_mcpg/render_mcpg.cpp:137-185 plus the g-buffer spec constants of
 * src/gbuffer/gbuffer.cpp:76-115, as one POD block. */
typedef struct {
    int32_t reference_mode;     /* MERIAN_QUAKE_REFERENCE_MODE */
    int32_t adaptive_grid_type; /* 0 exponential, 1 quadratic */
    int32_t spp;                /* SURFACE_SPP */
    int32_t max_path_length;    /* MAX_PATH_LENGTH */
    int32_t use_light_cache_tail;
    float fov_tan_alpha_half;
    float sun_w[3];
    float sun_color[3];
    int32_t volume_spp;
    int32_t volume_use_light_cache;
    float draine_g, draine_a;
    int32_t mc_samples;
    float mc_samples_adaptive_prob;
    int32_t distance_mc_samples;
    int32_t mc_fast_recovery;
    int32_t lc_grid_type;
    uint32_t lc_buffer_size;
    float lc_grid_steps_per_unit_size, lc_grid_tan_alpha_half, lc_grid_min_width, lc_grid_power;
    uint32_t mc_adaptive_buffer_size;
    float mc_adaptive_grid_tan_alpha_half, mc_adaptive_grid_min_width, mc_adaptive_grid_power,
        mc_adaptive_grid_steps_per_unit_size;
    uint32_t mc_static_buffer_size;
    float mc_static_grid_width;
    int32_t distance_mc_grid_width;
    float volume_max_t;
    float surf_bsdf_p, volume_phase_p, dir_guide_prior, dist_guide_p;
    uint32_t distance_mc_vertex_state_count;
    uint32_t seed;
    /* g-buffer node */
    int32_t gbuffer_hide_sun; /* src/gbuffer/gbuffer.cpp:99 */
    /* named quirk switches */
    int32_t quirk_lc_max_wo_p; /* 1 = keep `max(wo_p, 10)` of mcpg.comp:170 (default) */
    int32_t quirk_n16_wrap;    /* 1 = wrap N*N to 16 bit as GLSL uint16 arithmetic would (mc.glsl:26) */
    int32_t volume_forward_project; /* render_mcpg.hpp:153 */
    int32_t enable_albedo_mipmap, enable_emission_mipmap; /* g-buffer node, src/gbuffer/gbuffer.cpp:49-50,79-81 */
    int32_t debug_output_connected, debug_output_selector; /* DEBUG_OUTPUT_CONNECTED / _SELECTOR, render_mcpg.cpp:172-173 */
    int32_t freeze_learning; /* test hook: learning computations and RNG draws run, the stores to MC / LC / distance state do not */
    int32_t log_learning;    /* test hook: every PROPOSED learning write is appended to the learning log (orc_learn_log) */
} orc_params_t;

void orc_params_header_defaults(orc_params_t* p); /* src/render_mcpg/render_mcpg.hpp:108-166 */
void orc_params_json_defaults(orc_params_t* p);   /* res/default_config.json:599-638 */

enum {
    ORC_GEO_OPAQUE = 1, /* all triangles opaque: no any-hit alpha test (quake_node.cpp:869-871) */
};
enum {
    ORC_TEX_SRGB = 1,   /* decode sRGB -> linear on fetch (quake_node.hpp:93-95) */
    ORC_TEX_LINEAR = 2, /* bilinear magnification (else nearest) */
    ORC_TEX_MIPMAP = 4, /* has a mip chain (TEXPREF_MIPMAP, quake_node.cpp:698): used by the first hit's textureGrad */
};

enum {
    ORC_OUT_IRRADIANCE = 0,      /* mcpg: RGBA32F rgb = mean radiance, a = second moment */
    ORC_OUT_GB_ALBEDO = 1,       /* gbuffer: RGBA16F */
    ORC_OUT_GB_IRRADIANCE = 2,   /* gbuffer: RGBA16F first-hit emission */
    ORC_OUT_GB_MV = 3,           /* gbuffer: RG16F */
    ORC_OUT_GBUFFER = 4,         /* 16 B/px: enc_normal u32, linear_z f32, grad_z 2xf16, vel_z f32 */
    ORC_OUT_HITS = 5,            /* 40 B/px CompressedHit (res/shader/hit.glsl.h:19-30) */
    ORC_OUT_VOLUME = 6,          /* mcpg "volume" RGBA32F (volume.comp:237) */
    ORC_OUT_VOLUME_DEPTH = 7,    /* mcpg "volume_depth" R16F (volume.comp:211) */
    ORC_OUT_VOLUME_MV = 8,       /* mcpg "volume_mv" RG16F (render_mcpg.cpp:284-311) */
    ORC_OUT_DEBUG = 9,           /* mcpg "debug" RGBA16F (mcpg.comp:212-277), written when debug_output_connected */
    ORC_OUT_COUNT
};

typedef struct {
    uint64_t rays, nodes, tris, segments, guided_segments, lc_touches, mc_updates_accepted,
        mc_updates_dropped, mc_state_reads;
} orc_counters_t;

typedef struct orc_ctx orc_ctx;

orc_ctx* orc_create(const orc_params_t* p);
void orc_destroy(orc_ctx* c);
int orc_set_params(orc_ctx* c, const orc_params_t* p);
int orc_set_geometry(orc_ctx* c, int slot, const float* vtx, const float* prev_vtx, uint32_t n_vtx,
                     const uint32_t* idx, const orc_ext_t* ext, uint32_t n_tri, uint32_t flags);
int orc_set_texture(orc_ctx* c, uint32_t texnum, uint32_t w, uint32_t h, const uint8_t* rgba8,
                    uint32_t flags);
/* accel: 0 = brute force over all triangles, 1 = plain binary median-split BVH */
int orc_commit(orc_ctx* c, int accel);
/* allocate + zero all state (render_mcpg.cpp:221-226) for a W x H frame */
int orc_connect(orc_ctx* c, uint32_t w, uint32_t h);
/* one frame: g-buffer pass, surface pass, update pass (render_mcpg.cpp:243-277).
 * threads > 1 is only deterministic in reference mode. */
int orc_process(orc_ctx* c, const orc_uniform_t* u, int render, int threads);
/* the same with the update pass spread over the worker threads too (unordered, as the reference's one-thread-per-slot dispatch):
 * the CPU baseline of bench.py; tests use orc_process, whose update pass is sequential and deterministic */
int orc_process_mt(orc_ctx* c, const orc_uniform_t* u, int render, int threads, int parallel_update);
const void* orc_output(orc_ctx* c, int which, size_t* bytes);
void orc_get_counters(orc_ctx* c, orc_counters_t* out, int reset);
/* learning state, oracle layout: which 0 = Markov chains (52 B: id u32, 3 unused f32, w_tgt f32x3, sum_w, w_cos, mv f16x3,
 * 2 B padding, T f32, N u16, hash u16), 1 = light cache (24 B: hash, lock, irr f16x3, N u16, ok u32, cancel u32),
 * 2 = distance Markov chains (16 B: sum_w f32, N u32, m0 f32, m1 f32).
 * Returns a pointer into the context (valid until the next orc_connect) and the entry count. */
void* orc_debug_state(orc_ctx* c, int which, size_t* count, size_t* entry_bytes);

/* ---- learning-write log (test hook; the product library has the same log behind "debug: log learning writes") ----
 * With params.log_learning set, every learning write a path PROPOSES is appended as one 64-byte record of 16 dwords
 * (with freeze_learning also set nothing is stored, so the log is a deterministic function of the given state):
 *   kind 1, a queued Markov-chain update (mc.glsl:159-184): the MqUpdate layout of the product's queue --
 *           pos[3] weight target[3] id normal[3] T  mv0|mv1<<16  mv2|rank<<16 (rank 0 here)  slot  kind
 *   kind 2, a light-cache store (light_cache.glsl:66-84): chk, rekeyed?, irr0|irr1<<16, irr2|N<<16, ... [14] = cell, [15] = kind
 *   kind 3, a fast-recovery invalidation (mcpg.comp:175-178, volume.comp:226-229): [14] = slot, [15] = kind
 *   kind 4, a distance-chain store (volume.comp:213-215): sum_w, N, m0, m1, ... [14] = index, [15] = kind
 * orc_learn_log returns the records written since the last reset (count in *n); records beyond the capacity set by
 * orc_learn_log_reset are counted but not kept. */
const uint32_t* orc_learn_log(orc_ctx* c, size_t* n);
int orc_learn_log_reset(orc_ctx* c, size_t capacity);
/* Update application alone (compute_updates.comp:56-124) on caller-given queue contents: n records in the kind-1 layout
 * above, each with its slot and its arrival rank (dense 0..k-1 per slot, k <= 10).  Slots are applied in ascending order.
 * touches (may be NULL): receives up to touch_cap (slot, cell) pairs, one per table entry a slot's application reads or
 * writes -- lets a test find slots whose applications interfere; *n_touches = pairs produced. */
int orc_debug_apply_updates(orc_ctx* c, const uint32_t* records, size_t n, const orc_uniform_t* u, uint32_t* touches, size_t touch_cap, size_t* n_touches);

/* ---- post chain: the graph's "accum" / "volume accum" nodes, albedo re-modulation and the "add" node
 * (res/default_config.json:21-133,404-435,473-497).  merian's node sources are absent from the reference tree: the
 * arithmetic is DEFINED in DESIGN.md section 3 ("post chain") and restated here and in mq_post.hip independently.
 * params: which 0 = accum, 1 = volume accum; six floats: alpha, max history, normal threshold (radians), depth
 * threshold, enable motion vectors, reuse border.  Outputs: 0 accum RGBA32F, 1 accum history R32F, 2 volume accum
 * RGBA32F, 3 volume accum history R32F, 4 final RGBA32F. */
int orc_post_set_params(orc_ctx* c, int which, const float* six);
int orc_post_process(orc_ctx* c); /* after orc_process of the same frame */
int orc_post_set_add_restir(orc_ctx* c, int on); /* final += ReSTIR irradiance * albedo (the product's "add: restir irradiance") */
void orc_post_clear(orc_ctx* c);
const void* orc_post_output(orc_ctx* c, int which, size_t* bytes);

/* ---- ReSTIR DI node: src/render_restir/renderer_restir.cpp:129-251, res/shader/render_restir/{restir_di.glsl,
 * restir_di_common.glsl, restir_di_generate_samples_bsdf.comp, restir_di_temporal_reuse.comp, restir_di_spatial_reuse.comp,
 * restir_di_shade.comp, restir_di_clear.comp}.  Runs on the g-buffer outputs orc_process left for the same frame. */
typedef struct {
    int32_t spp; uint32_t seed; int32_t visibility_shade;
    float temporal_normal_reject_cos, temporal_depth_reject, spatial_normal_reject_cos, spatial_depth_reject;
    int32_t temporal_clamp_m, spatial_radius, temporal_bias_correction, spatial_bias_correction;
    float boiling_filter_strength;
    int32_t spatial_reuse_iterations; /* the property's value (0 = no spatial pass) */
    int32_t apply_mv, temporal_reuse_enable;
} orc_restir_params_t;
int orc_restir_process(orc_ctx* c, const orc_restir_params_t* r, const orc_uniform_t* u, int render, int threads);
/* which: 0 irradiance RGBA32F, 1 moments RG32F, 2 reservoirs (64 B per pixel) */
const void* orc_restir_output(orc_ctx* c, int which, size_t* bytes);

/* closest-hit queries (raytrace.glsl:82-119 semantics: back-face cull, alpha any-hit, tmin 0,
 * tmax 1e4).  out_prim = (slot << 28 | prim) or 0xffffffff on miss. */
int orc_trace_rays(orc_ctx* c, const float* org, const float* dir, uint32_t n, uint32_t* out_prim,
                   float* out_t, float* out_uv);

/* scalar math / shading known-answer entry points; `op` values below. in/out are float arrays,
 * n elements of `arity` inputs each. */
enum {
    ORC_OP_EXP2 = 0,      /* 1 -> 1 */
    ORC_OP_LOG2 = 1,      /* 1 -> 1 */
    ORC_OP_SINCOS2PI = 2, /* 1 -> 2 */
    ORC_OP_POW = 3,       /* 2 -> 1 */
    ORC_OP_F2H2F = 4,     /* 1 -> 1 round trip through half */
    ORC_OP_ENC_DEC_NORMAL = 5, /* 3 -> 4 (decoded xyz, encoded bits as float-punned u32) */
    ORC_OP_BSDF_SAMPLE = 6, /* wi3 n3 rough xi3 (10) -> wo3 pdf value (5) */
    ORC_OP_VMF_SAMPLE = 7,  /* mu3 kappa xi2 (6) -> w3 pdf (4) */
    ORC_OP_XORSHIFT = 8,    /* seed-as-float-punned (1) -> 4 successive uniforms (4) */
    ORC_OP_PCG4D16 = 9,     /* 4 punned u32 -> 1 punned u32 */
    ORC_OP_SKY = 10,        /* w3 (3) -> rgb (3); uses ctx params, no sky textures */
    ORC_OP_HASHGRID = 11,   /* pos3 normal3 level width-as-float size-punned (9) -> idx, chk punned (2) */
    ORC_OP_LDR_TO_HDR = 12, /* 3 -> 3 */
    ORC_OP_CAMERA = 13,     /* px py W H fwd3 up3 tan (11) -> dir3 + pixel2 roundtrip (5) */
    ORC_OP_DRAINE = 14,     /* wi3 g a xi2 (7) -> wo3, pdf (4) */
    ORC_OP_DISTANCE = 15,   /* mu_t tmax xi, gauss mu sigma xi2 (7) -> t, pdf_t, gauss x, gauss pdf (4) */
    ORC_OP_TEX_SAMPLE = 16, /* texnum-as-float s t (3) -> rgba (4): REPEAT addressing, nearest / bilinear, sRGB decode */
    ORC_OP_SKY_TEX = 17,    /* w3, sky_rt_bk sky_lf_ft sky_up_dn (punned u32), cl_time (7) -> rgb (3): textured skies */
    ORC_OP_TEX_GRAD = 18,   /* texnum s t dsdx dtdx dsdy dtdy (7) -> rgba (4): textureGrad with the mip chain */
    ORC_OP_COUNT = 19
};
int orc_math_eval(orc_ctx* c, int op, const float* in, float* out, uint32_t n);
int orc_op_arity(int op, int* n_in, int* n_out);

#ifdef __cplusplus
}
#endif
#endif
