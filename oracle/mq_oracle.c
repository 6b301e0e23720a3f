/*
 * mq_oracle.c -- CPU ORACLE (test infrastructure; see mq_oracle.h for the rules and the
 * "parity unpinned" statement).  A plain-C restatement of the reference's hot path:
 *
 *   g-buffer first hit   res/shader/gbuffer/gbuffer.comp:75-131
 *   trace_ray + any-hit  res/shader/raytrace.glsl:25-65,82-119,156-311
 *   hit (de)compression  res/shader/hit.glsl.h:34-53
 *   surface estimator    res/shader/render_mcpg/mcpg.comp:39-210
 *   Markov-chain states  res/shader/render_mcpg/mc.glsl:17-222, grid.h:6-35
 *   light cache          res/shader/render_mcpg/light_cache.glsl:13-84
 *   update application   res/shader/render_mcpg/compute_updates.comp:41-124
 *   pass order           src/render_mcpg/render_mcpg.cpp:221-277
 *
 * Traversal is deliberately naive: brute force over all triangles, or a plain binary
 * median-split BVH.  Both return the identical closest hit (smallest t, ties -> smallest
 * (slot, prim) key).
 */
#define _GNU_SOURCE
#include "mq_oracle.h"
#include "orc_math.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>

#define T_MAX 10000.0f          /* res/shader/config.h:11 */
#define ALPHA_THRESHOLD 0.666f  /* res/shader/config.h:13 */
#define MAX_GLTEXTURES 4096     /* res/shader/config.h:5 */
#define MAX_GEOMETRIES 16       /* res/shader/config.h:6 */
#define MAT_FLAGS_WATER 4
#define MAT_FLAGS_SKY 5
#define MAT_FLAGS_WATERFALL 6
#define MAT_FLAGS_SPRITE 7
#define MAT_FLAGS_TELE 3
#define MAT_FLAGS_SOLID 8
#define ML_MAX_N 1024           /* mc.glsl:2 */
#define ML_MIN_ALPHA 0.01f      /* mc.glsl:3 */
#define LIGHT_CACHE_MAX_N 128   /* light_cache.glsl:1 */
#define LIGHT_CACHE_MIN_ALPHA 0.01f
#define MAX_UPDATES 10          /* grid.h:29-34 */
#define BARY_EPS 3.814697265625e-06f /* 2^-18: closes cracks between adjacent triangles */

/* ---------------------------------------------------------------- data */

typedef struct {
    float* vtx; float* prev_vtx; uint32_t n_vtx;
    uint32_t* idx; orc_ext_t* ext; uint32_t n_tri; uint32_t flags;
} geo_t;

/* mip: levels 1.. of the chain as linear RGBA32F, level k at mip + mip_off[k] floats (NULL without ORC_TEX_MIPMAP) */
typedef struct { uint32_t w, h, flags; uint8_t* px; uint32_t levels; float* mip; size_t mip_off[16]; } tex_t;

typedef struct { v3 v0, v1, v2; uint32_t key; uint32_t opaque; } tri_t;
typedef struct { v3 bmin, bmax; uint32_t left, count; } bnode_t; /* count>0: leaf [left,left+count) */

/* grid.h:6-21 */
typedef struct {
    uint32_t id; float tgt_change, w_change, cos_change;
    v3 w_tgt; float sum_w, w_cos;
    uint16_t mv[3]; float T; uint16_t N; uint16_t hash;
} mcstate_t;
/* grid.h:23-35 (the reference stores one per state slot; the oracle keeps them sparsely) */
typedef struct {
    float T; uint16_t mv[MAX_UPDATES][3]; uint32_t ids[MAX_UPDATES]; float weights[MAX_UPDATES];
    v3 targets[MAX_UPDATES], positions[MAX_UPDATES], normals[MAX_UPDATES];
} mcupdate_t;
/* grid.h:37-46 */
typedef struct { uint32_t hash, lock; uint16_t irr[3]; uint16_t N; uint32_t ok, cancel; } lcvertex_t;

/* hit.glsl.h:6-17 */
typedef struct { v3 pos, prev_pos, wi, normal; uint32_t enc_geonormal; v3 albedo; float roughness; } hit_t;
/* hit.glsl.h:19-30, 40 bytes scalar layout */
typedef struct { float pos[3]; uint16_t mv[3]; uint16_t _pad; uint32_t wi, normal, enc_geonormal; uint16_t albedo[3]; uint16_t roughness; } chit_t;

typedef struct { uint32_t enc_normal; float linear_z; uint16_t grad_z[2]; float vel_z; } gbuf_t;
/* grid.h:48-52 */
typedef struct { float sum_w; uint32_t N; float m0, m1; } distmc_t;
#define DISTANCE_ML_MAX_N 1024 /* mc_distance.glsl:2 */
#define DISTANCE_ML_MIN_ALPHA 0.01f

struct orc_ctx {
    orc_params_t p;
    geo_t geo[MAX_GEOMETRIES];
    tex_t* tex; /* MAX_GLTEXTURES */
    float srgb_lut[256];
    /* accel */
    int accel;
    tri_t* tris; uint32_t n_tris;
    bnode_t* nodes; uint32_t n_nodes;
    /* frame state */
    uint32_t W, H;
    orc_uniform_t u;
    mcstate_t* mc; uint32_t mc_total;
    lcvertex_t* lc;
    uint32_t* upd_count; uint32_t* upd_rec; mcupdate_t* upd_pool; uint32_t upd_pool_cap; uint32_t upd_pool_used;
    uint32_t* upd_touched; uint32_t upd_touched_n;
    /* outputs */
    float* irradiance; uint16_t* gb_albedo; uint16_t* gb_irr; uint16_t* gb_mv; gbuf_t* gbuffer; chit_t* hits;
    /* volume pass (volume.comp, mc_distance.glsl, volume_forward_project.comp) */
    float* volume; uint16_t* volume_depth; uint16_t* prev_volume_depth; uint16_t* volume_mv;
    uint16_t* debug; /* RGBA16F */
    distmc_t* dist_mc; uint32_t dist_mc_n;
    uint64_t iteration;
    orc_counters_t ctr;
    int racy; /* threads > 1 in guided mode */
    /* learning-write log (params.log_learning) and the touch list of orc_debug_apply_updates */
    uint32_t* llog; size_t llog_cap, llog_n;
    uint32_t* touch; size_t touch_cap, touch_n;
    /* ReSTIR DI node */
    uint8_t* rs_out; uint8_t* rs_pong; uint8_t* rs_prev; gbuf_t* rs_prev_gb; float* rs_irr; float* rs_mom; uint64_t rs_iteration;
    /* post chain */
    float post_par[2][6];
    float* post_out[2]; float* post_hist[2]; float* post_prev_out[2]; float* post_prev_hist[2]; gbuf_t* post_prev_gb; float* post_final;
    int post_first; int volume_ran; int post_add_restir; /* the `add` node's input for the ReSTIR node's irradiance (times albedo) */
    struct orc_pool* pool; /* persistent worker threads of the multi-threaded passes */
    int parallel_update;   /* orc_process_mt flag: the update pass over the workers too (racy, like the reference's dispatch) */
};

/* ---------------------------------------------------------------- params */

void orc_params_header_defaults(orc_params_t* p) { /* render_mcpg.hpp:108-166, gbuffer.hpp:75-77 */
    memset(p, 0, sizeof *p);
    p->reference_mode = 0; p->adaptive_grid_type = 0; p->spp = 1; p->max_path_length = 3;
    p->use_light_cache_tail = 0; p->fov_tan_alpha_half = 1.0f;
    p->sun_w[0] = p->sun_w[1] = p->sun_w[2] = 0.57735026919f;
    p->volume_spp = 0; p->volume_use_light_cache = 0;
    p->mc_samples = 5; p->mc_samples_adaptive_prob = 0.7f; p->distance_mc_samples = 3; p->mc_fast_recovery = 1;
    p->lc_grid_type = 0; p->lc_buffer_size = 4000000; p->lc_grid_steps_per_unit_size = 6.0f;
    p->lc_grid_tan_alpha_half = 0.002f; p->lc_grid_min_width = 0.01f; p->lc_grid_power = 2.0f;
    p->mc_adaptive_buffer_size = 32777259; p->mc_adaptive_grid_tan_alpha_half = 0.003f;
    p->mc_adaptive_grid_min_width = 0.01f; p->mc_adaptive_grid_power = 4.0f; p->mc_adaptive_grid_steps_per_unit_size = 6.0f;
    p->mc_static_buffer_size = 800009; p->mc_static_grid_width = 25.3f; p->distance_mc_grid_width = 25;
    p->volume_max_t = 1000.0f; p->surf_bsdf_p = 0.15f; p->volume_phase_p = 0.3f; p->dir_guide_prior = 0.2f; p->dist_guide_p = 0.0f;
    p->distance_mc_vertex_state_count = 10; p->seed = 0;
    { double d = 25.0; p->draine_g = (float)exp(-2.20679 / (d + 3.91029) - 0.428934); p->draine_a = (float)exp(3.62489 - 8.29288 / (d + 5.52825)); }
    p->gbuffer_hide_sun = 1; p->quirk_lc_max_wo_p = 1; p->quirk_n16_wrap = 1; /* both on = what the reference's shaders compute (mcpg.comp:170; mc.glsl:26 with the uint16_t N of grid.h:19) */ p->volume_forward_project = 1;
    p->enable_albedo_mipmap = 1; p->enable_emission_mipmap = 1; /* src/gbuffer/gbuffer.hpp defaults; res/default_config.json:530-531 */
}
void orc_params_json_defaults(orc_params_t* p) { /* default_config.json:599-638 */
    orc_params_header_defaults(p);
    p->surf_bsdf_p = 0.1f; p->lc_buffer_size = 4000037; p->lc_grid_tan_alpha_half = 0.005f; p->lc_grid_type = 1;
    p->dir_guide_prior = 0.3f; p->volume_phase_p = 0.1f; p->mc_adaptive_grid_power = 1.7320508f;
    p->mc_adaptive_grid_steps_per_unit_size = 1.0f; p->mc_adaptive_grid_tan_alpha_half = 0.002f;
    p->dist_guide_p = 0.9f; p->spp = 2; p->volume_spp = 2; p->volume_use_light_cache = 1; p->volume_max_t = 10000.0f;
    { double d = 7.0; p->draine_g = (float)exp(-2.20679 / (d + 3.91029) - 0.428934); p->draine_a = (float)exp(3.62489 - 8.29288 / (d + 5.52825)); }
}

/* ---------------------------------------------------------------- ctx */

orc_ctx* orc_create(const orc_params_t* p) {
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof *c);
    if (!c) return NULL;
    { /* the "accum" / "volume accum" nodes of res/default_config.json:404-428,473-497 */
        const float a[6] = {0.951f, INFINITY, 0.645771861076355f, 0.026403000578284264f, 1.0f, 1.0f}, v[6] = {0.902f, INFINITY, 3.1415927410125732f, 0.28402701020240784f, 1.0f, 1.0f};
        memcpy(c->post_par[0], a, sizeof a); memcpy(c->post_par[1], v, sizeof v);
    }
    if (p) c->p = *p; else orc_params_header_defaults(&c->p);
    c->tex = (tex_t*)calloc(MAX_GLTEXTURES, sizeof(tex_t));
    for (int i = 0; i < 256; i++) { /* sRGB EOTF, evaluated in double then rounded once */
        double v = i / 255.0;
        double l = v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4);
        c->srgb_lut[i] = (float)l;
    }
    return c;
}
static void pool_destroy(orc_ctx* c);
static void free_state(orc_ctx* c) {
    for (int k = 0; k < 2; k++) { free(c->post_out[k]); free(c->post_hist[k]); free(c->post_prev_out[k]); free(c->post_prev_hist[k]); c->post_out[k] = c->post_hist[k] = c->post_prev_out[k] = c->post_prev_hist[k] = NULL; }
    free(c->post_prev_gb); free(c->post_final); c->post_prev_gb = NULL; c->post_final = NULL;
    free(c->rs_out); free(c->rs_pong); free(c->rs_prev); free(c->rs_prev_gb); free(c->rs_irr); free(c->rs_mom);
    c->rs_out = c->rs_pong = c->rs_prev = NULL; c->rs_prev_gb = NULL; c->rs_irr = c->rs_mom = NULL;
    free(c->llog); c->llog = NULL; c->llog_cap = c->llog_n = 0;
    free(c->mc); free(c->lc); free(c->upd_count); free(c->upd_rec); free(c->upd_pool); free(c->upd_touched);
    free(c->irradiance); free(c->gb_albedo); free(c->gb_irr); free(c->gb_mv); free(c->gbuffer); free(c->hits);
    free(c->volume); free(c->volume_depth); free(c->prev_volume_depth); free(c->volume_mv); free(c->dist_mc); free(c->debug); c->debug = NULL;
    c->volume = NULL; c->volume_depth = c->prev_volume_depth = c->volume_mv = NULL; c->dist_mc = NULL;
    c->mc = NULL; c->lc = NULL; c->upd_count = c->upd_rec = NULL; c->upd_pool = NULL; c->upd_touched = NULL;
    c->irradiance = NULL; c->gb_albedo = c->gb_irr = c->gb_mv = NULL; c->gbuffer = NULL; c->hits = NULL;
}
void orc_destroy(orc_ctx* c) {
    if (!c) return;
    pool_destroy(c);
    for (int i = 0; i < MAX_GEOMETRIES; i++) { free(c->geo[i].vtx); free(c->geo[i].prev_vtx); free(c->geo[i].idx); free(c->geo[i].ext); }
    for (int i = 0; i < MAX_GLTEXTURES; i++) { free(c->tex[i].px); free(c->tex[i].mip); }
    free(c->tex); free(c->tris); free(c->nodes);
    free_state(c);
    free(c);
}
int orc_set_params(orc_ctx* c, const orc_params_t* p) { c->p = *p; return 0; }

static void* dup_mem(const void* src, size_t n) { void* d = malloc(n ? n : 1); if (d && src) memcpy(d, src, n); return d; }

int orc_set_geometry(orc_ctx* c, int slot, const float* vtx, const float* prev_vtx, uint32_t n_vtx,
                     const uint32_t* idx, const orc_ext_t* ext, uint32_t n_tri, uint32_t flags) {
    if (slot < 0 || slot >= MAX_GEOMETRIES) return -1;
    geo_t* g = &c->geo[slot];
    free(g->vtx); free(g->prev_vtx); free(g->idx); free(g->ext);
    memset(g, 0, sizeof *g);
    if (n_tri == 0) return 0;
    g->vtx = (float*)dup_mem(vtx, (size_t)n_vtx * 12);
    g->prev_vtx = (float*)dup_mem(prev_vtx ? prev_vtx : vtx, (size_t)n_vtx * 12);
    g->idx = (uint32_t*)dup_mem(idx, (size_t)n_tri * 12);
    g->ext = (orc_ext_t*)dup_mem(ext, (size_t)n_tri * sizeof(orc_ext_t));
    g->n_vtx = n_vtx; g->n_tri = n_tri; g->flags = flags;
    for (uint32_t i = 0; i < n_tri * 3; i++) if (idx[i] >= n_vtx) return -2;
    return 0;
}
static void build_mips(const orc_ctx* c, tex_t* t);
int orc_set_texture(orc_ctx* c, uint32_t texnum, uint32_t w, uint32_t h, const uint8_t* rgba8, uint32_t flags) {
    if (texnum >= MAX_GLTEXTURES) return -1;
    tex_t* t = &c->tex[texnum];
    free(t->px); free(t->mip); t->px = NULL; t->mip = NULL; t->w = t->h = 0; t->levels = 1;
    if (!rgba8 || !w || !h) return 0;
    t->px = (uint8_t*)dup_mem(rgba8, (size_t)w * h * 4);
    t->w = w; t->h = h; t->flags = flags;
    if (flags & ORC_TEX_MIPMAP) build_mips(c, t);
    return 0;
}

/* ---------------------------------------------------------------- textures */

typedef struct { float r, g, b, a; } v4;

/* REPEAT addressing without integer division: u = s - floor(s) in [0,1], then scale by the size */
static inline v4 texel(const orc_ctx* c, const tex_t* t, int x, int y) { /* x, y already in range */
    const uint8_t* p = t->px + 4 * ((size_t)y * t->w + (size_t)x);
    v4 r;
    if (t->flags & ORC_TEX_SRGB) { r.r = c->srgb_lut[p[0]]; r.g = c->srgb_lut[p[1]]; r.b = c->srgb_lut[p[2]]; }
    else { r.r = (float)p[0] * (1.0f / 255.0f); r.g = (float)p[1] * (1.0f / 255.0f); r.b = (float)p[2] * (1.0f / 255.0f); }
    r.a = (float)p[3] * (1.0f / 255.0f);
    return r;
}
static inline int tex_nearest_coord(float s, float fw, int w) {
    float u = s - floorf(s);
    int i = (int)floorf(u * fw);
    return i > w - 1 ? w - 1 : i;
}
/* bilinear footprint along one axis: texel indices i0, i1 (wrapped) and the weight of i1 */
static inline void tex_linear_coord(float s, float fw, int w, int* i0, int* i1, float* f) {
    float u = s - floorf(s);
    float x = u * fw - 0.5f;
    float x0 = floorf(x);
    *f = x - x0;
    int a = (int)x0, b = a + 1;
    if (a < 0) a += w;
    if (b >= w) b -= w;
    *i0 = a; *i1 = b;
}
/* textureLod(img_tex[texnum], st, 0) with REPEAT wrap; nearest or bilinear per texture.
 * A texture slot that was never uploaded samples as opaque mid grey. */
static v4 tex_sample(const orc_ctx* c, uint32_t texnum, float s, float t) {
    if (texnum > MAX_GLTEXTURES - 1) texnum = MAX_GLTEXTURES - 1;
    const tex_t* tx = &c->tex[texnum];
    if (!tx->px) { v4 g = {0.5f, 0.5f, 0.5f, 1.0f}; return g; }
    float fw = (float)tx->w, fh = (float)tx->h;
    if (!(tx->flags & ORC_TEX_LINEAR)) return texel(c, tx, tex_nearest_coord(s, fw, (int)tx->w), tex_nearest_coord(t, fh, (int)tx->h));
    int x0, x1, y0, y1; float fx, fy;
    tex_linear_coord(s, fw, (int)tx->w, &x0, &x1, &fx);
    tex_linear_coord(t, fh, (int)tx->h, &y0, &y1, &fy);
    v4 a = texel(c, tx, x0, y0), b = texel(c, tx, x1, y0), d = texel(c, tx, x0, y1), e = texel(c, tx, x1, y1);
    v4 r;
    r.r = omix(omix(a.r, b.r, fx), omix(d.r, e.r, fx), fy);
    r.g = omix(omix(a.g, b.g, fx), omix(d.g, e.g, fx), fy);
    r.b = omix(omix(a.b, b.b, fx), omix(d.b, e.b, fx), fy);
    r.a = omix(omix(a.a, b.a, fx), omix(d.a, e.a, fx), fy);
    return r;
}

/* ---- mip chain + textureGrad (first hit only, raytrace.glsl:232-245,299-303) ----------------------
 * DEFINITION (merian's createTextureFromRGBA8 and the hardware sampler are absent): level k+1 is the 2x2 box
 * filter of level k evaluated on LINEAR float texels, ((a + b) + (c + d)) * 0.25 with source indices clamped,
 * level sizes max(1, size >> 1), down to 1x1; no 8-bit requantisation.  LOD as in the Vulkan specification:
 * rho = max over the two screen axes of the length of (ds * w, dt * h), lambda = log2(rho) clamped to the
 * chain; lambda <= 0 (or a NaN footprint) takes the magnification path = tex_sample; otherwise bilinear
 * (min filter LINEAR, quake_node.cpp:697) in the two nearest levels, mixed by frac(lambda). */
static inline uint32_t mip_dim(uint32_t d, uint32_t k) { uint32_t v = d >> k; return v ? v : 1u; }
static void build_mips(const orc_ctx* c, tex_t* t) {
    uint32_t levels = 1; while (mip_dim(t->w, levels - 1) > 1 || mip_dim(t->h, levels - 1) > 1) levels++;
    if (levels > 16) levels = 16;
    t->levels = levels;
    size_t total = 0;
    for (uint32_t k = 1; k < levels; k++) { t->mip_off[k] = total; total += (size_t)mip_dim(t->w, k) * mip_dim(t->h, k) * 4; }
    if (!total) return;
    t->mip = (float*)malloc(total * sizeof(float));
    /* level 0 decoded exactly as a fetch decodes it */
    float* prev = (float*)malloc((size_t)t->w * t->h * 4 * sizeof(float));
    for (uint32_t y = 0; y < t->h; y++) for (uint32_t x = 0; x < t->w; x++) { v4 v = texel(c, t, (int)x, (int)y); float* o = prev + 4 * ((size_t)y * t->w + x); o[0] = v.r; o[1] = v.g; o[2] = v.b; o[3] = v.a; }
    uint32_t pw = t->w, ph = t->h;
    for (uint32_t k = 1; k < levels; k++) {
        uint32_t w = mip_dim(t->w, k), h = mip_dim(t->h, k);
        float* dst = t->mip + t->mip_off[k];
        for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) {
            uint32_t x0 = 2 * x < pw ? 2 * x : pw - 1, x1 = 2 * x + 1 < pw ? 2 * x + 1 : pw - 1;
            uint32_t y0 = 2 * y < ph ? 2 * y : ph - 1, y1 = 2 * y + 1 < ph ? 2 * y + 1 : ph - 1;
            for (int ch = 0; ch < 4; ch++) {
                float a = prev[4 * ((size_t)y0 * pw + x0) + ch], b = prev[4 * ((size_t)y0 * pw + x1) + ch];
                float d = prev[4 * ((size_t)y1 * pw + x0) + ch], e = prev[4 * ((size_t)y1 * pw + x1) + ch];
                dst[4 * ((size_t)y * w + x) + ch] = ((a + b) + (d + e)) * 0.25f;
            }
        }
        if (k == 1) free(prev);
        prev = dst; pw = w; ph = h;
    }
    if (levels == 1) free(prev);
}
static inline v4 mip_texel(const orc_ctx* c, const tex_t* t, uint32_t level, int x, int y) {
    if (level == 0) return texel(c, t, x, y);
    const float* p = t->mip + t->mip_off[level] + 4 * ((size_t)y * mip_dim(t->w, level) + (size_t)x);
    v4 r = {p[0], p[1], p[2], p[3]};
    return r;
}
static v4 tex_bilinear_level(const orc_ctx* c, const tex_t* tx, uint32_t level, float s, float t) {
    int w = (int)mip_dim(tx->w, level), h = (int)mip_dim(tx->h, level);
    int x0, x1, y0, y1; float fx, fy;
    tex_linear_coord(s, (float)w, w, &x0, &x1, &fx);
    tex_linear_coord(t, (float)h, h, &y0, &y1, &fy);
    v4 a = mip_texel(c, tx, level, x0, y0), b = mip_texel(c, tx, level, x1, y0), d = mip_texel(c, tx, level, x0, y1), e = mip_texel(c, tx, level, x1, y1);
    v4 r;
    r.r = omix(omix(a.r, b.r, fx), omix(d.r, e.r, fx), fy);
    r.g = omix(omix(a.g, b.g, fx), omix(d.g, e.g, fx), fy);
    r.b = omix(omix(a.b, b.b, fx), omix(d.b, e.b, fx), fy);
    r.a = omix(omix(a.a, b.a, fx), omix(d.a, e.a, fx), fy);
    return r;
}
static v4 tex_sample_grad(const orc_ctx* c, uint32_t texnum, float s, float t, float dsdx, float dtdx, float dsdy, float dtdy) {
    if (texnum > MAX_GLTEXTURES - 1) texnum = MAX_GLTEXTURES - 1;
    const tex_t* tx = &c->tex[texnum];
    if (!tx->px || tx->levels <= 1) return tex_sample(c, texnum, s, t);
    float fw = (float)tx->w, fh = (float)tx->h;
    float ax = dsdx * fw, ay = dtdx * fh, bx = dsdy * fw, by = dtdy * fh;
    float rx = sqrtf(ax * ax + ay * ay), ry = sqrtf(bx * bx + by * by);
    float rho = rx > ry ? rx : ry; /* NaN compares false: a NaN footprint falls through to level 0 below unless ry is the NaN */
    if (!(rho > 1.0f)) return tex_sample(c, texnum, s, t);
    float lambda = orc_log2(rho);
    float top = (float)(tx->levels - 1);
    if (!(lambda < top)) lambda = top;
    float fl = floorf(lambda);
    uint32_t lo = (uint32_t)fl, hi = lo + 1 < tx->levels ? lo + 1 : lo;
    float f = lambda - fl;
    v4 c0 = tex_bilinear_level(c, tx, lo, s, t);
    if (hi == lo || !(f > 0.0f)) return c0;
    v4 c1 = tex_bilinear_level(c, tx, hi, s, t);
    v4 r = {omix(c0.r, c1.r, f), omix(c0.g, c1.g, f), omix(c0.b, c1.b, f), omix(c0.a, c1.a, f)};
    return r;
}
/* textureGather(tex, st, 3).r : alpha of footprint texel (i0, j1) */
static float tex_gather_alpha_r(const orc_ctx* c, uint32_t texnum, float s, float t) {
    if (texnum > MAX_GLTEXTURES - 1) texnum = MAX_GLTEXTURES - 1;
    const tex_t* tx = &c->tex[texnum];
    if (!tx->px) return 1.0f;
    int x0, x1, y0, y1; float fx, fy;
    tex_linear_coord(s, (float)tx->w, (int)tx->w, &x0, &x1, &fx);
    tex_linear_coord(t, (float)tx->h, (int)tx->h, &y0, &y1, &fy);
    return texel(c, tx, x0, y1).a;
}

/* ---------------------------------------------------------------- intersection */

/* Moeller-Trumbore, front faces only: the geometric normal is cross(v2-v0, v1-v0)
 * (raytrace.glsl:221-223) and a ray hits only if dot(dir, normal) < 0 (raytrace.glsl:73,85). */
static inline int tri_isect(v3 o, v3 d, v3 v0, v3 v1, v3 v2, float* t, float* u, float* v) {
    v3 e1 = vsub(v1, v0), e2 = vsub(v2, v0);
    v3 pv = vcross(d, e2);
    float det = vdot(e1, pv);
    if (!(det < 0.0f)) return 0;
    float inv = 1.0f / det;
    v3 tv = vsub(o, v0);
    float uu = vdot(tv, pv) * inv;
    if (!(uu >= -BARY_EPS) || !(uu <= 1.0f + BARY_EPS)) return 0;
    v3 qv = vcross(tv, e1);
    float vv = vdot(d, qv) * inv;
    if (!(vv >= -BARY_EPS) || !(uu + vv <= 1.0f + BARY_EPS)) return 0;
    float tt = vdot(e2, qv) * inv;
    if (!(tt > 0.0f)) return 0;
    *t = tt; *u = uu; *v = vv;
    return 1;
}

typedef struct { uint32_t key; float t, u, v; } rayhit_t;

static inline const orc_ext_t* ext_of(const orc_ctx* c, uint32_t key) { return &c->geo[key >> 28].ext[key & 0x0fffffffu]; }

/* any-hit confirmation, raytrace.glsl:100-118 */
static int anyhit_confirm(const orc_ctx* c, uint32_t key, float u, float v) {
    const orc_ext_t* e = ext_of(c, key);
    uint32_t flags = e->texnum_fb_flags >> 12, alpha = e->texnum_alpha >> 12;
    if (flags > 0 && flags < 7) return 1;
    if (alpha != 0) return orc_rh((float)(alpha - 1) / 14.0f) >= ALPHA_THRESHOLD;
    float b0 = 1.0f - u - v;
    float s = orc_h2f(e->st[0]) * b0 + orc_h2f(e->st[2]) * u + orc_h2f(e->st[4]) * v;
    float t = orc_h2f(e->st[1]) * b0 + orc_h2f(e->st[3]) * u + orc_h2f(e->st[5]) * v;
    return tex_gather_alpha_r(c, e->texnum_alpha & 0xfffu, s, t) >= ALPHA_THRESHOLD;
}

static inline void consider_range(const orc_ctx* c, const tri_t* tr, v3 o, v3 d, float tmin, float tmax, rayhit_t* best, orc_counters_t* ctr) {
    float t, u, v;
    ctr->tris++;
    if (!tri_isect(o, d, tr->v0, tr->v1, tr->v2, &t, &u, &v)) return;
    if (!(t < tmax) || !(t > tmin)) return;
    if (t < best->t || (t == best->t && tr->key < best->key)) {
        if (!tr->opaque && !anyhit_confirm(c, tr->key, u, v)) return;
        best->t = t; best->u = u; best->v = v; best->key = tr->key;
    }
}
static inline void consider(const orc_ctx* c, const tri_t* tr, v3 o, v3 d, float tmax, rayhit_t* best, orc_counters_t* ctr) {
    float t, u, v;
    ctr->tris++;
    if (!tri_isect(o, d, tr->v0, tr->v1, tr->v2, &t, &u, &v)) return;
    if (!(t < tmax)) return;
    if (t < best->t || (t == best->t && tr->key < best->key)) {
        if (!tr->opaque && !anyhit_confirm(c, tr->key, u, v)) return;
        best->t = t; best->u = u; best->v = v; best->key = tr->key;
    }
}

static inline int slab(const bnode_t* n, v3 o, v3 id, float tbest, float* tnear) {
    float tx0 = (n->bmin.x - o.x) * id.x, tx1 = (n->bmax.x - o.x) * id.x;
    float ty0 = (n->bmin.y - o.y) * id.y, ty1 = (n->bmax.y - o.y) * id.y;
    float tz0 = (n->bmin.z - o.z) * id.z, tz1 = (n->bmax.z - o.z) * id.z;
    float tn = omax(omax(omin(tx0, tx1), omin(ty0, ty1)), omax(omin(tz0, tz1), 0.0f));
    float tf = omin(omin(omax(tx0, tx1), omax(ty0, ty1)), omin(omax(tz0, tz1), tbest * 1.000001f + 1e-6f));
    *tnear = tn;
    return tn <= tf;
}

static void closest_hit(const orc_ctx* c, v3 o, v3 d, float tmax, rayhit_t* best, orc_counters_t* ctr) {
    best->key = 0xffffffffu; best->t = INFINITY; best->u = best->v = 0.0f;
    ctr->rays++;
    if (!c->accel || !c->nodes) {
        for (uint32_t i = 0; i < c->n_tris; i++) consider(c, &c->tris[i], o, d, tmax, best, ctr);
        return;
    }
    v3 id;
    id.x = 1.0f / (fabsf(d.x) > 1e-20f ? d.x : (d.x < 0 ? -1e-20f : 1e-20f));
    id.y = 1.0f / (fabsf(d.y) > 1e-20f ? d.y : (d.y < 0 ? -1e-20f : 1e-20f));
    id.z = 1.0f / (fabsf(d.z) > 1e-20f ? d.z : (d.z < 0 ? -1e-20f : 1e-20f));
    uint32_t stack[128]; int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const bnode_t* n = &c->nodes[stack[--sp]];
        float tn;
        ctr->nodes++;
        if (!slab(n, o, id, omin(best->t, tmax), &tn)) continue;
        if (n->count) { for (uint32_t i = 0; i < n->count; i++) consider(c, &c->tris[n->left + i], o, d, tmax, best, ctr); continue; }
        float t0, t1;
        int h0 = slab(&c->nodes[n->left], o, id, omin(best->t, tmax), &t0);
        int h1 = slab(&c->nodes[n->left + 1], o, id, omin(best->t, tmax), &t1);
        if (h0 && h1) { if (t0 < t1) { stack[sp++] = n->left + 1; stack[sp++] = n->left; } else { stack[sp++] = n->left; stack[sp++] = n->left + 1; } }
        else if (h0) stack[sp++] = n->left; else if (h1) stack[sp++] = n->left + 1;
        if (sp > 120) sp = 120; /* cannot happen for a median split of < 2^60 tris */
    }
}

/* ---------------------------------------------------------------- accel build */

static void tri_bounds(const tri_t* t, v3* mn, v3* mx) {
    mn->x = omin(t->v0.x, omin(t->v1.x, t->v2.x)); mx->x = omax(t->v0.x, omax(t->v1.x, t->v2.x));
    mn->y = omin(t->v0.y, omin(t->v1.y, t->v2.y)); mx->y = omax(t->v0.y, omax(t->v1.y, t->v2.y));
    mn->z = omin(t->v0.z, omin(t->v1.z, t->v2.z)); mx->z = omax(t->v0.z, omax(t->v1.z, t->v2.z));
}
static int g_axis;
static int cmp_centroid(const void* a, const void* b) {
    const tri_t* x = (const tri_t*)a; const tri_t* y = (const tri_t*)b;
    float cx, cy;
    if (g_axis == 0) { cx = x->v0.x + x->v1.x + x->v2.x; cy = y->v0.x + y->v1.x + y->v2.x; }
    else if (g_axis == 1) { cx = x->v0.y + x->v1.y + x->v2.y; cy = y->v0.y + y->v1.y + y->v2.y; }
    else { cx = x->v0.z + x->v1.z + x->v2.z; cy = y->v0.z + y->v1.z + y->v2.z; }
    if (cx < cy) return -1; if (cx > cy) return 1;
    return x->key < y->key ? -1 : (x->key > y->key ? 1 : 0);
}
static void build_rec(orc_ctx* c, uint32_t node, uint32_t first, uint32_t count, float pad) {
    v3 mn = V3(INFINITY, INFINITY, INFINITY), mx = V3(-INFINITY, -INFINITY, -INFINITY);
    v3 cmn = mn, cmx = mx;
    for (uint32_t i = 0; i < count; i++) {
        v3 a, b; tri_bounds(&c->tris[first + i], &a, &b);
        mn.x = omin(mn.x, a.x); mn.y = omin(mn.y, a.y); mn.z = omin(mn.z, a.z);
        mx.x = omax(mx.x, b.x); mx.y = omax(mx.y, b.y); mx.z = omax(mx.z, b.z);
        v3 ce = vadd(a, b);
        cmn.x = omin(cmn.x, ce.x); cmn.y = omin(cmn.y, ce.y); cmn.z = omin(cmn.z, ce.z);
        cmx.x = omax(cmx.x, ce.x); cmx.y = omax(cmx.y, ce.y); cmx.z = omax(cmx.z, ce.z);
    }
    bnode_t* n = &c->nodes[node];
    n->bmin = V3(mn.x - pad, mn.y - pad, mn.z - pad); n->bmax = V3(mx.x + pad, mx.y + pad, mx.z + pad);
    if (count <= 4) { n->left = first; n->count = count; return; }
    v3 ext = vsub(cmx, cmn);
    int axis = ext.x >= ext.y && ext.x >= ext.z ? 0 : (ext.y >= ext.z ? 1 : 2);
    g_axis = axis;
    qsort(&c->tris[first], count, sizeof(tri_t), cmp_centroid);
    uint32_t half = count / 2;
    uint32_t l = c->n_nodes; c->n_nodes += 2;
    n = &c->nodes[node]; n->left = l; n->count = 0;
    build_rec(c, l, first, half, pad);
    build_rec(c, l + 1, first + half, count - half, pad);
}

int orc_commit(orc_ctx* c, int accel) {
    free(c->tris); free(c->nodes); c->tris = NULL; c->nodes = NULL; c->n_nodes = 0;
    uint32_t total = 0;
    for (int s = 0; s < MAX_GEOMETRIES; s++) total += c->geo[s].n_tri;
    c->n_tris = total; c->accel = accel;
    c->tris = (tri_t*)malloc(sizeof(tri_t) * (total ? total : 1));
    uint32_t k = 0; float maxabs = 1.0f;
    for (int s = 0; s < MAX_GEOMETRIES; s++) {
        geo_t* g = &c->geo[s];
        for (uint32_t i = 0; i < g->n_tri; i++) {
            tri_t* t = &c->tris[k++];
            const float* a = g->vtx + 3 * g->idx[3 * i], *b = g->vtx + 3 * g->idx[3 * i + 1], *d = g->vtx + 3 * g->idx[3 * i + 2];
            t->v0 = V3(a[0], a[1], a[2]); t->v1 = V3(b[0], b[1], b[2]); t->v2 = V3(d[0], d[1], d[2]);
            t->key = ((uint32_t)s << 28) | i; t->opaque = (g->flags & ORC_GEO_OPAQUE) ? 1u : 0u;
            for (int j = 0; j < 3; j++) { maxabs = omax(maxabs, fabsf(a[j])); maxabs = omax(maxabs, fabsf(b[j])); maxabs = omax(maxabs, fabsf(d[j])); }
        }
    }
    if (accel && total) {
        c->nodes = (bnode_t*)malloc(sizeof(bnode_t) * (2 * (size_t)total + 2));
        c->n_nodes = 1;
        build_rec(c, 0, 0, total, omax(1e-4f, maxabs * 4.76837158203125e-07f));
    }
    return 0;
}

/* ---------------------------------------------------------------- state */

int orc_connect(orc_ctx* c, uint32_t w, uint32_t h) {
    free_state(c);
    c->W = w; c->H = h;
    size_t px = (size_t)w * h;
    c->mc_total = c->p.mc_adaptive_buffer_size + c->p.mc_static_buffer_size; /* render_mcpg.cpp:59,88 */
    c->mc = (mcstate_t*)calloc(c->mc_total, sizeof(mcstate_t));
    c->lc = (lcvertex_t*)calloc(c->p.lc_buffer_size, sizeof(lcvertex_t));
    c->upd_count = (uint32_t*)calloc(c->mc_total, 4);
    c->upd_rec = (uint32_t*)calloc(c->mc_total, 4);
    size_t segs = px * (size_t)(c->p.spp > 0 ? c->p.spp : 1) * (size_t)(c->p.max_path_length > 1 ? c->p.max_path_length - 1 : 1);
    segs += px * (size_t)(c->p.volume_spp > 0 ? c->p.volume_spp : 0); /* the volume pass queues updates too */
    c->upd_pool_cap = (uint32_t)(segs < c->mc_total ? segs : c->mc_total);
    c->upd_pool = (mcupdate_t*)calloc(c->upd_pool_cap ? c->upd_pool_cap : 1, sizeof(mcupdate_t));
    c->upd_touched = (uint32_t*)calloc(c->upd_pool_cap ? c->upd_pool_cap : 1, 4);
    c->irradiance = (float*)calloc(px, 16);
    c->gb_albedo = (uint16_t*)calloc(px, 8); c->gb_irr = (uint16_t*)calloc(px, 8); c->gb_mv = (uint16_t*)calloc(px, 4);
    c->gbuffer = (gbuf_t*)calloc(px, sizeof(gbuf_t)); c->hits = (chit_t*)calloc(px, sizeof(chit_t));
    c->volume = (float*)calloc(px, 16); c->volume_depth = (uint16_t*)calloc(px, 2); c->prev_volume_depth = (uint16_t*)calloc(px, 2); c->volume_mv = (uint16_t*)calloc(px, 4); c->debug = (uint16_t*)calloc(px, 8);
    { uint32_t gw = c->p.distance_mc_grid_width > 0 ? (uint32_t)c->p.distance_mc_grid_width : 25u; /* render_mcpg.cpp:80-82 */
      c->dist_mc_n = (w / gw + 2) * (h / gw + 2) * 10u; c->dist_mc = (distmc_t*)calloc(c->dist_mc_n, sizeof(distmc_t)); }
    for (int k = 0; k < 2; k++) { c->post_out[k] = (float*)calloc(px, 16); c->post_hist[k] = (float*)calloc(px, 4); c->post_prev_out[k] = (float*)calloc(px, 16); c->post_prev_hist[k] = (float*)calloc(px, 4); }
    c->rs_out = (uint8_t*)calloc(px, 64); c->rs_pong = (uint8_t*)calloc(px, 64); c->rs_prev = (uint8_t*)calloc(px, 64); c->rs_prev_gb = (gbuf_t*)calloc(px, sizeof(gbuf_t));
    c->rs_irr = (float*)calloc(px, 16); c->rs_mom = (float*)calloc(px, 8); c->rs_iteration = 0;
    c->post_prev_gb = (gbuf_t*)calloc(px, sizeof(gbuf_t)); c->post_final = (float*)calloc(px, 16); c->post_first = 1; c->volume_ran = 0;
    c->iteration = 0;
    if (!c->mc || !c->lc || !c->upd_count || !c->upd_rec || !c->upd_pool || !c->irradiance || !c->hits) return -1;
    return 0;
}

const void* orc_output(orc_ctx* c, int which, size_t* bytes) {
    size_t px = (size_t)c->W * c->H;
    switch (which) {
    case ORC_OUT_IRRADIANCE: if (bytes) *bytes = px * 16; return c->irradiance;
    case ORC_OUT_GB_ALBEDO: if (bytes) *bytes = px * 8; return c->gb_albedo;
    case ORC_OUT_GB_IRRADIANCE: if (bytes) *bytes = px * 8; return c->gb_irr;
    case ORC_OUT_GB_MV: if (bytes) *bytes = px * 4; return c->gb_mv;
    case ORC_OUT_GBUFFER: if (bytes) *bytes = px * sizeof(gbuf_t); return c->gbuffer;
    case ORC_OUT_HITS: if (bytes) *bytes = px * sizeof(chit_t); return c->hits;
    case ORC_OUT_VOLUME: if (bytes) *bytes = px * 16; return c->volume;
    case ORC_OUT_VOLUME_DEPTH: if (bytes) *bytes = px * 2; return c->volume_depth;
    case ORC_OUT_VOLUME_MV: if (bytes) *bytes = px * 4; return c->volume_mv;
    case ORC_OUT_DEBUG: if (bytes) *bytes = px * 8; return c->debug;
    }
    return NULL;
}
void* orc_debug_state(orc_ctx* c, int which, size_t* count, size_t* entry_bytes) {
    if (which == 0) { *count = (size_t)c->p.mc_adaptive_buffer_size + c->p.mc_static_buffer_size; *entry_bytes = sizeof(mcstate_t); return c->mc; }
    if (which == 1) { *count = c->p.lc_buffer_size; *entry_bytes = sizeof(lcvertex_t); return c->lc; }
    if (which == 2) { *count = c->dist_mc_n; *entry_bytes = sizeof(distmc_t); return c->dist_mc; }
    *count = 0; *entry_bytes = 0; return NULL;
}
void orc_get_counters(orc_ctx* c, orc_counters_t* out, int reset) { if (out) *out = c->ctr; if (reset) memset(&c->ctr, 0, sizeof c->ctr); }

/* ---------------------------------------------------------------- sky + trace_ray */

typedef struct { const orc_ctx* c; orc_counters_t ctr; uint32_t rng; v3 sun_color; } tls_t;

static inline v3 cam_x(const orc_ctx* c) { return V3(c->u.cam_x[0], c->u.cam_x[1], c->u.cam_x[2]); }

/* learning-write log: one 16-dword record per proposed write (layouts: mq_oracle.h) */
static void llog_append(orc_ctx* c, const uint32_t rec[16]) {
    size_t at = __atomic_fetch_add(&c->llog_n, (size_t)1, __ATOMIC_RELAXED);
    if (at < c->llog_cap) memcpy(c->llog + 16 * at, rec, 64);
}
static void llog_simple(orc_ctx* c, uint32_t kind, uint32_t index, uint32_t a, uint32_t b, uint32_t d, uint32_t e) {
    uint32_t r[16]; memset(r, 0, sizeof r);
    r[0] = a; r[1] = b; r[2] = d; r[3] = e; r[14] = index; r[15] = kind;
    llog_append(c, r);
}

/* raytrace.glsl:25-60 */
static v3 get_sky(const orc_ctx* c, v3 w, v3 sun_color) {
    v3 sun = V3(c->p.sun_w[0], c->p.sun_w[1], c->p.sun_w[2]);
    float a = 0.5f * (1.0f + vdot(sun, w));
    float a2 = a * a;
    float glow = 0.5f * (a2 * a2) + 5.0f * orc_vmf_pdf(w, sun, 3000.0f);
    v3 emm = orc_rh3(vscale(sun_color, orc_rh(glow)));
    const orc_uniform_t* u = &c->u;
    if ((u->sky_lf_ft & 0xffffu) == 0xffffu) { /* classic two-layer quake sky */
        float az = fabsf(w.z);
        float s = 0.5f + w.x / az, t = 0.5f + w.y / az;
        float tm = u->cl_time * 0.12f;
        v4 bck = tex_sample(c, u->sky_rt_bk & 0xffffu, s + 0.5f * tm, t + 0.5f * tm);
        v4 fnt = tex_sample(c, u->sky_rt_bk >> 16, s + tm, t + tm);
        v3 tex = V3(omix(bck.r, fnt.r, fnt.a), omix(bck.g, fnt.g, fnt.a), omix(bck.b, fnt.b, fnt.a));
        emm = orc_rh3(V3(10.0f * (orc_exp2(3.5f * orc_rh(tex.x)) - 1.0f), 10.0f * (orc_exp2(3.5f * orc_rh(tex.y)) - 1.0f), 10.0f * (orc_exp2(3.5f * orc_rh(tex.z)) - 1.0f)));
    } else {
        float ax = fabsf(w.x), ay = fabsf(w.y), az = fabsf(w.z);
        int side_i; /* cubemap_side: dominant axis, +x 0, -x 1, +y 2, -y 3, +z 4, -z 5 */
        if (ax >= ay && ax >= az) side_i = w.x >= 0 ? 0 : 1; else if (ay >= az) side_i = w.y >= 0 ? 2 : 3; else side_i = w.z >= 0 ? 4 : 5;
        uint32_t side = 0; float s = 0, t = 0;
        switch (side_i) {
        case 0: side = u->sky_rt_bk & 0xffffu; s = 0.5f + 0.5f * -w.y / ax; t = 0.5f + 0.5f * -w.z / ax; break;
        case 1: side = u->sky_lf_ft & 0xffffu; s = 0.5f + 0.5f * w.y / ax; t = 0.5f + 0.5f * -w.z / ax; break;
        case 2: side = u->sky_rt_bk >> 16; s = 0.5f + 0.5f * w.x / ay; t = 0.5f + 0.5f * -w.z / ay; break;
        case 3: side = u->sky_lf_ft >> 16; s = 0.5f + 0.5f * -w.x / ay; t = 0.5f + 0.5f * -w.z / ay; break;
        case 4: side = u->sky_up_dn & 0xffffu; s = 0.5f + 0.5f * -w.y / az; t = 0.5f + 0.5f * w.x / az; break;
        default: side = u->sky_up_dn >> 16; s = 0.5f + 0.5f * -w.y / az; t = 0.5f + 0.5f * -w.x / az; break;
        }
        if (side < MAX_GLTEXTURES) { v4 tx = tex_sample(c, side, s, t); emm = orc_rh3(V3(emm.x + orc_rh(tx.r), emm.y + orc_rh(tx.g), emm.z + orc_rh(tx.b))); }
    }
    return emm;
}

static inline v3 rd3(const float* p, uint32_t i) { return V3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }

/* raytrace.glsl:156-311.  throughput / contribution / albedo are float16 in the reference: values
 * are rounded to half at every store. `hit` carries ray origin (pos) and direction (wi) in. */
/* `diff`: first hit only (MERIAN_QUAKE_FIRST_HIT, raytrace.glsl:153-156): directions of the camera rays one pixel
 * to the right and below, diff[0] = r_x, diff[1] = r_y; NULL everywhere else. */
static void trace_ray(tls_t* tl, v3* throughput, v3* contribution, hit_t* hit, v3 sun_color, const v3* diff) {
    const orc_ctx* c = tl->c;
    rayhit_t rh;
    closest_hit(c, hit->pos, hit->wi, T_MAX, &rh, &tl->ctr);
    float tq = rh.key == 0xffffffffu ? T_MAX : rh.t;
    float tr = orc_rh(orc_transmittance(tq, c->u.cam_x[3], c->p.volume_max_t));
    *throughput = orc_rh3(vscale(*throughput, tr));
    hit->roughness = orc_rh(0.6f);
    const orc_ext_t* e = rh.key == 0xffffffffu ? NULL : ext_of(c, rh.key);
    uint32_t flags = e ? (uint32_t)(e->texnum_fb_flags >> 12) : 0;
    if (!e || flags == MAT_FLAGS_SKY) { /* :170-194 */
        v3 sky = get_sky(c, hit->wi, sun_color);
        v3 add = orc_rh3(vmul(*throughput, sky));
        *contribution = e ? orc_rh3(vadd(*contribution, add)) : add;
        hit->albedo = sky;
        hit->pos = vadd(hit->pos, vscale(hit->wi, T_MAX)); hit->prev_pos = hit->pos;
        hit->normal = vneg(hit->wi); hit->enc_geonormal = orc_encode_normal(hit->normal);
        return;
    }
    const geo_t* g = &c->geo[rh.key >> 28]; uint32_t prim = rh.key & 0x0fffffffu;
    float b0 = 1.0f - rh.u - rh.v, b1 = rh.u, b2 = rh.v;
    float st0s = orc_h2f(e->st[0]), st0t = orc_h2f(e->st[1]), st1s = orc_h2f(e->st[2]), st1t = orc_h2f(e->st[3]), st2s = orc_h2f(e->st[4]), st2t = orc_h2f(e->st[5]);
    float s = st0s * b0 + st1s * b1 + st2s * b2, t = st0t * b0 + st1t * b1 + st2t * b2;
    if (flags > 0 && flags < 5) { /* :198-204 warp (DEFINED: quake turbulence in normalised st) */
        float ws = s + 0.125f * orc_sin(8.0f * t + c->u.cl_time), wt = t + 0.125f * orc_sin(8.0f * s + c->u.cl_time);
        s = ws; t = wt;
        if (flags == MAT_FLAGS_WATER) {
            float as = 0.02f * orc_sin(20.0f * t + 1.7f * c->u.cl_time), at = 0.02f * orc_sin(20.0f * s + 1.3f * c->u.cl_time);
            s += as; t += at;
            hit->roughness = orc_rh(0.4f);
        }
    }
    uint32_t i0 = g->idx[3 * prim], i1 = g->idx[3 * prim + 1], i2 = g->idx[3 * prim + 2];
    v3 p0 = rd3(g->vtx, i0), p1 = rd3(g->vtx, i1), p2 = rd3(g->vtx, i2);
    hit->pos = vadd(vadd(vscale(p0, b0), vscale(p1, b1)), vscale(p2, b2));
    v3 du = vsub(p2, p0), dv = vsub(p1, p0);
    hit->normal = vnormalize(vcross(du, dv));
    hit->enc_geonormal = orc_encode_normal(hit->normal);
    hit->prev_pos = vadd(vadd(vscale(rd3(g->prev_vtx, i0), b0), vscale(rd3(g->prev_vtx, i1), b1)), vscale(rd3(g->prev_vtx, i2), b2));
    /* :232-239 texture-space footprint of the pixel.  DEFINITIONS (merian-shaders ray_diff_* / pseudoinverse are
     * absent): Igehy transfer of the differential {dO = 0, dD = r_x} over distance t onto the plane of the hit,
     * dO' = t r - (dot(t r, n) / dot(D, n)) D; pseudoinverse of the 3x2 edge matrix [du dv] by the normal equations;
     * st_dudv holds the HALF-precision edge differences of the texture coordinates (f16mat2, :208-209). */
    float gxs = 0.0f, gxt = 0.0f, gys = 0.0f, gyt = 0.0f;
    const int use_grad = diff && (c->p.enable_albedo_mipmap || c->p.enable_emission_mipmap);
    if (use_grad) {
        v3 n = hit->normal, D = hit->wi;
        float dn = vdot(D, n);
        float a = vdot(du, du), b = vdot(du, dv), cc = vdot(dv, dv);
        float det = a * cc - b * b;
        float d0x = orc_rh(st2s - st0s), d0y = orc_rh(st2t - st0t), d1x = orc_rh(st1s - st0s), d1y = orc_rh(st1t - st0t);
        for (int k = 0; k < 2; k++) {
            v3 dO = vscale(diff[k], rh.t);
            dO = vsub(dO, vscale(D, vdot(dO, n) / dn));
            float pu = vdot(du, dO), pv = vdot(dv, dO);
            float ca = (cc * pu - b * pv) / det, cb = (a * pv - b * pu) / det;
            float gs = d0x * ca + d1x * cb, gt = d0y * ca + d1y * cb;
            if (k == 0) { gxs = gs; gxt = gt; } else { gys = gs; gyt = gt; }
        }
    }
    v4 at = (use_grad && c->p.enable_albedo_mipmap) ? tex_sample_grad(c, e->texnum_alpha & 0xfffu, s, t, gxs, gxt, gys, gyt)
                                                     : tex_sample(c, e->texnum_alpha & 0xfffu, s, t);
    v3 albedo_tex = orc_rh3(V3(orc_pow(orc_rh(at.r), 1.0f / 1.2f), orc_pow(orc_rh(at.g), 1.0f / 1.2f), orc_pow(orc_rh(at.b), 1.0f / 1.2f)));
    if (e->n1_brush == 0xffffffffu) { /* :249-274 */
        uint32_t tn_norm = e->n0_gloss_norm >> 16, tn_gloss = e->n0_gloss_norm & 0xffffu;
        if (tn_norm > 0 && tn_norm < MAX_GLTEXTURES) {
            v4 nt = tex_sample(c, tn_norm, s, t);
            v3 tn = V3((nt.r - 0.5f) * 2.0f, (nt.g - 0.5f) * 2.0f, (nt.b - 0.5f) * 2.0f);
            float d0x = orc_rh(st2s - st0s), d0y = orc_rh(st2t - st0t), d1x = orc_rh(st1s - st0s), d1y = orc_rh(st1t - st0t);
            float det = orc_rh(orc_rh(d0x * d1y) - orc_rh(d1x * d0y));
            if (fabsf(det) > 1e-8f) {
                v3 du2 = vnormalize(vscale(vsub(vscale(du, d1y), vscale(dv, d0y)), 1.0f / det));
                dv = vneg(vnormalize(vscale(vadd(vscale(du, -d1x), vscale(dv, d0x)), 1.0f / det)));
                du = du2;
            }
            v3 gn = hit->normal;
            hit->normal = vnormalize(vadd(vadd(vscale(du, tn.x), vscale(dv, tn.y)), vscale(gn, tn.z)));
            v3 r = vsub(hit->wi, vscale(hit->normal, 2.0f * vdot(hit->wi, hit->normal)));
            if (vdot(r, gn) < 0.0f) hit->normal = vnormalize(vadd(vneg(hit->wi), vnormalize(vsub(r, vscale(gn, vdot(gn, r))))));
        }
        if (tn_gloss > 0 && tn_gloss < MAX_GLTEXTURES) hit->roughness = orc_rh(tex_sample(c, tn_gloss, s, t).r);
    } else if (flags == MAT_FLAGS_SOLID) { /* :275-278 */
        uint32_t a = e->n0_gloss_norm, b = e->n1_brush;
        hit->albedo = orc_rh3(V3(orc_rh((float)(a & 0xff)) / 255.0f, orc_rh((float)((a >> 8) & 0xff)) / 255.0f, orc_rh((float)((a >> 16) & 0xff)) / 255.0f));
        v3 em = orc_ldr_to_hdr(orc_rh3(V3(orc_rh((float)(b & 0xff)) / 255.0f, orc_rh((float)((b >> 8) & 0xff)) / 255.0f, orc_rh((float)((b >> 16) & 0xff)) / 255.0f)));
        *contribution = orc_rh3(vadd(*contribution, orc_rh3(vmul(*throughput, em))));
        return;
    }
    if (flags == MAT_FLAGS_WATERFALL) { /* :288-310 */
        hit->albedo = albedo_tex;
        *contribution = orc_rh3(vadd(*contribution, orc_rh3(vmul(*throughput, hit->albedo))));
    } else if (flags == MAT_FLAGS_SPRITE || flags == MAT_FLAGS_TELE) {
        hit->albedo = orc_ldr_to_hdr(albedo_tex);
        *contribution = orc_rh3(vadd(*contribution, orc_rh3(vmul(*throughput, hit->albedo))));
    } else {
        uint32_t fb = e->texnum_fb_flags & 0xfffu;
        hit->albedo = albedo_tex;
        if (fb > 0 && fb < MAX_GLTEXTURES) {
            v4 ft = (use_grad && c->p.enable_emission_mipmap) ? tex_sample_grad(c, fb, s, t, gxs, gxt, gys, gyt) : tex_sample(c, fb, s, t);
            v3 em = orc_ldr_to_hdr(orc_rh3(V3(ft.r, ft.g, ft.b)));
            if (em.x > 0.0f || em.y > 0.0f || em.z > 0.0f) {
                *contribution = orc_rh3(vadd(*contribution, orc_rh3(vmul(*throughput, em))));
                hit->albedo = em;
            }
        }
    }
}

/* ---------------------------------------------------------------- g-buffer pass */

static void compress_hit(const hit_t* h, chit_t* o) { /* hit.glsl.h:34-43 */
    o->pos[0] = h->pos.x; o->pos[1] = h->pos.y; o->pos[2] = h->pos.z;
    o->mv[0] = orc_f2h(h->pos.x - h->prev_pos.x); o->mv[1] = orc_f2h(h->pos.y - h->prev_pos.y); o->mv[2] = orc_f2h(h->pos.z - h->prev_pos.z);
    o->_pad = 0;
    o->wi = orc_encode_normal(h->wi); o->normal = orc_encode_normal(h->normal); o->enc_geonormal = h->enc_geonormal;
    o->albedo[0] = orc_f2h(h->albedo.x); o->albedo[1] = orc_f2h(h->albedo.y); o->albedo[2] = orc_f2h(h->albedo.z);
    o->roughness = orc_f2h(h->roughness);
}
static void decompress_hit(const chit_t* c, hit_t* h) { /* hit.glsl.h:45-53 */
    h->pos = V3(c->pos[0], c->pos[1], c->pos[2]);
    h->prev_pos = V3(c->pos[0] - orc_h2f(c->mv[0]), c->pos[1] - orc_h2f(c->mv[1]), c->pos[2] - orc_h2f(c->mv[2]));
    h->wi = orc_decode_normal(c->wi); h->normal = orc_decode_normal(c->normal); h->enc_geonormal = c->enc_geonormal;
    h->albedo = V3(orc_h2f(c->albedo[0]), orc_h2f(c->albedo[1]), orc_h2f(c->albedo[2]));
    h->roughness = orc_h2f(c->roughness);
}

/* gbuffer.comp:75-131, with the mip-mapped first-hit texturing of raytrace.glsl:232-245,299-303: r_x / r_y are the
 * directions of the camera rays one pixel to the right / below ("enable albedo mipmap" / "enable emission mipmap") */
static void gbuffer_pixel(tls_t* tl, uint32_t px, uint32_t py) {
    orc_ctx* c = (orc_ctx*)tl->c;
    const orc_uniform_t* u = &c->u;
    size_t idx = (size_t)py * c->W + px;
    float W = (float)c->W, H = (float)c->H, tan_half = c->p.fov_tan_alpha_half;
    v3 up = V3(u->cam_u[0], u->cam_u[1], u->cam_u[2]), fwd = V3(u->cam_w[0], u->cam_w[1], u->cam_w[2]);
    v3 r_x = orc_camera_ray_dir((float)px + 1.0f, (float)py, W, H, up, fwd, tan_half);
    v3 r_y = orc_camera_ray_dir((float)px, (float)py + 1.0f, W, H, up, fwd, tan_half);
    hit_t h; memset(&h, 0, sizeof h);
    h.wi = orc_camera_ray_dir((float)px, (float)py, W, H, up, fwd, tan_half);
    h.pos = cam_x(c);
    v3 incident = V3(0, 0, 0), thr = V3(1, 1, 1);
    v3 sun = c->p.gbuffer_hide_sun ? V3(0, 0, 0) : V3(c->p.sun_color[0], c->p.sun_color[1], c->p.sun_color[2]);
    const v3 diff[2] = {r_x, r_y};
    trace_ray(tl, &thr, &incident, &h, sun, diff);
    uint16_t* oi = c->gb_irr + 4 * idx;
    oi[0] = orc_f2h(incident.x); oi[1] = orc_f2h(incident.y); oi[2] = orc_f2h(incident.z); oi[3] = orc_f2h(1.0f);
    float keep = (incident.x >= 1e-5f || incident.y >= 1e-5f || incident.z >= 1e-5f) ? 0.0f : 1.0f; /* :107 */
    h.albedo = orc_rh3(vmul(orc_rh3(vscale(h.albedo, keep)), thr));
    uint16_t* oa = c->gb_albedo + 4 * idx;
    oa[0] = orc_f2h(h.albedo.x); oa[1] = orc_f2h(h.albedo.y); oa[2] = orc_f2h(h.albedo.z); oa[3] = orc_f2h(1.0f);
    { /* :111-115 */
        v3 old_dir = vsub(h.prev_pos, V3(u->prev_cam_x[0], u->prev_cam_x[1], u->prev_cam_x[2]));
        float opx, opy;
        orc_camera_pixel(old_dir, W, H, V3(u->prev_cam_u[0], u->prev_cam_u[1], u->prev_cam_u[2]), V3(u->prev_cam_w[0], u->prev_cam_w[1], u->prev_cam_w[2]), tan_half, &opx, &opy);
        c->gb_mv[2 * idx] = orc_f2h(opx - (float)px); c->gb_mv[2 * idx + 1] = orc_f2h(opy - (float)py);
    }
    compress_hit(&h, &c->hits[idx]);
    { /* :123-130 */
        v3 gn = orc_decode_normal(h.enc_geonormal);
        v3 cp = cam_x(c);
        float lz = vlen(vsub(cp, h.pos));
        float num = vdot(gn, vsub(h.pos, cp));
        gbuf_t* gb = &c->gbuffer[idx];
        gb->enc_normal = orc_encode_normal(h.normal); gb->linear_z = lz;
        gb->grad_z[0] = orc_f2h(num / vdot(gn, vsub(r_x, h.wi)) - lz); gb->grad_z[1] = orc_f2h(num / vdot(gn, vsub(r_y, h.wi)) - lz);
        gb->vel_z = vlen(vsub(V3(u->prev_cam_x[0], u->prev_cam_x[1], u->prev_cam_x[2]), h.prev_pos)) - lz;
    }
}

/* ---------------------------------------------------------------- grids */

static inline float X(tls_t* tl) { return orc_xorshift(&tl->rng); }

static uint32_t grid_level(int type, float steps, float tan_half, float minw, float power, v3 cam, v3 pos) {
    float w = 2.0f * tan_half * vlen(vsub(cam, pos));
    float lv;
    if (type == 0) lv = steps * orc_log(omax(w, minw) / minw) / orc_log(power); /* mc.glsl:65, light_cache.glsl:16 */
    else lv = steps * orc_pow(omax(w - minw, 0.0f), 1.0f / power);               /* mc.glsl:67, light_cache.glsl:18 */
    return (uint32_t)floorf(lv + 0.5f);
}
static float grid_width(int type, float steps, float minw, float power, uint32_t level) {
    if (type == 0) return minw * orc_pow(power, (float)level / steps); /* mc.glsl:73 */
    return orc_pow((float)level / steps, power) + minw;                /* mc.glsl:75 */
}

/* mc.glsl:83-88 */
static void mc_adaptive_buffer_index(tls_t* tl, v3 pos, v3 normal, uint32_t* index, uint16_t* hash) {
    const orc_params_t* p = &tl->c->p;
    uint32_t level = grid_level(p->adaptive_grid_type, p->mc_adaptive_grid_steps_per_unit_size, p->mc_adaptive_grid_tan_alpha_half, p->mc_adaptive_grid_min_width, p->mc_adaptive_grid_power, cam_x(tl->c), pos);
    float xi = X(tl);
    level += orc_level_jitter(xi); /* mc.glsl:70 */
    float width = grid_width(p->adaptive_grid_type, p->mc_adaptive_grid_steps_per_unit_size, p->mc_adaptive_grid_min_width, p->mc_adaptive_grid_power, level);
    i3 g = orc_grid_idx_interpolate(pos, width, X(tl));
    *index = orc_hash_grid_normal_level(g, normal, level, p->mc_adaptive_buffer_size);
    *hash = (uint16_t)orc_hash2_grid_level(g, level);
}
/* mc.glsl:117-121 */
static void mc_static_buffer_index(tls_t* tl, v3 pos, uint32_t* index, uint16_t* hash) {
    const orc_params_t* p = &tl->c->p;
    i3 g = orc_grid_idx_interpolate(pos, p->mc_static_grid_width, X(tl));
    *index = orc_hash_grid(g, p->mc_static_buffer_size) + p->mc_adaptive_buffer_size;
    *hash = (uint16_t)orc_hash2_grid(g);
}

static inline v3 mc_state_pos(const mcstate_t* s) { return s->sum_w > 0.0f ? vscale(s->w_tgt, 1.0f / s->sum_w) : s->w_tgt; } /* mc.glsl:22 */
static inline v3 mc_state_dir(const mcstate_t* s, v3 pos) { return vnormalize(vsub(mc_state_pos(s), pos)); }                 /* mc.glsl:20 */
static inline float mc_state_mean_cos(const orc_params_t* p, const mcstate_t* s, v3 pos) {                                    /* mc.glsl:24-26 */
    v3 d = vsub(pos, mc_state_pos(s));
    float prior = omax(0.0001f, p->dir_guide_prior / vdot(d, d));
    uint32_t nn = (uint32_t)s->N * (uint32_t)s->N;
    if (p->quirk_n16_wrap) nn &= 0xffffu;
    float n2 = (float)nn;
    return (n2 * oclamp(s->w_cos / s->sum_w, 0.0f, 0.9999999f)) / (n2 + prior);
}
static inline float mc_state_kappa(const orc_params_t* p, const mcstate_t* s, v3 pos) { /* mc.glsl:43-46 */
    float r = mc_state_mean_cos(p, s, pos);
    return (3.0f * r - r * r * r) / (1.0f - r * r);
}
static int mc_light_missing(const orc_params_t* p, const mcstate_t* s, float mc_f, v3 wo, v3 pos) { /* mc.glsl:28-41 */
    if (mc_f > 1e-3f * s->sum_w) return 0;
    float co = vdot(wo, mc_state_dir(s, pos));
    if (co < 0.9f + 0.1f * mc_state_mean_cos(p, s, pos)) return 0;
    return 1;
}
static inline mcstate_t mc_state_new(tls_t* tl) { /* mc.glsl:17 */
    mcstate_t s; memset(&s, 0, sizeof s);
    s.id = (uint32_t)(X(tl) * 4294967296.0f);
    return s;
}
static inline v3 h3(const uint16_t* h) { return V3(orc_h2f(h[0]), orc_h2f(h[1]), orc_h2f(h[2])); }
static void mc_finalize_load(const orc_ctx* c, mcstate_t* s, uint16_t hash, int is_static, v3 pos, v3 normal) { /* mc.glsl:90-96,130-135 */
    int bad = s->sum_w < 0.0f || hash != s->hash;
    if (!bad && is_static) bad = !(vdot(normal, mc_state_dir(s, pos)) > 0.0f);
    if (bad) s->sum_w = 0.0f;
    float k = s->sum_w * (c->u.cl_time - s->T);
    s->w_tgt = vadd(s->w_tgt, vscale(h3(s->mv), k));
}

/* mc.glsl:159-184, 207-222 */
static void mc_state_add_sample(tls_t* tl, const mcstate_t* st, v3 pos, float w, v3 target, v3 target_mv, v3 normal, uint32_t mc_buffer_index) {
    orc_ctx* c = (orc_ctx*)tl->c;
    uint32_t index = mc_buffer_index;
    if (index == 0xffffffffu) { uint16_t h; mc_adaptive_buffer_index(tl, pos, normal, &index, &h); }
    uint32_t r[16];
    r[0] = f2u(pos.x); r[1] = f2u(pos.y); r[2] = f2u(pos.z); r[3] = f2u(w);
    r[4] = f2u(target.x); r[5] = f2u(target.y); r[6] = f2u(target.z); r[7] = st->id;
    r[8] = f2u(normal.x); r[9] = f2u(normal.y); r[10] = f2u(normal.z); r[11] = f2u(c->u.cl_time);
    r[12] = (uint32_t)orc_f2h(target_mv.x) | ((uint32_t)orc_f2h(target_mv.y) << 16); r[13] = (uint32_t)orc_f2h(target_mv.z);
    r[14] = index; r[15] = 1u;
    if (c->p.log_learning && c->p.freeze_learning) llog_append(c, r); /* proposed, never queued: no arrival rank */
    if (c->p.freeze_learning) return;
    uint32_t old = __atomic_fetch_add(&c->upd_count[index], 1u, __ATOMIC_RELAXED);
    if (c->p.log_learning) { r[13] |= (old < 0xffffu ? old : 0xffffu) << 16; llog_append(c, r); } /* with the arrival rank (>= 10: dropped by the cap) */
    if (old >= MAX_UPDATES) { __atomic_fetch_sub(&c->upd_count[index], 1u, __ATOMIC_RELAXED); tl->ctr.mc_updates_dropped++; return; }
    uint32_t rec;
    if (old == 0) {
        rec = __atomic_fetch_add(&c->upd_pool_used, 1u, __ATOMIC_RELAXED);
        if (rec >= c->upd_pool_cap) { __atomic_fetch_sub(&c->upd_count[index], 1u, __ATOMIC_RELAXED); return; }
        c->upd_touched[rec] = index;
        c->upd_pool[rec].T = c->u.cl_time;
        __atomic_store_n(&c->upd_rec[index], rec + 1, __ATOMIC_RELEASE);
    } else {
        uint32_t r1;
        while ((r1 = __atomic_load_n(&c->upd_rec[index], __ATOMIC_ACQUIRE)) == 0) { /* racy mode only */ }
        rec = r1 - 1;
    }
    mcupdate_t* up = &c->upd_pool[rec];
    up->normals[old] = normal;
    up->mv[old][0] = orc_f2h(target_mv.x); up->mv[old][1] = orc_f2h(target_mv.y); up->mv[old][2] = orc_f2h(target_mv.z);
    up->ids[old] = st->id; up->targets[old] = target; up->weights[old] = w; up->positions[old] = pos;
    tl->ctr.mc_updates_accepted++;
}

/* ---------------------------------------------------------------- light cache */

static void lc_address(tls_t* tl, uint32_t level, v3 pos, v3 normal, uint32_t* idx, uint32_t* chk) {
    const orc_params_t* p = &tl->c->p;
    float width = grid_width(p->lc_grid_type, p->lc_grid_steps_per_unit_size, p->lc_grid_min_width, p->lc_grid_power, level);
    i3 g = orc_grid_idx_interpolate(pos, width, X(tl)); /* light_cache.glsl:29 */
    *idx = orc_hash_grid_normal_level(g, normal, level, p->lc_buffer_size);
    *chk = orc_hash2_grid_level(g, level);
}
static inline int h_bad(uint16_t h) { return (h & 0x7c00u) == 0x7c00u; } /* inf or nan */
/* light_cache.glsl:31-45 */
static void light_cache_get_level(tls_t* tl, v3* irr, uint16_t* N, uint32_t level, v3 pos, v3 normal) {
    uint32_t idx, chk;
    lc_address(tl, level, pos, normal, &idx, &chk);
    const lcvertex_t* v = &tl->c->lc[idx];
    tl->ctr.lc_touches++;
    if (v->hash == chk && !h_bad(v->irr[0]) && !h_bad(v->irr[1]) && !h_bad(v->irr[2])) { *irr = h3(v->irr); *N = v->N; }
    else { *irr = V3(0, 0, 0); *N = 0; }
}
static inline uint32_t lc_level(const orc_ctx* c, v3 pos) {
    const orc_params_t* p = &c->p;
    return grid_level(p->lc_grid_type, p->lc_grid_steps_per_unit_size, p->lc_grid_tan_alpha_half, p->lc_grid_min_width, p->lc_grid_power, cam_x(c), pos);
}
/* light_cache.glsl:47-52 */
static v3 light_cache_get(tls_t* tl, v3 pos, v3 normal) {
    v3 irr; uint16_t N;
    light_cache_get_level(tl, &irr, &N, lc_level(tl->c, pos), pos, normal);
    return irr;
}
/* light_cache.glsl:54-84 */
static void light_cache_update(tls_t* tl, v3 pos, v3 normal, v3 irr) {
    orc_ctx* c = (orc_ctx*)tl->c;
    uint32_t level = lc_level(c, pos), idx, chk;
    lc_address(tl, level, pos, normal, &idx, &chk);
    lcvertex_t* cell = &c->lc[idx];
    tl->ctr.lc_touches++;
    if (c->p.freeze_learning) { /* every RNG draw and computation of the update, none of its stores */
        if (c->u.frame == 0u) return; /* frame 0: the lock word (0) equals the frame number, the update is cancelled before any draw */
        lcvertex_t v = *cell;
        int rekey = v.hash != chk || h_bad(v.irr[0]) || h_bad(v.irr[1]) || h_bad(v.irr[2]);
        if (rekey) {
            v3 ci; uint16_t cn; light_cache_get_level(tl, &ci, &cn, level + 1, pos, normal);
            v.irr[0] = orc_f2h(ci.x); v.irr[1] = orc_f2h(ci.y); v.irr[2] = orc_f2h(ci.z); v.N = cn;
        }
        if (c->p.log_learning) {
            v.N = (uint16_t)(v.N + 1 < LIGHT_CACHE_MAX_N ? v.N + 1 : LIGHT_CACHE_MAX_N);
            float a = omax(1.0f / (float)v.N, LIGHT_CACHE_MIN_ALPHA);
            v3 cur = h3(v.irr);
            llog_simple(c, 2u, idx, chk, (uint32_t)rekey,
                        (uint32_t)orc_f2h(omix(cur.x, irr.x, a)) | ((uint32_t)orc_f2h(omix(cur.y, irr.y, a)) << 16),
                        (uint32_t)orc_f2h(omix(cur.z, irr.z, a)) | ((uint32_t)v.N << 16));
        }
        return;
    }
    uint32_t old = __atomic_exchange_n(&cell->lock, c->u.frame, __ATOMIC_ACQ_REL);
    if (old == c->u.frame) { __atomic_fetch_add(&cell->cancel, 1u, __ATOMIC_RELAXED); return; }
    lcvertex_t v = *cell;
    if (v.hash != chk || h_bad(v.irr[0]) || h_bad(v.irr[1]) || h_bad(v.irr[2])) {
        v3 ci; uint16_t cn;
        light_cache_get_level(tl, &ci, &cn, level + 1, pos, normal);
        v.irr[0] = orc_f2h(ci.x); v.irr[1] = orc_f2h(ci.y); v.irr[2] = orc_f2h(ci.z); v.N = cn;
        v.hash = chk;
    }
    v.N = (uint16_t)(v.N + 1 < LIGHT_CACHE_MAX_N ? v.N + 1 : LIGHT_CACHE_MAX_N);
    float a = omax(1.0f / (float)v.N, LIGHT_CACHE_MIN_ALPHA);
    v3 cur = h3(v.irr);
    v.irr[0] = orc_f2h(omix(cur.x, irr.x, a)); v.irr[1] = orc_f2h(omix(cur.y, irr.y, a)); v.irr[2] = orc_f2h(omix(cur.z, irr.z, a));
    cell->hash = v.hash; cell->irr[0] = v.irr[0]; cell->irr[1] = v.irr[1]; cell->irr[2] = v.irr[2]; cell->N = v.N;
    if (c->p.log_learning) llog_simple(c, 2u, idx, chk, 0u, (uint32_t)v.irr[0] | ((uint32_t)v.irr[1] << 16), (uint32_t)v.irr[2] | ((uint32_t)v.N << 16));
    __atomic_fetch_add(&cell->ok, 1u, __ATOMIC_RELAXED);
    __atomic_store_n(&cell->lock, 0u, __ATOMIC_RELEASE);
}

/* ---------------------------------------------------------------- surface estimator */

#define MAX_MC_SAMPLES 32
static inline int finite3(v3 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }

/* mcpg.comp:212-277: the nine debug views, evaluated with the pixel's RNG state after its samples.
 * DEFINITIONS: grid_idx_closest(p, w) = floor(p / w + 0.5); oklch_to_rgb / acos / exp as in orc_math.h. */
static void debug_view(tls_t* tl, size_t idx, v3 irr, float second_moment, const chit_t* fh) {
    orc_ctx* c = (orc_ctx*)tl->c;
    const orc_params_t* p = &c->p;
    hit_t h; decompress_hit(fh, &h);
    v3 out = V3(0, 0, 0);
    mcstate_t st; memset(&st, 0, sizeof st);
    const int sel = p->debug_output_selector;
    if (sel == 1 || sel == 2 || sel == 6 || sel == 7 || sel == 8) { /* mc_adaptive_load, mc.glsl:98-103 */
        uint32_t bi; uint16_t hash;
        mc_adaptive_buffer_index(tl, h.pos, h.normal, &bi, &hash);
        st = c->mc[bi];
        mc_finalize_load(c, &st, hash, 0, h.pos, h.normal);
    }
    switch (sel) {
    case 0: out = vscale(light_cache_get(tl, h.pos, h.normal), 5.0f); break;
    case 1: out = V3(st.sum_w * 0.1f, st.sum_w * 0.1f, st.sum_w * 0.1f); break;
    case 2: { v3 d = mc_state_dir(&st, h.pos); out = V3((d.x + 1.0f) / 2.0f, (d.y + 1.0f) / 2.0f, (d.z + 1.0f) / 2.0f); break; }
    case 3: {
        uint32_t level = grid_level(p->adaptive_grid_type, p->mc_adaptive_grid_steps_per_unit_size, p->mc_adaptive_grid_tan_alpha_half, p->mc_adaptive_grid_min_width, p->mc_adaptive_grid_power, cam_x(c), h.pos);
        float width = grid_width(p->adaptive_grid_type, p->mc_adaptive_grid_steps_per_unit_size, p->mc_adaptive_grid_min_width, p->mc_adaptive_grid_power, level);
        i3 g = orc_grid_idx_interpolate(h.pos, width, 0.5f); /* closest cell */
        uint32_t seed = orc_hash2_grid(g);
        float x0 = orc_xorshift(&seed), x1 = orc_xorshift(&seed);
        float L = orc_exp(0.001f * -vlen(vsub(h.pos, cam_x(c)))) * (0.0f + x0 * 1.0f) + 0.2f;
        out = orc_oklch_to_rgb(V3(L, 0.2f, 6.28318548202514648f * x1));
        break; }
    case 4: out = irr; break;
    case 5: out = V3(orc_luminance(irr), second_moment, 0.0f); break;
    case 6: { float v = st.sum_w > 0.0f ? 1.0f - oclamp(orc_acos(st.w_cos / st.sum_w) * ORC_INV_PI, 0.0f, 1.0f) : 0.0f; out = V3(v, v, v); break; }
    case 7: { float v = (float)st.N / (float)ML_MAX_N; out = V3(v, v, v); break; }
    case 8: out = V3(orc_h2f(st.mv[0]), orc_h2f(st.mv[1]), orc_h2f(st.mv[2])); break;
    default: break;
    }
    uint16_t* d = c->debug + 4 * idx;
    d[0] = orc_f2h(out.x); d[1] = orc_f2h(out.y); d[2] = orc_f2h(out.z); d[3] = orc_f2h(1.0f);
}

/* mcpg.comp:39-210 */
static void mcpg_pixel(tls_t* tl, uint32_t px, uint32_t py) {
    orc_ctx* c = (orc_ctx*)tl->c;
    const orc_params_t* p = &c->p;
    size_t idx = (size_t)py * c->W + px;
    tl->rng = orc_pcg4d16(px, py, c->u.frame, p->seed); /* :40 */
    float second_moment = 0.0f; v3 irr = V3(0, 0, 0);
    const chit_t* fh = &c->hits[idx];
    v3 sun = V3(p->sun_color[0], p->sun_color[1], p->sun_color[2]);
    int refmode = p->reference_mode || p->surf_bsdf_p == 1.0f; /* render_mcpg.cpp:139-140 */
    int K = p->mc_samples < MAX_MC_SAMPLES ? p->mc_samples : MAX_MC_SAMPLES;
    if (orc_h2f(fh->albedo[0]) >= 1e-7f || orc_h2f(fh->albedo[1]) >= 1e-7f || orc_h2f(fh->albedo[2]) >= 1e-7f) /* :44 */
    for (int s = 0; s < p->spp; s++) {
        hit_t cur; decompress_hit(fh, &cur);
        v3 thr = V3(1, 1, 1), f = V3(0, 0, 0); float pp = 1.0f;
        for (int segment = 1; segment < p->max_path_length; segment++) {
            v3 wo; float wodotn, wo_p = 0.0f;
            float alpha = orc_roughness_to_alpha(cur.roughness);
            mcstate_t mc_state; memset(&mc_state, 0, sizeof mc_state);
            uint32_t mc_buffer_index = 0xffffffffu; float score_sum = 0.0f;
            tl->ctr.segments++;
            if (refmode) { /* :59-64 */
                float x0 = X(tl), x1 = X(tl), x2 = X(tl);
                wo = orc_bsdf_sample(cur.wi, cur.normal, alpha, x0, x1, x2);
                wodotn = vdot(wo, cur.normal);
                if (wodotn <= 1e-3f || vdot(wo, orc_decode_normal(cur.enc_geonormal)) <= 1e-3f) break;
                wo_p = orc_bsdf_pdf(cur.wi, wo, cur.normal, alpha);
            } else { /* :67-137 */
                float scores[MAX_MC_SAMPLES] = {0}; v3 vdir[MAX_MC_SAMPLES]; float vk[MAX_MC_SAMPLES] = {0};
                memset(vdir, 0, sizeof vdir);
                tl->ctr.guided_segments++;
                for (int i = 0; i < K; i++) {
                    int adaptive = X(tl) < p->mc_samples_adaptive_prob;
                    uint32_t bi; uint16_t hash;
                    v3 lp = s == 0 ? cur.prev_pos : cur.pos;
                    if (adaptive) mc_adaptive_buffer_index(tl, lp, cur.normal, &bi, &hash);
                    else mc_static_buffer_index(tl, lp, &bi, &hash);
                    mcstate_t st = c->mc[bi];
                    tl->ctr.mc_state_reads++;
                    mc_finalize_load(c, &st, hash, !adaptive, cur.pos, cur.normal);
                    score_sum += st.sum_w;
                    v3 d = mc_state_dir(&st, cur.pos); float kk = mc_state_kappa(p, &st, cur.pos);
                    if (X(tl) < st.sum_w / score_sum) { /* NaN compares false */
                        mc_state = st; mc_buffer_index = bi;
                        vdir[i] = vdir[0]; vk[i] = vk[0]; scores[i] = scores[0];
                        scores[0] = st.sum_w; vdir[0] = d; vk[0] = kk;
                    } else { scores[i] = st.sum_w; vdir[i] = d; vk[i] = kk; }
                }
                if (score_sum == 0.0f || X(tl) < p->surf_bsdf_p) { /* :113-117 */
                    float x0 = X(tl), x1 = X(tl), x2 = X(tl);
                    wo = orc_bsdf_sample(cur.wi, cur.normal, alpha, x0, x1, x2);
                    mc_state = mc_state_new(tl);
                    mc_buffer_index = 0xffffffffu;
                } else {
                    float x0 = X(tl), x1 = X(tl);
                    wo = orc_vmf_sample(vdir[0], vk[0], x0, x1);
                }
                wodotn = vdot(wo, cur.normal);
                if (wodotn <= 1e-3f || vdot(wo, orc_decode_normal(cur.enc_geonormal)) <= 1e-3f) break; /* :124 */
                if (score_sum > 0.0f) {
                    for (int i = 0; i < K; i++) wo_p += scores[i] * orc_vmf_pdf(wo, vdir[i], vk[i]);
                    wo_p /= score_sum;
                }
                wo_p = (score_sum > 0.0f ? p->surf_bsdf_p : 1.0f) * orc_bsdf_pdf(cur.wi, wo, cur.normal, alpha) + (1.0f - p->surf_bsdf_p) * wo_p; /* :135 */
            }
            hit_t next; memset(&next, 0, sizeof next);
            next.wi = wo;
            next.pos = vsub(cur.pos, vscale(cur.wi, 1e-3f)); /* :144 */
            v3 incident = V3(0, 0, 0), throughput = V3(1, 1, 1);
            trace_ray(tl, &throughput, &incident, &next, sun, NULL);
            v3 lc_incident; /* :149 */
            if ((incident.x > 0.0f || incident.y > 0.0f || incident.z > 0.0f) || (p->use_light_cache_tail == 0 && p->max_path_length == 2)) lc_incident = incident;
            else lc_incident = orc_rh3(vmul(throughput, light_cache_get(tl, next.pos, next.normal)));
            float bsdf = orc_bsdf_times_wodotn(cur.wi, wo, cur.normal, alpha, 0.02f); /* :153 */
            thr = vscale(thr, bsdf);
            if (p->use_light_cache_tail) f = vmul(thr, segment < p->max_path_length - 1 ? incident : lc_incident);
            else f = vmul(thr, incident);
            pp *= wo_p;
            thr = vmul(thr, throughput);
            if (!refmode) { /* :165-181 */
                float mc_f = orc_luminance(vscale(vscale(lc_incident, bsdf), 1.0f / wo_p));
                if (isfinite(mc_f)) {
                    float den = p->quirk_lc_max_wo_p ? omax(wo_p, 10.0f) : omax(wo_p, 1e-6f);
                    light_cache_update(tl, cur.pos, cur.normal, vscale(vscale(vmul(lc_incident, vscale(cur.albedo, ORC_INV_PI)), wodotn), 1.0f / den));
                    if (X(tl) * score_sum < mc_f * (float)p->mc_samples) {
                        v3 mv = orc_rh3(vscale(vsub(next.pos, next.prev_pos), 1.0f / c->u.cam_w[3]));
                        mc_state_add_sample(tl, &mc_state, cur.pos, mc_f, next.pos, mv, cur.normal, mc_buffer_index);
                    } else if (p->mc_fast_recovery && mc_buffer_index != 0xffffffffu && mc_light_missing(p, &mc_state, mc_f, wo, cur.pos)) {
                        if (p->log_learning) llog_simple(c, 3u, mc_buffer_index, 0u, 0u, 0u, 0u);
                        if (!p->freeze_learning) c->mc[mc_buffer_index].sum_w = 0.0f; /* :177 */
                    }
                }
            }
            thr = vmul(thr, next.albedo); /* :184 */
            cur = next;
            if ((thr.x < 1e-7f && thr.y < 1e-7f && thr.z < 1e-7f) || (f.x > 1e-7f || f.y > 1e-7f || f.z > 1e-7f)) break;
        }
        v3 contrib = vscale(f, 1.0f / pp); /* :193 */
        if (finite3(contrib)) { irr = vadd(irr, contrib); float l = orc_luminance(contrib); second_moment += l * l; }
    }
    if (p->spp > 0) { float inv = 1.0f / (float)p->spp; irr = vscale(irr, inv); second_moment *= inv; }
    float* o = c->irradiance + 4 * idx;
    o[0] = irr.x; o[1] = irr.y; o[2] = irr.z; o[3] = second_moment;
    if (p->debug_output_connected) debug_view(tl, idx, irr, second_moment, fh);
}

/* ---------------------------------------------------------------- update application */

/* compute_updates.comp:41-54 */
static void mc_update(mcstate_t* s, v3 pos, float w, v3 target, const uint16_t* mv) {
    s->N = (uint16_t)(s->N + 1 < ML_MAX_N ? s->N + 1 : ML_MAX_N);
    float alpha = omax(1.0f / (float)s->N, ML_MIN_ALPHA);
    s->sum_w = omix(s->sum_w, w, alpha);
    s->w_tgt = V3(omix(s->w_tgt.x, w * target.x, alpha), omix(s->w_tgt.y, w * target.y, alpha), omix(s->w_tgt.z, w * target.z, alpha));
    /* :51 reads the state after :49-50 assigned sum_w and w_tgt */
    float co = omax(0.0f, vdot(vnormalize(vsub(target, pos)), mc_state_dir(s, pos)));
    s->w_cos = omin(omix(s->w_cos, w * co, alpha), s->sum_w);
    s->mv[0] = mv[0]; s->mv[1] = mv[1]; s->mv[2] = mv[2];
}

/* compute_updates.comp:56-124 for one slot */
static void apply_slot(tls_t* tl, uint32_t slot) {
    orc_ctx* c = (orc_ctx*)tl->c;
    uint32_t count = c->upd_count[slot];
    if (!count) return;
    if (count > MAX_UPDATES) count = MAX_UPDATES; /* SURVEY D.2 */
    const mcupdate_t* up = &c->upd_pool[c->upd_rec[slot] - 1];
    tl->rng = orc_pcg4d16(slot, 0, c->u.frame, c->p.seed); /* :62 */
    mcstate_t mc_state = c->mc[slot];
#define TOUCH(cell) do { if (c->touch) { if (c->touch_n < c->touch_cap) { c->touch[2 * c->touch_n] = slot; c->touch[2 * c->touch_n + 1] = (cell); } c->touch_n++; } } while (0)
    TOUCH(slot);
    float sum = 0.0f; v3 pos = V3(0, 0, 0), normal = V3(0, 0, 0);
    mcstate_t new_state; memset(&new_state, 0, sizeof new_state); int picked = 0;
    for (uint32_t i = 0; i < count; i++) {
        mcstate_t st = mc_state;
        if (mc_state.id != up->ids[i]) st = mc_state_new(tl);
        mc_update(&st, up->positions[i], up->weights[i], up->targets[i], up->mv[i]);
        if (mc_state.id == st.id) mc_state = st;
        sum += st.sum_w;
        if (X(tl) < st.sum_w / sum) { new_state = st; pos = up->positions[i]; normal = up->normals[i]; picked = 1; }
    }
    new_state.T = up->T;
    if (picked) for (uint32_t i = 0; i < count; i++) {
        { uint32_t bi; uint16_t hash; mc_static_buffer_index(tl, pos, &bi, &hash); TOUCH(bi);
          new_state.hash = hash; mcstate_t old = c->mc[bi];
          if (old.id == new_state.id || X(tl) < new_state.sum_w / (new_state.sum_w + old.sum_w)) c->mc[bi] = new_state; }
        { uint32_t bi; uint16_t hash; mc_adaptive_buffer_index(tl, pos, normal, &bi, &hash); TOUCH(bi);
          new_state.hash = hash; mcstate_t old = c->mc[bi];
          if (old.id == new_state.id || X(tl) < new_state.sum_w / (new_state.sum_w + old.sum_w)) c->mc[bi] = new_state; }
    }
    c->upd_count[slot] = 0; c->upd_rec[slot] = 0;
#undef TOUCH
}
static int cmp_u32(const void* a, const void* b) { uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : x > y; }


/* ---------------------------------------------------------------- volume pass */

/* mc_distance.glsl:10-16 */
static void distance_normal_dist(const distmc_t* s, float* mu, float* sigma) {
    float den = s->sum_w > 0.0f ? s->sum_w : 1.0f;
    float m0 = s->m0 / den, m1 = s->m1 / den;
    float sg = sqrtf(omax(m1 - m0 * m0, 0.0f));
    float n2 = (float)(s->N * s->N);
    *mu = m0; *sigma = (n2 * sg + 0.2f) / (n2 + 0.2f);
}
/* mc_distance.glsl:19-27 */
static void distance_add_sample(distmc_t* s, float dist, float w) {
    s->N = s->N + 1 < DISTANCE_ML_MAX_N ? s->N + 1 : DISTANCE_ML_MAX_N;
    float alpha = omax(1.0f / (float)s->N, DISTANCE_ML_MIN_ALPHA);
    s->sum_w = omix(s->sum_w, w, alpha);
    s->m0 = omix(s->m0, w * dist, alpha); s->m1 = omix(s->m1, w * (dist * dist), alpha);
}
/* mc_distance.glsl:29-44: address of a random state of the stochastically chosen grid vertex */
static uint32_t distance_mc_index(tls_t* tl, float px, float py, uint32_t grid_max_x) {
    const orc_params_t* p = &tl->c->p;
    float inv = 1.0f / (float)p->distance_mc_grid_width;
    float xi = X(tl);
    int gx = (int)floorf(px * inv + xi), gy = (int)floorf(py * inv + xi);
    uint32_t st = (uint32_t)(X(tl) * (float)p->distance_mc_vertex_state_count);
    uint32_t v = (uint32_t)gx + (grid_max_x + 1u) * (uint32_t)gy;
    uint32_t idx = v * 10u + st; /* MAX_DISTANCE_MC_VERTEX_STATE_COUNT states per vertex (config.h:22) */
    return idx < tl->c->dist_mc_n ? idx : tl->c->dist_mc_n - 1;
}

/* volume_forward_project.comp:17-53 */
static void forward_project_pixel(orc_ctx* c, uint32_t px, uint32_t py) {
    const orc_uniform_t* u = &c->u;
    float W = (float)c->W, H = (float)c->H, th = c->p.fov_tan_alpha_half;
    float prev_depth = orc_h2f(c->prev_volume_depth[(size_t)py * c->W + px]);
    v3 pwi = orc_camera_ray_dir((float)px, (float)py, W, H, V3(u->prev_cam_u[0], u->prev_cam_u[1], u->prev_cam_u[2]), V3(u->prev_cam_w[0], u->prev_cam_w[1], u->prev_cam_w[2]), th);
    v3 ppos = vadd(V3(u->prev_cam_x[0], u->prev_cam_x[1], u->prev_cam_x[2]), vscale(pwi, prev_depth));
    float fx, fy;
    orc_camera_pixel(vsub(ppos, cam_x(c)), W, H, V3(u->cam_u[0], u->cam_u[1], u->cam_u[2]), V3(u->cam_w[0], u->cam_w[1], u->cam_w[2]), th, &fx, &fy);
    float rx = floorf(fx + 0.5f), ry = floorf(fy + 0.5f);
    if (!(rx >= 0.0f && ry >= 0.0f && rx < W && ry < H)) return;
    if (prev_depth < 50.0f) return;
    int nx = (int)rx, ny = (int)ry;
    size_t o = 2 * ((size_t)ny * c->W + (size_t)nx);
    c->volume_mv[o] = orc_f2h((float)px - rx); c->volume_mv[o + 1] = orc_f2h((float)py - ry);
}

/* volume.comp:34-238 */
static void volume_pixel(tls_t* tl, uint32_t px, uint32_t py) {
    orc_ctx* c = (orc_ctx*)tl->c;
    const orc_params_t* p = &c->p;
    const orc_uniform_t* u = &c->u;
    size_t idx = (size_t)py * c->W + px;
    const uint32_t grid_max_x = c->W / (uint32_t)p->distance_mc_grid_width + 1u;
    const float mu_t = u->cam_x[3];
    const v3 mu_s = V3(u->prev_cam_x[3], u->prev_cam_w[3], u->prev_cam_u[3]);
    tl->rng = orc_pcg4d16(px, py, u->frame, p->seed);
    v3 irr = V3(0, 0, 0); float second_moment = 0.0f;
    const gbuf_t* gb = &c->gbuffer[idx];
    const float linear_z = gb->linear_z;
    const v3 first_n = orc_decode_normal(gb->enc_normal);
    const v3 first_wi = orc_camera_ray_dir((float)px, (float)py, (float)c->W, (float)c->H, V3(u->cam_u[0], u->cam_u[1], u->cam_u[2]), V3(u->cam_w[0], u->cam_w[1], u->cam_w[2]), p->fov_tan_alpha_half);
    const float mvx = orc_h2f(c->volume_mv[2 * idx]), mvy = orc_h2f(c->volume_mv[2 * idx + 1]);
    const v3 sun = V3(p->sun_color[0], p->sun_color[1], p->sun_color[2]);
    int KD = p->distance_mc_samples < MAX_MC_SAMPLES ? p->distance_mc_samples : MAX_MC_SAMPLES;
    int K = p->mc_samples < MAX_MC_SAMPLES ? p->mc_samples : MAX_MC_SAMPLES;
    const float tmax_v = omin(linear_z, p->volume_max_t);
    if (mu_t > 0.0f)
    for (int s = 0; s < p->volume_spp; s++) {
        float pd = 0.0f, t = 0.0f;
        distmc_t dstate; memset(&dstate, 0, sizeof dstate);
        float dist_score_sum = 0.0f;
        { /* camera-distance sampling, :58-104 */
            const float xi_max = orc_transmittance_xi_max(tmax_v, mu_t);
            float scores[MAX_MC_SAMPLES] = {0}, nmu[MAX_MC_SAMPLES] = {0}, nsg[MAX_MC_SAMPLES] = {0};
            for (int i = 0; i < KD; i++) {
                distmc_t st;
                if (s == 0) {
                    float qx = oclamp((float)px + mvx, 0.0f, (float)c->W - 1.0f), qy = oclamp((float)py + mvy, 0.0f, (float)c->H - 1.0f);
                    st = c->dist_mc[distance_mc_index(tl, qx, qy, grid_max_x)];
                    distance_normal_dist(&st, &nmu[i], &nsg[i]);
                    nmu[i] -= vdot(vsub(cam_x(c), V3(u->prev_cam_x[0], u->prev_cam_x[1], u->prev_cam_x[2])), first_wi);
                } else {
                    st = c->dist_mc[distance_mc_index(tl, (float)px, (float)py, grid_max_x)];
                    distance_normal_dist(&st, &nmu[i], &nsg[i]);
                }
                scores[i] = st.sum_w * (st.sum_w > 0.0f ? 1.0f : 0.0f) * (nmu[i] < linear_z ? 1.0f : 0.0f);
                dist_score_sum += scores[i];
                if (X(tl) < scores[i] / dist_score_sum) {
                    dstate = st;
                    float x0 = X(tl), x1 = X(tl);
                    t = orc_sample_normal_box_muller(nmu[i], nsg[i], x0, x1);
                }
            }
            if (p->dist_guide_p < X(tl) || dist_score_sum == 0.0f) t = orc_transmittance_sample2(mu_t, X(tl), xi_max);
            else if (t >= tmax_v || t <= 0.0f) continue;
            if (dist_score_sum > 0.0f) {
                for (int i = 0; i < KD; i++) pd += scores[i] * orc_sample_normal_pdf(nmu[i], nsg[i], t);
                pd /= dist_score_sum;
            }
            pd = (dist_score_sum > 0.0f ? (1.0f - p->dist_guide_p) : 1.0f) * orc_transmittance_pdf2(t, mu_t, xi_max) + p->dist_guide_p * pd;
        }
        v3 cur_pos = vadd(cam_x(c), vscale(first_wi, t)); /* :109 */
        v3 cur_normal = first_n;                          /* :111 (unused by the lookups below, kept for clarity) */
        (void)cur_normal;
        v3 wo; float wo_p = 0.0f, score_sum = 0.0f;
        mcstate_t mc_state; memset(&mc_state, 0, sizeof mc_state);
        uint32_t mc_buffer_index = 0xffffffffu;
        { /* :119-177 */
            float scores[MAX_MC_SAMPLES] = {0}, vk[MAX_MC_SAMPLES] = {0}; v3 vdir[MAX_MC_SAMPLES]; memset(vdir, 0, sizeof vdir);
            for (int i = 0; i < K; i++) {
                int adaptive = X(tl) < p->mc_samples_adaptive_prob;
                uint32_t bi; uint16_t hash;
                if (adaptive) mc_adaptive_buffer_index(tl, cur_pos, vneg(first_wi), &bi, &hash);
                else mc_static_buffer_index(tl, cur_pos, &bi, &hash);
                mcstate_t st = c->mc[bi];
                tl->ctr.mc_state_reads++;
                { /* mc_adaptive_finalize_load / two-argument mc_static_finalize_load, mc.glsl:90-96,123-128 */
                    if (st.sum_w < 0.0f || hash != st.hash) st.sum_w = 0.0f;
                    float k = st.sum_w * (u->cl_time - st.T);
                    st.w_tgt = vadd(st.w_tgt, vscale(h3(st.mv), k));
                }
                score_sum += st.sum_w;
                v3 d = mc_state_dir(&st, cur_pos); float kk = mc_state_kappa(p, &st, cur_pos);
                if (X(tl) < st.sum_w / score_sum) {
                    mc_state = st; mc_buffer_index = bi;
                    vdir[i] = vdir[0]; vk[i] = vk[0]; scores[i] = scores[0];
                    scores[0] = st.sum_w; vdir[0] = d; vk[0] = kk;
                } else { scores[i] = st.sum_w; vdir[i] = d; vk[i] = kk; }
            }
            if (score_sum == 0.0f || X(tl) < p->volume_phase_p) {
                float x0 = X(tl), x1 = X(tl);
                wo = orc_draine_sample(x0, x1, first_wi, p->draine_g, p->draine_a);
                mc_state = mc_state_new(tl);
                mc_buffer_index = 0xffffffffu;
            } else {
                float x0 = X(tl), x1 = X(tl);
                wo = orc_vmf_sample(vdir[0], vk[0], x0, x1);
            }
            if (score_sum > 0.0f) {
                for (int i = 0; i < K; i++) wo_p += scores[i] * orc_vmf_pdf(wo, vdir[i], vk[i]);
                wo_p /= score_sum;
            }
            wo_p = (score_sum > 0.0f ? p->volume_phase_p : 1.0f) * orc_draine_eval(vdot(first_wi, wo), p->draine_g, p->draine_a) + (1.0f - p->volume_phase_p) * wo_p;
        }
        pd *= wo_p;
        hit_t next; memset(&next, 0, sizeof next);
        next.wi = wo; next.pos = cur_pos;
        v3 incident = V3(0, 0, 0), throughput = V3(1, 1, 1);
        trace_ray(tl, &throughput, &incident, &next, sun, NULL);
        if (p->volume_use_light_cache && !(incident.x > 0.0f || incident.y > 0.0f || incident.z > 0.0f))
            incident = orc_rh3(vmul(throughput, light_cache_get(tl, next.pos, next.normal))); /* :188-192 */
        const float phase = orc_draine_eval(vdot(first_wi, wo), p->draine_g, p->draine_a);
        const float tr = orc_transmittance(t, mu_t, p->volume_max_t);
        v3 contrib = vscale(vmul(vscale(incident, phase), mu_s), tr / pd); /* :195 */
        if (finite3(contrib)) {
            irr = vadd(irr, contrib);
            float l = orc_luminance(contrib);
            second_moment += l * l;
            distance_add_sample(&dstate, t, l); /* :202 */
            if (s == p->volume_spp - 1) c->volume_depth[idx] = orc_f2h(dstate.sum_w > 0.0f ? dstate.m0 / dstate.sum_w : linear_z);
            if (X(tl) < l / (dist_score_sum / (float)p->distance_mc_samples)) { /* :213 */
                uint32_t di = distance_mc_index(tl, (float)px, (float)py, grid_max_x);
                if (p->log_learning) llog_simple(c, 4u, di, f2u(dstate.sum_w), dstate.N, f2u(dstate.m0), f2u(dstate.m1));
                if (!p->freeze_learning) c->dist_mc[di] = dstate;
            }
            const float mc_f = orc_luminance(vscale(vscale(incident, phase), 1.0f / wo_p)); /* :218 */
            if (X(tl) < mc_f / (score_sum / (float)p->mc_samples)) {
                float x0 = X(tl), x1 = X(tl);
                v3 jn = orc_sample_cos_frame(vneg(first_wi), x0, x1);
                v3 mv = orc_rh3(vscale(vsub(next.pos, next.prev_pos), 1.0f / u->cam_w[3]));
                mc_state_add_sample(tl, &mc_state, cur_pos, mc_f, next.pos, mv, jn, mc_buffer_index);
            } else if (p->mc_fast_recovery && mc_buffer_index != 0xffffffffu && mc_light_missing(p, &mc_state, mc_f, wo, cur_pos)) {
                if (p->log_learning) llog_simple(c, 3u, mc_buffer_index, 0u, 0u, 0u, 0u);
                if (!p->freeze_learning) c->mc[mc_buffer_index].sum_w = 0.0f; /* volume.comp:228 */
            }
        }
    }
    float inv = 1.0f / (float)(p->volume_spp > 1 ? p->volume_spp : 1);
    float* o = c->volume + 4 * idx;
    o[0] = irr.x * inv; o[1] = irr.y * inv; o[2] = irr.z * inv; o[3] = second_moment * inv;
}

/* ---------------------------------------------------------------- frame driver */

/* Worker pool: threads are created once per context and woken per pass (they used to be created and joined three times per
 * frame); rows -- and, with the parallel-update flag, update slots -- are claimed from a shared counter, so a thread
 * that drew cheap rows (sky) takes more of them.  Which thread renders a pixel never changes what the pixel computes. */
typedef struct orc_pool {
    pthread_mutex_t m; pthread_cond_t go, done;
    int n, active, pending, stop, pass; uint64_t gen;
    pthread_t th[256]; struct pool_arg { struct orc_pool* pool; int tid; } arg[256]; orc_counters_t ctr[256];
    orc_ctx* c; uint32_t next, upd_n;
} orc_pool;
static void pool_work(orc_pool* P, int tid) {
    orc_ctx* c = P->c;
    tls_t tl; memset(&tl, 0, sizeof tl); tl.c = c;
    if (P->pass == 3) { /* update pass: chunks of touched slots */
        for (;;) {
            uint32_t i0 = __atomic_fetch_add(&P->next, 64u, __ATOMIC_RELAXED);
            if (i0 >= P->upd_n) break;
            uint32_t i1 = i0 + 64u < P->upd_n ? i0 + 64u : P->upd_n;
            for (uint32_t i = i0; i < i1; i++) apply_slot(&tl, c->upd_touched[i]);
        }
    } else for (;;) {
        uint32_t y = __atomic_fetch_add(&P->next, 1u, __ATOMIC_RELAXED);
        if (y >= c->H) break;
        for (uint32_t x = 0; x < c->W; x++) { if (P->pass == 0) gbuffer_pixel(&tl, x, y); else if (P->pass == 1) mcpg_pixel(&tl, x, y); else volume_pixel(&tl, x, y); }
    }
    P->ctr[tid] = tl.ctr;
}
static void* pool_thread(void* arg) {
    orc_pool* P = ((struct pool_arg*)arg)->pool;
    const int tid = ((struct pool_arg*)arg)->tid;
    uint64_t seen = 0;
    for (;;) {
        pthread_mutex_lock(&P->m);
        while (P->gen == seen && !P->stop) pthread_cond_wait(&P->go, &P->m);
        if (P->stop) { pthread_mutex_unlock(&P->m); return NULL; }
        seen = P->gen;
        const int work = tid < P->active;
        pthread_mutex_unlock(&P->m);
        if (work) pool_work(P, tid);
        pthread_mutex_lock(&P->m);
        if (--P->pending == 0) pthread_cond_signal(&P->done);
        pthread_mutex_unlock(&P->m);
    }
}
static void pool_destroy(orc_ctx* c) {
    orc_pool* P = c->pool;
    if (!P) return;
    pthread_mutex_lock(&P->m); P->stop = 1; pthread_cond_broadcast(&P->go); pthread_mutex_unlock(&P->m);
    for (int i = 0; i < P->n; i++) pthread_join(P->th[i], NULL);
    pthread_mutex_destroy(&P->m); pthread_cond_destroy(&P->go); pthread_cond_destroy(&P->done);
    free(P); c->pool = NULL;
}
static orc_pool* pool_get(orc_ctx* c, int threads) {
    if (c->pool && c->pool->n >= threads) return c->pool;
    pool_destroy(c);
    orc_pool* P = (orc_pool*)calloc(1, sizeof *P);
    if (!P) return NULL;
    pthread_mutex_init(&P->m, NULL); pthread_cond_init(&P->go, NULL); pthread_cond_init(&P->done, NULL);
    P->c = c;
    for (int i = 0; i < threads; i++) { P->arg[i].pool = P; P->arg[i].tid = i; if (pthread_create(&P->th[i], NULL, pool_thread, &P->arg[i]) != 0) break; P->n++; }
    c->pool = P;
    return P;
}
static void acc_ctr(orc_counters_t* a, const orc_counters_t* b) {
    a->rays += b->rays; a->nodes += b->nodes; a->tris += b->tris; a->segments += b->segments; a->guided_segments += b->guided_segments;
    a->lc_touches += b->lc_touches; a->mc_updates_accepted += b->mc_updates_accepted; a->mc_updates_dropped += b->mc_updates_dropped; a->mc_state_reads += b->mc_state_reads;
}
static void pool_run(orc_ctx* c, int pass, int threads, uint32_t upd_n) {
    orc_pool* P = pool_get(c, threads);
    if (!P || P->n < 1) return;
    pthread_mutex_lock(&P->m);
    P->pass = pass; P->next = 0; P->upd_n = upd_n; P->active = threads < P->n ? threads : P->n; P->pending = P->n;
    memset(P->ctr, 0, sizeof P->ctr);
    P->gen++;
    pthread_cond_broadcast(&P->go);
    while (P->pending) pthread_cond_wait(&P->done, &P->m);
    pthread_mutex_unlock(&P->m);
    for (int i = 0; i < P->active; i++) acc_ctr(&c->ctr, &P->ctr[i]);
}
static void run_pass(orc_ctx* c, int pass, int threads) {
    if (threads < 1) threads = 1; if (threads > 256) threads = 256;
    if (threads == 1) { /* the calling thread, rows in order: the sequential frames the tests' learning runs rely on */
        tls_t tl; memset(&tl, 0, sizeof tl); tl.c = c;
        for (uint32_t y = 0; y < c->H; y++)
            for (uint32_t x = 0; x < c->W; x++) { if (pass == 0) gbuffer_pixel(&tl, x, y); else if (pass == 1) mcpg_pixel(&tl, x, y); else volume_pixel(&tl, x, y); }
        acc_ctr(&c->ctr, &tl.ctr);
    } else pool_run(c, pass, threads, 0);
}

/* update pass, render_mcpg.cpp:270-277: every touched slot, ascending slot order (one thread: deterministic); with
 * orc_process_mt's parallel-update flag the slots are spread over the workers -- compute_updates.comp is one GPU thread per
 * slot, unordered -- which is what bench.py's CPU baseline times */
static void run_update_pass(orc_ctx* c, int threads) {
    uint32_t n = c->upd_pool_used < c->upd_pool_cap ? c->upd_pool_used : c->upd_pool_cap;
    if (c->parallel_update && threads > 1 && n > 4096u) pool_run(c, 3, threads > 256 ? 256 : threads, n);
    else {
        qsort(c->upd_touched, n, 4, cmp_u32);
        tls_t tl; memset(&tl, 0, sizeof tl); tl.c = c;
        for (uint32_t i = 0; i < n; i++) apply_slot(&tl, c->upd_touched[i]);
        acc_ctr(&c->ctr, &tl.ctr);
    }
    c->upd_pool_used = 0;
}

const uint32_t* orc_learn_log(orc_ctx* c, size_t* n) { if (n) *n = c->llog_n; return c->llog; }
int orc_learn_log_reset(orc_ctx* c, size_t capacity) {
    if (capacity != c->llog_cap) { free(c->llog); c->llog = (uint32_t*)malloc((capacity ? capacity : 1) * 64); c->llog_cap = c->llog ? capacity : 0; }
    c->llog_n = 0;
    return c->llog ? 0 : -1;
}

int orc_debug_apply_updates(orc_ctx* c, const uint32_t* records, size_t n, const orc_uniform_t* u, uint32_t* touches, size_t touch_cap, size_t* n_touches) {
    if (!c->mc || c->upd_pool_used != 0) return -1; /* needs a connected context with an empty queue */
    c->u = *u;
    for (size_t i = 0; i < n; i++) {
        const uint32_t* r = records + 16 * i;
        const uint32_t slot = r[14], rank = r[13] >> 16;
        if (slot >= c->mc_total || rank >= MAX_UPDATES) return -2;
        uint32_t rec;
        if (c->upd_rec[slot] == 0) {
            rec = c->upd_pool_used++;
            if (rec >= c->upd_pool_cap) return -3;
            c->upd_touched[rec] = slot; c->upd_rec[slot] = rec + 1;
        } else rec = c->upd_rec[slot] - 1;
        mcupdate_t* up = &c->upd_pool[rec];
        if (rank == 0) up->T = u2f(r[11]);
        up->positions[rank] = V3(u2f(r[0]), u2f(r[1]), u2f(r[2])); up->weights[rank] = u2f(r[3]);
        up->targets[rank] = V3(u2f(r[4]), u2f(r[5]), u2f(r[6])); up->ids[rank] = r[7];
        up->normals[rank] = V3(u2f(r[8]), u2f(r[9]), u2f(r[10]));
        up->mv[rank][0] = (uint16_t)(r[12] & 0xffffu); up->mv[rank][1] = (uint16_t)(r[12] >> 16); up->mv[rank][2] = (uint16_t)(r[13] & 0xffffu);
        c->upd_count[slot]++;
    }
    c->touch = touches; c->touch_cap = touches ? touch_cap : 0; c->touch_n = 0;
    run_update_pass(c, 1);
    if (n_touches) *n_touches = c->touch_n;
    c->touch = NULL; c->touch_cap = c->touch_n = 0;
    return 0;
}

int orc_process(orc_ctx* c, const orc_uniform_t* u, int render, int threads) {
    if (!c->irradiance) return -1;
    c->u = *u;
    size_t px = (size_t)c->W * c->H;
    if (!render) { /* clear.comp:15-23, gbuffer.comp:83-90 */
        memset(c->irradiance, 0, px * 16); memset(c->gb_albedo, 0, px * 8); memset(c->gb_irr, 0, px * 8); memset(c->gb_mv, 0, px * 4);
        memset(c->gbuffer, 0, px * sizeof(gbuf_t)); memset(c->volume, 0, px * 16);
        return 0;
    }
    run_pass(c, 0, threads);
    run_pass(c, 1, threads);
    run_update_pass(c, threads);
    /* volume passes, render_mcpg.cpp:280-320: copy mv, forward-project, single-scatter estimator.
     * Its Markov-chain updates stay queued until the next frame's update pass. */
    if (c->p.volume_spp > 0) {
        memcpy(c->prev_volume_depth, c->volume_depth, px * 2); /* delay-1 feedback connector, default_config.json:243-248 */
        memcpy(c->volume_mv, c->gb_mv, px * 4);
        if (c->p.volume_forward_project && c->iteration != 0)
            for (uint32_t y = 0; y < c->H; y++) for (uint32_t x = 0; x < c->W; x++) forward_project_pixel(c, x, y);
        run_pass(c, 2, threads);
        c->volume_ran = u->cam_x[3] > 0.0f;
        /* keep the touched-slot list of the volume pass for the next frame's update pass */
    } else { memset(c->volume, 0, px * 16); c->volume_ran = 0; }
    c->iteration++;
    return 0;
}

int orc_process_mt(orc_ctx* c, const orc_uniform_t* u, int render, int threads, int parallel_update) {
    c->parallel_update = parallel_update;
    const int r = orc_process(c, u, render, threads);
    c->parallel_update = 0;
    return r;
}

/* ---------------------------------------------------------------- ReSTIR DI node */

/* ReSTIRDIReservoir, restir_di_reservoir.glsl.h:8-27: 64 bytes in the scalar layout */
typedef struct { uint32_t M; float w, p_target; float pos[3], normal[3], mv[3]; float T; uint16_t rad[3]; uint16_t _pad; uint32_t flags; } reservoir_t;
static reservoir_t res_init(void) { reservoir_t r; memset(&r, 0, sizeof r); return r; }
static inline v3 a3(const float* p) { return V3(p[0], p[1], p[2]); }
static inline void s3(float* p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
static void res_discard(reservoir_t* r) { r->w = 0.0f; r->flags = 0; r->rad[0] = r->rad[1] = r->rad[2] = 0; } /* restir_di.glsl:54-58 */
static void res_take_sample(reservoir_t* r, const reservoir_t* x) { memcpy(r->pos, x->pos, 36); r->T = x->T; memcpy(r->rad, x->rad, 6); r->flags = x->flags; }
/* restir_di_common.glsl:7-18.  DEFINED: pow(d, 2) = d * d; yuv_luminance_f16 = luminance of the half radiance, rounded to half */
static float restir_target_pdf(const reservoir_t* y, const hit_t* surface) {
    v3 dv = vsub(a3(y->pos), surface->pos);
    v3 wo = vnormalize(dv);
    float wodotn = vdot(wo, surface->normal);
    if (wodotn <= 0.0f) return 0.0f;
    float bsdf = orc_bsdf_times_wodotn(surface->wi, wo, surface->normal, orc_roughness_to_alpha(surface->roughness), 0.02f);
    float dist = vlen(dv);
    return ((omax(vdot(a3(y->normal), vneg(wo)), 0.0f) / (dist * dist)) * bsdf) * orc_rh(orc_luminance(h3(y->rad)));
}
static int res_add_sample(tls_t* tl, reservoir_t* r, const reservoir_t* x, float p_sample, float p_target) { /* restir_di.glsl:69-88 */
    float w = p_target / p_sample;
    r->w += w; r->M += 1;
    if (X(tl) * r->w < w) { r->p_target = p_target; res_take_sample(r, x); return 1; }
    return 0;
}
static int res_combine_finalized(tls_t* tl, reservoir_t* r, const reservoir_t* o, float p_target_x_y) { /* restir_di.glsl:125-141 */
    r->M += o->M;
    float w = (p_target_x_y * o->w) * (float)o->M;
    r->w += w;
    if (X(tl) * r->w < w) { r->p_target = p_target_x_y; res_take_sample(r, o); return 1; }
    return 0;
}
static void res_finalize(reservoir_t* r) { float den = (float)r->M * r->p_target; r->w = den > 0.0f ? r->w / den : 0.0f; }                       /* :146-149 */
static void res_finalize_custom(reservoir_t* r, float num, float den) { den *= r->p_target; r->w = den > 0.0f ? (r->w * num) / den : 0.0f; } /* :153-156 */
/* DEFINED (merian-shaders/reprojection.glsl is absent): normals within the threshold, depths within a fraction of the larger one */
static int reprojection_valid(v3 n, v3 pn, float cos_reject, float z, float vel_z, float pz, float depth_reject) {
    float ze = z + vel_z;
    return vdot(n, pn) >= cos_reject && fabsf(ze - pz) <= depth_reject * omax(ze, pz);
}
/* trace_visibility, raytrace.glsl:66-150: tmin 1e-3, tmax |to - from| - 2e-3; a sky surface in between does not occlude */
static int trace_visibility(tls_t* tl, v3 from, v3 to) {
    const orc_ctx* c = tl->c;
    v3 wo = vsub(to, from);
    float len = vlen(wo);
    v3 d = vnormalize(wo);
    float tmin = 1e-3f, tmax = omax(1e-3f, len - 2.0f * 1e-3f);
    rayhit_t best; best.key = 0xffffffffu; best.t = INFINITY; best.u = best.v = 0.0f;
    tl->ctr.rays++;
    if (!c->accel || !c->nodes) { for (uint32_t i = 0; i < c->n_tris; i++) consider_range(c, &c->tris[i], from, d, tmin, tmax, &best, &tl->ctr); }
    else {
        v3 id;
        id.x = 1.0f / (fabsf(d.x) > 1e-20f ? d.x : (d.x < 0 ? -1e-20f : 1e-20f));
        id.y = 1.0f / (fabsf(d.y) > 1e-20f ? d.y : (d.y < 0 ? -1e-20f : 1e-20f));
        id.z = 1.0f / (fabsf(d.z) > 1e-20f ? d.z : (d.z < 0 ? -1e-20f : 1e-20f));
        uint32_t stack[128]; int sp = 0; stack[sp++] = 0;
        while (sp) {
            const bnode_t* n = &c->nodes[stack[--sp]]; float tn;
            if (!slab(n, from, id, omin(best.t, tmax), &tn)) continue;
            if (n->count) { for (uint32_t i = 0; i < n->count; i++) consider_range(c, &c->tris[n->left + i], from, d, tmin, tmax, &best, &tl->ctr); continue; }
            if (sp < 120) { stack[sp++] = n->left; stack[sp++] = n->left + 1; }
        }
    }
    if (best.key == 0xffffffffu) return 1;
    return (ext_of(c, best.key)->texnum_fb_flags >> 12) == MAT_FLAGS_SKY;
}

typedef struct { orc_ctx* c; const orc_restir_params_t* r; int pass, tid, nthreads; uint8_t* res_a; const uint8_t* res_read; orc_counters_t ctr; } rjob_t;
#define RES(buf, i) ((reservoir_t*)((buf) + 64 * (size_t)(i)))
static void restir_generate_pixel(tls_t* tl, const rjob_t* j, uint32_t px, uint32_t py) { /* restir_di_generate_samples_bsdf.comp:23-62 */
    orc_ctx* c = j->c; const orc_restir_params_t* R = j->r;
    size_t idx = (size_t)py * c->W + px;
    tl->rng = orc_pcg4d16(px, py, c->u.frame * 4u + 0u, R->seed);
    reservoir_t r = res_init();
    hit_t first; decompress_hit(&c->hits[idx], &first);
    v3 sun = V3(c->p.sun_color[0], c->p.sun_color[1], c->p.sun_color[2]);
    if (first.albedo.x >= 1e-7f || first.albedo.y >= 1e-7f || first.albedo.z >= 1e-7f)
        for (int s = 0; s < R->spp; s++) {
            float alpha = orc_roughness_to_alpha(first.roughness);
            float x0 = X(tl), x1 = X(tl), x2 = X(tl);
            v3 wo = orc_bsdf_sample(first.wi, first.normal, alpha, x0, x1, x2);
            float wodotn = vdot(wo, first.normal);
            if (vdot(wo, orc_decode_normal(first.enc_geonormal)) <= 1e-3f || wodotn <= 1e-3f) continue;
            hit_t next; memset(&next, 0, sizeof next);
            next.wi = wo; next.pos = vsub(first.pos, vscale(first.wi, 1e-3f));
            v3 incident = V3(0, 0, 0), throughput = V3(1, 1, 1);
            trace_ray(tl, &throughput, &incident, &next, sun, NULL);
            float dist = vlen(vsub(next.pos, first.pos));
            float geo = omax(vdot(next.normal, vneg(wo)), 0.0f) / (dist * dist);
            reservoir_t x = res_init();
            s3(x.pos, next.pos); s3(x.normal, next.normal); s3(x.mv, vscale(vsub(next.pos, next.prev_pos), 1.0f / c->u.cam_w[3])); x.T = c->u.cl_time;
            x.rad[0] = orc_f2h(incident.x); x.rad[1] = orc_f2h(incident.y); x.rad[2] = orc_f2h(incident.z); x.flags = 1u;
            res_add_sample(tl, &r, &x, geo * orc_bsdf_pdf(first.wi, wo, first.normal, alpha), restir_target_pdf(&x, &first));
        }
    res_finalize(&r);
    *RES(j->res_a, idx) = r;
}
/* restir_di_temporal_reuse.comp:71-146 for one 8x8 tile (the boiling filter, :37-69, averages over the workgroup: lane order) */
static void restir_temporal_tile(tls_t* tl, const rjob_t* j, uint32_t tx, uint32_t ty) {
    orc_ctx* c = j->c; const orc_restir_params_t* R = j->r;
    reservoir_t rr[64]; int active[64];
    for (int lane = 0; lane < 64; lane++) {
        uint32_t px = tx * 8u + ((uint32_t)lane & 7u), py = ty * 8u + ((uint32_t)lane >> 3);
        active[lane] = px < c->W && py < c->H;
        rr[lane] = res_init();
        if (!active[lane]) continue;
        size_t idx = (size_t)py * c->W + px;
        reservoir_t* r = &rr[lane];
        tl->rng = orc_pcg4d16(px, py, c->u.frame * 4u + 1u, R->seed);
        const reservoir_t cur = *RES(j->res_a, idx);
        res_combine_finalized(tl, r, &cur, cur.p_target);
        float qx = floorf(((float)px + orc_h2f(c->gb_mv[2 * idx])) + 0.5f), qy = floorf(((float)py + orc_h2f(c->gb_mv[2 * idx + 1])) + 0.5f);
        active[lane] = qx >= 0.0f && qy >= 0.0f && qx < (float)c->W && qy < (float)c->H;
        if (!active[lane]) continue;
        size_t q = (size_t)(uint32_t)qy * c->W + (uint32_t)qx;
        const gbuf_t* g = &c->gbuffer[idx]; const gbuf_t* pg = &c->rs_prev_gb[q];
        active[lane] = reprojection_valid(orc_decode_normal(g->enc_normal), orc_decode_normal(pg->enc_normal), R->temporal_normal_reject_cos, g->linear_z, g->vel_z, pg->linear_z, R->temporal_depth_reject);
        if (!active[lane]) continue;
        hit_t center; decompress_hit(&c->hits[idx], &center);
        reservoir_t prev = *RES(c->rs_prev, q);
        if (R->apply_mv == 1) { s3(prev.pos, vadd(a3(prev.pos), vscale(a3(prev.mv), c->u.cl_time - prev.T))); prev.T = c->u.cl_time; }
        if (R->temporal_clamp_m > 0) prev.M = prev.M < (uint32_t)R->temporal_clamp_m ? prev.M : (uint32_t)R->temporal_clamp_m;
        int selected_prev = res_combine_finalized(tl, r, &prev, restir_target_pdf(&prev, &center));
        if (R->temporal_bias_correction == 0) res_finalize(r);
        else { /* :110-138 */
            float pi = r->p_target, pi_sum = r->p_target * (float)cur.M;
            hit_t psurf; decompress_hit(&c->hits[q], &psurf); /* surface_at(prev_pixel): this frame's record at that pixel, as the reference reads it */
            float temporal_p = restir_target_pdf(r, &psurf);
            if (temporal_p > 0.0f) {
                if (R->temporal_bias_correction == 2 && !trace_visibility(tl, center.pos, a3(r->pos))) temporal_p = 0.0f;
                if (R->temporal_bias_correction == 3) temporal_p = 0.0f;
            }
            pi = selected_prev ? temporal_p : pi;
            pi_sum += temporal_p * (float)prev.M;
            res_finalize_custom(r, pi, pi_sum);
        }
    }
    if (R->boiling_filter_strength > 1e-6f) {
        float mult = 10.0f / R->boiling_filter_strength - 9.0f, sum = 0.0f; uint32_t count = 0;
        for (int lane = 0; lane < 64; lane++) if (active[lane]) { sum += rr[lane].w; count += rr[lane].w > 0.0f ? 1u : 0u; }
        float avg = count > 0 ? sum / (float)count : 0.0f;
        for (int lane = 0; lane < 64; lane++) if (active[lane] && rr[lane].w > avg * mult) res_discard(&rr[lane]);
    }
    for (int lane = 0; lane < 64; lane++) if (active[lane]) {
        uint32_t px = tx * 8u + ((uint32_t)lane & 7u), py = ty * 8u + ((uint32_t)lane >> 3);
        *RES(j->res_a, (size_t)py * c->W + px) = rr[lane];
    }
}
static void restir_spatial_pixel(tls_t* tl, const rjob_t* j, uint32_t px, uint32_t py) { /* restir_di_spatial_reuse.comp:26-101 */
    orc_ctx* c = j->c; const orc_restir_params_t* R = j->r;
    size_t idx = (size_t)py * c->W + px;
    int NI = R->spatial_reuse_iterations < 1 ? 1 : (R->spatial_reuse_iterations > 7 ? 7 : R->spatial_reuse_iterations);
    tl->rng = orc_pcg4d16(px, py, c->u.frame * 4u + 2u, R->seed);
    reservoir_t r = res_init();
    const reservoir_t cur = *RES(j->res_read, idx);
    res_combine_finalized(tl, &r, &cur, cur.p_target);
    hit_t center; decompress_hit(&c->hits[idx], &center);
    const gbuf_t* g = &c->gbuffer[idx];
    int selected = -1; uint32_t nq[7];
    for (int i = 0; i < NI; i++) {
        float x0 = X(tl), x1 = X(tl);
        float nx = floorf(((float)px + (float)R->spatial_radius * (2.0f * x0 - 1.0f)) + 0.5f), ny = floorf(((float)py + (float)R->spatial_radius * (2.0f * x1 - 1.0f)) + 0.5f);
        nq[i] = 0xffffffffu;
        if (!(nx >= 0.0f && ny >= 0.0f && nx < (float)c->W && ny < (float)c->H)) continue;
        uint32_t q = (uint32_t)ny * c->W + (uint32_t)nx;
        const gbuf_t* ng = &c->gbuffer[q];
        if (!reprojection_valid(orc_decode_normal(g->enc_normal), orc_decode_normal(ng->enc_normal), R->spatial_normal_reject_cos, g->linear_z, g->vel_z, ng->linear_z, R->spatial_depth_reject)) continue;
        nq[i] = q;
        const reservoir_t nb = *RES(j->res_read, q);
        if (res_combine_finalized(tl, &r, &nb, restir_target_pdf(&nb, &center))) selected = i;
    }
    if (R->spatial_bias_correction == 0) res_finalize(&r);
    else { /* :75-97 */
        float pi = r.p_target, pi_sum = r.p_target * (float)cur.M;
        for (int i = 0; i < NI; i++) {
            if (nq[i] == 0xffffffffu) continue;
            hit_t ns; decompress_hit(&c->hits[nq[i]], &ns);
            float spatial_p = restir_target_pdf(&r, &ns);
            if (R->spatial_bias_correction == 2 && spatial_p > 0.0f && !trace_visibility(tl, ns.pos, a3(r.pos))) spatial_p = 0.0f;
            pi = selected == i ? spatial_p : pi;
            pi_sum += spatial_p * (float)RES(j->res_read, nq[i])->M;
        }
        res_finalize_custom(&r, pi, pi_sum);
    }
    *RES(j->res_a, idx) = r;
}
static void restir_shade_pixel(tls_t* tl, const rjob_t* j, uint32_t px, uint32_t py) { /* restir_di_shade.comp:21-62 */
    orc_ctx* c = j->c; const orc_restir_params_t* R = j->r;
    size_t idx = (size_t)py * c->W + px;
    reservoir_t r = *RES(j->res_a, idx);
    v3 irr = V3(0, 0, 0);
    v3 sun = V3(c->p.sun_color[0], c->p.sun_color[1], c->p.sun_color[2]);
    if (r.flags & 1u) {
        hit_t first; decompress_hit(&c->hits[idx], &first);
        v3 dv = vsub(a3(r.pos), first.pos);
        v3 wo = vnormalize(dv);
        hit_t next; memset(&next, 0, sizeof next);
        next.wi = wo; next.pos = vsub(first.pos, vscale(first.wi, 1e-3f));
        v3 incident = V3(0, 0, 0), throughput = V3(1, 1, 1);
        trace_ray(tl, &throughput, &incident, &next, sun, NULL);
        float d_sample = vlen(dv), d_hit = vlen(vsub(first.pos, next.pos));
        if (R->visibility_shade && fabsf(d_sample - d_hit) / omax(d_sample, d_hit) > 0.1f) { res_discard(&r); *RES(j->res_a, idx) = r; }
        float bsdf = orc_bsdf_times_wodotn(first.wi, wo, first.normal, orc_roughness_to_alpha(first.roughness), 0.02f);
        if (isfinite(r.w)) irr = vscale(vscale(vscale(h3(r.rad), bsdf), r.w), omax(vdot(a3(r.normal), vneg(wo)), 0.0f) / (d_sample * d_sample));
    }
    float* o = c->rs_irr + 4 * idx; o[0] = irr.x; o[1] = irr.y; o[2] = irr.z; o[3] = 1.0f;
    float l = orc_luminance(irr);
    c->rs_mom[2 * idx] = l; c->rs_mom[2 * idx + 1] = l * l;
}
static void* restir_worker(void* arg) {
    rjob_t* j = (rjob_t*)arg; orc_ctx* c = j->c;
    tls_t tl; memset(&tl, 0, sizeof tl); tl.c = c;
    const uint32_t tiles_x = (c->W + 7) / 8, tiles_y = (c->H + 7) / 8;
    for (uint32_t t = (uint32_t)j->tid; t < tiles_x * tiles_y; t += (uint32_t)j->nthreads) {
        uint32_t tx = t % tiles_x, ty = t / tiles_x;
        if (j->pass == 1) { restir_temporal_tile(&tl, j, tx, ty); continue; }
        for (int lane = 0; lane < 64; lane++) {
            uint32_t px = tx * 8u + ((uint32_t)lane & 7u), py = ty * 8u + ((uint32_t)lane >> 3);
            if (px >= c->W || py >= c->H) continue;
            if (j->pass == 0) restir_generate_pixel(&tl, j, px, py); else if (j->pass == 2) restir_spatial_pixel(&tl, j, px, py); else restir_shade_pixel(&tl, j, px, py);
        }
    }
    j->ctr = tl.ctr;
    return NULL;
}
static void restir_pass(orc_ctx* c, const orc_restir_params_t* r, int pass, uint8_t* res_a, const uint8_t* res_read, int threads) {
    if (threads < 1) threads = 1; if (threads > 256) threads = 256;
    rjob_t jobs[256]; pthread_t th[256];
    for (int i = 0; i < threads; i++) { jobs[i].c = c; jobs[i].r = r; jobs[i].pass = pass; jobs[i].tid = i; jobs[i].nthreads = threads; jobs[i].res_a = res_a; jobs[i].res_read = res_read; memset(&jobs[i].ctr, 0, sizeof(orc_counters_t)); }
    if (threads == 1) restir_worker(&jobs[0]);
    else { for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, restir_worker, &jobs[i]); for (int i = 0; i < threads; i++) pthread_join(th[i], NULL); }
    for (int i = 0; i < threads; i++) acc_ctr(&c->ctr, &jobs[i].ctr);
}
/* RendererRESTIR::process, renderer_restir.cpp:129-251 */
int orc_restir_process(orc_ctx* c, const orc_restir_params_t* r, const orc_uniform_t* u, int render, int threads) {
    if (!c->rs_out) return -1;
    c->u = *u;
    const size_t px = (size_t)c->W * c->H;
    if (!render) { memset(c->rs_out, 0, px * 64); memset(c->rs_irr, 0, px * 16); memset(c->rs_mom, 0, px * 8); } /* restir_di_clear.comp */
    else {
        const int spatial = r->spatial_reuse_iterations > 0;
        uint8_t* a = spatial ? c->rs_pong : c->rs_out; /* the ping-pong of renderer_restir.cpp:136-146,213-250 */
        restir_pass(c, r, 0, a, NULL, threads);
        if (r->temporal_reuse_enable && c->rs_iteration > 0) restir_pass(c, r, 1, a, NULL, threads);
        if (spatial) restir_pass(c, r, 2, c->rs_out, c->rs_pong, threads);
        restir_pass(c, r, 3, c->rs_out, NULL, threads);
    }
    memcpy(c->rs_prev, c->rs_out, px * 64); memcpy(c->rs_prev_gb, c->gbuffer, px * sizeof(gbuf_t)); /* the graph's delay-1 inputs */
    c->rs_iteration++;
    return 0;
}
const void* orc_restir_output(orc_ctx* c, int which, size_t* bytes) {
    size_t px = (size_t)c->W * c->H;
    if (which == 0) { if (bytes) *bytes = px * 16; return c->rs_irr; }
    if (which == 1) { if (bytes) *bytes = px * 8; return c->rs_mom; }
    if (which == 2) { if (bytes) *bytes = px * 64; return c->rs_out; }
    return NULL;
}

/* ---------------------------------------------------------------- post chain (definitions: DESIGN.md section 3) */

int orc_post_set_add_restir(orc_ctx* c, int on) { c->post_add_restir = on != 0; return 0; }
int orc_post_set_params(orc_ctx* c, int which, const float* six) { if (which < 0 || which > 1) return -1; memcpy(c->post_par[which], six, 24); return 0; }
void orc_post_clear(orc_ctx* c) { c->post_first = 1; }
const void* orc_post_output(orc_ctx* c, int which, size_t* bytes) {
    size_t px = (size_t)c->W * c->H;
    switch (which) {
    case 0: case 2: if (bytes) *bytes = px * 16; return c->post_out[which / 2];
    case 1: case 3: if (bytes) *bytes = px * 4; return c->post_hist[which / 2];
    case 4: if (bytes) *bytes = px * 16; return c->post_final;
    }
    return NULL;
}
static void accumulate(orc_ctx* c, int k, const float* src, const uint16_t* mv) {
    const float alpha = c->post_par[k][0], max_history = c->post_par[k][1], cos_thr = (float)cos((double)c->post_par[k][2]), depth_thr = c->post_par[k][3];
    const int enable_mv = c->post_par[k][4] != 0.0f && mv != NULL, reuse_border = c->post_par[k][5] != 0.0f;
    const uint32_t W = c->W, H = c->H;
    for (uint32_t iy = 0; iy < H; iy++) for (uint32_t ix = 0; ix < W; ix++) {
        const size_t i = (size_t)iy * W + ix;
        const float* s = src + 4 * i;
        float h = 1.0f, o[4] = {s[0], s[1], s[2], s[3]};
        if (!c->post_first) {
            float mx = 0.0f, my = 0.0f;
            if (enable_mv) { mx = orc_h2f(mv[2 * i]); my = orc_h2f(mv[2 * i + 1]); }
            float qx = floorf(((float)ix + mx) + 0.5f), qy = floorf(((float)iy + my) + 0.5f);
            int valid = qx >= 0.0f && qy >= 0.0f && qx < (float)W && qy < (float)H;
            if (!valid && reuse_border && qx == qx && qy == qy) { qx = oclamp(qx, 0.0f, (float)W - 1.0f); qy = oclamp(qy, 0.0f, (float)H - 1.0f); valid = 1; }
            if (valid) {
                const size_t q = (size_t)(uint32_t)qy * W + (uint32_t)qx;
                const gbuf_t* g = &c->gbuffer[i]; const gbuf_t* pg = &c->post_prev_gb[q];
                const float ze = g->linear_z + g->vel_z, zp = pg->linear_z;
                valid = vdot(orc_decode_normal(g->enc_normal), orc_decode_normal(pg->enc_normal)) >= cos_thr && fabsf(ze - zp) <= depth_thr * omax(ze, zp);
                if (valid) {
                    h = omin(c->post_prev_hist[k][q] + 1.0f, max_history);
                    const float a = omax(1.0f / h, 1.0f - alpha);
                    const float* p = c->post_prev_out[k] + 4 * q;
                    for (int ch = 0; ch < 4; ch++) o[ch] = omix(p[ch], s[ch], a);
                }
            }
        }
        memcpy(c->post_out[k] + 4 * i, o, 16); c->post_hist[k][i] = h;
    }
}
int orc_post_process(orc_ctx* c) {
    if (!c->post_final) return -1;
    const size_t px = (size_t)c->W * c->H;
    accumulate(c, 0, c->irradiance, c->gb_mv);
    accumulate(c, 1, c->volume, c->volume_ran ? c->volume_mv : NULL); /* volume_mv exists only after a volume pass */
    for (size_t i = 0; i < px; i++) { /* add: accum * albedo (the denoiser node's re-modulation) + volume accum + first-hit emission */
        const float* a = c->post_out[0] + 4 * i; const float* v = c->post_out[1] + 4 * i; float* f = c->post_final + 4 * i;
        for (int ch = 0; ch < 3; ch++) {
            const float al = orc_h2f(c->gb_albedo[4 * i + ch]);
            f[ch] = (a[ch] * al + v[ch]) + orc_h2f(c->gb_irr[4 * i + ch]);
            if (c->post_add_restir && c->rs_irr) f[ch] = f[ch] + c->rs_irr[4 * i + ch] * al; /* config 5: + ReSTIR DI irradiance, re-modulated like the MCPG irradiance */
        }
        f[3] = 1.0f;
    }
    for (int k = 0; k < 2; k++) { memcpy(c->post_prev_out[k], c->post_out[k], px * 16); memcpy(c->post_prev_hist[k], c->post_hist[k], px * 4); }
    memcpy(c->post_prev_gb, c->gbuffer, px * sizeof(gbuf_t));
    c->post_first = 0;
    return 0;
}

/* ---------------------------------------------------------------- ray queries + KATs */

int orc_trace_rays(orc_ctx* c, const float* org, const float* dir, uint32_t n, uint32_t* out_prim, float* out_t, float* out_uv) {
    orc_counters_t ctr; memset(&ctr, 0, sizeof ctr);
    for (uint32_t i = 0; i < n; i++) {
        rayhit_t h;
        closest_hit(c, V3(org[3 * i], org[3 * i + 1], org[3 * i + 2]), V3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), T_MAX, &h, &ctr);
        out_prim[i] = h.key; out_t[i] = h.key == 0xffffffffu ? T_MAX : h.t;
        if (out_uv) { out_uv[2 * i] = h.u; out_uv[2 * i + 1] = h.v; }
    }
    acc_ctr(&c->ctr, &ctr);
    return 0;
}

static const int k_arity[ORC_OP_COUNT][2] = {{1, 1}, {1, 1}, {1, 2}, {2, 1}, {1, 1}, {3, 4}, {10, 5}, {6, 4}, {1, 4}, {4, 1}, {3, 3}, {9, 2}, {3, 3}, {11, 5}, {7, 4}, {7, 4}, {3, 4}, {7, 3}, {7, 4}};
int orc_op_arity(int op, int* n_in, int* n_out) {
    if (op < 0 || op >= ORC_OP_COUNT) return -1;
    *n_in = k_arity[op][0]; *n_out = k_arity[op][1];
    return 0;
}
int orc_math_eval(orc_ctx* c, int op, const float* in, float* out, uint32_t n) {
    int ni, no;
    if (orc_op_arity(op, &ni, &no)) return -1;
    for (uint32_t k = 0; k < n; k++) {
        const float* a = in + (size_t)k * ni; float* o = out + (size_t)k * no;
        switch (op) {
        case ORC_OP_EXP2: o[0] = orc_exp2(a[0]); break;
        case ORC_OP_LOG2: o[0] = orc_log2(a[0]); break;
        case ORC_OP_SINCOS2PI: orc_sincos2pi(a[0], &o[0], &o[1]); break;
        case ORC_OP_POW: o[0] = orc_pow(a[0], a[1]); break;
        case ORC_OP_F2H2F: o[0] = orc_rh(a[0]); break;
        case ORC_OP_ENC_DEC_NORMAL: { uint32_t e = orc_encode_normal(V3(a[0], a[1], a[2])); v3 d = orc_decode_normal(e); o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = u2f(e); break; }
        case ORC_OP_BSDF_SAMPLE: {
            v3 wi = V3(a[0], a[1], a[2]), nn = V3(a[3], a[4], a[5]); float al = orc_roughness_to_alpha(a[6]);
            v3 wo = orc_bsdf_sample(wi, nn, al, a[7], a[8], a[9]);
            o[0] = wo.x; o[1] = wo.y; o[2] = wo.z; o[3] = orc_bsdf_pdf(wi, wo, nn, al); o[4] = orc_bsdf_times_wodotn(wi, wo, nn, al, 0.02f); break; }
        case ORC_OP_VMF_SAMPLE: { v3 mu = V3(a[0], a[1], a[2]); v3 w = orc_vmf_sample(mu, a[3], a[4], a[5]); o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = orc_vmf_pdf(w, mu, a[3]); break; }
        case ORC_OP_XORSHIFT: { uint32_t s = f2u(a[0]); for (int i = 0; i < 4; i++) o[i] = orc_xorshift(&s); break; }
        case ORC_OP_PCG4D16: o[0] = u2f(orc_pcg4d16(f2u(a[0]), f2u(a[1]), f2u(a[2]), f2u(a[3]))); break;
        case ORC_OP_SKY: { orc_uniform_t save = c->u; c->u.sky_lf_ft = 0xfffe; c->u.sky_rt_bk = 0xffffffffu; c->u.sky_up_dn = 0xffffffffu;
            v3 s = get_sky(c, V3(a[0], a[1], a[2]), V3(c->p.sun_color[0], c->p.sun_color[1], c->p.sun_color[2])); c->u = save; o[0] = s.x; o[1] = s.y; o[2] = s.z; break; }
        case ORC_OP_HASHGRID: { i3 g = orc_grid_idx_interpolate(V3(a[0], a[1], a[2]), a[7], 0.5f); uint32_t lv = (uint32_t)a[6];
            o[0] = u2f(orc_hash_grid_normal_level(g, V3(a[3], a[4], a[5]), lv, f2u(a[8]))); o[1] = u2f(orc_hash2_grid_level(g, lv)); break; }
        case ORC_OP_LDR_TO_HDR: { v3 r = orc_ldr_to_hdr(V3(a[0], a[1], a[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
        case ORC_OP_CAMERA: { v3 fwd = V3(a[4], a[5], a[6]), up = V3(a[7], a[8], a[9]);
            v3 d = orc_camera_ray_dir(a[0], a[1], a[2], a[3], up, fwd, a[10]); o[0] = d.x; o[1] = d.y; o[2] = d.z;
            orc_camera_pixel(d, a[2], a[3], up, fwd, a[10], &o[3], &o[4]); break; }
        case ORC_OP_DRAINE: { v3 wi = V3(a[0], a[1], a[2]); v3 w = orc_draine_sample(a[5], a[6], wi, a[3], a[4]);
            o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = orc_draine_eval(vdot(wi, w), a[3], a[4]); break; }
        case ORC_OP_DISTANCE: { float xm = orc_transmittance_xi_max(a[1], a[0]); o[0] = orc_transmittance_sample2(a[0], a[2], xm); o[1] = orc_transmittance_pdf2(o[0], a[0], xm);
            o[2] = orc_sample_normal_box_muller(a[3], a[4], a[5], a[6]); o[3] = orc_sample_normal_pdf(a[3], a[4], o[2]); break; }
        case ORC_OP_SKY_TEX: { orc_uniform_t save = c->u; c->u.sky_rt_bk = f2u(a[3]); c->u.sky_lf_ft = f2u(a[4]); c->u.sky_up_dn = f2u(a[5]); c->u.cl_time = a[6];
            v3 s = get_sky(c, V3(a[0], a[1], a[2]), V3(c->p.sun_color[0], c->p.sun_color[1], c->p.sun_color[2])); c->u = save; o[0] = s.x; o[1] = s.y; o[2] = s.z; break; }
        case ORC_OP_TEX_GRAD: { v4 x = tex_sample_grad(c, (uint32_t)a[0], a[1], a[2], a[3], a[4], a[5], a[6]); o[0] = x.r; o[1] = x.g; o[2] = x.b; o[3] = x.a; break; }
        case ORC_OP_TEX_SAMPLE: { v4 x = tex_sample(c, (uint32_t)a[0], a[1], a[2]); o[0] = x.r; o[1] = x.g; o[2] = x.b; o[3] = x.a; break; }
        }
    }
    return 0;
}
