/*
 * mq.h -- C ABI of libmqhip.so: the MI355X-native (HIP / gfx950) replacement for merian-quake's
 * `GBuffer` + `Renderer (MCPG)` render nodes.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  Every entry point cites the reference
 * interface it stands in for (file:line relative to the merian-quake tree).  INTEGRATION.md shows
 * the merian `Node` subclass a maintainer would write on top of this header.
 *
 * Conventions:
 *  - every call returns 0 on success or a negative MQ_E* code; mq_last_error() has the text;
 *  - a context is single-caller (merian calls process() from the main thread only,
 *    src/merian-quake.cpp:272-275); calls are asynchronous w.r.t. the device unless stated;
 *  - scene upload calls copy; the caller keeps ownership of its arrays;
 *  - output pointers stay valid until the next mq_connect()/mq_destroy();
 *  - there is NO CPU fallback: every device entry point fails with MQ_ENODEVICE without a GPU.
 */
#ifndef MQ_H
#define MQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MQ_ABI_VERSION 3

enum {
    MQ_OK = 0,
    MQ_EINVAL = -1,    /* bad argument */
    MQ_ENODEVICE = -2, /* no HIP device / host-only context */
    MQ_EHIP = -3,      /* a HIP call failed */
    MQ_ESTATE = -4,    /* call sequence error (e.g. process before connect) */
    MQ_ENOMEM = -5,
    MQ_EIO = -6,       /* file could not be read / parsed */
    MQ_EUNKNOWN_KEY = -7
};

/* res/shader/config.h:5-6 */
#define MQ_MAX_GLTEXTURES 4096
#define MQ_MAX_GEOMETRIES 16

/* VertexExtraData, src/game/quake_helpers.hpp:10-34 (28 bytes, one per triangle) */
typedef struct mq_ext {
    uint16_t texnum_alpha;
    uint16_t texnum_fb_flags;
    uint32_t n0_gloss_norm;
    uint32_t n1_brush;
    uint32_t n2;
    uint16_t st[6];
} mq_ext;

/* UniformData push constant, res/shader/scene_info.glsl.h:18-32 == src/game/quake_node.hpp:42-60
 * (124 bytes) */
typedef struct mq_uniform {
    float cam_x[4];      /* xyz camera position, w = fog mu_t */
    float cam_w[4];      /* xyz forward, w = time diff (1 if paused) */
    float cam_u[4];      /* xyz up */
    float prev_cam_x[4]; /* w = mu_s.r */
    float prev_cam_w[4]; /* w = mu_s.g */
    float prev_cam_u[4]; /* w = mu_s.b */
    uint32_t sky_rt_bk, sky_lf_ft, sky_up_dn;
    float cl_time;
    uint32_t frame;
    uint32_t player;
    uint32_t rt_config;
} mq_uniform;

/* QuakeNode::ConstantData, src/game/quake_node.hpp:62-69 */
typedef struct mq_constants {
    float sun_color[3];
    float sun_direction[3];
    float fov;
    float fov_tan_alpha_half;
    float volume_max_t;
} mq_constants;

/* geometry flags: instance flags of src/game/quake_node.cpp:869-871,891-892 */
enum { MQ_GEO_OPAQUE = 1, MQ_GEO_STATIC = 2 };
/* texture flags: src/game/quake_node.hpp:86-108 (sRGB unless *_norm/_gloss), sampler choice */
enum { MQ_TEX_SRGB = 1, MQ_TEX_LINEAR = 2, MQ_TEX_MIPMAP = 4 /* TEXPREF_MIPMAP, quake_node.cpp:698: mip chain for the first hit's textureGrad */ };

/* named outputs of the two nodes: src/render_mcpg/render_mcpg.cpp:42-52, src/gbuffer/gbuffer.cpp:27-43 */
enum {
    MQ_OUT_IRRADIANCE = 0,    /* "irradiance" RGBA32F: rgb mean radiance, a = luminance 2nd moment */
    MQ_OUT_GB_ALBEDO = 1,     /* gbuffer "albedo" RGBA16F */
    MQ_OUT_GB_IRRADIANCE = 2, /* gbuffer "irradiance" RGBA16F (first-hit emission) */
    MQ_OUT_GB_MV = 3,         /* gbuffer "mv" RG16F */
    MQ_OUT_GBUFFER = 4,       /* gbuffer "gbuffer": 16 B/pixel */
    MQ_OUT_HITS = 5,          /* gbuffer "hits": 40 B/pixel CompressedHit, res/shader/hit.glsl.h:19-30 */
    MQ_OUT_TILES = 6,         /* this rank's irradiance tiles, tile-major (multi-GPU exchange buffer) */
    MQ_OUT_VOLUME = 7,        /* "volume" RGBA32F: single-scatter radiance + 2nd moment, render_mcpg.cpp:44-45 */
    MQ_OUT_VOLUME_DEPTH = 8,  /* "volume_depth" R16F, render_mcpg.cpp:46-47 */
    MQ_OUT_VOLUME_MV = 9,     /* "volume_mv" RG16F, render_mcpg.cpp:48-50 */
    MQ_OUT_VOLUME_TILES = 10, /* this rank's "volume" tiles, tile-major (second multi-GPU exchange buffer, configs with volume spp > 0) */
    MQ_OUT_DEBUG = 11,        /* "debug" RGBA16F (render_mcpg.cpp:51-52, mcpg.comp:212-277); written when the property "debug output connected" is set */
    /* the post chain, mq_post_process: res/default_config.json nodes "accum", "volume accum", "add" */
    MQ_OUT_ACCUM = 12,                /* accum "out" RGBA32F: temporally accumulated irradiance (a = accumulated 2nd moment) */
    MQ_OUT_ACCUM_HISTORY = 13,        /* accum "history" R32F: frames accumulated per pixel */
    MQ_OUT_VOLUME_ACCUM = 14,         /* volume accum "out" RGBA32F */
    MQ_OUT_VOLUME_ACCUM_HISTORY = 15, /* volume accum "history" R32F */
    MQ_OUT_FINAL = 16,                /* add "out" RGBA32F: accum * albedo + volume accum + first-hit emission */
    /* the ReSTIR DI node, mq_restir_process: src/render_restir/renderer_restir.cpp:75-96 */
    MQ_OUT_RESTIR_IRRADIANCE = 17,    /* "irradiance" RGBA32F (direct light at the first hit, albedo excluded; a = 1) */
    MQ_OUT_RESTIR_MOMENTS = 18,       /* "moments" RG32F: luminance, luminance^2 */
    MQ_OUT_RESTIR_RESERVOIRS = 19,    /* "reservoirs": 64 B/pixel ReSTIRDIReservoir, res/shader/render_restir/restir_di_reservoir.glsl.h:8-27 */
    MQ_OUT_VOLUME_DEPTH_TILES = 20,   /* this rank's "volume_depth" tiles, tile-major R16F (third exchange buffer: configs with volume spp > 0 AND "volume forward project",
                                       * whose scatter reads last frame's volume_depth of EVERY pixel, render_mcpg.cpp:296-311) */
    MQ_OUT_COUNT
};

typedef struct mq_io_desc {
    uint32_t width, height;
    size_t bytes[MQ_OUT_COUNT];          /* size of each output buffer */
    uint32_t bytes_per_pixel[MQ_OUT_COUNT];
    size_t state_bytes_markovchain, state_bytes_lightcache, state_bytes_update_queue, state_bytes_volume_distancemc;
} mq_io_desc;

/* work counters of the last MQ_COUNT-enabled frame (algorithmic-bytes inputs) */
typedef struct mq_counters {
    uint64_t rays, nodes, tris, segments, guided_segments, lc_touches, mc_updates_accepted,
        mc_updates_dropped, mc_state_reads, pixels;
    uint64_t queue_rays, queue_nodes, queue_tris; /* the bounce-ray traversal kernel alone */
    uint64_t queue_overflow;                       /* != 0 since connect (never expected): bit 0 a ray queue, bit 1 the update queue ran out of room, bit 2 the packet walk's stack;
                                                    * bit 3: on a rank of a row partition a reprojected pixel lay beyond the rows the rank holds (raise "band: reprojection halo") */
} mq_counters;

typedef struct mq_ctx mq_ctx;

/* ---- lifecycle ---- */
/* device >= 0: HIP device ordinal.  device < 0: host-only context (scene, BVH, properties; every
 * device call returns MQ_ENODEVICE). */
int mq_create(mq_ctx** out, int device);
void mq_destroy(mq_ctx* ctx);
const char* mq_last_error(const mq_ctx* ctx);
int mq_abi_version(void);

/* ---- properties(): src/render_mcpg/render_mcpg.cpp:419-578, src/gbuffer/gbuffer.cpp (hide sun) --
 * Keys are the reference's own strings ("BSDF Prob", "mc samples", "adaptive grid buf size",
 * "LC grid type", "reference mode", "seed", ...; full list: mq_property_name()).  Options
 * ("adaptive grid type", "LC grid type", "debug output") take the option index or its string via
 * mq_set_property_str.  Returns 1 if the change needs a reconnect (NEEDS_RECONNECT,
 * render_mcpg.cpp:567-575), 0 if it only refreshes the kernel parameter block. */
int mq_set_property(mq_ctx* ctx, const char* key, double value);
int mq_set_property_str(mq_ctx* ctx, const char* key, const char* value);
int mq_get_property(const mq_ctx* ctx, const char* key, double* value);
int mq_property_count(void);
const char* mq_property_name(int index);
/* what the reference's Properties visitor is called with for this key: MQ_PROP_BOOL config_bool, _INT config_int, _UINT config_uint,
 * _FLOAT config_float / config_percent / config_angle, _OPTION config_options (its choices: mq_property_option(index, 0..), NULL past the end) */
enum { MQ_PROP_BOOL = 0, MQ_PROP_INT = 1, MQ_PROP_UINT = 2, MQ_PROP_FLOAT = 3, MQ_PROP_OPTION = 4 };
int mq_property_type(int index);
const char* mq_property_option(int index, int option);
/* load the "properties" object of a node from a merian-quake graph JSON (res/default_config.json
 * layout): node_name e.g. "render_markovchain" or "gbuffer". */
int mq_load_properties_json(mq_ctx* ctx, const char* json_text, const char* node_name);
void mq_properties_header_defaults(mq_ctx* ctx); /* src/render_mcpg/render_mcpg.hpp:108-166 */
void mq_properties_json_defaults(mq_ctx* ctx);   /* res/default_config.json:599-638 */

/* ---- scene inputs: connectors vtx/prev_vtx/idx/ext/textures/tlas, render_mcpg.hpp:61-70 ---- */
int mq_scene_set_geometry(mq_ctx* ctx, int slot, const float* vtx, const float* prev_vtx,
                          uint32_t n_vtx, const uint32_t* idx, const mq_ext* ext, uint32_t n_tri,
                          uint32_t flags);
int mq_scene_set_texture(mq_ctx* ctx, uint32_t texnum, uint32_t w, uint32_t h,
                         const uint8_t* rgba8, uint32_t flags);
/* (re)builds the compressed wide BVH and uploads; stands in for merian's "Acceleration Structure
 * Builder" node fed by tlas_info (res/default_config.json:3-20,400-403).  As in the reference
 * (src/game/quake_node.cpp:847-983: static geometry is built at map load, per-frame geometry every
 * frame) slots flagged MQ_GEO_STATIC form one tree that is rebuilt and uploaded only when one of them
 * (or a texture) changed; the other slots form a second tree, stored behind the first, that every
 * commit rebuilds -- a commit after changing only non-static slots rewrites just that part of the
 * device arrays.  Rays visit the second tree after the first.
 * A commit of per-frame geometry does NOT wait for the frames in flight: the device arrays hold three regions for the
 * per-frame part, the commit writes the one those frames do not read (asynchronously, from pinned staging memory, on a
 * stream of its own) and waits only for the last launch that read it, three commits ago -- the host builds the tree of
 * frame n + 1 while the device renders frame n.  A commit that changes static geometry or textures, the first commit,
 * one whose per-frame part outgrows its region, and per-frame geometry without any static geometry wait for the device. */
int mq_scene_commit(mq_ctx* ctx);
/* nodes [0, static_nodes) / triangles [0, static_tris) of mq_scene_get_bvh are the static tree (root 0); the
 * per-frame tree follows (root = node static_nodes) */
int mq_scene_layout(const mq_ctx* ctx, uint64_t* static_nodes, uint64_t* static_tris);
/* how many commits took the full path and how many only rewrote the per-frame part */
int mq_scene_commit_counts(const mq_ctx* ctx, uint32_t* full, uint32_t* per_frame);
/* how many of the per-frame commits did not wait for the device (see mq_scene_commit) */
int mq_scene_commit_async_count(const mq_ctx* ctx, uint32_t* n);
/* how many per-frame commits had their tree built on the device (property "per-frame BVH": "host" | "device" | "auto"; the
 * reference's per-frame acceleration structure is built by the Vulkan driver on the GPU, quake_node.cpp:896-983 + the graph's
 * builder node).  "auto", the default: on the device from 12 288 per-frame triangles on (below that the host's SAH tree is built in time and is the better tree).  Results do not depend on who builds a tree. */
int mq_scene_commit_device_count(const mq_ctx* ctx, uint32_t* n);
/* QuakeRenderInfo::constant + constant_data_update, src/game/quake_node.hpp:62-84 */
int mq_set_constants(mq_ctx* ctx, const mq_constants* c);
int mq_get_constants(const mq_ctx* ctx, mq_constants* out);
/* read back what the context holds (host copies) */
int mq_scene_get_geometry(const mq_ctx* ctx, int slot, const float** vtx, const float** prev_vtx,
                          uint32_t* n_vtx, const uint32_t** idx, const mq_ext** ext, uint32_t* n_tri,
                          uint32_t* flags);
int mq_scene_get_texture(const mq_ctx* ctx, uint32_t texnum, uint32_t* w, uint32_t* h,
                         const uint8_t** rgba8, uint32_t* flags);
/* committed acceleration structure, for inspection: 80-byte nodes, 48-byte triangles (leaf order) */
int mq_scene_get_bvh(const mq_ctx* ctx, const void** nodes, uint64_t* n_nodes, const void** tris, uint64_t* n_tris);
/* ... and the 64-byte leaf records the traversal reads: one or two triangles that share an edge as four vertices
 * (float v[4][3]; u32 key0, key1, tri0, sel -- merian-quake_amd/csrc/mq_types.h MqLeafRec); a node's tri_base and the
 * offsets in its meta bytes count these records, record.tri0 is its first triangle in the triangle array */
int mq_scene_get_leaves(const mq_ctx* ctx, const void** leaves, uint64_t* n_leaves);
int mq_scene_stats(const mq_ctx* ctx, uint64_t* n_tris, uint64_t* n_nodes, uint64_t* bvh_bytes,
                   float* sah_cost);

/* ---- describe_outputs / on_connected / process: render_mcpg.cpp:36-115,117-320 ---- */
int mq_describe(const mq_ctx* ctx, uint32_t width, uint32_t height, mq_io_desc* out);
int mq_connect(mq_ctx* ctx, uint32_t width, uint32_t height);
/* one frame on `stream` (a hipStream_t, NULL = default stream).  render == 0 runs the clear pass
 * (render_mcpg.cpp:243-250).  Iteration 0 zero-fills all learning state (render_mcpg.cpp:221-226). */
int mq_process(mq_ctx* ctx, const mq_uniform* u, int render, void* stream);
int mq_sync(mq_ctx* ctx);
int mq_map_output(mq_ctx* ctx, int which, void** dev_ptr, size_t* bytes);
int mq_read_output(mq_ctx* ctx, int which, void* host_dst, size_t bytes); /* sync + D2H copy */
/* device time of the last frame's hot-path kernels (hipEvent pair on the process stream) */
int mq_last_frame_ms(mq_ctx* ctx, float* total_ms, float* render_ms, float* update_ms);
/* accumulated device time of every frame since the last reset (hipEvent triplets recorded on the
 * process stream, resolved lazily so frames stay in flight): MERIAN_PROFILE_SCOPE_GPU "surface",
 * render_mcpg.cpp:255 */
int mq_timing_reset(mq_ctx* ctx);
int mq_timing_get(mq_ctx* ctx, uint32_t* frames, double* render_ms_sum, double* update_ms_sum);
/* The per-launch split below needs an event between every two launches, which costs a few microseconds each:
 * record them on every `every`-th frame only (default 1 = every frame; frame 0 after a reset is always one).
 * mq_timing_get covers all frames, _get_detail / _get_rounds the mq_timing_detail_frames() frames with events. */
int mq_timing_set_interval(mq_ctx* ctx, uint32_t every);
int mq_timing_detail_frames(mq_ctx* ctx, uint32_t* frames);
/* the render time split by kernel class: primary (first hit), trace (BVH traversal of bounce rays), bounce (shading/guiding) */
int mq_timing_get_detail(mq_ctx* ctx, double* primary_ms_sum, double* trace_ms_sum, double* bounce_ms_sum);
/* per launch: entry 0 = (trace of the primary rays, first-hit shading), entry 1 + r = (trace, bounce) of
 * round r; at most MQ_TIMING_ROUNDS entries.  Sums over the frames since the last reset. */
#define MQ_TIMING_ROUNDS 9
int mq_timing_get_rounds(mq_ctx* ctx, double* trace_ms_sum, double* shade_ms_sum, int n);
/* enable work counting for subsequent frames (separate kernel instantiation, slower) */
int mq_enable_counters(mq_ctx* ctx, int on);
int mq_get_counters(mq_ctx* ctx, mq_counters* out);
int mq_reset_state(mq_ctx* ctx); /* next process() behaves like iteration 0 */
/* Profiling builds (-DMQ_PROF) only: shader clocks per kernel code section, summed over all waves
 * since the last reset, followed by 64 histogram bins (rays by loop iterations in the queue kernel, bins of 8);
 * all zero in a product build.  Section ids: tools/prof_sections.py. */
/* Learning state in the device layout: which 0 = Markov-chain table (64 B per state: w_tgt f32x3, sum_w, w_cos, T, id u32,
 * N | hash << 16 u32, mv f16x3, padding), 1 = light cache (16 B per cell: hash u32, lock u32, irradiance f16x3, N u16),
 * 2 = distance Markov chains (16 B: sum_w f32, N u32, m0 f32, m1 f32); read only, kept while the property "debug: LC lock
 * statistics" is set (the reference's dump statistics, render_mcpg.cpp:354-416): 3 = per light-cache cell update_succeeded,
 * update_canceled (2 x u32, grid.h:44-45; the light cache then runs the reference's try-lock, light_cache.glsl:59-64),
 * 4 = per Markov-chain slot last_update_count (u32, grid.h:25).
 * `bytes` must equal the table size.  Writing needs one processed frame (the first frame zeroes the tables).
 * Test hook, with the property "debug: freeze learning": a guided frame from a given state is deterministic. */
int mq_debug_state_read(mq_ctx* ctx, int which, void* dst_host, size_t bytes);
int mq_debug_state_write(mq_ctx* ctx, int which, const void* src_host, size_t bytes);
/* Learning-write log (test hook; property "debug: log learning writes").  While the property is set every learning
 * write a path PROPOSES during mq_process is appended as one 64-byte record of 16 dwords; with "debug: freeze learning"
 * also set nothing is stored, so the log is a deterministic function of the given state.  The log holds one frame.
 *   kind 1, a queued Markov-chain update (mc.glsl:159-184), in the layout of the update queue itself:
 *           pos[3] weight target[3] id normal[3] T  mv0|mv1<<16  mv2|rank<<16 (rank 0 in the log)  slot  kind
 *   kind 2, a light-cache store (light_cache.glsl:66-84): chk, rekeyed?, irr0|irr1<<16, irr2|N<<16, ... [14] = cell, [15] = kind
 *   kind 3, a fast-recovery invalidation (mcpg.comp:175-178, volume.comp:226-229): [14] = slot, [15] = kind
 *   kind 4, a distance-chain store (volume.comp:213-215): sum_w, N, m0, m1, ... [14] = index, [15] = kind
 * *n_records = records proposed by the last frame (may exceed what the log could hold). */
int mq_debug_learn_log_read(mq_ctx* ctx, void* dst_host, size_t cap_records, size_t* n_records);
/* The update pass alone (render_mcpg.cpp:261-277, compute_updates.comp:56-124) on caller-given queue contents: n records
 * in the kind-1 layout above with slot and arrival rank (< 10, distinct per slot) filled in.  Synchronous. */
int mq_debug_apply_updates(mq_ctx* ctx, const void* records, uint32_t n, const mq_uniform* u);
#define MQ_PROF_SECTION_COUNT 40
int mq_debug_section_clocks(mq_ctx* ctx, uint64_t* out, int n, int reset);

/* ---- the post chain: temporal accumulation + albedo re-modulation + composition ----
 * Stands in for the reference graph's "accum" / "volume accum" (merian Accumulate) nodes, the albedo re-modulation of
 * its denoiser nodes and the "add" node (res/default_config.json:21-133,404-435,473-497); the arithmetic is defined
 * in merian-quake_amd/csrc/mq_post.hip (merian's node sources are not part of the reference tree).
 * Call after mq_process of the same frame, on the same stream.  Properties: "accum: alpha", "accum: max history",
 * "accum: normal threshold", "accum: depth threshold", "accum: enable motion vectors", "accum: reuse border" and the
 * same six with the prefix "volume accum: " (mq_load_properties_json(ctx, json, "accum") reads them from a graph file).
 * On a rank of a partitioned frame (mq_set_partition, world > 1) the chain covers the rank's row band, see "row partition" below. */
int mq_post_process(mq_ctx* ctx, void* stream);
/* the nodes' "clear event": the next mq_post_process starts a new history */
int mq_post_clear(mq_ctx* ctx);

/* ---- the ReSTIR DI render node: "Renderer (ReSTIR)" of src/merian-quake.cpp:191-199 ----
 * RendererRESTIR::process, src/render_restir/renderer_restir.cpp:129-251: generate samples -> [temporal reuse] ->
 * [spatial reuse] -> shade (or the clear pass when render == 0), on the g-buffer outputs (hits, gbuffer, mv) that the
 * mq_process call of the same frame left on this context (run it with "spp" 0 if only the g-buffer node is wanted).
 * Properties: the reference's keys (renderer_restir.cpp:253-325) with the prefix "restir: " -- "restir: spp",
 * "restir: seed", "restir: randomize seed", "restir: enable temporal reuse", "restir: temporal normal threshold"
 * (radians), "restir: temporal depth threshold", "restir: temporal clamp m", "restir: temporal bias correction",
 * "restir: boiling filter strength", "restir: apply mv", "restir: spatial reuse iterations", "restir: spatial normal
 * threshold", "restir: spatial depth threshold", "restir: spatital radius" (sic), "restir: spatial bias correction",
 * "restir: shade visibility".  On a rank of a partitioned frame the node covers the rank's row band, see "row partition" below. */
int mq_restir_process(mq_ctx* ctx, const mq_uniform* u, int render, void* stream);

/* ---- multi-GPU framebuffer sharding (no reference counterpart) ----
 * Rank r of `world` renders the 8x8-pixel tiles t with t % world == r into MQ_OUT_TILES
 * (tile-major, 64 RGBA32F pixels per tile).  After an all-gather of the per-rank buffers (done by
 * the caller, e.g. RCCL through torch.distributed), mq_untile() scatters the gathered buffer
 * (rank-major) into the full MQ_OUT_IRRADIANCE image. */
int mq_set_partition(mq_ctx* ctx, int rank, int world);
int mq_tiles_per_rank(const mq_ctx* ctx, uint32_t* tiles, size_t* bytes);
int mq_untile(mq_ctx* ctx, const void* gathered_dev, void* stream);
/* the same into a caller-owned W*H RGBA32F device image (an exchange that overlaps the next frame must not write
 * into MQ_OUT_IRRADIANCE, which the next frame's kernels are filling) */
int mq_untile_to(mq_ctx* ctx, const void* gathered_dev, void* image_dev, void* stream);
/* the same for the gathered MQ_OUT_VOLUME_TILES buffers -> MQ_OUT_VOLUME */
int mq_untile_volume(mq_ctx* ctx, const void* gathered_dev, void* stream);
/* ... and for the gathered MQ_OUT_VOLUME_DEPTH_TILES buffers -> MQ_OUT_VOLUME_DEPTH: with it every rank forward-projects from ALL pixels of
 * the last frame, as one rank does (without it a rank projects from its own pixels only: an approximation) */
int mq_untile_volume_depth(mq_ctx* ctx, const void* gathered_dev, void* stream);

/* ---- row partition of the ReSTIR DI node and the post chain (no reference counterpart; BASELINE config 5 on N GPUs) ----
 * Temporal reuse, spatial reuse and temporal accumulation read OTHER pixels (the reprojected one, neighbours within the spatial
 * radius: restir_di_temporal_reuse.comp:71-146, restir_di_spatial_reuse.comp:38-66), which the interleaved tiles of the MCPG
 * node put on other ranks.  These two nodes therefore cut the image into bands of whole tile rows: rank r of mq_set_partition
 * owns the pixel rows [row_begin, row_end) -- it runs spatial reuse, shading, accumulation and composition there and holds the
 * result rows of MQ_OUT_RESTIR_* / MQ_OUT_ACCUM* / MQ_OUT_FINAL (images stay full-size and row-indexed on every rank) --,
 * generates and temporally reuses reservoirs on [reuse_begin, reuse_end) (its rows widened by "restir: spatital radius"), and
 * needs last frame's reservoirs, accumulated images and g-buffer on [need_begin, need_end) (widened again by the property
 * "band: reprojection halo", the reach of reprojection in pixel rows; a reprojected pixel beyond it counts as "no history"
 * and raises bit 3 of mq_counters::queue_overflow).  The g-buffer of those rows the rank renders itself (mq_restir_process does,
 * or mq_band_gbuffer when only the post chain runs); what it cannot recompute it gets from the rows' owners once per frame:
 *     for every other rank s:  rows of [need_begin, need_end) that rank s owns:
 *         copy  rows * row_bytes  from  s's send_base + row * row_bytes  to  this rank's recv_base + row * row_bytes
 * -- point-to-point (ncclSend/ncclRecv between band neighbours over xGMI; merian-quake_amd/mq_bands.py does it with
 * torch.distributed.batch_isend_irecv), after mq_restir_process / mq_post_process of frame n and before those of frame n + 1.
 * Frame order on a rank:  mq_process (interleaved tiles) -> all-gather + mq_untile of the radiance tiles -> mq_restir_process
 * -> mq_post_process -> halo exchange (-> all-gather of the rows of MQ_OUT_FINAL if one rank wants the whole image). */
typedef struct mq_band { uint32_t row_begin, row_end, reuse_begin, reuse_end, need_begin, need_end; } mq_band;
/* the bands of ANY rank of a `world`-way row partition of a width x height image under this context's properties (host arithmetic) */
int mq_band_layout(const mq_ctx* ctx, uint32_t width, uint32_t height, int rank, int world, mq_band* out);
/* the g-buffer node's outputs on this rank's rows [need_begin, need_end) for the frame mq_process last started (mq_restir_process
 * does this itself; call it before mq_post_process when the ReSTIR node is not used) */
int mq_band_gbuffer(mq_ctx* ctx, const mq_uniform* u, void* stream);
enum { MQ_HALO_RESTIR_RESERVOIRS = 0, /* 64 B/pixel: MQ_OUT_RESTIR_RESERVOIRS -> the node's delay-1 input "reservoirs" */
       MQ_HALO_ACCUM = 1, MQ_HALO_ACCUM_HISTORY = 2, MQ_HALO_VOLUME_ACCUM = 3, MQ_HALO_VOLUME_ACCUM_HISTORY = 4, /* 16 / 4 B/pixel: the accumulate nodes' prev_out / prev_history */
       MQ_HALO_COUNT = 5 };
int mq_map_halo(mq_ctx* ctx, int which, void** send_base, void** recv_base, size_t* row_bytes);

/* ---- closest-hit ray queries against the committed scene (raytrace.glsl:82-119 semantics) ---- */
int mq_trace_rays(mq_ctx* ctx, const float* org_host, const float* dir_host, uint32_t n,
                  uint32_t* prim_host, float* t_host, float* uv_host);
/* achievable HBM read rate of this GPU: best of `reps` passes of a streaming-read kernel over a
 * fresh `bytes`-sized buffer (use >= 1 GiB: beyond the 256 MB Infinity Cache), in GB/s */
int mq_measure_stream_read(mq_ctx* ctx, size_t bytes, int reps, double* gb_per_s);
/* device-side evaluation of the shading primitives, for known-answer tests against the oracle */
int mq_math_eval(mq_ctx* ctx, int op, const float* in_host, float* out_host, uint32_t n);

/* ---- scene sources ---- */
/* seeded synthetic "BSP-like" scenes: "synth_start", "synth_sepulcher", "synth_tears",
 * "synth_azad" (SURVEY 8d).  Fills geometry slots, textures, constants. */
int mq_synth_scene(mq_ctx* ctx, const char* name, uint32_t seed);
/* deterministic fly-through camera for a synthetic scene: fills a uniform for frame index f */
int mq_synth_camera(const mq_ctx* ctx, uint32_t frame, mq_uniform* out);
/* Quake BSP29 / BSP2 world model + palette (768-byte file or NULL for a built-in grey ramp) */
int mq_load_bsp(mq_ctx* ctx, const char* bsp_path, const char* palette_path);

/* ---- per-frame geometry producers: QuakeNode::update_dynamic_geo, src/game/quake_node.cpp:896-983 ----
 * The reference collects, every frame, the view model, the visible and static entities and the particles into ONE
 * non-opaque geometry (add_geo / add_particles, src/game/quake_helpers.cpp:50-652) from quakespasm's live structures.
 * Here the same quantities come in through plain structs; models are read from id Software's own file formats.
 *     mq_dyn_begin(ctx); mq_dyn_add_*(...) ...; mq_dyn_end(ctx, slot); mq_scene_commit(ctx);   -- once per frame */
typedef struct mq_view { float origin[3], forward[3], right[3], up[3]; } mq_view; /* r_refdef.vieworg, AngleVectors(r_refdef.viewangles) */
enum { MQ_PT_FIRE = 3, MQ_PT_EXPLODE2 = 5 }; /* quakespasm ptype_t values the heuristics of quake_helpers.cpp:95-113 look at */
typedef struct mq_particle { /* particle_t: org, the fork's mv_prev_origin, vel, d_8to24table[color], type; seed = the fork's per-particle RNG seed (p->die with "reproducible renders", quake_helpers.cpp:82-84) */
    float org[3], prev_org[3], vel[3];
    uint32_t color_rgba;
    int32_t type;
    uint32_t seed;
} mq_particle;
typedef struct mq_alias_instance { /* what add_geo_alias takes from entity_t / lerpdata_t, quake_helpers.cpp:244-303 */
    float origin[3], angles[3];           /* lerpdata.origin / lerpdata.angles (pitch, yaw, roll in degrees, as R_SetupEntityTransform leaves them) */
    float prev_origin[3], prev_angles[3]; /* the same one frame ago (the fork's mv_prev_origin / mv_prev_angles before their sign flip) */
    int32_t pose1, pose2; float blend, prev_blend; /* lerpdata.pose1 / pose2 / blend, ent->mv_prev_blend */
    int32_t skin;                         /* ent->skinnum */
    float fovscale;                       /* view model only: tan(fov / 2) if fov > 90 (quake_helpers.cpp:244-246), else 0 or 1 */
} mq_alias_instance;
typedef struct mq_sprite_instance { float origin[3], prev_origin[3], angles[3]; float scale; int32_t frame; } mq_sprite_instance;
int mq_dyn_begin(mq_ctx* ctx);
int mq_dyn_add_particles(mq_ctx* ctx, const mq_particle* particles, uint32_t n, const mq_view* view, uint32_t texnum_blood, uint32_t texnum_explosion, double cl_time, double prev_cl_time); /* add_particles, quake_helpers.cpp:50-216 */
int mq_dyn_add_alias(mq_ctx* ctx, int alias_model, const mq_alias_instance* inst);                                   /* add_geo_alias, :218-359 */
/* n entities at once, on the library's worker pool (the reference runs add_geo_alias under a parallel_for over the visible entities,
 * quake_node.cpp:904-938): the same triangles, in the same order, as n calls of mq_dyn_add_alias */
int mq_dyn_add_alias_batch(mq_ctx* ctx, const int* alias_models, const mq_alias_instance* inst, uint32_t n);
int mq_dyn_add_sprite(mq_ctx* ctx, int sprite_model, const mq_sprite_instance* inst, const mq_view* view);           /* add_geo_sprite, :471-626 */
int mq_dyn_add_brush_model(mq_ctx* ctx, int bsp_model, const float origin[3], const float angles[3], const float prev_origin[3], const float prev_angles[3]); /* add_geo_brush for an entity, :362-469 */
int mq_dyn_end(mq_ctx* ctx, int slot); /* -> mq_scene_set_geometry(slot, ..., flags 0): alpha tests apply, rebuilt every frame (quake_node.cpp:969-981) */
int mq_bsp_model_count(const mq_ctx* ctx); /* brush models of the BSP loaded by mq_load_bsp (model 0 = the world, already in slots 0 / 1) */
/* MDL ("IDPO" version 6) / SPR ("IDSP" version 1) files; their pictures become textures first_texnum.. (next free number in *next_texnum) */
int mq_load_mdl(mq_ctx* ctx, const char* path, const char* palette_path, uint32_t first_texnum, int* alias_model, uint32_t* next_texnum);
int mq_load_spr(mq_ctx* ctx, const char* path, const char* palette_path, uint32_t first_texnum, int* sprite_model, uint32_t* next_texnum);

/* ---- per-frame uniform and constants producer: QuakeNode::process, src/game/quake_node.cpp:742-824 ----------------
 * What the Quake node reads from quakespasm each frame, as a plain struct; mq_uniform_update turns the PREVIOUS frame's
 * uniform in *u into this frame's (previous camera = last camera, time difference, fog coefficients, sky texture numbers,
 * player flags), mq_constants_fov the two field-of-view constants.  Pure host arithmetic, no context. */
typedef struct mq_frame_state {
    float vieworg[3], viewangles[3]; /* r_refdef.vieworg / viewangles (pitch, yaw, roll in degrees) */
    double cl_time;                  /* cl.time */
    uint32_t frame;                  /* frames since the last worldspawn */
    int32_t render;                  /* render_info.render: a map is loaded and drawn */
    int32_t has_player, weapon, waterlevel; /* sv_player != nullptr, sv_player->v.weapon / waterlevel (demos have no player) */
    int32_t sky_mode;                /* 0: no sky textures, 1: skybox (sky[0..5] = rt bk lf ft up dn), 2: classic (sky[0] solid, sky[1] alpha layer) */
    uint16_t sky[6], notexture;      /* texture numbers; notexture->texnum fills what a mode leaves open */
    int32_t mu_overwrite;            /* the node's "overwrite mu_t / mu_s" switch with its values, else Quake's fog */
    float mu_t, mu_s_div_mu_t[3];
    float fog_density, fog_color[3]; /* Fog_GetDensity(), Fog_GetColor() */
} mq_frame_state;
enum { MQ_PLAYER_FLAGS_TORCH = 1, MQ_PLAYER_FLAGS_UNDERWATER = 2 }; /* res/shader/config.h:39-40 */
int mq_uniform_update(mq_uniform* u, const mq_frame_state* in); /* quake_node.cpp:768-824 */
int mq_constants_fov(mq_constants* k, float fov_x_degrees);     /* quake_node.cpp:762-765 */

#ifdef __cplusplus
}
#endif
#endif /* MQ_H */
