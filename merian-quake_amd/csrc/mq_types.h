// mq_types.h -- POD layouts shared by the host code and the HIP kernels of libmqhip.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#include "../../include/mq.h"

#define MQ_T_MAX 10000.0f          // res/shader/config.h:11
#define MQ_ALPHA_THRESHOLD 0.666f  // res/shader/config.h:13
#define MQ_MAT_FLAGS_LAVA 1
#define MQ_MAT_FLAGS_SLIME 2
#define MQ_MAT_FLAGS_TELE 3        // res/shader/config.h:26-35
#define MQ_MAT_FLAGS_WATER 4
#define MQ_MAT_FLAGS_SKY 5
#define MQ_MAT_FLAGS_WATERFALL 6
#define MQ_MAT_FLAGS_SPRITE 7
#define MQ_MAT_FLAGS_SOLID 8
#define MQ_ML_MAX_N 1024           // res/shader/render_mcpg/mc.glsl:2
#define MQ_ML_MIN_ALPHA 0.01f      // mc.glsl:3
#define MQ_LC_MAX_N 128            // light_cache.glsl:1
#define MQ_LC_MIN_ALPHA 0.01f      // light_cache.glsl:2
#define MQ_MAX_UPDATES 10          // grid.h:29-34
#define MQ_MAX_MC_SAMPLES 30       // the reference's range of "mc samples" / "dist mc samples" (render_mcpg.cpp:460,494); the lobes live in LDS sized at run time
#define MQ_BARY_EPS 3.814697265625e-06f
#define MQ_NIL 0xffffffffu
#define MQ_WIDTH_LUT 48
// Control words.  Every counter that many waves bump is SHARDED 16 ways, one 128-byte line per shard
// (atomics on one line serialise at about 88/us on this chip, whichever word of the line they hit):
// shard s of a queue owns every 16th block of 64 positions.
#define MQ_SHARDS 16
#define MQ_SHARD_STRIDE 32                                  // words between the counters of two shards (128 B)
#define MQ_CTRL_GROUP (MQ_SHARDS * MQ_SHARD_STRIDE)         // words of one sharded counter
#define MQ_CTRL_UPDATES MQ_CTRL_GROUP                       // update-queue tails (group 1; group 0 holds the overflow flag)
// Ray queues: rounds run one after the other (round r's queue is filled by the shading launch of round r - 1 and drained by
// the trace + shading launches of round r), so TWO sets of control words alternate by round parity, however many rounds a
// frame has (spp, max path length and volume spp go up to 15 each, render_mcpg.cpp:487-493): the trace launch of round r,
// which runs between the last reader of round r - 1's words and the first writer of round r + 1's, zeroes the other set.
#define MQ_CTRL_QUEUE0 (2 * MQ_CTRL_GROUP)                  // ray-queue tails, one group per round parity
#define MQ_CTRL_HEAD0 (MQ_CTRL_QUEUE0 + 2 * MQ_CTRL_GROUP)  // traversal fetch heads (64-entry blocks), one group per round parity
#define MQ_CTRL_WORDS (MQ_CTRL_HEAD0 + 2 * MQ_CTRL_GROUP)
#define MQ_QTAILS(round) (MQ_CTRL_QUEUE0 + ((round) & 1) * MQ_CTRL_GROUP)
#define MQ_QHEADS(round) (MQ_CTRL_HEAD0 + ((round) & 1) * MQ_CTRL_GROUP)

// 80-byte compressed 8-wide BVH node (Ylitie et al. 2017 layout).
struct MqNode {
    float px, py, pz;          // quantisation origin
    uint8_t ex, ey, ez, imask; // per-axis scale exponent (biased), internal-child mask
    uint32_t child_base;       // index of first internal child
    uint32_t tri_base;         // index of the node's first leaf record (MqLeafRec)
    uint8_t meta[8];
    uint8_t qlox[8], qloy[8], qloz[8];
    uint8_t qhix[8], qhiy[8], qhiz[8];
};
static_assert(sizeof(MqNode) == 80, "node must be 80 bytes");

// 48-byte triangle record, stored in BVH leaf order.
struct MqTri {
    float v0[3], v1[3], v2[3];
    uint32_t key;   // slot << 28 | prim
    uint32_t flags; // bit0: needs any-hit alpha test, bit1: has distinct prev_vtx
    uint32_t pad;
};
static_assert(sizeof(MqTri) == 48, "triangle must be 48 bytes");
#define MQ_TRI_ANYHIT 1u
#define MQ_TRI_DYNAMIC 2u

// 64-byte LEAF RECORD of the traversal: one or two triangles that share an edge -- the two halves of a brush quad, two
// neighbours of a polygon's fan -- as FOUR vertices instead of six.  A leaf slot of a node (one bit of its hit mask)
// addresses one record: four 16-byte loads and one loop step for two triangle tests, where two 48-byte MqTri records cost six
// loads and two steps.  Triangle A = (v[0], v[1], v[2]); triangle B = (v[s0], v[s1], v[s2]) with two-bit selectors: the
// vertices keep the ORDER they have in the index buffer (the intersection arithmetic, and with it every bit of t, u, v, is
// that of the uploaded triangle).  The MqTri / MqShadeRec arrays (shading) stay: A is triangle tri0 of them, B is tri0 + 1.
struct MqLeafRec {
    float v[4][3];
    uint32_t key0, key1; // slot << 28 | prim of A and B (tie break of equal hit distances)
    uint32_t tri0;
    uint32_t sel;        // bits 0..5: s0 | s1 << 2 | s2 << 4; bit 8: B present; bit 16 / 17: A / B needs the any-hit alpha test
};
static_assert(sizeof(MqLeafRec) == 64, "leaf record must be 64 bytes");
#define MQ_LEAF_HAS_B 0x100u
#define MQ_LEAF_SEL_FAN 0x38u // B = (v0, v2, v3): the second triangle of a quad (a, b, c)(a, c, d) or of a fan -- nearly every record

struct MqTexDesc {
    uint32_t offset; // texel offset into the texel pool; MQ_NIL if the slot is empty
    uint16_t w, h;
    uint32_t flags;  // MQ_TEX_* in bits 0..7; number of mip levels (>= 1) in bits 8..15, level k follows level k-1 in the pool
};

// 64-byte shading record per triangle, in BVH triangle order (same index as MqTri): the triangle's
// 28-byte extra data (scene_info.glsl.h:7-16) with the descriptors of its albedo and fullbright
// textures resolved at commit time, so that a hit needs ONE dependent fetch (triangle + record, issued
// together) before the texel reads instead of triangle -> extra data -> descriptor -> texels.
struct MqShadeRec {
    uint32_t ext[7];
    uint32_t pad0;
    MqTexDesc albedo; // tex[min(texnum, MQ_MAX_GLTEXTURES - 1)]
    uint32_t pad1;
    MqTexDesc fb;     // tex[fullbright texnum] (valid when the texnum is in range; raytrace.glsl:296)
    uint32_t pad2;
};
static_assert(sizeof(MqShadeRec) == 64, "shade record must be 64 bytes");

// 64-byte Markov-chain state (reference MCState is 52 B scalar, grid.h:6-21; the three
// *_change debug fields are never written by this fork and are dropped).
struct MqMCState {
    float w_tgt[3];
    float sum_w;
    float w_cos;
    float T;
    uint32_t id;
    uint32_t n_hash; // N | hash << 16
    uint16_t mv[3];
    uint16_t pad0;
    uint32_t pad[6];
};
static_assert(sizeof(MqMCState) == 64, "mc state must be 64 bytes");

// 16-byte light-cache cell (reference LightCacheVertex is 24 B, grid.h:37-46; its two statistics
// counters live in global counters instead).
struct MqLCCell {
    uint32_t hash;
    uint32_t lock;
    uint16_t irr[3];
    uint16_t N;
};
static_assert(sizeof(MqLCCell) == 16, "lc cell must be 16 bytes");

// 64-byte queued Markov-chain update (one element of the reference's 512-byte MCUpdate slot,
// grid.h:23-35).  Entries of one slot are chained through `next`.
struct MqUpdate {
    float pos[3];
    float weight;
    float target[3];
    uint32_t id;
    float normal[3];
    float T;
    uint16_t mv[3];
    uint16_t rank;   // arrival rank within the slot (0..9): the update pass replays a slot's entries in this order
    uint32_t slot;
    uint32_t next;   // index+1 of the previously pushed entry of this slot, 0 = end
};
static_assert(sizeof(MqUpdate) == 64, "update must be 64 bytes");

// The macro table of src/render_mcpg/render_mcpg.cpp:137-185 as a kernel parameter block.
struct MqParams {
    int32_t reference_mode, adaptive_grid_type, spp, max_path_length, use_light_cache_tail;
    float fov_tan_alpha_half;
    float sun_w[3], sun_color[3];
    int32_t volume_spp, volume_use_light_cache;
    float draine_g, draine_a;
    int32_t mc_samples;
    float mc_samples_adaptive_prob;
    int32_t distance_mc_samples, mc_fast_recovery, lc_grid_type;
    uint32_t lc_buffer_size;
    float lc_grid_steps_per_unit_size, lc_grid_tan_alpha_half, lc_grid_min_width, lc_grid_power;
    uint32_t mc_adaptive_buffer_size;
    float mc_adaptive_grid_tan_alpha_half, mc_adaptive_grid_min_width, mc_adaptive_grid_power,
        mc_adaptive_grid_steps_per_unit_size;
    uint32_t mc_static_buffer_size;
    float mc_static_grid_width;
    int32_t distance_mc_grid_width;
    float volume_max_t, surf_bsdf_p, volume_phase_p, dir_guide_prior, dist_guide_p;
    uint32_t distance_mc_vertex_state_count, seed;
    int32_t gbuffer_hide_sun, quirk_lc_max_wo_p, quirk_n16_wrap;
    int32_t debug_output_selector;
    int32_t volume_forward_project;
    int32_t enable_albedo_mipmap, enable_emission_mipmap; // g-buffer node, gbuffer.cpp:49-50,79-81
    int32_t debug_output_connected; // DEBUG_OUTPUT_CONNECTED, render_mcpg.cpp:182-183 (selector: debug_output_selector above)
    int32_t freeze_learning; // test hook: every learning computation and RNG draw runs, the stores to MC / LC / distance state do not
    int32_t lc_lock_protocol; // "debug: LC lock statistics": the reference's per-cell try-lock (light_cache.glsl:59-64,82-83) with its success / cancel counters
    int32_t lc_try_lock;      // "LC try-lock": the try-lock alone (contended updates are cancelled as in the reference), no counters
    int32_t log_learning;    // test hook: every PROPOSED learning write is appended to MqFrame::learn_log (layouts: include/mq.h, mq_debug_learn_log_read)
    // derived on the host with the same float operations the kernels would use (mq_api.cpp props_to_params)
    float mc_static_inv_width;
    float mc_inv_width_lut[MQ_WIDTH_LUT]; // 1 / width(level) of the adaptive MC grid
    float lc_inv_width_lut[MQ_WIDTH_LUT]; // 1 / width(level) of the light-cache grid
    // constants of grid_level() that depend on the parameters only, evaluated on the host with the kernels' own float
    // code (mq_log, IEEE division): log(power) and 1 / power of the adaptive MC grid and of the light-cache grid
    float mc_log_power, mc_inv_power, lc_log_power, lc_inv_power;
};

struct MqGeoDev {
    const mq_ext* ext;
    const uint32_t* idx;
    const float* prev_vtx;
};

// Everything a kernel needs to see the scene.
struct MqSceneDev {
    const MqNode* nodes;
    const MqTri* tris;
    const MqLeafRec* leaves; // the traversal's leaf records (MqNode::tri_base counts these)
    const MqShadeRec* shade; // one per triangle, same order as tris
    MqGeoDev geo[MQ_MAX_GEOMETRIES];
    const MqTexDesc* tex;
    const float4* texels; // linear RGBA32F, decoded at commit
    uint32_t n_nodes, n_tris;
    uint32_t dyn_root; // root of the per-frame tree (visited after the static tree), MQ_NIL if there is none
};

// 16-byte distance Markov-chain state, grid.h:48-52
struct MqDistMC { float sum_w; uint32_t N; float m0, m1; };

#define MQ_PROF_SECTIONS 40
struct MqCountersDev {
    unsigned long long rays, nodes, tris, segments, guided_segments, lc_touches, mc_updates_accepted,
        mc_updates_dropped, mc_state_reads, pixels, lc_ok, lc_cancel, q_rays, q_nodes, q_tris, q_paths;
    unsigned long long prof[MQ_PROF_SECTIONS]; // -DMQ_PROF builds: shader clocks per code section, summed over waves
    unsigned long long ray_hist[64];           // -DMQ_PROF builds: rays by loop iterations spent in the queue kernel (bins of 8)
};

// Per-frame launch block of the render kernel.
struct MqFrame {
    mq_uniform u;
    uint32_t W, H;
    uint32_t tiles_x, tiles_y;
    uint32_t n_local_tiles; // tiles this rank renders
    uint32_t slot_begin, slot_end; // the pixel slots (64 per local tile) of the sub-pipeline this launch belongs to; [0, n_local_tiles * 64) = the whole rank
    uint32_t rank, world;
    // pixel slot -> tile of the image: local tile l is global tile l * tile_mul + tile_add.  (world, rank): the interleaved
    // partition of the MCPG node; (1, first tile of a band of tile rows): the g-buffer of a row band (ReSTIR / post chain on a rank)
    uint32_t tile_mul, tile_add;
    uint32_t gbuffer_only; // the first-hit kernel writes the g-buffer node's outputs only (no estimator, no radiance, no rays)
    // outputs
    float* irradiance;     // W*H*4 (full image, linear index) -- written for local tiles only
    float* tiles_out;      // n_local_tiles*64*4
    float* volume_tiles_out; // n_local_tiles*64*4: tile-major copy of `volume`
    uint16_t* vdepth_tiles_out; // n_local_tiles*64: tile-major copy of `volume_depth` (the forward projection of the next frame reads every pixel's)
    uint16_t* debug;       // W*H*4 half: "debug" image (mcpg.comp:212-277), when connected
    uint32_t* debug_rng;   // W*H: the pixel's RNG state after its samples, kept for the debug view
    uint16_t* gb_albedo;   // W*H*4 half
    uint16_t* gb_irr;      // W*H*4 half
    uint16_t* gb_mv;       // W*H*2 half
    uint32_t* gbuffer;     // W*H*4 dwords
    uint32_t* hits;        // W*H*10 dwords
    float* volume;         // W*H*4 (volume.comp:237)
    uint16_t* volume_depth;      // W*H half (volume.comp:211)
    uint16_t* prev_volume_depth; // W*H half: last frame's volume_depth (delay-1 feedback)
    uint16_t* volume_mv;   // W*H*2 half
    uint32_t* fp_winner;   // W*H: forward projection, per TARGET pixel the largest linear index + 1 of the pixels that project onto it (0: none)
    float4* dist_mc;       // distance Markov chains: (sum_w, N, m0, m1) per state, 10 states per grid vertex
    uint32_t dist_mc_n;
    // learning state
    MqMCState* mc;
    MqLCCell* lc;
    uint32_t* upd_count;   // per mc slot: entries queued this frame (soft cap at enqueue, exact cap in the update pass)
    uint32_t* upd_head;    // per mc slot, index+1 of the newest queue entry
    MqUpdate* queue;
    uint32_t* active;      // slots with queued updates, listed by the link pass for the apply pass (queue_cap entries, sharded like the queues)
    uint32_t* active_ctrl; // its 16 sharded tails (one MQ_CTRL_GROUP of words), zeroed by the frame's first-hit kernel
    uint32_t queue_cap;
    // control words: see MQ_CTRL_* (sharded tails of the update queue and of every round's ray queue, fetch heads).
    // ctrl: the rank's block (overflow flags, update-queue tails); qctrl: the block of this launch's sub-pipeline (its
    // ray-queue tails and fetch heads per round)
    uint32_t* ctrl;
    uint32_t ctrl_words;   // all control words of the rank (every sub-pipeline's block)
    uint32_t* qctrl;
    // wavefront state: 160-byte path records per pixel slot, rays / hits per queue position,
    // ping-pong queues of pixel slots
    uint4* paths;          // field-major: field k of pixel slot s at paths[k * n_slots + s]
    uint32_t n_slots;      // pixel slots of this rank (tiles * 64)
    float4* rays;
    uint4* ray_hits;
    uint4* prim_hits;      // closest hits of the camera rays, per pixel slot (mq_primary_trace_kernel -> mq_primary_kernel); one buffer per frame parity
    uint32_t* queue_slots[2];
    uint32_t ray_cap;      // positions available in rays / ray_hits / queue_slots (2x the pixel slots: shard imbalance margin)
    MqCountersDev* counters;
    uint32_t count_stats;  // != 0: kernels without a COUNT instantiation may bump `counters` too
    // traversal stack spill area: MQ_SPILL_ENTRIES 8-byte entries per resident lane
    unsigned long long* stack_spill;
    unsigned long long* cam_spill; // the camera-ray kernel's: MQ_SPILL_ENTRIES entries per PIXEL SLOT (it launches one wave per tile, and beside other kernels)
    // statistics of the reference's dumps (render_mcpg.cpp:354-416), kept only while "debug: LC lock statistics" is set:
    uint2* lc_stats;          // per light-cache cell: update_succeeded, update_canceled (grid.h:44-45)
    uint32_t* last_upd_count; // per Markov-chain slot: last_update_count (grid.h:25, compute_updates.comp:121)
    // learning-write log (property "debug: log learning writes"): 64-byte records, count bumped per record
    uint4* learn_log;
    uint32_t* learn_log_count;
    uint32_t learn_log_cap;
    // dynamic LDS of the shading kernels: 8-byte rows of 64 lanes per wave (lobe storage: 3 rows per Markov-chain sample)
    uint32_t lds_rows2;
    uint32_t shade_block;  // threads per block of the kernels that keep lobes in LDS: 256, or fewer when lds_rows2 rows per wave would not fit four waves into a block's LDS
};

// ---- ReSTIR DI node (mq_restir.h) ----
struct MqRestirParams { // the specialisation constants of renderer_restir.cpp:163-176
    int32_t spp;
    uint32_t seed;
    int32_t visibility_shade;
    float temporal_normal_reject_cos, temporal_depth_reject, spatial_normal_reject_cos, spatial_depth_reject;
    int32_t temporal_clamp_m, spatial_radius, temporal_bias_correction, spatial_bias_correction;
    float boiling_filter_strength;
    int32_t spatial_reuse_iterations, apply_mv;
};

struct MqRestirFrame {
    mq_uniform u;
    uint32_t W, H, tiles_x, n_tiles;
    // Row bands (mq_set_partition with world > 1): a launch covers the global tiles [tile_begin, tile_end) -- whole tile rows --;
    // pixel slots (scratch, queue entries) are numbered from slot_tile0; rows [row_lo, row_hi) are those whose g-buffer and
    // previous-frame state this rank holds: a reprojected pixel outside them raises overflow bit 3 (flags[0] |= 8) and
    // counts as "no history".  One rank: [0, n_tiles), 0, [0, H).
    uint32_t tile_begin, tile_end, slot_tile0, row_lo, row_hi;
    uint32_t* flags;
    const uint32_t* hits;      // gbuffer "hits" (CompressedHit, 10 dwords per pixel)
    const uint4* gbuffer;      // gbuffer "gbuffer"
    const uint4* prev_gbuffer; // the same, one frame ago
    const uint32_t* mv;        // gbuffer "mv" (RG16F)
    const uint4* prev_reservoirs; // "reservoirs" output of the previous frame
    uint4* res_a;              // binding 1 of the ping-pong set: `reservoirs` (read-write)
    const uint4* res_read;     // binding 0: `reservoirs_spatial_read`
    float4* irradiance;        // RGBA32F
    float2* moments;           // RG32F
    unsigned long long* stack_spill;
};

