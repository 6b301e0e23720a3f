// mq_api.cpp -- implementation of the C ABI in include/mq.h: the HIP-stream render node that
// stands in for merian-quake's GBuffer + RendererMarkovChain nodes.
//
//   describe / connect   src/render_mcpg/render_mcpg.cpp:36-115, src/gbuffer/gbuffer.cpp:23-66
//   process              src/render_mcpg/render_mcpg.cpp:117-320, src/gbuffer/gbuffer.cpp:68-128
//   properties           src/render_mcpg/render_mcpg.cpp:419-578
//
// There is no CPU rendering path in this library: without a HIP device every device entry point
// returns MQ_ENODEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <random>
#include <string>
#include <vector>

#include <dlfcn.h>

#include "mq_host.h"
#include "mq_devbvh.h"
#include "mq_device.h" // host-callable grid_width(): the per-level tables must carry the kernels' own float results

// launchers implemented in mq_kernels.hip
int mq_launch_primary(const MqSceneDev& sc, const MqParams& P, const MqFrame& F, bool guided, bool count, int grid, hipStream_t s);
int mq_launch_primary_trace(const MqSceneDev& sc, const MqParams& P, const MqFrame& F, bool packet, int grid, hipStream_t s);
int mq_packet_stack_entries();
int mq_launch_trace_queue(const MqSceneDev& sc, const MqFrame& F, int round, bool count, int grid, hipStream_t s);
int mq_launch_bounce(const MqSceneDev& sc, const MqParams& P, const MqFrame& F, int round, bool guided, bool count, int grid, hipStream_t s);
int mq_launch_apply(const MqParams& P, const MqFrame& F, int grid, uint32_t sequential_mc_total, hipStream_t s);
int mq_launch_debug_view(const MqParams& P, const MqFrame& F, int grid, hipStream_t s);
int mq_stack_lds_entries();
int mq_launch_stream_read(const void* src, size_t bytes, uint32_t* sink, int grid, hipStream_t s);
int mq_resident_blocks(bool guided, size_t shade_lds_bytes, int shade_block, int out[4]);
int mq_launch_clear(const MqFrame& F, hipStream_t s);
int mq_launch_untile(const void* gathered, void* image, uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles, uint32_t world, uint32_t tiles_per_rank, hipStream_t s);
int mq_launch_untile16(const void* gathered, void* image, uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles, uint32_t world, uint32_t tiles_per_rank, hipStream_t s);
int mq_launch_trace(const MqSceneDev& sc, const float* org, const float* dir, uint32_t n, uint32_t* prim, float* t, float* uv, unsigned long long* spill, int grid, hipStream_t s);
int mq_launch_math(const MqSceneDev& sc, const MqParams& P, int op, int ni, int no, const float* in, float* out, uint32_t n, hipStream_t s);
int mq_launch_forward_project(const MqParams& P, const MqFrame& F, int grid, hipStream_t s);
int mq_launch_volume_sample(const MqSceneDev& sc, const MqParams& P, const MqFrame& F, int smp, int round, bool count, int grid, hipStream_t s);
int mq_launch_volume_shade(const MqSceneDev& sc, const MqParams& P, const MqFrame& F, int smp, int round, bool count, int grid, hipStream_t s);
int mq_launch_volume_finish(const MqParams& P, const MqFrame& F, int grid, hipStream_t s);
// mq_restir.h
int mq_restir_resident_blocks(int out[4]);
int mq_launch_restir_wavefront(const MqSceneDev& sc, const MqParams& P, const MqRestirParams& R, const MqRestirFrame& F, const MqFrame& FQ, int which, int round, int grid, hipStream_t s);
int mq_launch_restir(const MqSceneDev& sc, const MqParams& P, const MqRestirParams& R, const MqRestirFrame& F, int pass, int grid, hipStream_t s);

// mq_post.hip
int mq_launch_accumulate(const float* accum_params6, uint32_t W, uint32_t H, const void* src, const void* mv, const void* gb, const void* prev_gb, const void* prev_out, const void* prev_hist, void* out, void* hist, int first, const uint32_t rows[4], uint32_t* flags, hipStream_t s);
int mq_launch_compose(uint32_t W, uint32_t row_begin, uint32_t row_end, const void* accum, const void* albedo, const void* vol, const void* emission, const void* direct, void* final_out, hipStream_t s);
int mq_render_block_size();
int mq_spill_entries();

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
};

// Named ranges for rocprofv3 --marker-trace, with the reference's own scope names (MERIAN_PROFILE_SCOPE_GPU "surface", "volume", "volume forward
// project", "copy mv for volume", "clear": render_mcpg.cpp:244,255,283,300,315; the ReSTIR node's "generate samples", "temporal reuse", "spatial
// reuse", "shade": renderer_restir.cpp:189-250).  roctx is looked up at run time (a profiler preloads it; MQ_ROCTX=1 loads it by hand): the
// library has no link-time dependency on it, and without it a range is two null-pointer tests.
struct Roctx {
    int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
    Roctx() {
        void* h = nullptr;
        for (const char* n : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!h && getenv("MQ_ROCTX")) for (const char* n : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) if ((h = dlopen(n, RTLD_NOW))) break;
        if (h) { push = (int (*)(const char*))dlsym(h, "roctxRangePushA"); pop = (int (*)())dlsym(h, "roctxRangePop"); }
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
struct RoctxRange {
    static const Roctx& api() { static const Roctx r; return r; }
    bool on;
    explicit RoctxRange(const char* name) : on(api().push != nullptr) { if (on) api().push(name); }
    void end() { if (on) { api().pop(); on = false; } }
    ~RoctxRange() { end(); }
    RoctxRange(const RoctxRange&) = delete; RoctxRange& operator=(const RoctxRange&) = delete;
};

struct mq_ctx {
    int device = -1;
    std::string err;
    MqProps props;
    mq_constants constants{};
    MqHostGeo geo[MQ_MAX_GEOMETRIES];
    std::vector<MqHostTex> tex;
    MqSynthInfo synth;
    MqProducerState producers;
    // committed scene (host copies kept for stats / debugging)
    std::vector<MqNode> nodes;
    std::vector<MqTri> tris;
    std::vector<MqLeafRec> leaves; // the traversal's leaf records, in the order MqNode::tri_base counts them
    float sah_cost = 0.0f;
    bool committed = false;
    // two trees under one root: static slots (MQ_GEO_STATIC) are rebuilt only when one of them changed, the
    // per-frame slots on every commit (quake_node.cpp:847-983); layout in mq_scene_commit.
    std::vector<MqNode> s_nodes; std::vector<MqTri> s_tris; std::vector<MqLeafRec> s_leaves; // the static tree as built
    float s_sah = 0.0f;
    uint32_t s_depth = 0, d_depth = 0; // levels of 8-wide nodes in the static / per-frame tree (the packet kernel's shared stack must hold them)
    bool static_dirty = true, tex_dirty = true;
    bool joined = false;               // both trees present
    uint32_t n_static_nodes = 0, n_static_tris = 0, n_static_leaves = 0; // the static part of nodes / tris / leaves
    uint32_t dev_static_nodes = 0, dev_static_tris = 0, dev_static_leaves = 0; // static part present in the device arrays (joined layout)
    bool dev_scene_valid = false;
    std::vector<MqTexDesc> texdesc;    // of the last full commit: shading records of per-frame triangles need them
    uint32_t commits_full = 0, commits_dynamic = 0;
    // device scene
    DevBuf d_nodes, d_tris, d_leaves, d_shade, d_texdesc, d_texels;
    DevBuf d_ext[MQ_MAX_GEOMETRIES], d_idx[MQ_MAX_GEOMETRIES], d_prev[MQ_MAX_GEOMETRIES];
    MqSceneDev scene{};
    // Per-frame geometry is DOUBLE-BUFFERED on the device: behind the static part of nodes / tris / leaves / shading records lie two
    // regions of dyn_cap_* entries, and a commit of per-frame geometry writes the one the frames in flight do not read (asynchronously,
    // from pinned staging memory, on its own stream) -- the host builds frame n + 1's tree while the device renders frame n.
    uint32_t dyn_cap_nodes = 0, dyn_cap_tris = 0; // entries per region (0: the device arrays hold no second region)
    static const int MQ_DYN_REGIONS = 3;          // regions of the per-frame part: a commit writes the one the last MQ_DYN_REGIONS - 1 commits did not (the host may prepare frame n + 2 while frame n renders)
    int dyn_parity = 0;                           // region the current scene (c->scene) points at
    DevBuf d_ext_x[MQ_DYN_REGIONS - 1][MQ_MAX_GEOMETRIES], d_idx_x[MQ_DYN_REGIONS - 1][MQ_MAX_GEOMETRIES], d_prev_x[MQ_DYN_REGIONS - 1][MQ_MAX_GEOMETRIES]; // the per-slot arrays of regions 1.. (region 0: d_ext / d_idx / d_prev)
    void* stage[MQ_DYN_REGIONS] = {}; size_t stage_bytes[MQ_DYN_REGIONS] = {}; // pinned staging memory per region
    hipStream_t up_stream = nullptr;
    hipEvent_t ev_uploaded = nullptr; bool uploaded_valid = false;       // the last asynchronous commit has landed
    hipEvent_t ev_scene_used[MQ_DYN_REGIONS] = {}; bool scene_used_valid[MQ_DYN_REGIONS] = {}; // last launch that read region p
    hipStream_t scene_used_stream[MQ_DYN_REGIONS] = {}; bool scene_used_mixed[MQ_DYN_REGIONS] = {}; // (read from more than one stream: the commit falls back to a device synchronisation)
    uint32_t commits_async = 0;
    std::vector<MqTri> flat_scratch;
    // the tree of the per-frame geometry built on the device (property "per-frame BVH"; mq_devbvh.hip)
    DevBuf d_db_scratch; uint32_t db_cap = 0;  // the builder's scratch for up to db_cap triangles
    uint32_t* db_ctr_host[MQ_DYN_REGIONS] = {}; // pinned copies of the builder's counters per region (node count, flags), read after the fact
    bool db_ctr_pending[MQ_DYN_REGIONS] = {};
    bool mirror_from_device = false; uint32_t db_tris[MQ_DYN_REGIONS] = {};
    uint32_t commits_device = 0;
    std::vector<MqNode> pend_nodes; std::vector<MqTri> pend_tris; std::vector<MqLeafRec> pend_leaves; bool mirror_pending = false; // per-frame trees of the last asynchronous commit, not yet in nodes / tris / leaves
    uint32_t n_dyn_nodes = 0, n_dyn_tris = 0, n_dyn_leaves = 0;
    // frame state
    bool connected = false;
    uint32_t W = 0, H = 0, tiles_x = 0, tiles_y = 0;
    int rank = 0, world = 1;
    uint32_t n_local_tiles = 0, tiles_per_rank = 0;
    DevBuf d_out[MQ_OUT_COUNT];
    DevBuf d_mc, d_lc, d_upd_count, d_upd_head, d_queue, d_active, d_active_ctrl, d_ctrl, d_counters, d_spill;
    int restir_occ[4] = {0, 0, 0, 0}; // resident blocks per CU of the ReSTIR pass kernels
    bool queues_dirty = true;      // the ray-queue control words have to be zeroed before the next frame uses them
    DevBuf d_paths, d_rays, d_ray_hits, d_qslots[2], d_debug_rng;
    DevBuf d_prev_vdepth, d_dist_mc, d_fp_winner;
    DevBuf d_restir_pong, d_restir_prev, d_restir_prev_gb; // ReSTIR: ping-pong partner of the "reservoirs" output, last frame's reservoirs and g-buffer (the graph's delay-1 inputs)
    uint64_t restir_iteration = 0; bool restir_seeded = false; uint32_t restir_seed_in_use = 0;
    DevBuf d_post_prev_gb, d_post_prev_out[2], d_post_prev_hist[2]; // post chain: last frame's g-buffer, accumulated images and histories (surface, volume)
    bool post_first = true;
    DevBuf d_lc_stats, d_last_upd; // "debug: LC lock statistics": allocated by the first frame that keeps them
    DevBuf d_learn_log, d_learn_count; // "debug: log learning writes": allocated by the first frame that logs
    uint32_t learn_log_cap = 0;
    uint32_t dist_mc_n = 0;
    uint32_t ray_cap = 0;
    uint32_t queue_cap = 0;
    uint32_t mc_total = 0, lc_total = 0;
    uint64_t iteration = 0;
    bool params_dirty = true;
    MqParams params{};
    bool count_enabled = false;
    int cu_count = 0, grid_blocks = 0;
    int grid_frame[4] = {0, 0, 0, 0}; // first hit, trace, bounce, camera rays: every block resident (see frame_grids)
    int grid_key = -1;             // lds_rows2 the grids were derived for
    static const int EV_RING = 32;
    static const int EV_PER = 4 + 2 * 8; // start, primary trace end, primary shade end, (trace end, bounce end) x up to 8 rounds, apply end
    hipEvent_t evr[EV_RING][EV_PER] = {};
    int ev_rounds[EV_RING] = {};
    double t_primary_sum = 0.0, t_trace_sum = 0.0, t_bounce_sum = 0.0;
    double t_round_trace[MQ_TIMING_ROUNDS] = {}, t_round_shade[MQ_TIMING_ROUNDS] = {};
    bool ev_pending[EV_RING] = {};
    int ev_slot = 0, ev_last = -1;
    double t_render_sum = 0.0, t_update_sum = 0.0; uint32_t t_frames = 0;
    float last_render_ms = 0.0f, last_update_ms = 0.0f;
    bool ev_valid = false;
    bool volume_outputs_zero = false; // "volume" and its tile copy hold zeros: no need to clear them again
    bool ev_detail[EV_RING] = {};     // the slot's frame recorded the per-launch events
    uint32_t ev_counter = 0, t_detail_frames = 0, timing_interval = 1;
    hipStream_t last_stream = nullptr;
    // sub-pipelines (property "pipelines"): the rank's pixel slots are cut into `subs` contiguous ranges, each rendered by
    // its own chain of launches on its own stream (sub 0 on the caller's); they join before the update pass
    static const int MAX_SUBS = 4;
    int subs = 1;                         // in effect since the last connect
    uint32_t sub_slot_begin[MAX_SUBS + 1] = {};
    uint32_t sub_ray_cap = 0;             // queue positions per sub-pipeline (ray_cap = subs * sub_ray_cap)
    hipStream_t side[MAX_SUBS - 1] = {};
    hipEvent_t ev_fork = nullptr, ev_join[MAX_SUBS - 1] = {};
    // camera rays of the NEXT frame beside the kernels of this one (property "overlap camera rays"): they depend on the
    // scene and the camera only.  Traced on pt_stream into the hit buffer of the frame's parity; ev_pt_done[p]: traced,
    // ev_shaded[p]: the first-hit kernel that read buffer p has finished (the buffer may be overwritten).
    hipStream_t pt_stream = nullptr;
    DevBuf d_prim_hits[2];
    DevBuf d_cam_spill[2];         // stack spill areas of the camera-ray kernel, per pixel slot: [0] the MCPG node's launches, [1] the row band's
    DevBuf d_band_hits;            // closest hits of the camera rays of a row band's g-buffer (ReSTIR node / post chain of a rank of a partitioned frame)
    uint32_t slot_tiles = 0;       // tiles the per-slot buffers (path records, hit buffers, ray queues) are sized for: the rank's share of interleaved tiles, or its widest row band
    bool band_gb_valid = false;    // the g-buffer of this rank's rows has been rendered for the frame mq_process last started
    hipEvent_t ev_pt_done[2] = {}, ev_shaded[2] = {};
    hipEvent_t ev_bounced = nullptr; bool bounced_valid = false; // the last bounce kernel of the previous frame has been issued ("update pass" overlap: the camera rays start behind it)
    bool shaded_valid[2] = {false, false};
    uint32_t frame_parity = 0;
    hipEvent_t ev_pt_t[EV_RING][2] = {}; // start / end of the camera-ray launch on pt_stream (frames with per-launch events)
    bool ev_pt_timed[EV_RING] = {};
    double t_pt_kernel_sum = 0.0;
    mq_ctx() : tex(MQ_MAX_GLTEXTURES) {}
};

MqHostGeo& mq_ctx_geo(mq_ctx* c, int slot) { c->static_dirty = true; return c->geo[slot]; }
MqHostTex& mq_ctx_tex(mq_ctx* c, uint32_t t) { c->tex_dirty = true; return c->tex[t]; }
mq_constants& mq_ctx_constants(mq_ctx* c) { return c->constants; }
MqSynthInfo& mq_ctx_synth(mq_ctx* c) { return c->synth; }
MqProducerState& mq_ctx_producers(mq_ctx* c) { return c->producers; }
int mq_ctx_fail(mq_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }
void mq_ctx_clear_scene(mq_ctx* c) {
    for (auto& g : c->geo) g = MqHostGeo();
    for (auto& t : c->tex) t = MqHostTex();
    c->synth = MqSynthInfo();
    c->producers = MqProducerState();
    c->committed = false; c->static_dirty = c->tex_dirty = true;
}

namespace {

int fail(mq_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }
#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((c), MQ_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

void dev_free(DevBuf& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
int dev_alloc(mq_ctx* c, DevBuf& b, size_t bytes) {
    dev_free(b);
    if (bytes == 0) bytes = 16;
    HIPCHK(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    // hipMalloc does not initialise: with MQ_DEBUG_POISON set every fresh buffer is filled with 0xff bytes
    // (NaN floats, MQ_NIL indices), so a read of memory nobody wrote shows up instead of "working by luck"
    static const bool poison = getenv("MQ_DEBUG_POISON") != nullptr;
    if (poison) HIPCHK(c, hipMemset(b.p, 0xff, bytes));
    return MQ_OK;
}
int dev_upload(mq_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    int r = dev_alloc(c, b, bytes);
    if (r) return r;
    if (bytes) HIPCHK(c, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return MQ_OK;
}

// buffers rewritten every frame: keep the allocation while it is large enough (b.bytes is then the capacity)
int dev_update(mq_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    if (!b.p || b.bytes < bytes) { int r = dev_alloc(c, b, bytes + bytes / 2 + 4096); if (r) return r; }
    if (bytes) HIPCHK(c, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return MQ_OK;
}

// ---- property table: the reference's key strings (render_mcpg.cpp:449-548, gbuffer.cpp) ---------
enum PType { PT_BOOL, PT_INT, PT_UINT, PT_FLOAT, PT_OPTION };
struct PropDesc { const char* key; PType type; size_t off; bool reconnect; const char* options[10]; };
#define POFF(f) offsetof(MqProps, f)
const PropDesc k_props[] = {
    {"randomize seed", PT_BOOL, POFF(randomize_seed), false, {}},
    {"seed", PT_UINT, POFF(seed), false, {}},
    {"reference mode", PT_BOOL, POFF(reference_mode), true, {}},
    {"ML Prior", PT_FLOAT, POFF(dir_guide_prior), false, {}},
    {"mc samples", PT_INT, POFF(mc_samples), false, {}},
    {"adaptive grid prob", PT_FLOAT, POFF(mc_samples_adaptive_prob), false, {}},
    {"adaptive grid type", PT_OPTION, POFF(mc_adaptive_grid_type), true, {"exponential", "quadratic"}},
    {"adaptive grid buf size", PT_UINT, POFF(mc_adaptive_buffer_size), true, {}},
    {"adaptive grid tan(alpha/2)", PT_FLOAT, POFF(mc_adaptive_grid_tan_alpha_half), false, {}},
    {"adaptive grid steps per unit", PT_FLOAT, POFF(mc_adaptive_grid_steps_per_unit_size), false, {}},
    {"adaptive grid min width", PT_FLOAT, POFF(mc_adaptive_grid_min_width), false, {}},
    {"adaptive grid power", PT_FLOAT, POFF(mc_adaptive_grid_power), false, {}},
    {"static grid buf size", PT_UINT, POFF(mc_static_buffer_size), true, {}},
    {"mc static width", PT_FLOAT, POFF(mc_static_grid_width), true, {}},
    {"mc fast recovery", PT_BOOL, POFF(mc_fast_recovery), false, {}},
    {"spp", PT_INT, POFF(spp), false, {}},
    {"max path length", PT_INT, POFF(max_path_length), false, {}},
    {"BSDF Prob", PT_FLOAT, POFF(surf_bsdf_p), false, {}},
    {"volume spp", PT_INT, POFF(volume_spp), false, {}},
    {"dist mc samples", PT_INT, POFF(distance_mc_samples), false, {}},
    {"dist mc grid width", PT_INT, POFF(distance_mc_grid_width), true, {}},
    {"dist mc states per vertex", PT_UINT, POFF(distance_mc_vertex_state_count), true, {}},
    {"particle size", PT_FLOAT, POFF(volume_particle_size_um), false, {}},
    {"dist guide p", PT_FLOAT, POFF(dist_guide_p), false, {}},
    {"Phase Prob", PT_FLOAT, POFF(volume_phase_p), false, {}},
    {"volume forward project", PT_BOOL, POFF(volume_forward_project), false, {}},
    {"surf: use LC", PT_BOOL, POFF(use_light_cache_tail), false, {}},
    {"volume: use LC", PT_BOOL, POFF(volume_use_light_cache), false, {}},
    {"LC grid type", PT_OPTION, POFF(lc_grid_type), true, {"exponential", "quadratic"}},
    {"LC buf size", PT_UINT, POFF(lc_buffer_size), true, {}},
    {"LC grid tan(alpha/2)", PT_FLOAT, POFF(lc_grid_tan_alpha_half), false, {}},
    {"LC grid steps per unit", PT_FLOAT, POFF(lc_grid_steps_per_unit_size), false, {}},
    {"LC grid min width", PT_FLOAT, POFF(lc_grid_min_width), false, {}},
    {"LC grid power", PT_FLOAT, POFF(lc_grid_power), false, {}},
    {"debug output", PT_OPTION, POFF(debug_output_selector), false, {"light cache", "mc weight", "mc mean direction", "mc grid", "irradiance", "moments", "mc cos", "mc N", "mc motion vectors"}},
    // GBuffer node (default_config.json:527-535)
    {"hide sun", PT_BOOL, POFF(hide_sun), false, {}},
    {"enable albedo mipmap", PT_BOOL, POFF(enable_albedo_mipmap), false, {}},
    {"enable emission mipmap", PT_BOOL, POFF(enable_emission_mipmap), false, {}},
    {"debug output connected", PT_BOOL, POFF(debug_output_connected), false, {}},
    // post chain: the Accumulate nodes "accum" / "volume accum" of the graph (res/default_config.json:404-428,473-497)
    {"accum: alpha", PT_FLOAT, POFF(accum_alpha), false, {}},
    {"accum: max history", PT_FLOAT, POFF(accum_max_history), false, {}},
    {"accum: normal threshold", PT_FLOAT, POFF(accum_normal_threshold), false, {}},
    {"accum: depth threshold", PT_FLOAT, POFF(accum_depth_threshold), false, {}},
    {"accum: enable motion vectors", PT_BOOL, POFF(accum_enable_mv), false, {}},
    {"accum: reuse border", PT_BOOL, POFF(accum_reuse_border), false, {}},
    {"volume accum: alpha", PT_FLOAT, POFF(vaccum_alpha), false, {}},
    {"volume accum: max history", PT_FLOAT, POFF(vaccum_max_history), false, {}},
    {"volume accum: normal threshold", PT_FLOAT, POFF(vaccum_normal_threshold), false, {}},
    {"volume accum: depth threshold", PT_FLOAT, POFF(vaccum_depth_threshold), false, {}},
    {"volume accum: enable motion vectors", PT_BOOL, POFF(vaccum_enable_mv), false, {}},
    {"volume accum: reuse border", PT_BOOL, POFF(vaccum_reuse_border), false, {}},
    // ReSTIR DI node: renderer_restir.cpp:253-325
    {"restir: randomize seed", PT_BOOL, POFF(restir_randomize_seed), false, {}},
    {"restir: seed", PT_UINT, POFF(restir_seed), false, {}},
    {"restir: spp", PT_INT, POFF(restir_spp), false, {}},
    {"restir: enable temporal reuse", PT_BOOL, POFF(restir_temporal_reuse), false, {}},
    {"restir: temporal normal threshold", PT_FLOAT, POFF(restir_temporal_normal_angle), false, {}},
    {"restir: temporal depth threshold", PT_FLOAT, POFF(restir_temporal_depth), false, {}},
    {"restir: temporal clamp m", PT_INT, POFF(restir_temporal_clamp_m), false, {}},
    {"restir: temporal bias correction", PT_OPTION, POFF(restir_temporal_bias), false, {"none", "basic", "raytraced", "raytraced previous bvh"}},
    {"restir: boiling filter strength", PT_FLOAT, POFF(restir_boiling), false, {}},
    {"restir: apply mv", PT_BOOL, POFF(restir_apply_mv), false, {}},
    {"restir: spatial reuse iterations", PT_INT, POFF(restir_spatial_iterations), false, {}},
    {"restir: spatial normal threshold", PT_FLOAT, POFF(restir_spatial_normal_angle), false, {}},
    {"restir: spatial depth threshold", PT_FLOAT, POFF(restir_spatial_depth), false, {}},
    {"restir: spatital radius", PT_INT, POFF(restir_spatial_radius), false, {}},
    {"restir: spatial bias correction", PT_OPTION, POFF(restir_spatial_bias), false, {"none", "basic", "raytraced"}},
    {"restir: shade visibility", PT_BOOL, POFF(restir_shade_visibility), false, {}},
    {"inline restir rays", PT_BOOL, POFF(restir_inline_rays), false, {}},
    // row partition of the ReSTIR node and the post chain over the ranks of mq_set_partition (no reference counterpart), and the `add` node's ReSTIR input
    {"band: reprojection halo", PT_INT, POFF(band_reprojection_halo), true, {}},
    {"add: restir irradiance", PT_BOOL, POFF(add_restir), false, {}}, // scheduling of this build: trace the generate / shade rays inside the pass kernels instead of through the queues
    {"debug: freeze learning", PT_BOOL, POFF(freeze_learning), false, {}},
    {"debug: log learning writes", PT_BOOL, POFF(log_learning), false, {}},
    {"debug: LC lock statistics", PT_BOOL, POFF(lc_lock_statistics), false, {}},
    {"LC try-lock", PT_BOOL, POFF(lc_try_lock), false, {}},
    {"per-frame BVH", PT_OPTION, POFF(dyn_bvh), false, {"host", "device", "auto"}},
    {"debug: sequential update pass", PT_BOOL, POFF(sequential_update_pass), false, {}},
    // scheduling of this build (no reference counterpart): number of concurrent sub-pipelines a frame is cut into
    {"pipelines", PT_INT, POFF(pipelines), true, {}},
    {"overlap camera rays", PT_OPTION, POFF(overlap_camera_rays), false, {"off", "auto", "always", "update pass", "last round", "last bounce"}},
    {"camera rays: frustum packets", PT_BOOL, POFF(packet_camera_rays), false, {}},
    // named quirk switches of this build (SURVEY Appendix D.4 / mc.glsl:26 uint16 arithmetic)
    {"quirk: LC max(wo_p,10)", PT_BOOL, POFF(quirk_lc_max_wo_p), false, {}},
    {"quirk: 16-bit N*N", PT_BOOL, POFF(quirk_n16_wrap), false, {}},
};
const int k_nprops = (int)(sizeof(k_props) / sizeof(k_props[0]));

const PropDesc* find_prop(const char* key) {
    for (int i = 0; i < k_nprops; i++) if (!strcmp(k_props[i].key, key)) return &k_props[i];
    return nullptr;
}
double prop_get(const MqProps& p, const PropDesc& d) {
    const char* b = (const char*)&p + d.off;
    switch (d.type) {
    case PT_BOOL: return *(const bool*)b ? 1.0 : 0.0;
    case PT_INT: case PT_OPTION: return (double)*(const int*)b;
    case PT_UINT: return (double)*(const uint32_t*)b;
    case PT_FLOAT: return (double)*(const float*)b;
    }
    return 0.0;
}
bool prop_set(MqProps& p, const PropDesc& d, double v) { // returns true if the value changed
    char* b = (char*)&p + d.off;
    switch (d.type) {
    case PT_BOOL: { bool n = v != 0.0; bool ch = *(bool*)b != n; *(bool*)b = n; return ch; }
    case PT_INT: case PT_OPTION: { int n = (int)std::llround(v); bool ch = *(int*)b != n; *(int*)b = n; return ch; }
    case PT_UINT: { uint32_t n = (uint32_t)std::llround(v); bool ch = *(uint32_t*)b != n; *(uint32_t*)b = n; return ch; }
    case PT_FLOAT: { float n = (float)v; bool ch = *(float*)b != n; *(float*)b = n; return ch; }
    }
    return false;
}

// ---- a small JSON reader: enough for merian-quake graph files ---------------------------------
struct JParser {
    const char* s; size_t n, i = 0; bool ok = true;
    JParser(const char* t) : s(t), n(strlen(t)) {}
    void ws() { while (i < n && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++; }
    bool eat(char c) { ws(); if (i < n && s[i] == c) { i++; return true; } return false; }
    std::string str() {
        std::string r; ws();
        if (i >= n || s[i] != '"') { ok = false; return r; }
        i++;
        while (i < n && s[i] != '"') { if (s[i] == '\\' && i + 1 < n) { i++; } r.push_back(s[i++]); }
        if (i < n) i++; else ok = false;
        return r;
    }
    void skip() { // skip any value
        ws();
        if (i >= n) { ok = false; return; }
        if (s[i] == '"') { str(); return; }
        if (s[i] == '{' || s[i] == '[') {
            char open = s[i], close = open == '{' ? '}' : ']'; i++;
            ws();
            if (i < n && s[i] == close) { i++; return; }
            for (;;) {
                if (open == '{') { str(); if (!eat(':')) { ok = false; return; } }
                skip(); if (!ok) return;
                if (eat(',')) continue;
                if (eat(close)) return;
                ok = false; return;
            }
        }
        while (i < n && s[i] != ',' && s[i] != '}' && s[i] != ']' && s[i] != ' ' && s[i] != '\n' && s[i] != '\r' && s[i] != '\t') i++;
    }
    // positions the cursor at the value of `key` inside the object starting at the cursor
    bool find_key(const std::string& key) {
        if (!eat('{')) return false;
        ws();
        if (i < n && s[i] == '}') return false;
        for (;;) {
            std::string k = str();
            if (!ok || !eat(':')) return false;
            if (k == key) return true;
            skip(); if (!ok) return false;
            if (eat(',')) continue;
            return false;
        }
    }
};

void props_to_params(mq_ctx* c) {
    const MqProps& q = c->props; MqParams& P = c->params;
    memset(&P, 0, sizeof P);
    P.reference_mode = (q.reference_mode || q.surf_bsdf_p == 1.0f) ? 1 : 0; // render_mcpg.cpp:139-140
    P.adaptive_grid_type = q.mc_adaptive_grid_type; P.spp = q.spp; P.max_path_length = q.max_path_length;
    P.use_light_cache_tail = q.use_light_cache_tail; P.fov_tan_alpha_half = c->constants.fov_tan_alpha_half;
    for (int k = 0; k < 3; k++) { P.sun_w[k] = c->constants.sun_direction[k]; P.sun_color[k] = c->constants.sun_color[k]; }
    P.volume_spp = q.volume_spp; P.volume_use_light_cache = q.volume_use_light_cache;
    P.draine_g = (float)std::exp(-2.20679 / ((double)q.volume_particle_size_um + 3.91029) - 0.428934); // render_mcpg.cpp:134
    P.draine_a = (float)std::exp(3.62489 - 8.29288 / ((double)q.volume_particle_size_um + 5.52825));   // render_mcpg.cpp:135
    P.mc_samples = q.mc_samples; P.mc_samples_adaptive_prob = q.mc_samples_adaptive_prob;
    P.distance_mc_samples = q.distance_mc_samples; P.mc_fast_recovery = q.mc_fast_recovery; P.lc_grid_type = q.lc_grid_type;
    P.lc_buffer_size = c->lc_total ? c->lc_total : q.lc_buffer_size;
    P.lc_grid_steps_per_unit_size = q.lc_grid_steps_per_unit_size; P.lc_grid_tan_alpha_half = q.lc_grid_tan_alpha_half;
    P.lc_grid_min_width = q.lc_grid_min_width; P.lc_grid_power = q.lc_grid_power;
    P.mc_adaptive_buffer_size = q.mc_adaptive_buffer_size; P.mc_adaptive_grid_tan_alpha_half = q.mc_adaptive_grid_tan_alpha_half;
    P.mc_adaptive_grid_min_width = q.mc_adaptive_grid_min_width; P.mc_adaptive_grid_power = q.mc_adaptive_grid_power;
    P.mc_adaptive_grid_steps_per_unit_size = q.mc_adaptive_grid_steps_per_unit_size;
    P.mc_static_buffer_size = q.mc_static_buffer_size; P.mc_static_grid_width = q.mc_static_grid_width;
    P.distance_mc_grid_width = q.distance_mc_grid_width; P.volume_max_t = c->constants.volume_max_t;
    P.surf_bsdf_p = q.surf_bsdf_p; P.volume_phase_p = q.volume_phase_p; P.dir_guide_prior = q.dir_guide_prior; P.dist_guide_p = q.dist_guide_p;
    P.distance_mc_vertex_state_count = q.distance_mc_vertex_state_count;
    if (q.randomize_seed) { std::random_device dev; std::mt19937 rng(dev()); c->props.seed = (uint32_t)rng(); } // render_mcpg.cpp:127-132
    P.seed = c->props.seed;
    P.gbuffer_hide_sun = q.hide_sun; P.quirk_lc_max_wo_p = q.quirk_lc_max_wo_p; P.quirk_n16_wrap = q.quirk_n16_wrap;
    P.debug_output_selector = q.debug_output_selector;
    P.volume_forward_project = q.volume_forward_project;
    P.enable_albedo_mipmap = q.enable_albedo_mipmap; P.enable_emission_mipmap = q.enable_emission_mipmap; P.freeze_learning = q.freeze_learning; P.log_learning = q.log_learning; P.lc_lock_protocol = q.lc_lock_statistics; P.lc_try_lock = q.lc_try_lock; P.debug_output_connected = q.debug_output_connected;
    P.mc_static_inv_width = 1.0f / P.mc_static_grid_width;
    for (uint32_t l = 0; l < MQ_WIDTH_LUT; l++) {
        P.mc_inv_width_lut[l] = 1.0f / grid_width(P.adaptive_grid_type, P.mc_adaptive_grid_steps_per_unit_size, P.mc_adaptive_grid_min_width, P.mc_adaptive_grid_power, l);
        P.lc_inv_width_lut[l] = 1.0f / grid_width(P.lc_grid_type, P.lc_grid_steps_per_unit_size, P.lc_grid_min_width, P.lc_grid_power, l);
    }
    P.mc_log_power = mq_log(P.mc_adaptive_grid_power); P.mc_inv_power = 1.0f / P.mc_adaptive_grid_power;
    P.lc_log_power = mq_log(P.lc_grid_power); P.lc_inv_power = 1.0f / P.lc_grid_power;
    c->params_dirty = false;
}

void free_frame_state(mq_ctx* c) {
    for (auto& b : c->d_out) dev_free(b);
    dev_free(c->d_mc); dev_free(c->d_lc); dev_free(c->d_upd_count); dev_free(c->d_upd_head); dev_free(c->d_queue); dev_free(c->d_active); dev_free(c->d_active_ctrl);
    dev_free(c->d_ctrl); dev_free(c->d_counters); dev_free(c->d_spill);
    dev_free(c->d_restir_pong); dev_free(c->d_restir_prev); dev_free(c->d_restir_prev_gb);
    dev_free(c->d_post_prev_gb); for (int k = 0; k < 2; k++) { dev_free(c->d_post_prev_out[k]); dev_free(c->d_post_prev_hist[k]); }
    dev_free(c->d_prev_vdepth); dev_free(c->d_dist_mc); dev_free(c->d_fp_winner); dev_free(c->d_learn_log); dev_free(c->d_learn_count); c->learn_log_cap = 0; dev_free(c->d_lc_stats); dev_free(c->d_last_upd);
    dev_free(c->d_prim_hits[0]); dev_free(c->d_prim_hits[1]); dev_free(c->d_band_hits); dev_free(c->d_cam_spill[0]); dev_free(c->d_cam_spill[1]); c->shaded_valid[0] = c->shaded_valid[1] = false; c->bounced_valid = false;
    dev_free(c->d_debug_rng); dev_free(c->d_paths); dev_free(c->d_rays); dev_free(c->d_ray_hits); dev_free(c->d_qslots[0]); dev_free(c->d_qslots[1]);
    c->connected = false;
}
void free_scene_dev(mq_ctx* c) {
    dev_free(c->d_nodes); dev_free(c->d_tris); dev_free(c->d_leaves); dev_free(c->d_shade); dev_free(c->d_texdesc); dev_free(c->d_texels);
    for (int s = 0; s < MQ_MAX_GEOMETRIES; s++) { dev_free(c->d_ext[s]); dev_free(c->d_idx[s]); dev_free(c->d_prev[s]); for (int k = 0; k < mq_ctx::MQ_DYN_REGIONS - 1; k++) { dev_free(c->d_ext_x[k][s]); dev_free(c->d_idx_x[k][s]); dev_free(c->d_prev_x[k][s]); } }
    c->dyn_cap_nodes = c->dyn_cap_tris = 0; c->dyn_parity = 0;
}

// Update-queue capacity: one entry per traced segment of the frame (every segment can queue at most one update) plus
// the slack the 16-way sharded layout needs -- a shard owns every 16th block of 64 positions, so the highest position
// in use is 1024 * ceil(longest tail / 64), which exceeds the entry count when the shards are out of balance.
size_t queue_entries_needed(const mq_ctx* c) {
    const size_t local_px = (size_t)c->tiles_per_rank * 64;
    const size_t segs = local_px * ((size_t)std::max(1, c->props.spp) * (size_t)std::max(1, c->props.max_path_length - 1) + (size_t)std::max(0, c->props.volume_spp));
    return segs + segs / 8 + (size_t)MQ_SHARDS * 64 * 4;
}

// ---- row bands: how the ReSTIR node and the post chain of a partitioned frame are cut ----------
// Rank r owns the tile rows [tiles_y * r / world, tiles_y * (r + 1) / world): it shades, accumulates and composes those pixels.
// Spatial reuse reads neighbours up to "restir: spatital radius" pixels away, so the rank generates and temporally reuses
// reservoirs on its rows widened by ceil(radius / 8) tile rows (`reuse`); temporal reuse and accumulation read last frame's
// state at a reprojected pixel, which may lie "band: reprojection halo" rows further out still (`need`): the rank renders
// the g-buffer of those rows itself and receives the other ranks' rows of last frame's reservoirs / accumulated images.
struct BandRows { uint32_t t0, t1, e0, e1, g0, g1; }; // tile rows: owned, reuse, need
BandRows band_rows(const MqProps& q, uint32_t H, int rank, int world, int radius_override = -1) {
    const uint32_t ty = (H + 7) / 8;
    BandRows b;
    b.t0 = (uint32_t)((uint64_t)ty * (uint32_t)rank / (uint32_t)world); b.t1 = (uint32_t)((uint64_t)ty * ((uint32_t)rank + 1u) / (uint32_t)world);
    const int radius = radius_override >= 0 ? radius_override : (q.restir_spatial_iterations > 0 ? std::max(0, q.restir_spatial_radius) : 0);
    const uint32_t hs = world > 1 ? ((uint32_t)radius + 7u) / 8u : 0u, hm = world > 1 ? ((uint32_t)std::max(0, q.band_reprojection_halo) + 7u) / 8u : 0u;
    b.e0 = b.t0 > hs ? b.t0 - hs : 0u; b.e1 = std::min(ty, b.t1 + hs);
    b.g0 = b.e0 > hm ? b.e0 - hm : 0u; b.g1 = std::min(ty, b.e1 + hm);
    if (b.t0 == b.t1) { b.e0 = b.e1 = b.g0 = b.g1 = b.t0; } // more ranks than tile rows: nothing to do on this one
    return b;
}

const uint32_t k_bpp[MQ_OUT_COUNT] = {16, 8, 8, 4, 16, 40, 16, 16, 2, 4, 16, 8, 16, 4, 16, 4, 16, 16, 8, 64, 2};

void fill_desc(const mq_ctx* c, uint32_t w, uint32_t h, mq_io_desc* d) {
    memset(d, 0, sizeof *d);
    d->width = w; d->height = h;
    size_t px = (size_t)w * h;
    uint32_t tx = (w + 7) / 8, ty = (h + 7) / 8, nt = tx * ty;
    uint32_t tpr = (nt + (uint32_t)c->world - 1) / (uint32_t)c->world;
    for (int i = 0; i < MQ_OUT_COUNT; i++) { d->bytes_per_pixel[i] = k_bpp[i]; d->bytes[i] = px * k_bpp[i]; }
    d->bytes[MQ_OUT_TILES] = (size_t)tpr * 64 * 16;
    d->bytes[MQ_OUT_VOLUME_TILES] = (size_t)tpr * 64 * 16;
    d->bytes[MQ_OUT_VOLUME_DEPTH_TILES] = (size_t)tpr * 64 * 2;
    size_t mc_total = (size_t)c->props.mc_adaptive_buffer_size + c->props.mc_static_buffer_size; // render_mcpg.cpp:59
    d->state_bytes_markovchain = mc_total * sizeof(MqMCState) + mc_total * 8; // states + per-slot count and chain head of the update queue
    d->state_bytes_lightcache = (size_t)c->props.lc_buffer_size * sizeof(MqLCCell);
    size_t local_px = (size_t)tpr * 64;
    size_t segs = local_px * (size_t)std::max(1, c->props.spp) * (size_t)std::max(1, c->props.max_path_length - 1);
    segs += local_px * (size_t)std::max(0, c->props.volume_spp); // the volume pass queues Markov-chain updates too
    d->state_bytes_update_queue = (segs + segs / 8 + (size_t)MQ_SHARDS * 64 * 4) * sizeof(MqUpdate); // + shard slack, queue_entries_needed()
    const uint32_t gw = (uint32_t)std::max(1, c->props.distance_mc_grid_width); // render_mcpg.cpp:80-82
    d->state_bytes_volume_distancemc = (size_t)(w / gw + 2) * (h / gw + 2) * 10 * sizeof(MqDistMC);
}

} // namespace

// ================================================================================================
extern "C" {

int mq_abi_version(void) { return MQ_ABI_VERSION; }

int mq_create(mq_ctx** out, int device) {
    if (!out) return MQ_EINVAL;
    *out = nullptr;
    mq_ctx* c = new (std::nothrow) mq_ctx();
    if (!c) return MQ_ENOMEM;
    c->device = -1;
    c->constants.fov = 90.0f; c->constants.fov_tan_alpha_half = 1.0f; c->constants.volume_max_t = 1000.0f;
    c->constants.sun_direction[0] = c->constants.sun_direction[1] = c->constants.sun_direction[2] = 0.57735026919f;
    if (device >= 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || device >= n) { delete c; return MQ_ENODEVICE; }
        if (hipSetDevice(device) != hipSuccess) { delete c; return MQ_ENODEVICE; }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete c; return MQ_ENODEVICE; }
        c->cu_count = prop.multiProcessorCount;
        c->device = device;
        for (auto& tr : c->evr) for (auto& e3 : tr) if (hipEventCreate(&e3) != hipSuccess) { delete c; return MQ_EHIP; }
        bool ok = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
        for (int k = 0; ok && k < mq_ctx::MAX_SUBS - 1; k++)
            ok = hipStreamCreateWithFlags(&c->side[k], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&c->ev_join[k], hipEventDisableTiming) == hipSuccess;
        { // its own hardware queue: streams of one priority share a small pool of queues round robin (this one landed on the
          // caller's queue and its launches ran in line with the frame); a lower priority has its own pool, and fill-in work is what it is
            int least = 0, greatest = 0;
            if (ok && (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || hipStreamCreateWithPriority(&c->pt_stream, hipStreamNonBlocking, least) != hipSuccess)) {
                (void)hipGetLastError(); // no priorities on this device: an ordinary stream (the overlap may then not happen, the results are the same)
                ok = hipStreamCreateWithFlags(&c->pt_stream, hipStreamNonBlocking) == hipSuccess;
            }
        }
        for (int k = 0; ok && k < 2; k++) ok = hipEventCreateWithFlags(&c->ev_pt_done[k], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_shaded[k], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&c->ev_bounced, hipEventDisableTiming) == hipSuccess;
        for (auto& pr : c->ev_pt_t) for (auto& e4 : pr) ok = ok && hipEventCreate(&e4) == hipSuccess;
        if (!ok) { delete c; return MQ_EHIP; }
    }
    *out = c;
    return MQ_OK;
}

void mq_destroy(mq_ctx* c) {
    if (!c) return;
    if (c->device >= 0) {
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        free_frame_state(c); free_scene_dev(c);
        for (auto& tr : c->evr) for (auto& e : tr) if (e) (void)hipEventDestroy(e);
        if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
        if (c->pt_stream) (void)hipStreamDestroy(c->pt_stream);
        for (int k = 0; k < 2; k++) { if (c->ev_pt_done[k]) (void)hipEventDestroy(c->ev_pt_done[k]); if (c->ev_shaded[k]) (void)hipEventDestroy(c->ev_shaded[k]); }
        if (c->ev_bounced) (void)hipEventDestroy(c->ev_bounced);
        if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
        if (c->ev_uploaded) (void)hipEventDestroy(c->ev_uploaded);
        for (int k = 0; k < mq_ctx::MQ_DYN_REGIONS; k++) { if (c->ev_scene_used[k]) (void)hipEventDestroy(c->ev_scene_used[k]); if (c->stage[k]) (void)hipHostFree(c->stage[k]); if (c->db_ctr_host[k]) (void)hipHostFree(c->db_ctr_host[k]); }
        dev_free(c->d_db_scratch);
        for (auto& pr : c->ev_pt_t) for (auto& e4 : pr) if (e4) (void)hipEventDestroy(e4);
        for (int k = 0; k < mq_ctx::MAX_SUBS - 1; k++) { if (c->ev_join[k]) (void)hipEventDestroy(c->ev_join[k]); if (c->side[k]) (void)hipStreamDestroy(c->side[k]); }
    }
    delete c;
}

const char* mq_last_error(const mq_ctx* c) { return c ? c->err.c_str() : "null context"; }

// ---- properties --------------------------------------------------------------------------------
int mq_property_count(void) { return k_nprops; }
const char* mq_property_name(int i) { return (i >= 0 && i < k_nprops) ? k_props[i].key : nullptr; }
int mq_property_type(int i) {
    if (i < 0 || i >= k_nprops) return MQ_EINVAL;
    switch (k_props[i].type) { case PT_BOOL: return MQ_PROP_BOOL; case PT_INT: return MQ_PROP_INT; case PT_UINT: return MQ_PROP_UINT; case PT_FLOAT: return MQ_PROP_FLOAT; default: return MQ_PROP_OPTION; }
}
const char* mq_property_option(int i, int k) { return (i >= 0 && i < k_nprops && k >= 0 && k < 10) ? k_props[i].options[k] : nullptr; }

int mq_set_property(mq_ctx* c, const char* key, double value) {
    if (!c || !key) return MQ_EINVAL;
    const PropDesc* d = find_prop(key);
    if (!d) return fail(c, MQ_EUNKNOWN_KEY, std::string("unknown property: ") + key);
    if ((!strcmp(key, "mc samples") || !strcmp(key, "dist mc samples")) && (value < 0 || value > MQ_MAX_MC_SAMPLES)) return fail(c, MQ_EINVAL, std::string(key) + " must be in [0, 30] (render_mcpg.cpp:460,494)");
    if ((!strcmp(key, "spp") || !strcmp(key, "max path length") || !strcmp(key, "volume spp") || !strcmp(key, "restir: spp")) && (value < 0 || value > 15))
        return fail(c, MQ_EINVAL, std::string(key) + " must be in [0, 15] (render_mcpg.cpp:487-493, renderer_restir.cpp:268)");
    if (d->type == PT_OPTION) { int nopt = 0; while (nopt < 10 && d->options[nopt]) nopt++; if (value < 0 || value >= nopt) return fail(c, MQ_EINVAL, std::string("option index out of range for ") + key); }
    bool changed = prop_set(c->props, *d, value);
    if (changed) c->params_dirty = true;
    if (changed && !strncmp(key, "restir: ", 8)) c->restir_seeded = false; // "recreate pipeline": a new seed if they are randomized (renderer_restir.cpp:152-158)
    if (changed && d->reconnect) { c->connected = false; return 1; } // NEEDS_RECONNECT, render_mcpg.cpp:567-575
    return 0;
}
int mq_set_property_str(mq_ctx* c, const char* key, const char* value) {
    if (!c || !key || !value) return MQ_EINVAL;
    const PropDesc* d = find_prop(key);
    if (!d) return fail(c, MQ_EUNKNOWN_KEY, std::string("unknown property: ") + key);
    if (d->type == PT_OPTION) {
        for (int i = 0; i < 10 && d->options[i]; i++) if (!strcmp(d->options[i], value)) return mq_set_property(c, key, i);
        return fail(c, MQ_EINVAL, std::string("unknown option '") + value + "' for " + key);
    }
    if (d->type == PT_BOOL) { if (!strcmp(value, "true")) return mq_set_property(c, key, 1); if (!strcmp(value, "false")) return mq_set_property(c, key, 0); }
    char* end = nullptr; double v = strtod(value, &end);
    if (end == value) return fail(c, MQ_EINVAL, std::string("not a number: ") + value);
    return mq_set_property(c, key, v);
}
int mq_get_property(const mq_ctx* c, const char* key, double* value) {
    if (!c || !key || !value) return MQ_EINVAL;
    const PropDesc* d = find_prop(key);
    if (!d) return MQ_EUNKNOWN_KEY;
    *value = prop_get(c->props, *d);
    return MQ_OK;
}
void mq_properties_header_defaults(mq_ctx* c) { if (!c) return; c->props = MqProps(); c->params_dirty = true; c->connected = false; }
void mq_properties_json_defaults(mq_ctx* c) { // res/default_config.json:527-535,599-638
    if (!c) return;
    MqProps p;
    p.surf_bsdf_p = 0.1f; p.lc_buffer_size = 4000037; p.lc_grid_min_width = 0.01f; p.lc_grid_power = 2.0f; p.lc_grid_steps_per_unit_size = 6.0f;
    p.lc_grid_tan_alpha_half = 0.005f; p.lc_grid_type = 1; p.dir_guide_prior = 0.3f; p.volume_phase_p = 0.1f;
    p.mc_adaptive_buffer_size = 32777259; p.mc_adaptive_grid_min_width = 0.01f; p.mc_adaptive_grid_power = 1.7320508f; p.mc_samples_adaptive_prob = 0.7f;
    p.mc_adaptive_grid_steps_per_unit_size = 1.0f; p.mc_adaptive_grid_tan_alpha_half = 0.002f; p.mc_adaptive_grid_type = 0;
    p.dist_guide_p = 0.9f; p.distance_mc_grid_width = 25; p.distance_mc_samples = 3; p.distance_mc_vertex_state_count = 10;
    p.max_path_length = 3; p.mc_fast_recovery = true; p.mc_samples = 5; p.mc_static_grid_width = 25.3f; p.volume_particle_size_um = 7.0f;
    p.randomize_seed = true; p.reference_mode = false; p.spp = 2; p.mc_static_buffer_size = 800009; p.use_light_cache_tail = false;
    p.volume_forward_project = true; p.volume_spp = 2; p.volume_use_light_cache = true;
    c->props = p; c->params_dirty = true; c->connected = false;
}

int mq_load_properties_json(mq_ctx* c, const char* json_text, const char* node_name) {
    if (!c || !json_text || !node_name) return MQ_EINVAL;
    // graph files keep nodes under {"graph": {"nodes": {...}}} or {"nodes": {...}}; accept both, or a bare properties object
    const char* paths[3][4] = {{"graph", "nodes", node_name, "properties"}, {"nodes", node_name, "properties", nullptr}, {node_name, "properties", nullptr, nullptr}};
    for (auto& path : paths) {
        JParser jp(json_text);
        bool found = true;
        for (int k = 0; k < 4 && path[k]; k++) if (!jp.find_key(path[k])) { found = false; break; }
        if (!found) continue;
        if (!jp.eat('{')) return fail(c, MQ_EIO, "properties is not an object");
        int applied = 0, reconnect = 0;
        jp.ws();
        if (jp.i < jp.n && jp.s[jp.i] == '}') return 0;
        for (;;) {
            std::string k = jp.str();
            if (!jp.ok || !jp.eat(':')) return fail(c, MQ_EIO, "malformed properties object");
            jp.ws();
            size_t v0 = jp.i; bool is_str = jp.i < jp.n && jp.s[jp.i] == '"';
            std::string sval;
            if (is_str) sval = jp.str(); else { jp.skip(); sval.assign(jp.s + v0, jp.i - v0); }
            if (!jp.ok) return fail(c, MQ_EIO, "malformed property value");
            const std::string prefixed = std::string(node_name) + ": " + k; // the post chain's nodes: "accum: alpha", "volume accum: alpha", ...
            const std::string& key = find_prop(prefixed.c_str()) ? prefixed : k;
            if (find_prop(key.c_str())) { int r = mq_set_property_str(c, key.c_str(), sval.c_str()); if (r < 0) return r; reconnect |= r; applied++; }
            if (jp.eat(',')) continue;
            break;
        }
        (void)applied;
        return reconnect;
    }
    return fail(c, MQ_EIO, std::string("node not found in json: ") + node_name);
}

// ---- scene -------------------------------------------------------------------------------------
int mq_scene_set_geometry(mq_ctx* c, int slot, const float* vtx, const float* prev_vtx, uint32_t n_vtx, const uint32_t* idx, const mq_ext* ext, uint32_t n_tri, uint32_t flags) {
    if (!c || slot < 0 || slot >= MQ_MAX_GEOMETRIES) return fail(c, MQ_EINVAL, "geometry slot out of range");
    MqHostGeo& g = c->geo[slot];
    if (((g.flags & MQ_GEO_STATIC) && g.n_tri()) || ((flags & MQ_GEO_STATIC) && n_tri)) c->static_dirty = true;
    g = MqHostGeo();
    c->committed = false;
    if (n_tri == 0) return MQ_OK;
    if (!vtx || !idx || !ext) return fail(c, MQ_EINVAL, "null geometry arrays");
    if (n_tri >= (1u << 28)) return fail(c, MQ_EINVAL, "too many triangles in one slot");
    for (uint32_t i = 0; i < 3 * n_tri; i++) if (idx[i] >= n_vtx) return fail(c, MQ_EINVAL, "index out of range");
    g.vtx.assign(vtx, vtx + 3 * (size_t)n_vtx);
    g.prev_vtx.assign(prev_vtx ? prev_vtx : vtx, (prev_vtx ? prev_vtx : vtx) + 3 * (size_t)n_vtx);
    g.idx.assign(idx, idx + 3 * (size_t)n_tri);
    g.ext.assign(ext, ext + n_tri);
    g.flags = flags;
    return MQ_OK;
}
int mq_scene_set_texture(mq_ctx* c, uint32_t texnum, uint32_t w, uint32_t h, const uint8_t* rgba8, uint32_t flags) {
    if (!c || texnum >= MQ_MAX_GLTEXTURES) return fail(c, MQ_EINVAL, "texnum out of range");
    if (w > 65535 || h > 65535) return fail(c, MQ_EINVAL, "texture too large");
    MqHostTex& t = c->tex[texnum];
    t = MqHostTex();
    c->committed = false; c->tex_dirty = true;
    if (!rgba8 || !w || !h) return MQ_OK;
    t.w = w; t.h = h; t.flags = flags; t.px.assign(rgba8, rgba8 + (size_t)w * h * 4);
    return MQ_OK;
}
int mq_scene_get_geometry(const mq_ctx* c, int slot, const float** vtx, const float** prev_vtx, uint32_t* n_vtx, const uint32_t** idx, const mq_ext** ext, uint32_t* n_tri, uint32_t* flags) {
    if (!c || slot < 0 || slot >= MQ_MAX_GEOMETRIES) return MQ_EINVAL;
    const MqHostGeo& g = c->geo[slot];
    if (vtx) *vtx = g.vtx.data(); if (prev_vtx) *prev_vtx = g.prev_vtx.data(); if (n_vtx) *n_vtx = (uint32_t)(g.vtx.size() / 3);
    if (idx) *idx = g.idx.data(); if (ext) *ext = g.ext.data(); if (n_tri) *n_tri = g.n_tri(); if (flags) *flags = g.flags;
    return MQ_OK;
}
int mq_scene_get_texture(const mq_ctx* c, uint32_t texnum, uint32_t* w, uint32_t* h, const uint8_t** rgba8, uint32_t* flags) {
    if (!c || texnum >= MQ_MAX_GLTEXTURES) return MQ_EINVAL;
    const MqHostTex& t = c->tex[texnum];
    if (w) *w = t.w; if (h) *h = t.h; if (rgba8) *rgba8 = t.px.empty() ? nullptr : t.px.data(); if (flags) *flags = t.flags;
    return MQ_OK;
}
static void materialize_mirror(const mq_ctx* cc) { // (the inspection calls take a const context; the mirror is a cache)
    mq_ctx* c = const_cast<mq_ctx*>(cc);
    if (!c->mirror_pending) return;
    if (c->mirror_from_device) { // a tree built on the device: read its region back (inspection only; waits for the device)
        const int p = c->dyn_parity;
        const size_t ns = c->n_static_nodes, ts = c->n_static_tris, ls = c->n_static_leaves, td = c->db_tris[p];
        const size_t on = ns + (size_t)p * c->dyn_cap_nodes, ot = ts + (size_t)p * c->dyn_cap_tris, ol = ls + (size_t)p * c->dyn_cap_tris;
        (void)hipSetDevice(c->device); (void)hipDeviceSynchronize();
        const size_t nd = td ? c->db_ctr_host[p][MQ_DB_NODES] : 0, ld = td ? c->db_ctr_host[p][MQ_DB_LEAVES] : 0;
        c->nodes.resize(ns + nd); c->tris.resize(ts + td); c->leaves.resize(ls + ld);
        if (nd) (void)hipMemcpy(c->nodes.data() + ns, (const MqNode*)c->d_nodes.p + on, nd * sizeof(MqNode), hipMemcpyDeviceToHost);
        if (td) (void)hipMemcpy(c->tris.data() + ts, (const MqTri*)c->d_tris.p + ot, td * sizeof(MqTri), hipMemcpyDeviceToHost);
        if (ld) (void)hipMemcpy(c->leaves.data() + ls, (const MqLeafRec*)c->d_leaves.p + ol, ld * sizeof(MqLeafRec), hipMemcpyDeviceToHost);
        for (size_t j = 0; j < nd; j++) { MqNode& n = c->nodes[ns + j]; n.child_base -= (uint32_t)(on - ns); n.tri_base -= (uint32_t)(ol - ls); } // the contiguous numbering
        for (size_t j = 0; j < ld; j++) c->leaves[ls + j].tri0 -= (uint32_t)(ot - ts);
        c->mirror_pending = false;
        return;
    }
    const size_t ns = c->n_static_nodes, ts = c->n_static_tris, ls = c->n_static_leaves, nd = c->pend_nodes.size(), td = c->pend_tris.size(), ld = c->pend_leaves.size();
    c->nodes.resize(ns + nd); c->tris.resize(ts + td); c->leaves.resize(ls + ld);
    for (size_t j = 0; j < nd; j++) { MqNode n = c->pend_nodes[j]; n.child_base += (uint32_t)ns; n.tri_base += (uint32_t)ls; c->nodes[ns + j] = n; }
    std::copy(c->pend_tris.begin(), c->pend_tris.end(), c->tris.begin() + (ptrdiff_t)ts);
    for (size_t j = 0; j < ld; j++) { MqLeafRec r = c->pend_leaves[j]; r.tri0 += (uint32_t)ts; c->leaves[ls + j] = r; }
    c->mirror_pending = false;
}
int mq_scene_get_bvh(const mq_ctx* c, const void** nodes, uint64_t* n_nodes, const void** tris, uint64_t* n_tris) {
    if (!c) return MQ_EINVAL;
    if (!c->committed) return MQ_ESTATE;
    materialize_mirror(c);
    if (nodes) *nodes = c->nodes.data(); if (n_nodes) *n_nodes = c->nodes.size(); if (tris) *tris = c->tris.data(); if (n_tris) *n_tris = c->tris.size();
    return MQ_OK;
}
int mq_scene_get_leaves(const mq_ctx* c, const void** leaves, uint64_t* n_leaves) {
    if (!c) return MQ_EINVAL;
    if (!c->committed) return MQ_ESTATE;
    materialize_mirror(c);
    if (leaves) *leaves = c->leaves.data(); if (n_leaves) *n_leaves = c->leaves.size();
    return MQ_OK;
}
int mq_scene_stats(const mq_ctx* c, uint64_t* n_tris, uint64_t* n_nodes, uint64_t* bvh_bytes, float* sah_cost) {
    if (!c) return MQ_EINVAL;
    materialize_mirror(c);
    if (n_tris) *n_tris = c->tris.size(); if (n_nodes) *n_nodes = c->nodes.size();
    if (bvh_bytes) *bvh_bytes = c->nodes.size() * sizeof(MqNode) + c->leaves.size() * sizeof(MqLeafRec); // what a traversal reads (the 48-byte triangles serve the shading)
    if (sah_cost) *sah_cost = c->sah_cost;
    return MQ_OK;
}

namespace {
void flatten_slots(mq_ctx* c, bool want_static, std::vector<MqTri>& flat) {
    size_t total = 0;
    for (int s = 0; s < MQ_MAX_GEOMETRIES; s++) if (((c->geo[s].flags & MQ_GEO_STATIC) != 0) == want_static) total += c->geo[s].n_tri();
    flat.resize(total);
    size_t at = 0;
    for (int s = 0; s < MQ_MAX_GEOMETRIES; s++) {
        MqHostGeo& g = c->geo[s];
        if (((g.flags & MQ_GEO_STATIC) != 0) != want_static) continue;
        g.dynamic = g.prev_vtx.size() == g.vtx.size() && !g.vtx.empty() && memcmp(g.prev_vtx.data(), g.vtx.data(), g.vtx.size() * 4) != 0;
        const uint32_t tflags = ((g.flags & MQ_GEO_OPAQUE) ? 0u : MQ_TRI_ANYHIT) | (g.dynamic ? MQ_TRI_DYNAMIC : 0u);
        MqTri* out = flat.data() + at;
        mq_parallel_for(g.n_tri(), 8192, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; i++) {
                MqTri t; memset(&t, 0, sizeof t);
                memcpy(t.v0, &g.vtx[3 * (size_t)g.idx[3 * i]], 12); memcpy(t.v1, &g.vtx[3 * (size_t)g.idx[3 * i + 1]], 12); memcpy(t.v2, &g.vtx[3 * (size_t)g.idx[3 * i + 2]], 12);
                t.key = ((uint32_t)s << 28) | (uint32_t)i;
                t.flags = tflags;
                out[i] = t;
            }
        });
        at += g.n_tri();
    }
}

void shade_records_into(const mq_ctx* c, const MqTri* tris, size_t n, MqShadeRec* recs);
void shade_records(const mq_ctx* c, const MqTri* tris, size_t n, std::vector<MqShadeRec>& recs) { recs.resize(n); shade_records_into(c, tris, n, recs.data()); }
void par_copy(void* dst, const void* src, size_t bytes) { // memcpy on the worker pool (one thread does not saturate the memory system)
    mq_parallel_for(bytes, (size_t)256 << 10, [&](size_t b, size_t e) { memcpy((char*)dst + b, (const char*)src + b, e - b); });
}
void shade_records_into(const mq_ctx* c, const MqTri* tris, size_t n, MqShadeRec* recs) {
    const std::vector<MqTexDesc>& desc = c->texdesc;
    mq_parallel_for(n, 8192, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) {
            const uint32_t key = tris[i].key;
            const mq_ext& x = c->geo[key >> 28].ext[key & 0x0fffffffu];
            MqShadeRec& q = recs[i]; memset(&q, 0, sizeof q);
            static_assert(sizeof(mq_ext) == 28, "extra data is 7 dwords");
            memcpy(q.ext, &x, 28);
            q.albedo = desc[std::min<uint32_t>(x.texnum_alpha & 0xfffu, MQ_MAX_GLTEXTURES - 1)];
            const uint32_t fb = x.texnum_fb_flags & 0xfffu;
            if (fb < MQ_MAX_GLTEXTURES) q.fb = desc[fb]; else { q.fb.offset = MQ_NIL; }
        }
    });
}

// per-slot arrays the kernels read for triangles with distinct previous positions (raytrace.glsl:226-228)
int upload_slot_arrays(mq_ctx* c, bool statics) {
    int r;
    for (int s = 0; s < MQ_MAX_GEOMETRIES; s++) {
        MqHostGeo& g = c->geo[s];
        if (((g.flags & MQ_GEO_STATIC) != 0) != statics) continue;
        c->scene.geo[s].ext = nullptr; c->scene.geo[s].idx = nullptr; c->scene.geo[s].prev_vtx = nullptr;
        if (!g.n_tri()) continue;
        if ((r = dev_update(c, c->d_ext[s], g.ext.data(), g.ext.size() * sizeof(mq_ext)))) return r;
        c->scene.geo[s].ext = (const mq_ext*)c->d_ext[s].p;
        if (g.dynamic) {
            if ((r = dev_update(c, c->d_idx[s], g.idx.data(), g.idx.size() * 4))) return r;
            if ((r = dev_update(c, c->d_prev[s], g.prev_vtx.data(), g.prev_vtx.size() * 4))) return r;
            c->scene.geo[s].idx = (const uint32_t*)c->d_idx[s].p; c->scene.geo[s].prev_vtx = (const float*)c->d_prev[s].p;
        }
    }
    return MQ_OK;
}

// Launches that read the scene: scene_ready() first (the last asynchronous commit of per-frame geometry must have landed),
// scene_used() behind the last of them (the next commit but one overwrites the region they read).
int scene_ready(mq_ctx* c, hipStream_t s) {
    if (c->uploaded_valid) HIPCHK(c, hipStreamWaitEvent(s, c->ev_uploaded, 0));
    return MQ_OK;
}
int scene_used(mq_ctx* c, hipStream_t s) {
    if (!c->up_stream) return MQ_OK; // no asynchronous commit so far: the first one synchronises with the device
    const int p = c->dyn_parity;
    if (c->scene_used_valid[p] && c->scene_used_stream[p] != s) c->scene_used_mixed[p] = true;
    HIPCHK(c, hipEventRecord(c->ev_scene_used[p], s));
    c->scene_used_valid[p] = true; c->scene_used_stream[p] = s;
    return MQ_OK;
}

// The streams and events of the asynchronous commits (created by the first of them).
int ensure_upload_stream(mq_ctx* c) {
    if (c->up_stream) return MQ_OK;
    HIPCHK(c, hipDeviceSynchronize()); // whatever read the scene so far is unknown to the events below
    { // a priority of its own = a hardware queue of its own: streams of one priority share a small pool of queues, and copies queued behind the frame's
      // kernels would land only when those are done -- too late for the next frame's camera rays, which run beside them and wait for the upload
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || hipStreamCreateWithPriority(&c->up_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
        }
    }
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_uploaded, hipEventDisableTiming));
    for (int k = 0; k < mq_ctx::MQ_DYN_REGIONS; k++) HIPCHK(c, hipEventCreateWithFlags(&c->ev_scene_used[k], hipEventDisableTiming));
    return MQ_OK;
}

// A commit of per-frame geometry whose tree is built ON THE DEVICE (property "per-frame BVH"): the flattened triangles and the
// per-slot arrays go up through the staging memory of region p, mq_devbvh.hip builds nodes, leaf records, triangles and shading
// records in place on the upload stream.  Same regions, same events as the host-built asynchronous commit.
int commit_per_frame_on_device(mq_ctx* c, const std::vector<MqTri>& flat) {
    const size_t ns = c->s_nodes.size(), ts = c->s_tris.size(), ls = c->s_leaves.size(), td = flat.size();
    int r;
    if ((r = ensure_upload_stream(c))) return r;
    static const bool times = getenv("MQ_DEBUG_COMMIT_TIMES") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    const int p = (c->dyn_parity + 1) % mq_ctx::MQ_DYN_REGIONS;
    const bool had_reader = c->scene_used_valid[p] || c->scene_used_mixed[p];
    if (c->scene_used_mixed[p]) { HIPCHK(c, hipDeviceSynchronize()); }
    else if (c->scene_used_valid[p]) HIPCHK(c, hipEventSynchronize(c->ev_scene_used[p]));
    c->scene_used_valid[p] = false; c->scene_used_mixed[p] = false;
    if (!had_reader) HIPCHK(c, hipStreamSynchronize(c->up_stream)); // (nobody rendered from this region since it was written: its upload -- the last reader of its staging memory -- may still be on its way)
    const auto t_1 = std::chrono::steady_clock::now();
    for (int k = p; k == p; k++) if (c->db_ctr_pending[k]) { // what the last build into this region reported (the frames that read it have finished: so has the copy of its counters)
        c->db_ctr_pending[k] = false;
        if (c->db_ctr_host[k][MQ_DB_ERR]) return fail(c, MQ_EHIP, "the device builder of the per-frame tree flagged an overflow (flags " + std::to_string(c->db_ctr_host[k][MQ_DB_ERR]) + ")");
    }
    const size_t on = ns + (size_t)p * c->dyn_cap_nodes, ot = ts + (size_t)p * c->dyn_cap_tris, ol = ls + (size_t)p * c->dyn_cap_tris;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // scratch of the builder: grown in steps (a growth waits for the device)
    if (c->db_cap < td) {
        const uint32_t cap = (uint32_t)std::max<size_t>(td + td / 2, 16384);
        const size_t bytes = al((size_t)cap * sizeof(MqTri)) + 4 * al((size_t)cap * 4) + al((size_t)cap * 2 * 4) + al((size_t)cap * 8) + al((size_t)cap * 2 * 24) + al((size_t)cap * 4)
                           + 3 * al((size_t)cap * 8) + al(MQ_DB_WORDS * 4) + al(mq_device_bvh_sort_bytes(cap));
        if ((r = dev_alloc(c, c->d_db_scratch, bytes))) return r;
        c->db_cap = cap;
    }
    for (int k = 0; k < mq_ctx::MQ_DYN_REGIONS; k++) if (!c->db_ctr_host[k]) HIPCHK(c, hipHostMalloc((void**)&c->db_ctr_host[k], MQ_DB_WORDS * 4, hipHostMallocDefault));
    size_t need = al(td * sizeof(MqTri));
    for (int sl = 0; sl < MQ_MAX_GEOMETRIES; sl++) {
        const MqHostGeo& g = c->geo[sl];
        if ((g.flags & MQ_GEO_STATIC) || !g.n_tri()) continue;
        need += al(g.ext.size() * sizeof(mq_ext)) + al(g.idx.size() * 4) + al(g.prev_vtx.size() * 4);
    }
    if (c->stage_bytes[p] < need) {
        if (c->stage[p]) HIPCHK(c, hipHostFree(c->stage[p]));
        c->stage[p] = nullptr; c->stage_bytes[p] = 0;
        HIPCHK(c, hipHostMalloc(&c->stage[p], need + need / 2 + 65536, hipHostMallocDefault));
        c->stage_bytes[p] = need + need / 2 + 65536;
    }
    char* st = (char*)c->stage[p]; size_t at = 0;
    auto push = [&](void* dev, const void* src, size_t bytes) -> int {
        if (!bytes) return MQ_OK;
        par_copy(st + at, src, bytes);
        HIPCHK(c, hipMemcpyAsync(dev, st + at, bytes, hipMemcpyHostToDevice, c->up_stream));
        at += al(bytes);
        return MQ_OK;
    };
    // carve the scratch
    MqDevBvh A; memset(&A, 0, sizeof A);
    { char* q = (char*)c->d_db_scratch.p; const size_t cap = c->db_cap;
      auto take = [&](size_t bytes) { char* x = q; q += al(bytes); return x; };
      A.in = (const MqTri*)take(cap * sizeof(MqTri));
      A.keys0 = (uint32_t*)take(cap * 4); A.keys1 = (uint32_t*)take(cap * 4); A.vals0 = (uint32_t*)take(cap * 4); A.vals1 = (uint32_t*)take(cap * 4);
      A.parent = (int*)take(cap * 2 * 4); A.child = (int2*)take(cap * 8); A.box = (float*)take(cap * 2 * 24); A.flag = (uint32_t*)take(cap * 4);
      A.queue0 = (uint2*)take(cap * 8); A.queue1 = (uint2*)take(cap * 8); A.leaf_at = (uint2*)take(cap * 8); A.ctr = (uint32_t*)take(MQ_DB_WORDS * 4); }
    void* sort_tmp = (char*)A.ctr + al(MQ_DB_WORDS * 4);
    const size_t sort_bytes = mq_device_bvh_sort_bytes(c->db_cap);
    if ((r = push((void*)A.in, flat.data(), td * sizeof(MqTri)))) return r;
    for (int sl = 0; sl < MQ_MAX_GEOMETRIES; sl++) { // per-slot arrays of region p (the builder reads the extra data for the shading records)
        MqHostGeo& g = c->geo[sl];
        if (g.flags & MQ_GEO_STATIC) continue;
        c->scene.geo[sl].ext = nullptr; c->scene.geo[sl].idx = nullptr; c->scene.geo[sl].prev_vtx = nullptr;
        if (!g.n_tri()) continue;
        DevBuf& be = p ? c->d_ext_x[p - 1][sl] : c->d_ext[sl]; DevBuf& bi = p ? c->d_idx_x[p - 1][sl] : c->d_idx[sl]; DevBuf& bp = p ? c->d_prev_x[p - 1][sl] : c->d_prev[sl];
        auto room = [&](DevBuf& b, size_t bytes) -> int { return (!b.p || b.bytes < bytes) ? dev_alloc(c, b, bytes + bytes / 2 + 4096) : MQ_OK; };
        if ((r = room(be, g.ext.size() * sizeof(mq_ext)))) return r;
        if ((r = push(be.p, g.ext.data(), g.ext.size() * sizeof(mq_ext)))) return r;
        c->scene.geo[sl].ext = (const mq_ext*)be.p;
        if (g.dynamic) {
            if ((r = room(bi, g.idx.size() * 4)) || (r = room(bp, g.prev_vtx.size() * 4))) return r;
            if ((r = push(bi.p, g.idx.data(), g.idx.size() * 4)) || (r = push(bp.p, g.prev_vtx.data(), g.prev_vtx.size() * 4))) return r;
            c->scene.geo[sl].idx = (const uint32_t*)bi.p; c->scene.geo[sl].prev_vtx = (const float*)bp.p;
        }
    }
    A.n = (uint32_t)td;
    A.nodes = (MqNode*)c->d_nodes.p; A.leaves = (MqLeafRec*)c->d_leaves.p; A.tris = (MqTri*)c->d_tris.p; A.shade = (MqShadeRec*)c->d_shade.p;
    A.node_base = (uint32_t)on; A.leaf_base = (uint32_t)ol; A.tri_base = (uint32_t)ot; A.node_cap = c->dyn_cap_nodes;
    A.sc = c->scene; // (with the per-slot pointers of region p set above)
    if (td) {
        int e = mq_launch_device_bvh(A, sort_tmp, sort_bytes, c->up_stream);
        if (e) return fail(c, MQ_EHIP, std::string("device BVH build: ") + hipGetErrorString((hipError_t)e));
        HIPCHK(c, hipMemcpyAsync(c->db_ctr_host[p], A.ctr, MQ_DB_WORDS * 4, hipMemcpyDeviceToHost, c->up_stream));
        c->db_ctr_pending[p] = true;
    }
    HIPCHK(c, hipEventRecord(c->ev_uploaded, c->up_stream));
    if (times) { auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "commit (per-frame, region %d, tree on the device, %zu triangles): wait for the region's last reader and the upload stream %.3f ms, stage + enqueue copies and the build %.3f\n", p, td, ms(t_0, t_1), ms(t_1, std::chrono::steady_clock::now())); }
    c->uploaded_valid = true;
    c->scene.n_nodes = (uint32_t)(on + (td ? 1 : 0)); c->scene.n_tris = (uint32_t)(ot + td);
    c->scene.dyn_root = td ? (uint32_t)on : MQ_NIL;
    c->dyn_parity = p;
    c->n_static_nodes = (uint32_t)ns; c->n_static_tris = (uint32_t)ts; c->n_static_leaves = (uint32_t)ls;
    c->joined = td != 0; c->sah_cost = c->s_sah; c->d_depth = td ? MQ_DB_LEVELS : 0; c->committed = true;
    c->db_tris[p] = (uint32_t)td;
    c->commits_dynamic++; c->commits_async++; c->commits_device++;
    c->mirror_pending = true; c->mirror_from_device = true;
    return MQ_OK;
}
} // namespace

int mq_scene_commit(mq_ctx* c) {
    if (!c) return MQ_EINVAL;
    const auto t_commit = std::chrono::steady_clock::now();
    std::string err;
    std::vector<MqTri>& flat = c->flat_scratch; // (kept from commit to commit: a per-frame commit does not zero-fill it again)
    const bool static_rebuilt = c->static_dirty;
    if (c->static_dirty) {
        flatten_slots(c, true, flat);
        if (!mq_build_cwbvh(flat, c->s_nodes, c->s_tris, c->s_leaves, &c->s_sah, err, &c->s_depth)) return fail(c, MQ_EINVAL, "bvh build: " + err);
        c->static_dirty = false;
        std::vector<MqTri>().swap(flat); // (the static triangles: not worth keeping)
    }
    std::vector<MqNode> d_nodes; std::vector<MqTri> d_tris; std::vector<MqLeafRec> d_leaves; float d_sah = 0.0f;
    flatten_slots(c, false, flat);
    { // the per-frame tree built on the device (property "per-frame BVH"): only for commits that can take the asynchronous path (see below)
        static const bool force_sync0 = getenv("MQ_DEBUG_COMMIT_SYNC") != nullptr;
        static const size_t auto_tris = getenv("MQ_DEVBVH_AUTO_TRIS") ? (size_t)atol(getenv("MQ_DEVBVH_AUTO_TRIS")) : 12288; // (where the device-built tree starts to win: 8 k triangles 1.86 ms host / 2.10 ms device per frame, 16 k 2.87 / 2.28, 65 k 6.7 / 2.9)
        const size_t ns0 = c->s_nodes.size(), ts0 = c->s_tris.size(), ls0 = c->s_leaves.size(), td0 = flat.size();
        const bool partial0 = !static_rebuilt && c->nodes.size() >= ns0 && c->tris.size() >= ts0 && c->leaves.size() >= ls0 && !c->tex_dirty && c->dev_scene_valid
            && c->dev_static_nodes == ns0 && c->dev_static_tris == ts0 && c->dev_static_leaves == ls0;
        const bool wanted = c->props.dyn_bvh == 1 || (c->props.dyn_bvh == 2 && td0 >= auto_tris);
        if (c->device >= 0 && wanted && partial0 && ns0 != 0 && !force_sync0 && c->dyn_cap_tris != 0 && td0 <= c->dyn_cap_tris && td0 <= c->dyn_cap_nodes) {
            HIPCHK(c, hipSetDevice(c->device));
            return commit_per_frame_on_device(c, flat);
        }
    }
    if (!mq_build_cwbvh(flat, d_nodes, d_tris, d_leaves, &d_sah, err, &c->d_depth)) return fail(c, MQ_EINVAL, "bvh build: " + err);
    const size_t ns = c->s_nodes.size(), nd = d_nodes.size(), ts = c->s_tris.size(), td = d_tris.size(), ls = c->s_leaves.size(), ld = d_leaves.size();
    // layout: the static tree as built, then the per-frame tree (indices offset); the traversal starts at node 0 and
    // visits the per-frame root (MqSceneDev::dyn_root) last.  Without static geometry the per-frame tree is the tree.
    const bool joined = ns != 0 && nd != 0;
    const bool in_place = !static_rebuilt && c->nodes.size() >= ns && c->tris.size() >= ts && c->leaves.size() >= ls; // the static part is where it was
    if (!in_place) { c->nodes = c->s_nodes; c->tris = c->s_tris; c->leaves = c->s_leaves; }
    c->mirror_pending = false; c->mirror_from_device = false;
    auto mirror = [&]() { // the host copy with the contiguous numbering (mq_scene_get_bvh): static part, then the per-frame part
        c->nodes.resize(ns + nd); c->tris.resize(ts + td); c->leaves.resize(ls + ld);
        for (size_t j = 0; j < nd; j++) { MqNode n = d_nodes[j]; n.child_base += (uint32_t)ns; n.tri_base += (uint32_t)ls; c->nodes[ns + j] = n; }
        std::copy(d_tris.begin(), d_tris.end(), c->tris.begin() + (ptrdiff_t)ts);
        for (size_t j = 0; j < ld; j++) { MqLeafRec r = d_leaves[j]; r.tri0 += (uint32_t)ts; c->leaves[ls + j] = r; }
    };
    c->n_dyn_nodes = (uint32_t)nd; c->n_dyn_tris = (uint32_t)td; c->n_dyn_leaves = (uint32_t)ld;
    c->n_static_nodes = (uint32_t)ns; c->n_static_tris = (uint32_t)ts; c->n_static_leaves = (uint32_t)ls;
    c->joined = joined;
    c->sah_cost = c->s_sah + d_sah;
    c->committed = true;
    if (c->device < 0) { mirror(); c->tex_dirty = false; return MQ_OK; } // host-only context: BVH available for inspection, nothing to upload
    HIPCHK(c, hipSetDevice(c->device));
    int r;
    const bool partial = in_place && !c->tex_dirty && c->dev_scene_valid && c->dev_static_nodes == ns && c->dev_static_tris == ts && c->dev_static_leaves == ls;
    // Per-frame geometry only, the usual case of a running game (quake_node.cpp:896-983 rebuilds it every frame): the static part
    // stays where it is, the per-frame tree, its triangles, leaf records, shading records and per-slot arrays go into the region of
    // the device arrays that the frames in flight do NOT read, asynchronously.  Nothing here waits for the device except for the
    // last launch that read that region, two commits ago.  (Without static geometry the per-frame tree starts at node 0 and there
    // is only one place for it: the synchronous path below.)
    static const bool force_sync = getenv("MQ_DEBUG_COMMIT_SYNC") != nullptr; // the A/B switch: wait for the device, write in place (round 2's commit)
    const bool fits = c->dyn_cap_tris != 0 && nd <= c->dyn_cap_nodes && td <= c->dyn_cap_tris && ld <= c->dyn_cap_tris;
    if (partial && ns != 0 && fits && !force_sync) {
        const int p = (c->dyn_parity + 1) % mq_ctx::MQ_DYN_REGIONS;
        if ((r = ensure_upload_stream(c))) return r;
        static const bool times = getenv("MQ_DEBUG_COMMIT_TIMES") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto t_a = now();
        const bool had_reader = c->scene_used_valid[p] || c->scene_used_mixed[p];
        if (c->scene_used_mixed[p]) { HIPCHK(c, hipDeviceSynchronize()); }
        else if (c->scene_used_valid[p]) HIPCHK(c, hipEventSynchronize(c->ev_scene_used[p]));
        c->scene_used_valid[p] = false; c->scene_used_mixed[p] = false;
        const auto t_b = now();
        if (!had_reader) HIPCHK(c, hipStreamSynchronize(c->up_stream)); // (nobody rendered from this region since it was written: its upload, the last reader of its staging memory, may still be on its way)
        const auto t_c = now();
        const size_t on = ns + (size_t)p * c->dyn_cap_nodes, ot = ts + (size_t)p * c->dyn_cap_tris, ol = ls + (size_t)p * c->dyn_cap_tris; // where region p starts
        // staging layout: nodes | tris | leaves | shading records | per-slot arrays, each 256-byte aligned
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        size_t need = al(nd * sizeof(MqNode)) + al(td * sizeof(MqTri)) + al(ld * sizeof(MqLeafRec)) + al(td * sizeof(MqShadeRec));
        for (int sl = 0; sl < MQ_MAX_GEOMETRIES; sl++) {
            const MqHostGeo& g = c->geo[sl];
            if ((g.flags & MQ_GEO_STATIC) || !g.n_tri()) continue;
            need += al(g.ext.size() * sizeof(mq_ext)) + al(g.idx.size() * 4) + al(g.prev_vtx.size() * 4);
        }
        if (c->stage_bytes[p] < need) {
            if (c->stage[p]) HIPCHK(c, hipHostFree(c->stage[p]));
            c->stage[p] = nullptr; c->stage_bytes[p] = 0;
            HIPCHK(c, hipHostMalloc(&c->stage[p], need + need / 2 + 65536, hipHostMallocDefault));
            c->stage_bytes[p] = need + need / 2 + 65536;
        }
        char* st = (char*)c->stage[p]; size_t at = 0;
        auto push = [&](void* dev, const void* src, size_t bytes) -> int { // stage, then copy on the upload stream
            if (!bytes) return MQ_OK;
            if (src) par_copy(st + at, src, bytes);
            HIPCHK(c, hipMemcpyAsync(dev, st + at, bytes, hipMemcpyHostToDevice, c->up_stream));
            at += al(bytes);
            return MQ_OK;
        };
        { // the per-frame tree with the indices of region p (the host mirror c->nodes / c->leaves keeps the contiguous numbering)
            MqNode* sn = (MqNode*)(st + at);
            for (size_t j = 0; j < nd; j++) { MqNode n = d_nodes[j]; n.child_base += (uint32_t)on; n.tri_base += (uint32_t)ol; sn[j] = n; }
            if ((r = push((MqNode*)c->d_nodes.p + on, nullptr, nd * sizeof(MqNode)))) return r;
            if ((r = push((MqTri*)c->d_tris.p + ot, d_tris.data(), td * sizeof(MqTri)))) return r;
            MqLeafRec* sl = (MqLeafRec*)(st + at);
            for (size_t j = 0; j < ld; j++) { MqLeafRec q = d_leaves[j]; q.tri0 += (uint32_t)ot; sl[j] = q; }
            if ((r = push((MqLeafRec*)c->d_leaves.p + ol, nullptr, ld * sizeof(MqLeafRec)))) return r;
            shade_records_into(c, d_tris.data(), td, (MqShadeRec*)(st + at)); // (straight into the staging memory)
            if ((r = push((MqShadeRec*)c->d_shade.p + ot, nullptr, td * sizeof(MqShadeRec)))) return r;
        }
        for (int sl = 0; sl < MQ_MAX_GEOMETRIES; sl++) { // per-slot arrays of region p
            MqHostGeo& g = c->geo[sl];
            if (g.flags & MQ_GEO_STATIC) continue;
            c->scene.geo[sl].ext = nullptr; c->scene.geo[sl].idx = nullptr; c->scene.geo[sl].prev_vtx = nullptr;
            if (!g.n_tri()) continue;
            DevBuf& be = p ? c->d_ext_x[p - 1][sl] : c->d_ext[sl]; DevBuf& bi = p ? c->d_idx_x[p - 1][sl] : c->d_idx[sl]; DevBuf& bp = p ? c->d_prev_x[p - 1][sl] : c->d_prev[sl];
            auto room = [&](DevBuf& b, size_t bytes) -> int { return (!b.p || b.bytes < bytes) ? dev_alloc(c, b, bytes + bytes / 2 + 4096) : MQ_OK; };
            if ((r = room(be, g.ext.size() * sizeof(mq_ext)))) return r;
            if ((r = push(be.p, g.ext.data(), g.ext.size() * sizeof(mq_ext)))) return r;
            c->scene.geo[sl].ext = (const mq_ext*)be.p;
            if (g.dynamic) {
                if ((r = room(bi, g.idx.size() * 4)) || (r = room(bp, g.prev_vtx.size() * 4))) return r;
                if ((r = push(bi.p, g.idx.data(), g.idx.size() * 4)) || (r = push(bp.p, g.prev_vtx.data(), g.prev_vtx.size() * 4))) return r;
                c->scene.geo[sl].idx = (const uint32_t*)bi.p; c->scene.geo[sl].prev_vtx = (const float*)bp.p;
            }
        }
        HIPCHK(c, hipEventRecord(c->ev_uploaded, c->up_stream));
        if (times) fprintf(stderr, "commit (per-frame, region %d): build %.3f ms, wait for the region's last reader %.3f, for the upload stream %.3f, stage + enqueue copies %.3f\n", p, ms(t_commit, t_a), ms(t_a, t_b), ms(t_b, t_c), ms(t_c, now()));
        c->uploaded_valid = true;
        c->scene.n_nodes = (uint32_t)(on + nd); c->scene.n_tris = (uint32_t)(ot + td);
        c->scene.dyn_root = nd ? (uint32_t)on : MQ_NIL;
        c->dyn_parity = p;
        c->commits_dynamic++; c->commits_async++;
        // the host mirror is brought up to date when somebody asks for it (materialize_mirror): the trees are kept as built
        c->pend_nodes.swap(d_nodes); c->pend_tris.swap(d_tris); c->pend_leaves.swap(d_leaves); c->mirror_pending = true;
        return MQ_OK;
    }
    mirror();
    HIPCHK(c, hipDeviceSynchronize());
    if (c->up_stream) for (int k = 0; k < mq_ctx::MQ_DYN_REGIONS; k++) { c->scene_used_valid[k] = false; c->scene_used_mixed[k] = false; c->db_ctr_pending[k] = false; }
    if (partial && (ns == 0 || (force_sync && fits)) && c->dyn_parity == 0 && c->d_nodes.bytes >= c->nodes.size() * sizeof(MqNode) && c->d_tris.bytes >= c->tris.size() * sizeof(MqTri) && c->d_shade.bytes >= c->tris.size() * sizeof(MqShadeRec)
        && c->d_leaves.bytes >= c->leaves.size() * sizeof(MqLeafRec)) { // per-frame geometry without a static tree: in place, behind a synchronisation
        std::vector<MqShadeRec> recs;
        shade_records(c, c->tris.data() + ts, td, recs);
        if (nd) HIPCHK(c, hipMemcpy((MqNode*)c->d_nodes.p + ns, c->nodes.data() + ns, nd * sizeof(MqNode), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy((MqTri*)c->d_tris.p + ts, c->tris.data() + ts, td * sizeof(MqTri), hipMemcpyHostToDevice));
        if (ld) HIPCHK(c, hipMemcpy((MqLeafRec*)c->d_leaves.p + ls, c->leaves.data() + ls, ld * sizeof(MqLeafRec), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy((MqShadeRec*)c->d_shade.p + ts, recs.data(), td * sizeof(MqShadeRec), hipMemcpyHostToDevice));
        if ((r = upload_slot_arrays(c, false))) return r;
        c->scene.n_nodes = (uint32_t)c->nodes.size(); c->scene.n_tris = (uint32_t)c->tris.size();
        c->scene.dyn_root = joined ? (uint32_t)ns : MQ_NIL;
        c->commits_dynamic++;
        return MQ_OK;
    }
    c->dev_scene_valid = false;
    free_scene_dev(c);
    memset(&c->scene, 0, sizeof c->scene); c->scene.dyn_root = MQ_NIL;
    // two regions for the per-frame part (see above), each with room to grow without another full upload; the first holds the
    // part committed now, right behind the static part -- the layout of the host mirror
    const size_t cap_tris = td + 16384, cap_nodes = std::max(nd + 8192, c->props.dyn_bvh ? cap_tris : (size_t)0); // (a device-built tree may have as many nodes as triangles)
    if ((r = dev_alloc(c, c->d_nodes, (ns + mq_ctx::MQ_DYN_REGIONS * cap_nodes) * sizeof(MqNode)))) return r;
    if ((r = dev_alloc(c, c->d_tris, (ts + mq_ctx::MQ_DYN_REGIONS * cap_tris) * sizeof(MqTri)))) return r;
    if ((r = dev_alloc(c, c->d_leaves, (ls + mq_ctx::MQ_DYN_REGIONS * cap_tris) * sizeof(MqLeafRec)))) return r; // (at most one record per triangle: a region holds cap_tris records)
    if (!c->leaves.empty()) HIPCHK(c, hipMemcpy(c->d_leaves.p, c->leaves.data(), c->leaves.size() * sizeof(MqLeafRec), hipMemcpyHostToDevice));
    c->dyn_cap_nodes = (uint32_t)cap_nodes; c->dyn_cap_tris = (uint32_t)cap_tris; c->dyn_parity = 0;
    if ((r = dev_alloc(c, c->d_shade, (ts + mq_ctx::MQ_DYN_REGIONS * cap_tris) * sizeof(MqShadeRec)))) return r;
    if (!c->nodes.empty()) HIPCHK(c, hipMemcpy(c->d_nodes.p, c->nodes.data(), c->nodes.size() * sizeof(MqNode), hipMemcpyHostToDevice));
    if (!c->tris.empty()) HIPCHK(c, hipMemcpy(c->d_tris.p, c->tris.data(), c->tris.size() * sizeof(MqTri), hipMemcpyHostToDevice));
    if ((r = upload_slot_arrays(c, true))) return r;
    if ((r = upload_slot_arrays(c, false))) return r;
    // Texel pool: linear RGBA32F.  Level 0 decoded once (sRGB through the 256-entry table, quake_node.hpp:93-95,
    // everything else x / 255: the values a per-fetch decode gives); MQ_TEX_MIPMAP textures are followed by their
    // mip chain, level k+1 = 2x2 box filter of level k on the float texels, ((a + b) + (c + d)) * 0.25 with clamped
    // source indices, sizes max(1, size >> 1) down to 1x1.
    std::vector<MqTexDesc> desc(MQ_MAX_GLTEXTURES);
    std::vector<float> lin;
    {
        float lut[256];
        for (int i = 0; i < 256; i++) { double v = i / 255.0; lut[i] = (float)(v <= 0.04045 ? v / 12.92 : std::pow((v + 0.055) / 1.055, 2.4)); }
        for (uint32_t t = 0; t < MQ_MAX_GLTEXTURES; t++) {
            const MqHostTex& tx = c->tex[t];
            if (tx.px.empty()) { desc[t].offset = MQ_NIL; desc[t].w = desc[t].h = 0; desc[t].flags = 0; continue; }
            const bool srgb = (tx.flags & MQ_TEX_SRGB) != 0;
            const size_t n = (size_t)tx.w * tx.h;
            size_t at = lin.size();
            desc[t].offset = (uint32_t)(at / 4); desc[t].w = (uint16_t)tx.w; desc[t].h = (uint16_t)tx.h;
            lin.resize(at + 4 * n);
            for (size_t i = 0; i < n; i++) {
                const uint8_t* p = &tx.px[4 * i];
                float* o = &lin[at + 4 * i];
                for (int k = 0; k < 3; k++) o[k] = srgb ? lut[p[k]] : (float)p[k] * (1.0f / 255.0f);
                o[3] = (float)p[3] * (1.0f / 255.0f);
            }
            uint32_t levels = 1;
            if (tx.flags & MQ_TEX_MIPMAP) {
                uint32_t pw = tx.w, ph = tx.h; size_t prev = at;
                while ((pw > 1 || ph > 1) && levels < 16) {
                    const uint32_t w = std::max(1u, pw >> 1), h = std::max(1u, ph >> 1);
                    const size_t dst = lin.size();
                    lin.resize(dst + (size_t)w * h * 4);
                    for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) {
                        const uint32_t x0 = std::min(2 * x, pw - 1), x1 = std::min(2 * x + 1, pw - 1), y0 = std::min(2 * y, ph - 1), y1 = std::min(2 * y + 1, ph - 1);
                        for (int ch = 0; ch < 4; ch++) {
                            const float a = lin[prev + 4 * ((size_t)y0 * pw + x0) + ch], b = lin[prev + 4 * ((size_t)y0 * pw + x1) + ch];
                            const float d = lin[prev + 4 * ((size_t)y1 * pw + x0) + ch], e = lin[prev + 4 * ((size_t)y1 * pw + x1) + ch];
                            lin[dst + 4 * ((size_t)y * w + x) + ch] = ((a + b) + (d + e)) * 0.25f;
                        }
                    }
                    prev = dst; pw = w; ph = h; levels++;
                }
            }
            desc[t].flags = (tx.flags & 0xffu) | (levels << 8);
        }
    }
    if ((r = dev_upload(c, c->d_texdesc, desc.data(), desc.size() * sizeof(MqTexDesc)))) return r;
    c->texdesc = desc;
    { // shading records in BVH triangle order
        std::vector<MqShadeRec> recs;
        shade_records(c, c->tris.data(), c->tris.size(), recs);
        if (!recs.empty()) HIPCHK(c, hipMemcpy(c->d_shade.p, recs.data(), recs.size() * sizeof(MqShadeRec), hipMemcpyHostToDevice));
    }
    if ((r = dev_upload(c, c->d_texels, lin.data(), lin.size() * 4))) return r;
    c->scene.nodes = (const MqNode*)c->d_nodes.p; c->scene.tris = (const MqTri*)c->d_tris.p; c->scene.leaves = (const MqLeafRec*)c->d_leaves.p; c->scene.shade = (const MqShadeRec*)c->d_shade.p;
    c->scene.tex = (const MqTexDesc*)c->d_texdesc.p; c->scene.texels = (const float4*)c->d_texels.p;
    c->scene.n_nodes = (uint32_t)c->nodes.size(); c->scene.n_tris = (uint32_t)c->tris.size();
    c->scene.dyn_root = joined ? (uint32_t)ns : MQ_NIL;
    c->tex_dirty = false; c->dev_scene_valid = true;
    c->dev_static_nodes = (uint32_t)ns; c->dev_static_tris = (uint32_t)ts; c->dev_static_leaves = (uint32_t)ls;
    c->commits_full++;
    return MQ_OK;
}

int mq_scene_layout(const mq_ctx* c, uint64_t* static_nodes, uint64_t* static_tris) {
    if (!c) return MQ_EINVAL;
    if (static_nodes) *static_nodes = c->n_static_nodes; if (static_tris) *static_tris = c->n_static_tris;
    return MQ_OK;
}

int mq_scene_commit_counts(const mq_ctx* c, uint32_t* full, uint32_t* per_frame) {
    if (!c) return MQ_EINVAL;
    if (full) *full = c->commits_full; if (per_frame) *per_frame = c->commits_dynamic;
    return MQ_OK;
}

int mq_scene_commit_async_count(const mq_ctx* c, uint32_t* n) {
    if (!c || !n) return MQ_EINVAL;
    *n = c->commits_async;
    return MQ_OK;
}

int mq_scene_commit_device_count(const mq_ctx* c, uint32_t* n) {
    if (!c || !n) return MQ_EINVAL;
    *n = c->commits_device;
    return MQ_OK;
}

int mq_set_constants(mq_ctx* c, const mq_constants* k) {
    if (!c || !k) return MQ_EINVAL;
    c->constants = *k; c->params_dirty = true; // constant_data_update -> pipeline refresh, render_mcpg.cpp:125
    return MQ_OK;
}

int mq_get_constants(const mq_ctx* c, mq_constants* out) { if (!c || !out) return MQ_EINVAL; *out = c->constants; return MQ_OK; }

// ---- describe / connect / process ---------------------------------------------------------------
int mq_describe(const mq_ctx* c, uint32_t w, uint32_t h, mq_io_desc* out) {
    if (!c || !out || !w || !h) return MQ_EINVAL;
    fill_desc(c, w, h, out);
    return MQ_OK;
}

int mq_set_partition(mq_ctx* c, int rank, int world) {
    if (!c || world < 1 || rank < 0 || rank >= world) return fail(c, MQ_EINVAL, "bad partition");
    if (rank != c->rank || world != c->world) c->connected = false;
    c->rank = rank; c->world = world;
    return MQ_OK;
}
int mq_tiles_per_rank(const mq_ctx* c, uint32_t* tiles, size_t* bytes) {
    if (!c || !c->W) return MQ_ESTATE;
    if (tiles) *tiles = c->tiles_per_rank; if (bytes) *bytes = (size_t)c->tiles_per_rank * 64 * 16;
    return MQ_OK;
}

int mq_connect(mq_ctx* c, uint32_t w, uint32_t h) {
    if (!c || !w || !h) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context: no HIP device");
    if (c->props.mc_samples > MQ_MAX_MC_SAMPLES) return fail(c, MQ_EINVAL, "mc samples exceeds kernel limit");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    free_frame_state(c);
    c->W = w; c->H = h; c->tiles_x = (w + 7) / 8; c->tiles_y = (h + 7) / 8;
    uint32_t nt = c->tiles_x * c->tiles_y;
    c->tiles_per_rank = (nt + (uint32_t)c->world - 1) / (uint32_t)c->world;
    c->n_local_tiles = (nt > (uint32_t)c->rank) ? (nt - (uint32_t)c->rank + (uint32_t)c->world - 1) / (uint32_t)c->world : 0;
    mq_io_desc d; fill_desc(c, w, h, &d);
    int r;
    for (int i = 0; i < MQ_OUT_COUNT; i++) { if ((r = dev_alloc(c, c->d_out[i], d.bytes[i]))) return r; HIPCHK(c, hipMemset(c->d_out[i].p, 0, c->d_out[i].bytes)); }
    c->mc_total = c->props.mc_adaptive_buffer_size + c->props.mc_static_buffer_size;
    c->lc_total = c->props.lc_buffer_size;
    if ((r = dev_alloc(c, c->d_mc, (size_t)c->mc_total * sizeof(MqMCState)))) return r;
    if ((r = dev_alloc(c, c->d_lc, (size_t)c->lc_total * sizeof(MqLCCell)))) return r;
    if ((r = dev_alloc(c, c->d_upd_count, (size_t)c->mc_total * 4))) return r;
    if ((r = dev_alloc(c, c->d_upd_head, (size_t)c->mc_total * 4))) return r;
    c->queue_cap = (uint32_t)std::min<size_t>(queue_entries_needed(c), 0x7fffffffu);
    if ((r = dev_alloc(c, c->d_queue, (size_t)c->queue_cap * sizeof(MqUpdate)))) return r;
    if ((r = dev_alloc(c, c->d_active, (size_t)c->queue_cap * 4))) return r;
    if ((r = dev_alloc(c, c->d_active_ctrl, (size_t)MQ_CTRL_GROUP * 4))) return r;
    HIPCHK(c, hipMemset(c->d_active_ctrl.p, 0, (size_t)MQ_CTRL_GROUP * 4));
    c->subs = std::min(std::max(1, c->props.pipelines), (int)mq_ctx::MAX_SUBS);
    c->queues_dirty = true;
    if ((r = dev_alloc(c, c->d_ctrl, (size_t)c->subs * MQ_CTRL_WORDS * 4))) return r; // block 0: the rank's (flags, update tails) + sub 0's queues; block k: sub k's queues
    if ((r = dev_alloc(c, c->d_counters, sizeof(MqCountersDev)))) return r;
    HIPCHK(c, hipMemset(c->d_counters.p, 0, sizeof(MqCountersDev)));
    c->grid_blocks = std::max(1, c->cu_count) * 8;
    // one region per sub-pipeline, indexed by block and thread of the (resident) grids.  The camera-ray launches have areas of their own,
    // indexed by PIXEL SLOT (allocated below): they launch one wave per tile, and they run BESIDE the previous frame's kernels (and the
    // ReSTIR / volume passes) on the caller's stream
    if ((r = dev_alloc(c, c->d_spill, (size_t)c->subs * c->grid_blocks * mq_render_block_size() * mq_spill_entries() * 8))) return r;
    // per-slot buffers: the rank's share of interleaved tiles (MCPG node), or -- a rank of a partitioned frame also runs the ReSTIR
    // node and the g-buffer on a band of rows -- its widest band: rows owned + the largest spatial radius (100) + the reprojection halo
    c->slot_tiles = c->tiles_per_rank;
    if (c->world > 1) for (int rk = 0; rk < c->world; rk++) { const BandRows b = band_rows(c->props, h, rk, c->world, 100); c->slot_tiles = std::max(c->slot_tiles, (b.g1 - b.g0) * c->tiles_x); }
    const size_t slots = (size_t)c->slot_tiles * 64;
    if (c->world > 1 && (r = dev_alloc(c, c->d_band_hits, slots * 16))) return r;
    // stack spill of the camera-ray kernel: 52 entries of 8 bytes per pixel slot (0.86 GB at 1920x1080, 3.5 GB at 3840x2160 of the 288 GB; touched
    // only by rays whose stacks outgrow the 12 LDS entries); a second area for the row band's camera rays of a partitioned frame, which may
    // run while the next frame's (overlapped) camera rays do
    if ((r = dev_alloc(c, c->d_cam_spill[0], slots * (size_t)mq_spill_entries() * 8))) return r;
    if (c->world > 1 && (r = dev_alloc(c, c->d_cam_spill[1], slots * (size_t)mq_spill_entries() * 8))) return r;
    c->band_gb_valid = false;
    if ((r = dev_alloc(c, c->d_paths, slots * 112))) return r; // 7 fields of 16 bytes (the volume pass uses 6)
    if ((r = dev_alloc(c, c->d_debug_rng, (size_t)c->W * c->H * 4))) return r;
    for (int k = 0; k < 2; k++) if ((r = dev_alloc(c, c->d_prim_hits[k], slots * 16))) return r;
    // sub-pipeline k renders the local tiles [n * k / subs, n * (k + 1) / subs); its queues hold 2x its pixel slots + 1024
    // positions (sharded queues interleave 16 tails: room for shard imbalance).  The regions are consecutive parts of the
    // same allocations, so a launch over the whole rank (the volume pass) uses them as ONE queue of ray_cap positions.
    for (int k = 0; k <= c->subs; k++) c->sub_slot_begin[k] = (uint32_t)((uint64_t)c->n_local_tiles * k / c->subs) * 64u;
    c->sub_ray_cap = (uint32_t)(2 * ((slots / 64 + c->subs - 1) / c->subs + 1) * 64 + 1024);
    c->ray_cap = c->sub_ray_cap * (uint32_t)c->subs;
    if ((r = dev_alloc(c, c->d_rays, (size_t)c->ray_cap * 32 * 2))) return r; // two buffers, by round parity (ray_buffer in the kernels)
    if ((r = dev_alloc(c, c->d_ray_hits, (size_t)c->ray_cap * 16))) return r;
    if ((r = dev_alloc(c, c->d_qslots[0], (size_t)c->ray_cap * 4))) return r;
    if ((r = dev_alloc(c, c->d_qslots[1], (size_t)c->ray_cap * 4))) return r;
    if ((r = dev_alloc(c, c->d_prev_vdepth, (size_t)w * h * 2))) return r;
    HIPCHK(c, hipMemset(c->d_prev_vdepth.p, 0, c->d_prev_vdepth.bytes));
    if ((r = dev_alloc(c, c->d_fp_winner, (size_t)w * h * 4))) return r;
    HIPCHK(c, hipMemset(c->d_fp_winner.p, 0, c->d_fp_winner.bytes)); // (the resolve pass leaves it zero again)
    if ((r = dev_alloc(c, c->d_post_prev_gb, (size_t)w * h * 16))) return r;
    for (int k = 0; k < 2; k++) { if ((r = dev_alloc(c, c->d_post_prev_out[k], (size_t)w * h * 16))) return r; if ((r = dev_alloc(c, c->d_post_prev_hist[k], (size_t)w * h * 4))) return r; }
    c->post_first = true;
    if ((r = dev_alloc(c, c->d_restir_pong, (size_t)w * h * 64))) return r;
    if ((r = dev_alloc(c, c->d_restir_prev, (size_t)w * h * 64))) return r;
    if ((r = dev_alloc(c, c->d_restir_prev_gb, (size_t)w * h * 16))) return r;
    // the delay-1 buffers start as zeros: a rank of a row partition whose caller has not (yet) delivered the other ranks' halo rows then
    // reads empty reservoirs / a zero history there -- "no history", never uninitialised memory
    HIPCHK(c, hipMemset(c->d_restir_prev.p, 0, c->d_restir_prev.bytes)); HIPCHK(c, hipMemset(c->d_restir_prev_gb.p, 0, c->d_restir_prev_gb.bytes));
    HIPCHK(c, hipMemset(c->d_post_prev_gb.p, 0, c->d_post_prev_gb.bytes));
    for (int k = 0; k < 2; k++) { HIPCHK(c, hipMemset(c->d_post_prev_out[k].p, 0, c->d_post_prev_out[k].bytes)); HIPCHK(c, hipMemset(c->d_post_prev_hist[k].p, 0, c->d_post_prev_hist[k].bytes)); }
    c->restir_iteration = 0; c->restir_seeded = false;
    c->dist_mc_n = (uint32_t)(d.state_bytes_volume_distancemc / sizeof(MqDistMC));
    if ((r = dev_alloc(c, c->d_dist_mc, d.state_bytes_volume_distancemc))) return r;
    c->iteration = 0; c->connected = true; c->params_dirty = true; c->volume_outputs_zero = true; // outputs were cleared above
    return MQ_OK;
}

static int drain_slot(mq_ctx* c, int slot) {
    if (!c->ev_pending[slot]) return MQ_OK;
    const int R = c->ev_rounds[slot], last = 3 + 2 * R;
    HIPCHK(c, hipEventSynchronize(c->evr[slot][last]));
    float prim = 0, tr = 0, bo = 0, ap = 0, x = 0;
    if (!c->ev_detail[slot]) { // only frame start, end of the render passes, end of the update passes
        HIPCHK(c, hipEventElapsedTime(&x, c->evr[slot][0], c->evr[slot][2 + 2 * R]));
        HIPCHK(c, hipEventElapsedTime(&ap, c->evr[slot][2 + 2 * R], c->evr[slot][last]));
        c->t_render_sum += x; c->t_update_sum += ap; c->t_frames++;
        if (slot == c->ev_last) { c->last_render_ms = x; c->last_update_ms = ap; }
        c->ev_pending[slot] = false;
        return MQ_OK;
    }
    c->t_detail_frames++;
    float ptrace = 0;
    HIPCHK(c, hipEventElapsedTime(&ptrace, c->evr[slot][0], c->evr[slot][1])); // mq_primary_trace_kernel (~0 in a counting frame: traced inline)
    HIPCHK(c, hipEventElapsedTime(&prim, c->evr[slot][1], c->evr[slot][2]));
    if (c->ev_pt_timed[slot]) { // traced on its own stream beside the previous frame: the launch's own duration; ev[0] -> ev[1] above is what the frame waited for it
        float k = 0; HIPCHK(c, hipEventElapsedTime(&k, c->ev_pt_t[slot][0], c->ev_pt_t[slot][1]));
        c->t_round_trace[0] += k; c->t_pt_kernel_sum += k;
    } else c->t_round_trace[0] += ptrace;
    c->t_round_shade[0] += prim;
    prim += ptrace; // "primary" = both launches of the first hit; "trace" = the queue kernel of the bounce rounds only
    for (int k = 0; k < R; k++) {
        HIPCHK(c, hipEventElapsedTime(&x, c->evr[slot][2 + 2 * k], c->evr[slot][3 + 2 * k])); tr += x; c->t_round_trace[1 + k] += x;
        HIPCHK(c, hipEventElapsedTime(&x, c->evr[slot][3 + 2 * k], c->evr[slot][4 + 2 * k])); bo += x; c->t_round_shade[1 + k] += x;
    }
    HIPCHK(c, hipEventElapsedTime(&ap, c->evr[slot][2 + 2 * R], c->evr[slot][last]));
    c->t_primary_sum += prim; c->t_trace_sum += tr; c->t_bounce_sum += bo;
    c->t_render_sum += prim + tr + bo; c->t_update_sum += ap; c->t_frames++;
    if (slot == c->ev_last) { c->last_render_ms = prim + tr + bo; c->last_update_ms = ap; }
    c->ev_pending[slot] = false;
    return MQ_OK;
}

int mq_reset_state(mq_ctx* c) { if (!c) return MQ_EINVAL; c->iteration = 0; return MQ_OK; }

// sub < 0: a launch over the whole rank (all queue regions as one); sub >= 0: sub-pipeline `sub`
static void fill_frame(mq_ctx* c, const mq_uniform* u, MqFrame& F, int sub = -1) {
    memset(&F, 0, sizeof F);
    F.u = *u; F.W = c->W; F.H = c->H; F.tiles_x = c->tiles_x; F.tiles_y = c->tiles_y;
    F.n_local_tiles = c->n_local_tiles; F.rank = (uint32_t)c->rank; F.world = (uint32_t)c->world;
    F.irradiance = (float*)c->d_out[MQ_OUT_IRRADIANCE].p; F.tiles_out = (float*)c->d_out[MQ_OUT_TILES].p; F.volume_tiles_out = (float*)c->d_out[MQ_OUT_VOLUME_TILES].p; F.vdepth_tiles_out = (uint16_t*)c->d_out[MQ_OUT_VOLUME_DEPTH_TILES].p;
    F.debug = (uint16_t*)c->d_out[MQ_OUT_DEBUG].p; F.debug_rng = (uint32_t*)c->d_debug_rng.p;
    F.gb_albedo = (uint16_t*)c->d_out[MQ_OUT_GB_ALBEDO].p; F.gb_irr = (uint16_t*)c->d_out[MQ_OUT_GB_IRRADIANCE].p;
    F.gb_mv = (uint16_t*)c->d_out[MQ_OUT_GB_MV].p; F.gbuffer = (uint32_t*)c->d_out[MQ_OUT_GBUFFER].p; F.hits = (uint32_t*)c->d_out[MQ_OUT_HITS].p;
    F.mc = (MqMCState*)c->d_mc.p; F.lc = (MqLCCell*)c->d_lc.p; F.upd_count = (uint32_t*)c->d_upd_count.p; F.upd_head = (uint32_t*)c->d_upd_head.p;
    F.queue = (MqUpdate*)c->d_queue.p; F.active = (uint32_t*)c->d_active.p; F.active_ctrl = (uint32_t*)c->d_active_ctrl.p; F.queue_cap = c->queue_cap; F.ctrl = (uint32_t*)c->d_ctrl.p; F.ctrl_words = (uint32_t)(c->d_ctrl.bytes / 4); F.counters = (MqCountersDev*)c->d_counters.p; F.count_stats = c->count_enabled ? 1u : 0u;
    const size_t k = sub < 0 ? 0 : (size_t)sub, qoff = k * c->sub_ray_cap;
    F.slot_begin = sub < 0 ? 0u : c->sub_slot_begin[sub]; F.slot_end = sub < 0 ? c->n_local_tiles * 64u : c->sub_slot_begin[sub + 1];
    F.qctrl = F.ctrl + k * MQ_CTRL_WORDS;
    F.stack_spill = (unsigned long long*)c->d_spill.p + k * c->grid_blocks * mq_render_block_size() * mq_spill_entries();
    F.paths = (uint4*)c->d_paths.p; F.n_slots = c->slot_tiles * 64u;
    F.tile_mul = (uint32_t)c->world; F.tile_add = (uint32_t)c->rank;
    F.rays = (float4*)c->d_rays.p + 4 * qoff; F.ray_hits = (uint4*)c->d_ray_hits.p + qoff; // rays: two buffers (round parity) of origins + directions per region
    F.prim_hits = (uint4*)c->d_prim_hits[c->frame_parity & 1].p; F.cam_spill = (unsigned long long*)c->d_cam_spill[0].p;
    F.queue_slots[0] = (uint32_t*)c->d_qslots[0].p + qoff; F.queue_slots[1] = (uint32_t*)c->d_qslots[1].p + qoff;
    F.ray_cap = sub < 0 ? c->ray_cap : c->sub_ray_cap;
    { // test hook: tell the kernels of a smaller queue than the one allocated, so that the overflow path (rays dropped, frame flagged) runs
        static const int div = getenv("MQ_DEBUG_RAY_CAP_DIV") ? std::max(1, atoi(getenv("MQ_DEBUG_RAY_CAP_DIV"))) : 1;
        if (div > 1) F.ray_cap = std::max(1024u, F.ray_cap / (uint32_t)div);
    }
    F.volume = (float*)c->d_out[MQ_OUT_VOLUME].p; F.volume_depth = (uint16_t*)c->d_out[MQ_OUT_VOLUME_DEPTH].p; F.volume_mv = (uint16_t*)c->d_out[MQ_OUT_VOLUME_MV].p;
    F.prev_volume_depth = (uint16_t*)c->d_prev_vdepth.p; F.fp_winner = (uint32_t*)c->d_fp_winner.p; F.dist_mc = (float4*)c->d_dist_mc.p; F.dist_mc_n = c->dist_mc_n;
    F.lc_stats = (uint2*)c->d_lc_stats.p; F.last_upd_count = (uint32_t*)c->d_last_upd.p;
    F.learn_log = (uint4*)c->d_learn_log.p; F.learn_log_count = (uint32_t*)c->d_learn_count.p; F.learn_log_cap = c->learn_log_cap;
    const int K = c->params.reference_mode ? 0 : std::max(0, c->params.mc_samples);
    const int KD = c->params.reference_mode || c->params.volume_spp <= 0 ? 0 : std::max(0, c->params.distance_mc_samples);
    F.lds_rows2 = (uint32_t)std::max(mq_stack_lds_entries(), std::max((6 * K + 1) / 2, (3 * KD + 1) / 2));
    F.shade_block = (uint32_t)mq_render_block_size();
    while (F.shade_block > 64 && (size_t)F.lds_rows2 * 64 * 8 * (F.shade_block / 64) > 64 * 1024) F.shade_block /= 2; // a block's LDS: at most 64 KB
}

#ifndef MQ_GRID_MODE
#define MQ_GRID_MODE 1
#endif
// Grid sizes of the three frame kernels.  They are grid-stride / persistent kernels, so the right
// grid is exactly the number of blocks the chip holds at once (CUs x resident blocks per CU at the
// kernel's register and LDS footprint): a larger grid runs as a ragged last wave of blocks.
static int frame_grids(mq_ctx* c, const MqFrame& F) {
    const int key = (int)F.lds_rows2 * 2 + (c->params.reference_mode ? 1 : 0);
    if (c->grid_key == key) return MQ_OK;
    int occ[4] = {0, 0, 0, 0};
    int e = mq_resident_blocks(!c->params.reference_mode, (size_t)F.lds_rows2 * 64 * 8 * (F.shade_block / 64), (int)F.shade_block, occ);
    if (e) return fail(c, MQ_EHIP, std::string("occupancy query: ") + hipGetErrorString((hipError_t)e));
    for (int i = 0; i < 4; i++) {
#if MQ_GRID_MODE == 0
        c->grid_frame[i] = c->grid_blocks;
#else
        c->grid_frame[i] = std::min(i == 1 || i == 3 ? c->grid_blocks : c->grid_blocks * (mq_render_block_size() / (int)F.shade_block), std::max(1, c->cu_count) * std::max(1, occ[i])); // (smaller shading blocks: more of them, the same number of threads at most)
#endif
        static const char* const names[4] = {"MQ_DEBUG_PRIMARY_BLOCKS_PER_CU", "MQ_DEBUG_TRACE_BLOCKS_PER_CU", "MQ_DEBUG_BOUNCE_BLOCKS_PER_CU", "MQ_DEBUG_CAMERA_BLOCKS_PER_CU"};
        if (const char* ev = getenv(names[i])) { int v = atoi(ev); if (v > 0) c->grid_frame[i] = std::min(c->grid_blocks, std::max(1, c->cu_count) * v); } // tuning experiments only
    }
    c->grid_key = key;
    return MQ_OK;
}

// "spp", "max path length" and "volume spp" do not need a reconnect (render_mcpg.cpp:567-575 lists what does), but
// the update queue is sized by them: grown here, between frames, keeping what the volume pass of the last frame queued.
static int ensure_queue(mq_ctx* c) {
    const uint32_t need = (uint32_t)std::min<size_t>(queue_entries_needed(c), 0x7fffffffu);
    if (need <= c->queue_cap) return MQ_OK;
    HIPCHK(c, hipDeviceSynchronize());
    DevBuf bigger;
    int r = dev_alloc(c, bigger, (size_t)need * sizeof(MqUpdate));
    if (r) return r;
    HIPCHK(c, hipMemcpy(bigger.p, c->d_queue.p, (size_t)c->queue_cap * sizeof(MqUpdate), hipMemcpyDeviceToDevice));
    DevBuf list; // (rebuilt by every link pass: nothing to keep)
    if ((r = dev_alloc(c, list, (size_t)need * 4))) { dev_free(bigger); return r; }
    dev_free(c->d_queue); dev_free(c->d_active);
    c->d_queue = bigger; c->d_active = list; c->queue_cap = need;
    return MQ_OK;
}
// the learning-write log: room for every write a frame can propose (per segment: one update or invalidation and one
// light-cache store; per volume sample: an update or invalidation and a distance store)
static int ensure_learn_log(mq_ctx* c) {
    const size_t local_px = (size_t)c->tiles_per_rank * 64;
    const size_t need = 2 * local_px * ((size_t)std::max(1, c->props.spp) * (size_t)std::max(1, c->props.max_path_length - 1) + (size_t)std::max(0, c->props.volume_spp)) + 1024;
    if (c->d_learn_log.p && c->learn_log_cap >= need) return MQ_OK;
    HIPCHK(c, hipDeviceSynchronize());
    int r = dev_alloc(c, c->d_learn_log, need * 64);
    if (!r && !c->d_learn_count.p) r = dev_alloc(c, c->d_learn_count, 16);
    if (r) return r;
    c->learn_log_cap = (uint32_t)std::min<size_t>(need, 0xffffffffu);
    return MQ_OK;
}

int mq_process(mq_ctx* c, const mq_uniform* u, int render, void* stream) {
    if (!c || !u) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context: no HIP device");
    if (!c->connected) return fail(c, MQ_ESTATE, "mq_process before mq_connect (or a property change needs a reconnect)");
    if (!c->committed) return fail(c, MQ_ESTATE, "mq_process before mq_scene_commit");
    hipStream_t s = (hipStream_t)stream;
    c->last_stream = s;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->params_dirty) props_to_params(c);
    { int r = ensure_queue(c); if (r) return r; }
    { int r = scene_ready(c, s); if (r) return r; }
    if (c->params.lc_lock_protocol && !c->d_lc_stats.p) { // statistics start at zero when they are switched on
        int r = dev_alloc(c, c->d_lc_stats, (size_t)c->lc_total * 8); if (!r) r = dev_alloc(c, c->d_last_upd, (size_t)c->mc_total * 4); if (r) return r;
        HIPCHK(c, hipMemsetAsync(c->d_lc_stats.p, 0, c->d_lc_stats.bytes, s)); HIPCHK(c, hipMemsetAsync(c->d_last_upd.p, 0, c->d_last_upd.bytes, s));
    }
    if (c->params.log_learning) { int r = ensure_learn_log(c); if (r) return r; HIPCHK(c, hipMemsetAsync(c->d_learn_count.p, 0, 16, s)); } // the log holds ONE frame
    MqFrame F; fill_frame(c, u, F);
    { int r = frame_grids(c, F); if (r) return r; }
    if (c->iteration == 0) { // render_mcpg.cpp:221-226
        HIPCHK(c, hipMemsetAsync(c->d_mc.p, 0, c->d_mc.bytes, s));
        HIPCHK(c, hipMemsetAsync(c->d_lc.p, 0, c->d_lc.bytes, s));
        HIPCHK(c, hipMemsetAsync(c->d_upd_count.p, 0, c->d_upd_count.bytes, s));
        HIPCHK(c, hipMemsetAsync(c->d_upd_head.p, 0, c->d_upd_head.bytes, s));
        HIPCHK(c, hipMemsetAsync(c->d_dist_mc.p, 0, c->d_dist_mc.bytes, s)); // volume_distancemc, render_mcpg.cpp:225
        HIPCHK(c, hipMemsetAsync(c->d_ctrl.p, 0, c->d_ctrl.bytes, s));
    }
    const bool first_iteration = c->iteration == 0;
    c->iteration++;
    c->band_gb_valid = false; // a new frame: the ReSTIR node / post chain of a partitioned frame render the g-buffer of their rows again
    if (!render) { // render_mcpg.cpp:243-250
        RoctxRange rr("clear");
        int e = mq_launch_clear(F, s);
        if (e) return fail(c, MQ_EHIP, std::string("clear launch: ") + hipGetErrorString((hipError_t)e));
        return MQ_OK;
    }
    // queue counters of every sub-pipeline; the update tail survives (volume-pass entries of the last frame).  After a guided
    // frame without volume passes the update pass has left them zero already (reset_queue_control).
    if (c->queues_dirty) HIPCHK(c, hipMemsetAsync((char*)c->d_ctrl.p + 4 * MQ_CTRL_QUEUE0, 0, ((size_t)c->subs * MQ_CTRL_WORDS - MQ_CTRL_QUEUE0) * 4, s));
    c->queues_dirty = true;
    if (c->count_enabled) HIPCHK(c, hipMemsetAsync(c->d_counters.p, 0, offsetof(MqCountersDev, prof), s));
    const bool guided = !c->params.reference_mode;
    // rounds: every sample needs at most (max_path_length - 1) traced segments, render_mcpg.cpp:142-143
    const int rounds = std::max(0, c->params.spp) * std::max(0, c->params.max_path_length - 1);
    const bool volume = c->params.volume_spp > 0 && u->cam_x[3] > 0.0f; // needs a medium: mu_t > 0
    const int timed = std::min(rounds, 8); // rounds beyond the 8th are not split out (their time lands in the update interval)
    const int slot = c->ev_slot;
    { int r = drain_slot(c, slot); if (r) return r; } // the slot's previous frame finished long ago
    hipEvent_t* ev = c->evr[slot];
    const bool detail = c->ev_counter++ % c->timing_interval == 0; // per-launch events on every k-th frame only: each one costs a few microseconds between two dependent launches
    c->ev_detail[slot] = detail;
    HIPCHK(c, hipEventRecord(ev[0], s));
    int e = 0;
    RoctxRange rr_surface("surface"); // render_mcpg.cpp:255
    // ---- surface pass: `subs` independent chains of launches (each over its own pixel slots, with its own queues) on
    // `subs` streams.  A launch of this pipeline ends with the tail of its longest ray or path on a nearly idle chip;
    // with several chains in flight the tail of one overlaps the body of another.  Chain 0 runs on the caller's stream
    // (and carries the per-launch timing events), the others fork from it here and join it before the update pass.
    const int S = c->subs;
    MqFrame FS[mq_ctx::MAX_SUBS];
    for (int k = 0; k < S; k++) fill_frame(c, u, FS[k], k);
    auto st = [&](int k) { return k == 0 ? s : c->side[k - 1]; };
    auto sub_grid = [&](int i) { return std::max(std::max(1, c->cu_count), c->grid_frame[i] / S); }; // the chains share the chip
    if (S > 1) {
        HIPCHK(c, hipEventRecord(c->ev_fork, s));
        for (int k = 1; k < S; k++) HIPCHK(c, hipStreamWaitEvent(st(k), c->ev_fork, 0));
    }
    // The camera rays of this frame need not wait for the previous frame: the host runs ahead of the device, so this
    // launch can execute beside the previous frame's kernels.  For correctness it waits only for the first-hit kernel that
    // last read the hit buffer of this parity (two frames ago); WHERE in the previous frame it starts is a matter of speed:
    const int ov = c->props.overlap_camera_rays; // off / auto / always / update pass / last round
    const bool overlap_pt = ov != 0 && !c->count_enabled;
    // Where the camera rays of the NEXT frame may start among this frame's launches T0 B0 T1 B1 ... link apply (they run on a
    // low-priority stream and fill what those leave idle): "always" = with the frame; "update pass" = behind the last bounce
    // kernel, i.e. beside link / apply only; "last round" = behind the second-to-last bounce kernel, i.e. beside the last
    // round and the update pass -- the drain of the last trace launch, the terminal bounce kernel (bound by gathers) and the
    // update pass (latency bound) leave the vector ALUs the camera rays need; the first round is issue bound itself.
    // Measured per frame at 1 / 2 / 4 / 8 ranks: off 2.03 / - / - / 0.61, always 2.04 / 1.17 / 0.70 / 0.46, update pass 2.00 / - / - / -,
    // last round 1.93 / 1.10 / 0.66 / 0.46 ms.  "auto": last round for a rank of a partitioned frame; update pass for a full
    // frame, where it keeps every kernel of the surface pass alone on the chip (their times stay those of the kernels) for
    // 3 % of the frame time.
    // Round 3: "last bounce" = behind the last TRACE launch, i.e. beside the terminal bounce kernel and the update pass only.  That kernel
    // waits for scattered gathers (TD busy 0.86, vector ALU 0.57) and the camera rays are pure ALU work: on a full frame it loses 2 us
    // (0.150 -> 0.152 ms) while the frame gains 0.05 ms (1.93 -> 1.88 ms, profiles/r03_o_camera_ray_start.txt), and every trace launch
    // still has the chip to itself -- which "last round" (1.875 ms) gives up.  "auto" for a full frame since then.
    const int mode = ov == 1 ? (c->world > 1 ? 4 : 5) : ov;
    const bool behind_bounces = mode == 3 || mode == 4 || mode == 5;
    static const int pt_behind_env = getenv("MQ_DEBUG_PT_BEHIND") ? atoi(getenv("MQ_DEBUG_PT_BEHIND")) : -1; // tuning experiments only
    const int pt_behind = std::min(std::max(pt_behind_env >= 0 ? pt_behind_env : (mode == 3 ? 0 : (mode == 5 ? 1 : 2)), 0), std::max(0, 2 * rounds - 1));
    const uint32_t parity = c->frame_parity & 1u;
    c->ev_pt_timed[slot] = overlap_pt && detail;
    if (overlap_pt) {
        { int r = scene_ready(c, c->pt_stream); if (r) return r; }
        if (c->shaded_valid[parity]) HIPCHK(c, hipStreamWaitEvent(c->pt_stream, c->ev_shaded[parity], 0));
        if (behind_bounces && c->bounced_valid) HIPCHK(c, hipStreamWaitEvent(c->pt_stream, c->ev_bounced, 0));
        if (detail) HIPCHK(c, hipEventRecord(c->ev_pt_t[slot][0], c->pt_stream));
    }
    auto join = [&]() -> int {
        for (int k = 1; k < S; k++) { HIPCHK(c, hipEventRecord(c->ev_join[k - 1], st(k))); HIPCHK(c, hipStreamWaitEvent(s, c->ev_join[k - 1], 0)); }
        return MQ_OK;
    };
    // a shared-stack entry per level with pending siblings, +1 for the per-frame root waiting at the bottom
    const bool packet = c->props.packet_camera_rays && (int)std::max(c->s_depth, c->d_depth) + 2 <= mq_packet_stack_entries();
    if (!c->count_enabled) // camera rays: traversal in its own launch (the counting instantiation of the primary kernel traces them inline)
        for (int k = 0; k < S; k++) {
            MqFrame FP = FS[k]; // (its stack spill area: MqFrame::cam_spill, one per pixel slot)
            // Camera rays: ONE WAVE PER TILE, every block launched (round 3).  The other frame kernels run resident grids with grid-stride
            // loops; here the hardware's block dispatch is the better scheduler -- tiles differ in cost (sky against interiors), a wave
            // that has finished its tile makes room at once, and blocks that come and go let the frame's own (higher-priority) launches
            // in: 0.45 -> 0.39 ms for the overlapped launch, 1.87 -> 1.80 ms per frame (profiles/r03_y_camera_ray_grid.txt).  The
            // packet kernel keeps its resident grid (one tile per wave per trip).
            static const bool cam_resident = getenv("MQ_DEBUG_CAMERA_GRID_RESIDENT") != nullptr; // the A/B switch
            const int cam_grid = (packet || cam_resident) ? (overlap_pt ? c->grid_frame[3] : sub_grid(3)) : (int)((FP.slot_end - FP.slot_begin + 255u) / 256u);
            e = mq_launch_primary_trace(c->scene, c->params, FP, packet, std::max(1, cam_grid), overlap_pt ? c->pt_stream : st(k));
            if (e) return fail(c, MQ_EHIP, std::string("primary trace launch: ") + hipGetErrorString((hipError_t)e));
        }
    if (overlap_pt) {
        if (detail) HIPCHK(c, hipEventRecord(c->ev_pt_t[slot][1], c->pt_stream));
        HIPCHK(c, hipEventRecord(c->ev_pt_done[parity], c->pt_stream));
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_pt_done[parity], 0));
        for (int k = 1; k < S; k++) HIPCHK(c, hipStreamWaitEvent(st(k), c->ev_pt_done[parity], 0));
    }
    if (detail) HIPCHK(c, hipEventRecord(ev[1], s));
    for (int k = 0; k < S; k++) {
        // first-hit shading: one thread per pixel slot, every block launched (0.315 -> 0.304 ms against the resident grid; same A/B).  The
        // counting instantiation traces inline and indexes its stack spill area by block: resident grid.
        static const bool prim_resident = getenv("MQ_DEBUG_PRIMARY_GRID_RESIDENT") != nullptr;
        const int prim_grid = (c->count_enabled || prim_resident) ? sub_grid(0) : std::max(1, (int)((FS[k].slot_end - FS[k].slot_begin + FS[k].shade_block - 1) / FS[k].shade_block));
        e = mq_launch_primary(c->scene, c->params, FS[k], guided, c->count_enabled, prim_grid, st(k));
        if (e) return fail(c, MQ_EHIP, std::string("primary launch: ") + hipGetErrorString((hipError_t)e));
    }
    if (rounds == 0) { int r = join(); if (r) return r; }
    if (detail || timed == 0) HIPCHK(c, hipEventRecord(ev[2], s));
    for (int r = 0; r < rounds; r++) {
        for (int k = 0; k < S; k++) {
            e = mq_launch_trace_queue(c->scene, FS[k], r, c->count_enabled, sub_grid(1), st(k));
            if (e) return fail(c, MQ_EHIP, std::string("trace launch: ") + hipGetErrorString((hipError_t)e));
        }
        if (r < timed && detail) HIPCHK(c, hipEventRecord(ev[3 + 2 * r], s));
        // launches of this frame in order: T0 B0 T1 B1 ...; `behind` counts back from the last one (0 = the last bounce kernel)
        if (overlap_pt && 2 * r == 2 * rounds - 1 - pt_behind) { HIPCHK(c, hipEventRecord(c->ev_bounced, s)); c->bounced_valid = true; }
        for (int k = 0; k < S; k++) {
            e = mq_launch_bounce(c->scene, c->params, FS[k], r, guided, c->count_enabled, sub_grid(2), st(k));
            if (e) return fail(c, MQ_EHIP, std::string("bounce launch: ") + hipGetErrorString((hipError_t)e));
        }
        if (r == rounds - 1) { int rr = join(); if (rr) return rr; } // every chain is done before the render interval ends
        if (overlap_pt && 2 * r + 1 == 2 * rounds - 1 - pt_behind) { HIPCHK(c, hipEventRecord(c->ev_bounced, s)); c->bounced_valid = true; }
        if (r < timed && (detail || r == timed - 1)) HIPCHK(c, hipEventRecord(ev[4 + 2 * r], s)); // the last one ends the render interval
    }
    rr_surface.end();
    if (c->params.debug_output_connected) { // mcpg.comp:212-277: part of the surface pass, i.e. before the update pass
        e = mq_launch_debug_view(c->params, F, c->grid_blocks, s);
        if (e) return fail(c, MQ_EHIP, std::string("debug view launch: ") + hipGetErrorString((hipError_t)e));
    }
    if (guided) { // render_mcpg.cpp:261-277
        RoctxRange rr("update");
        e = mq_launch_apply(c->params, F, std::max(1, c->cu_count) * 8, c->props.sequential_update_pass ? c->mc_total : 0u, s);
        if (e) return fail(c, MQ_EHIP, std::string("apply launch: ") + hipGetErrorString((hipError_t)e));
    }
    // ---- volume passes, render_mcpg.cpp:280-320 (their device time is part of the update interval) ----
    if (guided) c->queues_dirty = volume; // the update pass zeroed the control words of every queue (reset_queue_control): volume entries start at 0
    if (volume) {
        RoctxRange rr("volume");
        const size_t px = (size_t)c->W * c->H;
        HIPCHK(c, hipMemcpyAsync(c->d_prev_vdepth.p, c->d_out[MQ_OUT_VOLUME_DEPTH].p, px * 2, hipMemcpyDeviceToDevice, s)); // delay-1 feedback connector
        HIPCHK(c, hipMemcpyAsync(c->d_out[MQ_OUT_VOLUME_MV].p, c->d_out[MQ_OUT_GB_MV].p, px * 4, hipMemcpyDeviceToDevice, s)); // :284-288
        if (c->params.volume_forward_project && !first_iteration) { // :296-311
            // every pixel of the image projects (a rank of a partitioned frame too: the caller gathers the ranks' volume_depth tiles,
            // mq_untile_volume_depth; without that exchange the depths of the other ranks' pixels are stale zeros and project nothing)
            MqFrame FA = F; FA.tile_mul = 1u; FA.tile_add = 0u; FA.n_local_tiles = c->tiles_x * c->tiles_y;
            e = mq_launch_forward_project(c->params, FA, c->grid_blocks, s);
            if (e) return fail(c, MQ_EHIP, std::string("forward project launch: ") + hipGetErrorString((hipError_t)e));
        }
        for (int vs = 0; vs < c->params.volume_spp; vs++) {
            const int r = rounds + vs;
            e = mq_launch_volume_sample(c->scene, c->params, F, vs, r, c->count_enabled, c->grid_blocks, s);
            if (!e) e = mq_launch_trace_queue(c->scene, F, r, c->count_enabled, c->grid_blocks, s);
            if (!e) e = mq_launch_volume_shade(c->scene, c->params, F, vs, r, c->count_enabled, c->grid_blocks, s);
            if (e) return fail(c, MQ_EHIP, std::string("volume launch: ") + hipGetErrorString((hipError_t)e));
        }
        e = mq_launch_volume_finish(c->params, F, c->grid_blocks, s);
        if (e) return fail(c, MQ_EHIP, std::string("volume finish launch: ") + hipGetErrorString((hipError_t)e));
        c->volume_outputs_zero = false;
    } else if (!c->volume_outputs_zero) {
        HIPCHK(c, hipMemsetAsync(c->d_out[MQ_OUT_VOLUME].p, 0, c->d_out[MQ_OUT_VOLUME].bytes, s));
        HIPCHK(c, hipMemsetAsync(c->d_out[MQ_OUT_VOLUME_TILES].p, 0, c->d_out[MQ_OUT_VOLUME_TILES].bytes, s));
        c->volume_outputs_zero = true;
    }
    HIPCHK(c, hipEventRecord(ev[3 + 2 * timed], s));
    // the first-hit kernel has read this parity's hit buffer (recorded here, at the end of the frame, rather than behind that
    // kernel: an event between two dependent launches costs ~5 us, and the camera rays that wait for it are two frames away)
    if (overlap_pt) { HIPCHK(c, hipEventRecord(c->ev_shaded[parity], s)); c->shaded_valid[parity] = true; }
    { int r = scene_used(c, s); if (r) return r; }
    c->frame_parity++;
    c->ev_rounds[slot] = timed;
    c->ev_pending[slot] = true; c->ev_last = slot; c->ev_slot = (slot + 1) % mq_ctx::EV_RING;
    c->ev_valid = true;
    return MQ_OK;
}

int mq_sync(mq_ctx* c) {
    if (!c) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    return MQ_OK;
}
int mq_map_output(mq_ctx* c, int which, void** dev_ptr, size_t* bytes) {
    if (!c || which < 0 || which >= MQ_OUT_COUNT) return MQ_EINVAL;
    if (!c->d_out[which].p) return fail(c, MQ_ESTATE, "not connected");
    if (dev_ptr) *dev_ptr = c->d_out[which].p; if (bytes) *bytes = c->d_out[which].bytes;
    return MQ_OK;
}
int mq_read_output(mq_ctx* c, int which, void* dst, size_t bytes) {
    if (!c || !dst || which < 0 || which >= MQ_OUT_COUNT) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    if (!c->d_out[which].p) return fail(c, MQ_ESTATE, "not connected");
    if (bytes > c->d_out[which].bytes) return fail(c, MQ_EINVAL, "read larger than the output");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    HIPCHK(c, hipMemcpy(dst, c->d_out[which].p, bytes, hipMemcpyDeviceToHost));
    return MQ_OK;
}
int mq_last_frame_ms(mq_ctx* c, float* total_ms, float* render_ms, float* update_ms) {
    if (!c) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    if (!c->ev_valid || c->ev_last < 0) return fail(c, MQ_ESTATE, "no timed frame yet");
    { int r = drain_slot(c, c->ev_last); if (r) return r; }
    if (render_ms) *render_ms = c->last_render_ms; if (update_ms) *update_ms = c->last_update_ms; if (total_ms) *total_ms = c->last_render_ms + c->last_update_ms;
    return MQ_OK;
}
int mq_timing_reset(mq_ctx* c) {
    if (!c) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    for (int i = 0; i < mq_ctx::EV_RING; i++) { int r = drain_slot(c, i); if (r) return r; }
    c->t_render_sum = c->t_update_sum = 0.0; c->t_frames = 0; c->t_detail_frames = 0; c->ev_counter = 0;
    c->t_primary_sum = c->t_trace_sum = c->t_bounce_sum = 0.0; c->t_pt_kernel_sum = 0.0;
    for (int i = 0; i < MQ_TIMING_ROUNDS; i++) c->t_round_trace[i] = c->t_round_shade[i] = 0.0;
    return MQ_OK;
}
int mq_timing_set_interval(mq_ctx* c, uint32_t every) {
    if (!c || every == 0) return MQ_EINVAL;
    c->timing_interval = every;
    return MQ_OK;
}
int mq_timing_detail_frames(mq_ctx* c, uint32_t* frames) {
    if (!c) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    for (int i = 0; i < mq_ctx::EV_RING; i++) { int r = drain_slot(c, i); if (r) return r; }
    if (frames) *frames = c->t_detail_frames;
    return MQ_OK;
}
int mq_timing_get(mq_ctx* c, uint32_t* frames, double* render_ms_sum, double* update_ms_sum) {
    if (!c) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    for (int i = 0; i < mq_ctx::EV_RING; i++) { int r = drain_slot(c, i); if (r) return r; }
    if (frames) *frames = c->t_frames; if (render_ms_sum) *render_ms_sum = c->t_render_sum; if (update_ms_sum) *update_ms_sum = c->t_update_sum;
    return MQ_OK;
}
int mq_timing_get_detail(mq_ctx* c, double* primary_ms_sum, double* trace_ms_sum, double* bounce_ms_sum) {
    if (!c) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    for (int i = 0; i < mq_ctx::EV_RING; i++) { int r = drain_slot(c, i); if (r) return r; }
    if (primary_ms_sum) *primary_ms_sum = c->t_primary_sum; if (trace_ms_sum) *trace_ms_sum = c->t_trace_sum; if (bounce_ms_sum) *bounce_ms_sum = c->t_bounce_sum;
    return MQ_OK;
}
int mq_timing_get_rounds(mq_ctx* c, double* trace_ms_sum, double* shade_ms_sum, int n) {
    if (!c || n < 0) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    for (int i = 0; i < mq_ctx::EV_RING; i++) { int r = drain_slot(c, i); if (r) return r; }
    for (int i = 0; i < n && i < MQ_TIMING_ROUNDS; i++) { if (trace_ms_sum) trace_ms_sum[i] = c->t_round_trace[i]; if (shade_ms_sum) shade_ms_sum[i] = c->t_round_shade[i]; }
    return MQ_OK;
}
int mq_enable_counters(mq_ctx* c, int on) { if (!c) return MQ_EINVAL; c->count_enabled = on != 0; return MQ_OK; }
int mq_get_counters(mq_ctx* c, mq_counters* out) {
    if (!c || !out) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    if (!c->d_counters.p) return fail(c, MQ_ESTATE, "not connected");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    MqCountersDev d;
    HIPCHK(c, hipMemcpy(&d, c->d_counters.p, sizeof d, hipMemcpyDeviceToHost));
    out->rays = d.rays; out->nodes = d.nodes; out->tris = d.tris; out->segments = d.segments; out->guided_segments = d.guided_segments;
    out->lc_touches = d.lc_touches; out->mc_updates_accepted = d.mc_updates_accepted; out->mc_updates_dropped = d.mc_updates_dropped;
    out->mc_state_reads = d.mc_state_reads; out->pixels = d.pixels;
    out->queue_rays = d.q_rays; out->queue_nodes = d.q_nodes; out->queue_tris = d.q_tris;
    uint32_t flag = 0;
    HIPCHK(c, hipMemcpy(&flag, c->d_ctrl.p, 4, hipMemcpyDeviceToHost));
    out->queue_overflow = flag;
    return MQ_OK;
}

int mq_debug_section_clocks(mq_ctx* c, uint64_t* out, int n, int reset) {
    if (!c || !out || n < 0) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    if (!c->d_counters.p) return fail(c, MQ_ESTATE, "not connected");
    static_assert(MQ_PROF_SECTION_COUNT == MQ_PROF_SECTIONS, "section count");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    MqCountersDev d;
    HIPCHK(c, hipMemcpy(&d, c->d_counters.p, sizeof d, hipMemcpyDeviceToHost));
    for (int i = 0; i < n && i < MQ_PROF_SECTIONS + 64; i++) out[i] = i < MQ_PROF_SECTIONS ? d.prof[i] : d.ray_hist[i - MQ_PROF_SECTIONS];
    if (reset) HIPCHK(c, hipMemset((char*)c->d_counters.p + offsetof(MqCountersDev, prof), 0, sizeof d.prof + sizeof d.ray_hist));
    return MQ_OK;
}

// Learning state in the device layout (MqMCState 64 B / MqLCCell 16 B per entry): lets a test start the GPU
// from the oracle's learned tables.  Writing needs one processed frame (the first frame zeroes the tables).
static int state_buf(mq_ctx* c, int which, DevBuf** b) {
    if (!c->connected) return fail(c, MQ_ESTATE, "not connected");
    if (which == 0) *b = &c->d_mc; else if (which == 1) *b = &c->d_lc; else if (which == 2) *b = &c->d_dist_mc;
    else if ((which == 3 || which == 4) && c->d_lc_stats.p) *b = which == 3 ? &c->d_lc_stats : &c->d_last_upd;
    else if (which == 3 || which == 4) return fail(c, MQ_ESTATE, "no statistics: set \"debug: LC lock statistics\" and render a frame");
    else return fail(c, MQ_EINVAL, "state: 0 = Markov chains, 1 = light cache, 2 = distance Markov chains, 3 = light-cache lock statistics, 4 = last update counts");
    return MQ_OK;
}
int mq_debug_state_read(mq_ctx* c, int which, void* dst, size_t bytes) {
    if (!c || !dst) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    DevBuf* b = nullptr; int r = state_buf(c, which, &b); if (r) return r;
    if (bytes != b->bytes) return fail(c, MQ_EINVAL, "state size mismatch");
    HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(dst, b->p, bytes, hipMemcpyDeviceToHost));
    return MQ_OK;
}
int mq_debug_state_write(mq_ctx* c, int which, const void* src, size_t bytes) {
    if (!c || !src) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    DevBuf* b = nullptr; int r = state_buf(c, which, &b); if (r) return r;
    if (bytes != b->bytes) return fail(c, MQ_EINVAL, "state size mismatch");
    if (c->iteration == 0) return fail(c, MQ_ESTATE, "process one frame before writing state (the first frame zeroes it)");
    HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(b->p, src, bytes, hipMemcpyHostToDevice));
    return MQ_OK;
}

// The g-buffer node's outputs (hits, gbuffer, mv, albedo, first-hit emission) on the rows a rank of a partitioned frame needs for
// the ReSTIR node and the post chain (band_rows: owned + spatial radius + reprojection halo).  The MCPG node of that rank renders
// interleaved tiles; its row band is rendered here, by the same two kernels in their g-buffer-only form (camera rays, first-hit
// shading: a deterministic function of the pixel, so the values equal what the tiles' own launches wrote).  Once per frame.
static int ensure_band_gbuffer(mq_ctx* c, const mq_uniform* u, hipStream_t s) {
    if (c->world == 1 || c->band_gb_valid) return MQ_OK;
    const BandRows b = band_rows(c->props, c->H, c->rank, c->world);
    MqFrame F; fill_frame(c, u, F);
    { int r = frame_grids(c, F); if (r) return r; }
    F.tile_mul = 1u; F.tile_add = b.g0 * c->tiles_x;
    F.n_local_tiles = (b.g1 - b.g0) * c->tiles_x;
    if (F.n_local_tiles > c->slot_tiles) return fail(c, MQ_ESTATE, "row band larger than the buffers of this connect (reconnect after changing the partition)");
    F.slot_begin = 0u; F.slot_end = F.n_local_tiles * 64u;
    F.gbuffer_only = 1u;
    F.prim_hits = (uint4*)c->d_band_hits.p; F.cam_spill = (unsigned long long*)c->d_cam_spill[1].p;
    if (F.n_local_tiles) {
        { int r = scene_ready(c, s); if (r) return r; }
        int e = mq_launch_primary_trace(c->scene, c->params, F, false, (int)((F.slot_end + 255u) / 256u), s);
        if (!e) e = mq_launch_primary(c->scene, c->params, F, false, false, (int)((F.slot_end + F.shade_block - 1) / F.shade_block), s);
        if (e) return fail(c, MQ_EHIP, std::string("band g-buffer launch: ") + hipGetErrorString((hipError_t)e));
        { int r = scene_used(c, s); if (r) return r; }
    }
    c->band_gb_valid = true;
    return MQ_OK;
}

// ---- ReSTIR DI node (mq_restir.h): RendererRESTIR::process, src/render_restir/renderer_restir.cpp:129-251 ----------
// On a rank of a partitioned frame (mq_set_partition with world > 1) the node works on a band of rows (band_rows above):
// generate + temporal reuse on the rows it owns widened by the spatial radius, spatial reuse + shade on the rows it owns; the
// caller then moves the other ranks' rows of the new reservoirs into this rank's "previous reservoirs" (mq_map_halo).
int mq_restir_process(mq_ctx* c, const mq_uniform* u, int render, void* stream) {
    if (!c || !u) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context: no HIP device");
    if (!c->connected) return fail(c, MQ_ESTATE, "mq_restir_process before mq_connect");
    if (!c->committed) return fail(c, MQ_ESTATE, "mq_restir_process before mq_scene_commit");
    hipStream_t s = (hipStream_t)stream;
    c->last_stream = s;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->params_dirty) props_to_params(c);
    { int r = scene_ready(c, s); if (r) return r; }
    const MqProps& q = c->props;
    if (!c->restir_seeded) { // pipeline (re)creation, renderer_restir.cpp:152-158
        if (q.restir_randomize_seed) { std::random_device dev; std::mt19937 rng(dev()); c->props.restir_seed = (uint32_t)rng(); }
        c->restir_seed_in_use = c->props.restir_seed; c->restir_seeded = true;
    }
    if (render) { int r = ensure_band_gbuffer(c, u, s); if (r) return r; }
    MqRestirParams R;
    R.spp = q.restir_spp; R.seed = c->restir_seed_in_use; R.visibility_shade = q.restir_shade_visibility;
    R.temporal_normal_reject_cos = (float)std::cos((double)q.restir_temporal_normal_angle); R.temporal_depth_reject = q.restir_temporal_depth;
    R.spatial_normal_reject_cos = (float)std::cos((double)q.restir_spatial_normal_angle); R.spatial_depth_reject = q.restir_spatial_depth;
    R.temporal_clamp_m = q.restir_temporal_clamp_m; R.spatial_radius = q.restir_spatial_radius; R.temporal_bias_correction = q.restir_temporal_bias; R.spatial_bias_correction = q.restir_spatial_bias;
    R.boiling_filter_strength = q.restir_boiling; R.spatial_reuse_iterations = std::max(q.restir_spatial_iterations, 1); R.apply_mv = q.restir_apply_mv ? 1 : 0; // renderer_restir.cpp:175
    MqRestirFrame F; memset(&F, 0, sizeof F);
    F.u = *u; F.W = c->W; F.H = c->H; F.tiles_x = c->tiles_x; F.n_tiles = c->tiles_x * c->tiles_y;
    F.hits = (const uint32_t*)c->d_out[MQ_OUT_HITS].p; F.gbuffer = (const uint4*)c->d_out[MQ_OUT_GBUFFER].p; F.prev_gbuffer = (const uint4*)c->d_restir_prev_gb.p;
    F.mv = (const uint32_t*)c->d_out[MQ_OUT_GB_MV].p; F.prev_reservoirs = (const uint4*)c->d_restir_prev.p;
    F.irradiance = (float4*)c->d_out[MQ_OUT_RESTIR_IRRADIANCE].p; F.moments = (float2*)c->d_out[MQ_OUT_RESTIR_MOMENTS].p;
    F.stack_spill = (unsigned long long*)c->d_spill.p;
    F.flags = (uint32_t*)c->d_ctrl.p;
    // the rows of this rank: `wide` = generate + temporal reuse, `own` = spatial reuse + shade (one rank: both the whole image)
    const BandRows b = band_rows(q, c->H, c->rank, c->world);
    const uint32_t wide[2] = {b.e0 * c->tiles_x, b.e1 * c->tiles_x}, own[2] = {b.t0 * c->tiles_x, b.t1 * c->tiles_x};
    F.slot_tile0 = wide[0]; F.row_lo = b.g0 * 8u; F.row_hi = std::min(c->H, b.g1 * 8u);
    if (wide[1] - wide[0] > c->slot_tiles && c->world > 1) return fail(c, MQ_ESTATE, "row band larger than the buffers of this connect");
    auto rows = [&](const uint32_t t[2]) { F.tile_begin = t[0]; F.tile_end = t[1]; };
    uint4* const out = (uint4*)c->d_out[MQ_OUT_RESTIR_RESERVOIRS].p; uint4* const pong = (uint4*)c->d_restir_pong.p;
    if (!c->restir_occ[0]) { // resident blocks per CU of each pass kernel; the spill area holds grid_blocks >= any of these grids
        int e0 = mq_restir_resident_blocks(c->restir_occ);
        if (e0) return fail(c, MQ_EHIP, std::string("occupancy query: ") + hipGetErrorString((hipError_t)e0));
        for (int& o : c->restir_occ) o = std::max(1, o);
        if (const char* ev = getenv("MQ_DEBUG_RESTIR_BLOCKS_PER_CU")) for (int& o : c->restir_occ) o = std::max(1, atoi(ev)); // tuning experiments only
    }
    auto grid_of = [&](int pass) { return std::min(c->grid_blocks, std::max(1, c->cu_count) * c->restir_occ[pass]); };
    const int grid = grid_of(0);
    int e = 0;
    if (!render) { // renderer_restir.cpp:189-197: the clear pass writes set (1): `reservoirs` = the graph output
        F.res_a = out; F.res_read = pong;
        rows(own);
        RoctxRange rr("clear");
        e = mq_launch_restir(c->scene, c->params, R, F, 4, grid, s);
        if (e) return fail(c, MQ_EHIP, std::string("restir clear launch: ") + hipGetErrorString((hipError_t)e));
    } else {
        // the ping-pong of renderer_restir.cpp:136-146,213-250: the last writer before the shade pass is the graph output
        const bool spatial = q.restir_spatial_iterations > 0;
        F.res_a = spatial ? pong : out; F.res_read = spatial ? out : pong;
        // The generate and shade passes trace one closest-hit ray per pixel and sample: as a wavefront through the MCPG
        // node's queues and traversal kernel ("inline restir rays" = 0, the default), or inline in the pass kernels.
        const bool wavefront = !q.restir_inline_rays;
        MqFrame FQ;
        if (wavefront) {
            fill_frame(c, u, FQ);
            { int r = frame_grids(c, FQ); if (r) return r; }
            HIPCHK(c, hipMemsetAsync((char*)c->d_ctrl.p + 4 * MQ_CTRL_QUEUE0, 0, ((size_t)MQ_CTRL_WORDS - MQ_CTRL_QUEUE0) * 4, s)); // the rounds' queue tails and fetch heads
            c->queues_dirty = true; // the MCPG node's next frame starts from zeroed counters
        }
        auto traced_pass = [&](int which_a, int round) -> int { // request -> trace -> finish
            int e2 = mq_launch_restir_wavefront(c->scene, c->params, R, F, FQ, which_a, round, grid_of(which_a == 0 ? 0 : 3), s);
            if (!e2) e2 = mq_launch_trace_queue(c->scene, FQ, round, false, c->grid_frame[1], s);
            if (!e2) e2 = mq_launch_restir_wavefront(c->scene, c->params, R, F, FQ, which_a + 1, round, grid_of(which_a == 0 ? 0 : 3), s);
            return e2;
        };
        rows(wide);
        { RoctxRange rr("generate samples");
          if (wavefront) { for (int smp = 0; smp < std::max(1, R.spp) && !e; smp++) e = traced_pass(0, smp); }
          else e = mq_launch_restir(c->scene, c->params, R, F, 0, grid, s); }
        if (!e && q.restir_temporal_reuse && c->restir_iteration > 0) { RoctxRange rr("temporal reuse"); e = mq_launch_restir(c->scene, c->params, R, F, 1, grid_of(1), s); }
        rows(own);
        if (!e && spatial) { RoctxRange rr("spatial reuse"); F.res_a = out; F.res_read = pong; e = mq_launch_restir(c->scene, c->params, R, F, 2, grid_of(2), s); }
        F.res_a = out; F.res_read = pong;
        if (!e) { RoctxRange rr("shade"); e = wavefront ? traced_pass(2, std::max(1, R.spp)) : mq_launch_restir(c->scene, c->params, R, F, 3, grid_of(3), s); }
        if (e) return fail(c, MQ_EHIP, std::string("restir launch: ") + hipGetErrorString((hipError_t)e));
    }
    // the graph's delay-1 inputs of the next frame: "reservoirs" (the rows this rank owns; the others arrive from their owners,
    // mq_map_halo) and "prev_gbuffer" (every row it holds) (renderer_restir.hpp:73-74,86-87)
    const size_t r0 = (size_t)b.t0 * 8u * c->W, r1 = (size_t)std::min(c->H, b.t1 * 8u) * c->W; // pixels [r0, r1)
    const size_t g0 = (size_t)F.row_lo * c->W, g1 = (size_t)F.row_hi * c->W;
    if (r1 > r0) HIPCHK(c, hipMemcpyAsync((char*)c->d_restir_prev.p + r0 * 64, (const char*)out + r0 * 64, (r1 - r0) * 64, hipMemcpyDeviceToDevice, s));
    if (g1 > g0) HIPCHK(c, hipMemcpyAsync((char*)c->d_restir_prev_gb.p + g0 * 16, (const char*)c->d_out[MQ_OUT_GBUFFER].p + g0 * 16, (g1 - g0) * 16, hipMemcpyDeviceToDevice, s));
    c->restir_iteration++;
    return scene_used(c, s);
}

// ---- post chain (mq_post.hip): accum + volume accum + add ------------------------------------------------------
// On a rank of a partitioned frame: the rows the rank owns (band_rows), from the gathered `irradiance` / `volume` images (the
// caller's all-gather + mq_untile / mq_untile_volume come first); last frame's accumulated rows of the other ranks arrive
// through mq_map_halo.  The volume image then follows the g-buffer's motion vectors: the forward-projected `volume_mv` is a
// scatter over a rank's own tiles and is not exchanged.
int mq_post_clear(mq_ctx* c) { if (!c) return MQ_EINVAL; c->post_first = true; return MQ_OK; }
int mq_post_process(mq_ctx* c, void* stream) {
    if (!c) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context: no HIP device");
    if (!c->connected) return fail(c, MQ_ESTATE, "mq_post_process before mq_connect");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->world > 1 && !c->band_gb_valid) return fail(c, MQ_ESTATE, "mq_post_process on a rank of a partitioned frame needs the g-buffer of its rows: call mq_restir_process or mq_band_gbuffer first");
    const MqProps& q = c->props;
    const float A[2][6] = {{q.accum_alpha, q.accum_max_history, (float)std::cos((double)q.accum_normal_threshold), q.accum_depth_threshold, q.accum_enable_mv ? 1.0f : 0.0f, q.accum_reuse_border ? 1.0f : 0.0f},
                           {q.vaccum_alpha, q.vaccum_max_history, (float)std::cos((double)q.vaccum_normal_threshold), q.vaccum_depth_threshold, q.vaccum_enable_mv ? 1.0f : 0.0f, q.vaccum_reuse_border ? 1.0f : 0.0f}};
    const int src[2] = {MQ_OUT_IRRADIANCE, MQ_OUT_VOLUME}, mv[2] = {MQ_OUT_GB_MV, c->world > 1 ? MQ_OUT_GB_MV : MQ_OUT_VOLUME_MV}, out[2] = {MQ_OUT_ACCUM, MQ_OUT_VOLUME_ACCUM}, hist[2] = {MQ_OUT_ACCUM_HISTORY, MQ_OUT_VOLUME_ACCUM_HISTORY};
    const BandRows b = band_rows(q, c->H, c->rank, c->world);
    const uint32_t rows[4] = {b.t0 * 8u, std::min(c->H, b.t1 * 8u), b.g0 * 8u, std::min(c->H, b.g1 * 8u)};
    RoctxRange rr_post("accumulate + add");
    for (int k = 0; k < 2; k++) {
        // volume accum.mv <- render_markovchain.volume_mv: only written by frames with a volume pass; without one there is no motion to follow
        const bool have_vmv = k == 0 || (c->params.volume_spp > 0 && !c->volume_outputs_zero);
        float Ak[6]; memcpy(Ak, A[k], sizeof Ak); if (!have_vmv) Ak[4] = 0.0f;
        int e = mq_launch_accumulate(Ak, c->W, c->H, c->d_out[src[k]].p, c->d_out[mv[k]].p, c->d_out[MQ_OUT_GBUFFER].p, c->d_post_prev_gb.p, c->d_post_prev_out[k].p, c->d_post_prev_hist[k].p,
                                     c->d_out[out[k]].p, c->d_out[hist[k]].p, c->post_first ? 1 : 0, rows, (uint32_t*)c->d_ctrl.p, s);
        if (e) return fail(c, MQ_EHIP, std::string("accumulate launch: ") + hipGetErrorString((hipError_t)e));
    }
    int e = mq_launch_compose(c->W, rows[0], rows[1], c->d_out[MQ_OUT_ACCUM].p, c->d_out[MQ_OUT_GB_ALBEDO].p, c->d_out[MQ_OUT_VOLUME_ACCUM].p, c->d_out[MQ_OUT_GB_IRRADIANCE].p,
                              q.add_restir ? c->d_out[MQ_OUT_RESTIR_IRRADIANCE].p : nullptr, c->d_out[MQ_OUT_FINAL].p, s);
    if (e) return fail(c, MQ_EHIP, std::string("compose launch: ") + hipGetErrorString((hipError_t)e));
    // the graph's delay-1 connections (prev_out, prev_history: the rows this rank owns; prev_gbuffer: every row it holds)
    const size_t r0 = (size_t)rows[0] * c->W, r1 = (size_t)rows[1] * c->W, g0 = (size_t)rows[2] * c->W, g1 = (size_t)rows[3] * c->W;
    for (int k = 0; k < 2 && r1 > r0; k++) {
        HIPCHK(c, hipMemcpyAsync((char*)c->d_post_prev_out[k].p + r0 * 16, (const char*)c->d_out[out[k]].p + r0 * 16, (r1 - r0) * 16, hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync((char*)c->d_post_prev_hist[k].p + r0 * 4, (const char*)c->d_out[hist[k]].p + r0 * 4, (r1 - r0) * 4, hipMemcpyDeviceToDevice, s));
    }
    if (g1 > g0) HIPCHK(c, hipMemcpyAsync((char*)c->d_post_prev_gb.p + g0 * 16, (const char*)c->d_out[MQ_OUT_GBUFFER].p + g0 * 16, (g1 - g0) * 16, hipMemcpyDeviceToDevice, s));
    c->post_first = false;
    c->last_stream = s;
    return MQ_OK;
}

// ---- row partition of the ReSTIR node / post chain: layout query, the band's g-buffer alone, the halo buffers ----------
int mq_band_layout(const mq_ctx* c, uint32_t width, uint32_t height, int rank, int world, mq_band* out) {
    if (!c || !out || !width || !height || world < 1 || rank < 0 || rank >= world) return MQ_EINVAL;
    const BandRows b = band_rows(c->props, height, rank, world);
    out->row_begin = std::min(height, b.t0 * 8u); out->row_end = std::min(height, b.t1 * 8u);
    out->reuse_begin = std::min(height, b.e0 * 8u); out->reuse_end = std::min(height, b.e1 * 8u);
    out->need_begin = std::min(height, b.g0 * 8u); out->need_end = std::min(height, b.g1 * 8u);
    return MQ_OK;
}
int mq_band_gbuffer(mq_ctx* c, const mq_uniform* u, void* stream) {
    if (!c || !u) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context: no HIP device");
    if (!c->connected || !c->committed) return fail(c, MQ_ESTATE, "mq_band_gbuffer before mq_connect / mq_scene_commit");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->params_dirty) props_to_params(c);
    c->last_stream = (hipStream_t)stream;
    return ensure_band_gbuffer(c, u, (hipStream_t)stream);
}
int mq_map_halo(mq_ctx* c, int which, void** send_base, void** recv_base, size_t* row_bytes) {
    if (!c || which < 0 || which >= MQ_HALO_COUNT) return MQ_EINVAL;
    if (!c->connected) return fail(c, MQ_ESTATE, "not connected");
    static const int outs[MQ_HALO_COUNT] = {MQ_OUT_RESTIR_RESERVOIRS, MQ_OUT_ACCUM, MQ_OUT_ACCUM_HISTORY, MQ_OUT_VOLUME_ACCUM, MQ_OUT_VOLUME_ACCUM_HISTORY};
    DevBuf* const prev[MQ_HALO_COUNT] = {&c->d_restir_prev, &c->d_post_prev_out[0], &c->d_post_prev_hist[0], &c->d_post_prev_out[1], &c->d_post_prev_hist[1]};
    if (send_base) *send_base = c->d_out[outs[which]].p;
    if (recv_base) *recv_base = prev[which]->p;
    if (row_bytes) *row_bytes = (size_t)c->W * k_bpp[outs[which]];
    return MQ_OK;
}

int mq_debug_learn_log_read(mq_ctx* c, void* dst, size_t cap_records, size_t* n_records) {
    if (!c || !n_records) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    if (!c->d_learn_count.p) { *n_records = 0; return MQ_OK; }
    HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipDeviceSynchronize());
    uint32_t n = 0;
    HIPCHK(c, hipMemcpy(&n, c->d_learn_count.p, 4, hipMemcpyDeviceToHost));
    *n_records = n; // records proposed; those beyond the log's capacity were counted, not kept
    const size_t have = std::min<size_t>(n, c->learn_log_cap);
    if (dst && cap_records) HIPCHK(c, hipMemcpy(dst, c->d_learn_log.p, std::min(have, cap_records) * 64, hipMemcpyDeviceToHost));
    return MQ_OK;
}

// mq_link_kernel + mq_apply_kernel alone, on caller-given queue contents (the update pass of render_mcpg.cpp:261-277 /
// compute_updates.comp:56-124): records in the queue's own 64-byte layout with slot and arrival rank filled in.
int mq_debug_apply_updates(mq_ctx* c, const void* records, uint32_t n, const mq_uniform* u) {
    if (!c || !u || (!records && n)) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    if (!c->connected) return fail(c, MQ_ESTATE, "not connected");
    if (c->iteration == 0) return fail(c, MQ_ESTATE, "process one frame first (the first frame zeroes the tables)");
    if (n > c->queue_cap) return fail(c, MQ_EINVAL, "more records than the update queue holds");
    HIPCHK(c, hipSetDevice(c->device)); HIPCHK(c, hipDeviceSynchronize());
    if (c->params_dirty) props_to_params(c);
    // positions [0, n) of the sharded queue: shard s owns every 16th block of 64 positions, so its tail is the number
    // of those positions that fall into its blocks
    std::vector<uint32_t> ctrl(MQ_CTRL_GROUP, 0u);
    for (uint32_t p = 0; p < n; p += 64) ctrl[((p >> 6) & (MQ_SHARDS - 1)) * MQ_SHARD_STRIDE] += std::min(64u, n - p);
    std::vector<MqUpdate> recs((const MqUpdate*)records, (const MqUpdate*)records + n);
    for (auto& r : recs) { if (r.slot >= c->mc_total || r.rank >= MQ_MAX_UPDATES) return fail(c, MQ_EINVAL, "record with a bad slot or rank"); r.next = 0u; }
    if (n) HIPCHK(c, hipMemcpy(c->d_queue.p, recs.data(), (size_t)n * sizeof(MqUpdate), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy((uint32_t*)c->d_ctrl.p + MQ_CTRL_UPDATES, ctrl.data(), MQ_CTRL_GROUP * 4, hipMemcpyHostToDevice));
    MqFrame F; fill_frame(c, u, F);
    HIPCHK(c, hipMemsetAsync(c->d_active_ctrl.p, 0, (size_t)MQ_CTRL_GROUP * 4, nullptr)); // (a frame's first-hit kernel does this)
    int e = mq_launch_apply(c->params, F, std::max(1, c->cu_count) * 8, c->props.sequential_update_pass ? c->mc_total : 0u, nullptr);
    if (e) return fail(c, MQ_EHIP, std::string("apply launch: ") + hipGetErrorString((hipError_t)e));
    HIPCHK(c, hipMemsetAsync((uint32_t*)c->d_ctrl.p + MQ_CTRL_UPDATES, 0, MQ_CTRL_GROUP * 4, nullptr));
    HIPCHK(c, hipDeviceSynchronize());
    return MQ_OK;
}

int mq_untile(mq_ctx* c, const void* gathered_dev, void* stream) {
    if (!c || !gathered_dev) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    if (!c->connected) return fail(c, MQ_ESTATE, "not connected");
    HIPCHK(c, hipSetDevice(c->device));
    int e = mq_launch_untile(gathered_dev, c->d_out[MQ_OUT_IRRADIANCE].p, c->W, c->H, c->tiles_x, c->tiles_x * c->tiles_y, (uint32_t)c->world, c->tiles_per_rank, (hipStream_t)stream);
    if (e) return fail(c, MQ_EHIP, std::string("untile launch: ") + hipGetErrorString((hipError_t)e));
    return MQ_OK;
}

int mq_untile_to(mq_ctx* c, const void* gathered_dev, void* image_dev, void* stream) {
    if (!c || !gathered_dev || !image_dev) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    if (!c->connected) return fail(c, MQ_ESTATE, "not connected");
    HIPCHK(c, hipSetDevice(c->device));
    int e = mq_launch_untile(gathered_dev, image_dev, c->W, c->H, c->tiles_x, c->tiles_x * c->tiles_y, (uint32_t)c->world, c->tiles_per_rank, (hipStream_t)stream);
    if (e) return fail(c, MQ_EHIP, std::string("untile launch: ") + hipGetErrorString((hipError_t)e));
    return MQ_OK;
}
int mq_untile_volume_depth(mq_ctx* c, const void* gathered_dev, void* stream) {
    if (!c || !gathered_dev) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    if (!c->connected) return fail(c, MQ_ESTATE, "not connected");
    HIPCHK(c, hipSetDevice(c->device));
    int e = mq_launch_untile16(gathered_dev, c->d_out[MQ_OUT_VOLUME_DEPTH].p, c->W, c->H, c->tiles_x, c->tiles_x * c->tiles_y, (uint32_t)c->world, c->tiles_per_rank, (hipStream_t)stream);
    if (e) return fail(c, MQ_EHIP, std::string("untile launch: ") + hipGetErrorString((hipError_t)e));
    return MQ_OK;
}
int mq_untile_volume(mq_ctx* c, const void* gathered_dev, void* stream) {
    if (!c || !gathered_dev) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context");
    if (!c->connected) return fail(c, MQ_ESTATE, "not connected");
    HIPCHK(c, hipSetDevice(c->device));
    int e = mq_launch_untile(gathered_dev, c->d_out[MQ_OUT_VOLUME].p, c->W, c->H, c->tiles_x, c->tiles_x * c->tiles_y, (uint32_t)c->world, c->tiles_per_rank, (hipStream_t)stream);
    if (e) return fail(c, MQ_EHIP, std::string("untile launch: ") + hipGetErrorString((hipError_t)e));
    return MQ_OK;
}

// ---- queries -----------------------------------------------------------------------------------
int mq_trace_rays(mq_ctx* c, const float* org, const float* dir, uint32_t n, uint32_t* prim, float* t, float* uv) {
    if (!c || !org || !dir || !prim || !t) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context: no HIP device");
    if (!c->committed) return fail(c, MQ_ESTATE, "scene not committed");
    if (n == 0) return MQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf d_o, d_d, d_p, d_t, d_uv, d_sp;
    int grid = std::min<int>((int)((n + 255) / 256), std::max(1, c->cu_count) * 8);
    int r = 0;
    if (!r) r = dev_upload(c, d_o, org, (size_t)n * 12);
    if (!r) r = dev_upload(c, d_d, dir, (size_t)n * 12);
    if (!r) r = dev_alloc(c, d_p, (size_t)n * 4);
    if (!r) r = dev_alloc(c, d_t, (size_t)n * 4);
    if (!r) r = dev_alloc(c, d_uv, (size_t)n * 8);
    if (!r) r = dev_alloc(c, d_sp, (size_t)grid * 256 * mq_spill_entries() * 8);
    if (!r) r = scene_ready(c, nullptr);
    if (!r) { int e = mq_launch_trace(c->scene, (const float*)d_o.p, (const float*)d_d.p, n, (uint32_t*)d_p.p, (float*)d_t.p, (float*)d_uv.p, (unsigned long long*)d_sp.p, grid, nullptr); if (e) r = fail(c, MQ_EHIP, std::string("trace launch: ") + hipGetErrorString((hipError_t)e)); }
    if (!r && hipDeviceSynchronize() != hipSuccess) r = fail(c, MQ_EHIP, "trace kernel failed");
    if (!r && hipMemcpy(prim, d_p.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) r = fail(c, MQ_EHIP, "copy back");
    if (!r && hipMemcpy(t, d_t.p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) r = fail(c, MQ_EHIP, "copy back");
    if (!r && uv && hipMemcpy(uv, d_uv.p, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) r = fail(c, MQ_EHIP, "copy back");
    dev_free(d_o); dev_free(d_d); dev_free(d_p); dev_free(d_t); dev_free(d_uv); dev_free(d_sp);
    return r;
}

static const int k_arity[19][2] = {{1, 1}, {1, 1}, {1, 2}, {2, 1}, {1, 1}, {3, 4}, {10, 5}, {6, 4}, {1, 4}, {4, 1}, {3, 3}, {9, 2}, {3, 3}, {11, 5}, {7, 4}, {7, 4}, {3, 4}, {7, 3}, {7, 4}};
int mq_math_eval(mq_ctx* c, int op, const float* in, float* out, uint32_t n) {
    if (!c || !in || !out || op < 0 || op >= 19) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context: no HIP device");
    if (n == 0) return MQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->params_dirty) props_to_params(c);
    int ni = k_arity[op][0], no = k_arity[op][1];
    DevBuf d_in, d_out; int r = 0;
    if (!r) r = dev_upload(c, d_in, in, (size_t)n * ni * 4);
    if (!r) r = dev_alloc(c, d_out, (size_t)n * no * 4);
    if (!r) r = scene_ready(c, nullptr);
    if (!r) { int e = mq_launch_math(c->scene, c->params, op, ni, no, (const float*)d_in.p, (float*)d_out.p, n, nullptr); if (e) r = fail(c, MQ_EHIP, std::string("math launch: ") + hipGetErrorString((hipError_t)e)); }
    if (!r && hipDeviceSynchronize() != hipSuccess) r = fail(c, MQ_EHIP, "math kernel failed");
    if (!r && hipMemcpy(out, d_out.p, (size_t)n * no * 4, hipMemcpyDeviceToHost) != hipSuccess) r = fail(c, MQ_EHIP, "copy back");
    dev_free(d_in); dev_free(d_out);
    return r;
}

int mq_measure_stream_read(mq_ctx* c, size_t bytes, int reps, double* gb_per_s) {
    if (!c || !gb_per_s || reps < 1 || bytes < (1u << 20)) return MQ_EINVAL;
    if (c->device < 0) return fail(c, MQ_ENODEVICE, "host-only context: no HIP device");
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf buf, sink; int r = dev_alloc(c, buf, bytes);
    if (!r) r = dev_alloc(c, sink, 16);
    if (r) { dev_free(buf); dev_free(sink); return r; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = hipMemset(buf.p, 0x5a, bytes) == hipSuccess && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    const int grid = std::max(1, c->cu_count) * 16;
    float best = 0.0f;
    for (int i = 0; ok && i < reps + 1; i++) { // first pass warms up
        ok = hipEventRecord(e0, nullptr) == hipSuccess && mq_launch_stream_read(buf.p, bytes, (uint32_t*)sink.p, grid, nullptr) == 0 &&
             hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
        float ms = 0.0f;
        if (ok) ok = hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
        if (ok && i > 0 && ms > 0.0f) best = std::max(best, (float)(bytes / (ms * 1e-3) / 1e9));
    }
    if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1);
    dev_free(buf); dev_free(sink);
    if (!ok) return fail(c, MQ_EHIP, "stream-read measurement failed");
    *gb_per_s = best;
    return MQ_OK;
}

// ---- scene sources -----------------------------------------------------------------------------
int mq_synth_scene(mq_ctx* c, const char* name, uint32_t seed) {
    if (!c || !name) return MQ_EINVAL;
    std::string err;
    if (!mq_synth_generate(c, name, seed, err)) return fail(c, MQ_EINVAL, err);
    c->params_dirty = true;
    return MQ_OK;
}

static void catmull(const std::vector<float>& p, float t, float out[3], float tan[3]) {
    size_t n = p.size() / 3;
    float ft = std::floor(t);
    long i1 = (long)ft; float f = t - ft;
    auto at = [&](long i, int k) { long m = (long)n; long j = ((i % m) + m) % m; return p[3 * (size_t)j + k]; };
    for (int k = 0; k < 3; k++) {
        float p0 = at(i1 - 1, k), p1 = at(i1, k), p2 = at(i1 + 1, k), p3 = at(i1 + 2, k);
        float a = -0.5f * p0 + 1.5f * p1 - 1.5f * p2 + 0.5f * p3, b = p0 - 2.5f * p1 + 2.0f * p2 - 0.5f * p3, cc = -0.5f * p0 + 0.5f * p2;
        out[k] = ((a * f + b) * f + cc) * f + p1;
        tan[k] = (3.0f * a * f + 2.0f * b) * f + cc;
    }
}

int mq_synth_camera(const mq_ctx* c, uint32_t frame, mq_uniform* u) {
    if (!c || !u) return MQ_EINVAL;
    if (!c->synth.valid) return MQ_ESTATE;
    memset(u, 0, sizeof *u);
    const MqSynthInfo& s = c->synth;
    auto cam = [&](uint32_t f, float* x, float* w, float* up) {
        float pos[3], tan[3];
        catmull(s.path, 0.37f + s.speed * (float)f, pos, tan);
        float l = std::sqrt(tan[0] * tan[0] + tan[1] * tan[1]);
        if (l < 1e-6f) { tan[0] = 1; tan[1] = 0; l = 1; }
        // look along the path, slightly upward, so ceilings / sky and floors are both in view
        float fw[3] = {tan[0] / l, tan[1] / l, 0.12f};
        float fl = std::sqrt(fw[0] * fw[0] + fw[1] * fw[1] + fw[2] * fw[2]);
        for (int k = 0; k < 3; k++) fw[k] /= fl;
        // AngleVectors-style basis: right = fw x world_up, up = right x fw
        float r[3] = {fw[1], -fw[0], 0.0f};
        float rl = std::sqrt(r[0] * r[0] + r[1] * r[1]);
        r[0] /= rl; r[1] /= rl;
        float uu[3] = {r[1] * fw[2] - r[2] * fw[1], r[2] * fw[0] - r[0] * fw[2], r[0] * fw[1] - r[1] * fw[0]};
        for (int k = 0; k < 3; k++) { x[k] = pos[k]; w[k] = fw[k]; up[k] = uu[k]; }
    };
    cam(frame, u->cam_x, u->cam_w, u->cam_u);
    cam(frame ? frame - 1 : 0, u->prev_cam_x, u->prev_cam_w, u->prev_cam_u);
    u->cam_x[3] = s.mu_t; u->cam_w[3] = 1.0f / 60.0f; u->cam_u[3] = 0.0f; // TIME_DIFF, quake_node.cpp:787-790
    u->prev_cam_x[3] = s.mu_s[0]; u->prev_cam_w[3] = s.mu_s[1]; u->prev_cam_u[3] = s.mu_s[2];
    u->sky_rt_bk = s.sky_rt_bk; u->sky_lf_ft = 0xffffu; u->sky_up_dn = 0xffffffffu; // classic sky marker, raytrace.glsl:35
    u->cl_time = (float)frame / 60.0f; u->frame = frame; u->player = 0; u->rt_config = 0;
    return MQ_OK;
}

int mq_load_bsp(mq_ctx* c, const char* bsp_path, const char* palette_path) {
    if (!c || !bsp_path) return MQ_EINVAL;
    std::string err;
    try { if (!mq_bsp_load(c, bsp_path, palette_path, err)) return fail(c, MQ_EIO, err); }
    catch (const std::exception& e) { return fail(c, MQ_EIO, std::string("map file: ") + e.what()); } // no C++ exception crosses the C ABI
    c->params_dirty = true;
    return MQ_OK;
}

} // extern "C"
