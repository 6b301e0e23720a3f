// mq_host.h -- host-side internals of libmqhip (scene container, property table, launch glue).
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>
#include <functional>

#include "mq_types.h"

struct MqHostGeo {
    std::vector<float> vtx, prev_vtx;
    std::vector<uint32_t> idx;
    std::vector<mq_ext> ext;
    uint32_t flags = 0;
    bool dynamic = false; // prev_vtx differs from vtx
    uint32_t n_tri() const { return (uint32_t)(idx.size() / 3); }
};
struct MqHostTex {
    uint32_t w = 0, h = 0, flags = 0;
    std::vector<uint8_t> px;
};

struct MqSynthInfo {
    bool valid = false;
    std::vector<float> path; // camera control points xyz
    float eye_height = 0.0f;
    float speed = 0.0f;      // control points per frame
    float mu_t = 0.0f;
    float mu_s[3] = {0, 0, 0};
    uint32_t sky_rt_bk = 18u | (19u << 16); // classic sky layers: back | front << 16
};

// The reference's property set (src/render_mcpg/render_mcpg.hpp:108-166 + gbuffer.hpp:75-77)
struct MqProps {
    bool randomize_seed = true;
    uint32_t seed = 0;
    bool reference_mode = false;
    float dir_guide_prior = 0.2f;
    int mc_samples = 5;
    float mc_samples_adaptive_prob = 0.7f;
    int mc_adaptive_grid_type = 0;
    uint32_t mc_adaptive_buffer_size = 32777259;
    float mc_adaptive_grid_tan_alpha_half = 0.003f;
    float mc_adaptive_grid_steps_per_unit_size = 6.0f;
    float mc_adaptive_grid_min_width = 0.01f;
    float mc_adaptive_grid_power = 4.0f;
    uint32_t mc_static_buffer_size = 800009;
    float mc_static_grid_width = 25.3f;
    bool mc_fast_recovery = true;
    int spp = 1;
    int max_path_length = 3;
    float surf_bsdf_p = 0.15f;
    int volume_spp = 0;
    int distance_mc_samples = 3;
    int distance_mc_grid_width = 25;
    uint32_t distance_mc_vertex_state_count = 10;
    float volume_particle_size_um = 25.0f;
    float dist_guide_p = 0.0f;
    float volume_phase_p = 0.3f;
    bool volume_forward_project = true;
    bool use_light_cache_tail = false;
    bool volume_use_light_cache = false;
    int lc_grid_type = 0;
    uint32_t lc_buffer_size = 4000000;
    float lc_grid_tan_alpha_half = 0.002f;
    float lc_grid_steps_per_unit_size = 6.0f;
    float lc_grid_min_width = 0.01f;
    float lc_grid_power = 2.0f;
    int debug_output_selector = 0;
    // gbuffer node
    bool hide_sun = true;
    bool enable_albedo_mipmap = true;
    bool enable_emission_mipmap = true;
    bool debug_output_connected = false; // the reference derives this from the graph wiring (render_mcpg.cpp:182-183)
    bool freeze_learning = false; // test hook, not a reference property
    bool log_learning = false;    // test hook, not a reference property
    int dyn_bvh = 2;                 // "per-frame BVH": who builds the tree of the per-frame geometry: 0 host (SAH on the worker pool), 1 device (mq_devbvh.hip), 2 auto (device from 12 288 per-frame triangles on; MQ_DEVBVH_AUTO_TRIS)
    bool lc_try_lock = false;        // the reference's light-cache try-lock (contended updates cancelled, light_cache.glsl:59-64) instead of the lock-free 8-byte publish
    bool lc_lock_statistics = false; // the reference's light-cache try-lock with per-cell counters + last_update_count per slot (for the state dumps)
    int overlap_camera_rays = 1;      // scheduling of this build: the camera rays of frame n + 1 traced beside kernels of frame n, on a low-priority stream
                                      // with its own hardware queue.  0 off; 2 always: from the start of frame n; 3 update pass: beside frame n's link /
                                      // apply kernels only; 4 last round: from frame n's last trace launch on; 5 last bounce: beside frame n's terminal bounce kernel and
                                      // its update pass; 1 auto: 4 for a rank of a partitioned frame, 5 for a full frame.  Measurements: mq_process in mq_api.cpp
    bool packet_camera_rays = false;  // scheduling of this build: camera rays as one frustum packet per 8x8 tile (bit-identical; measured slower than the per-lane walk)
    int pipelines = 1; // scheduling of this build, not a reference property: sub-pipelines per frame (mq_api.cpp mq_process)
    bool sequential_update_pass = false; // test hook: the update pass in the reference's dispatch order, one slot after the other
    // post chain: the "accum" and "volume accum" nodes of res/default_config.json (header defaults = its values)
    float accum_alpha = 0.951f, accum_max_history = INFINITY, accum_normal_threshold = 0.645771861076355f, accum_depth_threshold = 0.026403000578284264f;
    bool accum_enable_mv = true, accum_reuse_border = true;
    float vaccum_alpha = 0.902f, vaccum_max_history = INFINITY, vaccum_normal_threshold = 3.1415927410125732f, vaccum_depth_threshold = 0.28402701020240784f;
    bool vaccum_enable_mv = true, vaccum_reuse_border = true;
    // ReSTIR DI node, src/render_restir/renderer_restir.hpp:108-127 (normal thresholds as the angles the UI shows, renderer_restir.cpp:276-279,303-306)
    int restir_spp = 1; uint32_t restir_seed = 0; bool restir_randomize_seed = true;
    bool restir_temporal_reuse = false; float restir_temporal_normal_angle = 0.28379410920832787f /* acos(0.96) */, restir_temporal_depth = 0.1f;
    int restir_temporal_clamp_m = 32 * 20, restir_temporal_bias = 0; float restir_boiling = 0.0f; bool restir_apply_mv = false;
    int restir_spatial_iterations = 0; float restir_spatial_normal_angle = 0.28379410920832787f, restir_spatial_depth = 0.1f;
    int restir_spatial_radius = 30, restir_spatial_bias = 0; bool restir_shade_visibility = false;
    int band_reprojection_halo = 64; // rows of last frame's state a rank of a row partition holds beyond what the spatial radius needs (the reach of temporal reprojection)
    bool add_restir = false;         // the `add` node's input for the ReSTIR node: final += restir irradiance * albedo (config 5: "ReSTIR DI + MCPG GI combined")
    bool restir_inline_rays = false; // scheduling of this build: generate / shade rays traced inside the pass kernels (the first implementation) instead of as a wavefront through the MCPG node's queues
    // named quirk switches: on = what the reference's shaders compute, off = the evident intent
    bool quirk_lc_max_wo_p = true; // mcpg.comp:170 `max(wo_p, 10)`
    bool quirk_n16_wrap = true;    // mc.glsl:26 `N * N` on a uint16_t (grid.h:19): wraps, 0 at N = 256 / 512 / 768 / 1024
};

// f(begin, end) over [0, n) in chunks of about `grain`, on the builder's worker pool (mq_bvh.cpp); the caller takes part.  Chunks must be independent.
void mq_parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)>& f);
bool mq_build_cwbvh(const std::vector<MqTri>& tris, std::vector<MqNode>& out_nodes, std::vector<MqTri>& out_tris, std::vector<MqLeafRec>& out_leaves, float* sah_cost, std::string& err, uint32_t* depth_out = nullptr);

struct mq_ctx;
bool mq_synth_generate(mq_ctx* ctx, const char* name, uint32_t seed, std::string& err);
bool mq_bsp_load(mq_ctx* ctx, const char* bsp_path, const char* palette_path, std::string& err);

// Per-frame geometry producers (mq_producers.cpp) and what the loaders keep for them
struct MqAliasModel { // an MDL file as quakespasm's GL_MakeAliasModelDisplayLists_VBO lays it out
    float scale[3], scale_origin[3];
    uint32_t skinwidth = 0, skinheight = 0, numverts = 0, numposes = 0;
    std::vector<uint8_t> trivertexes;   // numposes * numverts * 4 bytes: v[3], lightnormalindex
    std::vector<uint16_t> vertindex;    // per VBO vertex: index into a pose's vertices
    std::vector<float> st;              // per VBO vertex: s, t in texels
    std::vector<uint16_t> indexes;      // 3 per triangle, into the VBO vertices
    std::vector<uint32_t> skin_texnum, skin_fb_texnum, skin_norm_texnum, skin_gloss_texnum; // per skin
};
struct MqSpriteFrame { float up, down, left, right, smax, tmax; uint32_t texnum; bool alpha; };
struct MqSpriteModel { int32_t type = 0; std::vector<MqSpriteFrame> frames; };
struct MqProducerState {
    std::vector<MqHostGeo> bsp_models; // brush models 1.. of the loaded BSP, in model space (index 0 unused: the world)
    std::vector<MqAliasModel> alias;
    std::vector<MqSpriteModel> sprites;
    MqHostGeo pending;                 // the per-frame geometry being collected (mq_dyn_begin .. mq_dyn_end)
    bool collecting = false;
};
MqProducerState& mq_ctx_producers(mq_ctx* ctx);
int mq_ctx_fail(mq_ctx* ctx, int code, const std::string& msg);
bool mq_read_palette(const char* palette_path, uint8_t pal[768], std::string& err); // 768-byte palette.lmp, or the grey ramp for NULL

// host-side access used by the generators / loaders
MqHostGeo& mq_ctx_geo(mq_ctx* ctx, int slot);
MqHostTex& mq_ctx_tex(mq_ctx* ctx, uint32_t texnum);
mq_constants& mq_ctx_constants(mq_ctx* ctx);
MqSynthInfo& mq_ctx_synth(mq_ctx* ctx);
void mq_ctx_clear_scene(mq_ctx* ctx);
