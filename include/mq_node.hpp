// mq_node.hpp -- header-only C++ adapter: the five-method merian render-node lifecycle
// (describe_inputs / describe_outputs / on_connected / process / properties) implemented purely on
// the C ABI of mq.h.  It mirrors, method for method,
//   RendererMarkovChain  src/render_mcpg/render_mcpg.hpp:36-49, render_mcpg.cpp:27-115,117-320,419-578
//   GBuffer              src/gbuffer/gbuffer.hpp:27-40, gbuffer.cpp:23-128
// so that a merian `Node` subclass only forwards.  merian itself is not part
// of this repository; the small `Properties` visitor below has the same calls the reference uses
// (config_bool / config_int / config_uint / config_float / config_percent / config_options).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "mq.h"

namespace mq {

struct Error : std::runtime_error { // the reference reports failures as C++ exceptions (quake_node.cpp:453,549)
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// Visitor with the subset of merian::Properties the two nodes call (render_mcpg.cpp:449-548).
struct Properties {
    virtual ~Properties() = default;
    virtual bool config_bool(const std::string& id, bool& v) = 0;
    virtual bool config_int(const std::string& id, int32_t& v) = 0;
    virtual bool config_uint(const std::string& id, uint32_t& v) = 0;
    virtual bool config_float(const std::string& id, float& v) = 0;
    virtual bool config_percent(const std::string& id, float& v) { return config_float(id, v); }
    virtual bool config_options(const std::string& id, int& selected, const std::vector<std::string>& options) = 0;
};

enum NodeStatusFlags : uint32_t { NONE = 0, NEEDS_RECONNECT = 1 }; // render_mcpg.cpp:574

struct ConnectorDesc { std::string name; std::string format; size_t bytes; };

// QuakeNode::QuakeRenderInfo, src/game/quake_node.hpp:70-84
struct RenderInfo {
    mq_uniform uniform{};
    mq_constants constant{};
    bool render = true;
    bool constant_data_update = true;
};

class RendererMarkovChainHIP {
  public:
    explicit RendererMarkovChainHIP(int hip_device) {
        int r = mq_create(&ctx_, hip_device);
        if (r != MQ_OK) throw Error(r, "mq_create failed");
    }
    ~RendererMarkovChainHIP() { mq_destroy(ctx_); }
    RendererMarkovChainHIP(const RendererMarkovChainHIP&) = delete;
    RendererMarkovChainHIP& operator=(const RendererMarkovChainHIP&) = delete;

    mq_ctx* handle() { return ctx_; }

    // render_mcpg.cpp:27-34 + gbuffer.cpp:23-25: the graph inputs the fused node consumes
    std::vector<std::string> describe_inputs() const {
        return {"vtx", "prev_vtx", "idx", "ext", "textures", "tlas", "resolution", "render_info"};
    }
    // render_mcpg.cpp:36-103 + gbuffer.cpp:27-44: same names and pixel formats
    std::vector<ConnectorDesc> describe_outputs(uint32_t width, uint32_t height) {
        mq_io_desc d;
        check(mq_describe(ctx_, width, height, &d));
        width_ = width; height_ = height;
        return {{"irradiance", "R32G32B32A32Sfloat", d.bytes[MQ_OUT_IRRADIANCE]},
                {"volume", "R32G32B32A32Sfloat", d.bytes[MQ_OUT_VOLUME]},
                {"volume_depth", "R16Sfloat", d.bytes[MQ_OUT_VOLUME_DEPTH]},
                {"volume_mv", "R16G16Sfloat", d.bytes[MQ_OUT_VOLUME_MV]},
                {"debug", "R16G16B16A16Sfloat", d.bytes[MQ_OUT_DEBUG]},
                {"albedo", "R16G16B16A16Sfloat", d.bytes[MQ_OUT_GB_ALBEDO]},
                {"gbuffer.irradiance", "R16G16B16A16Sfloat", d.bytes[MQ_OUT_GB_IRRADIANCE]},
                {"mv", "R16G16Sfloat", d.bytes[MQ_OUT_GB_MV]},
                {"gbuffer", "GBuffer16B", d.bytes[MQ_OUT_GBUFFER]},
                {"hits", "CompressedHit40B", d.bytes[MQ_OUT_HITS]},
                {"markovchain", "buffer", d.state_bytes_markovchain},
                {"lightcache", "buffer", d.state_bytes_lightcache},
                {"volume_distancemc", "buffer", d.state_bytes_volume_distancemc},
                {"update_buffer", "buffer", d.state_bytes_update_queue}};
    }
    // render_mcpg.cpp:105-115: allocate persistent state, invalidate pipelines
    NodeStatusFlags on_connected() {
        check(mq_connect(ctx_, width_, height_));
        return NONE;
    }
    // scene inputs arrive as plain arrays (the reference binds them as descriptor arrays, render_mcpg.hpp:61-70)
    void set_geometry(int slot, const float* vtx, const float* prev_vtx, uint32_t n_vtx, const uint32_t* idx, const mq_ext* ext, uint32_t n_tri, uint32_t flags) {
        check(mq_scene_set_geometry(ctx_, slot, vtx, prev_vtx, n_vtx, idx, ext, n_tri, flags));
        scene_dirty_ = true;
    }
    void set_texture(uint32_t texnum, uint32_t w, uint32_t h, const uint8_t* rgba8, uint32_t flags) {
        check(mq_scene_set_texture(ctx_, texnum, w, h, rgba8, flags));
        scene_dirty_ = true;
    }
    // render_mcpg.cpp:117-320 / gbuffer.cpp:68-128: record this frame's work on `stream`
    void process(const RenderInfo& info, void* hip_stream) {
        if (info.constant_data_update) check(mq_set_constants(ctx_, &info.constant)); // render_mcpg.cpp:125
        if (scene_dirty_) { check(mq_scene_commit(ctx_)); scene_dirty_ = false; }
        check(mq_process(ctx_, &info.uniform, info.render ? 1 : 0, hip_stream));
    }
    void* output(int which, size_t* bytes = nullptr) {
        void* p = nullptr;
        check(mq_map_output(ctx_, which, &p, bytes));
        return p;
    }
    // render_mcpg.cpp:419-578: the same visitor serves UI, JSON load and JSON store.  The keys of this node are those without
    // a node prefix; the ReSTIR node's and the post chain's live under "restir: ", "accum: ", "volume accum: " (adapters below).
    NodeStatusFlags properties(Properties& config) { return visit_properties(config, ""); }

    // ---- per-frame geometry (QuakeNode::update_dynamic_geo, src/game/quake_node.cpp:896-983): see mq.h "producers" ----
    void dyn_begin() { check(mq_dyn_begin(ctx_)); }
    void dyn_add_particles(const mq_particle* p, uint32_t n, const mq_view& view, uint32_t texnum_blood, uint32_t texnum_explosion, double cl_time, double prev_cl_time) {
        check(mq_dyn_add_particles(ctx_, p, n, &view, texnum_blood, texnum_explosion, cl_time, prev_cl_time));
    }
    void dyn_add_alias(int model, const mq_alias_instance& inst) { check(mq_dyn_add_alias(ctx_, model, &inst)); }
    void dyn_add_alias_batch(const int* models, const mq_alias_instance* insts, uint32_t n) { check(mq_dyn_add_alias_batch(ctx_, models, insts, n)); } // the visible entities at once, on the worker pool
    void dyn_add_sprite(int model, const mq_sprite_instance& inst, const mq_view& view) { check(mq_dyn_add_sprite(ctx_, model, &inst, &view)); }
    void dyn_add_brush_model(int model, const float origin[3], const float angles[3], const float prev_origin[3], const float prev_angles[3]) {
        check(mq_dyn_add_brush_model(ctx_, model, origin, angles, prev_origin, prev_angles));
    }
    void dyn_end(int slot) { check(mq_dyn_end(ctx_, slot)); scene_dirty_ = true; }
    // the Quake node's per-frame uniform (QuakeNode::process, quake_node.cpp:768-824): `info.uniform` holds the previous frame's
    // on entry and this frame's on return; `info.render` follows the frame state
    void update_uniform(RenderInfo& info, const mq_frame_state& frame) { check(mq_uniform_update(&info.uniform, &frame)); info.render = frame.render != 0; }

    // shared by the adapters of the other nodes that live on this context
    NodeStatusFlags visit_properties(Properties& config, const std::string& prefix) {
        bool reconnect = false;
        for (int i = 0; i < mq_property_count(); i++) {
            const std::string full = mq_property_name(i);
            const bool prefixed = full.rfind("restir: ", 0) == 0 || full.rfind("accum: ", 0) == 0 || full.rfind("volume accum: ", 0) == 0;
            if (prefix.empty() ? prefixed : full.rfind(prefix, 0) != 0) continue;
            const std::string key = full.substr(prefix.size()); // the reference's own key string
            double cur = 0;
            check(mq_get_property(ctx_, full.c_str(), &cur));
            double next = cur;
            switch (mq_property_type(i)) {
            case MQ_PROP_BOOL: { bool b = cur != 0; config.config_bool(key, b); next = b; break; }
            case MQ_PROP_INT: { int32_t v = (int32_t)cur; config.config_int(key, v); next = v; break; }
            case MQ_PROP_UINT: { uint32_t u = (uint32_t)cur; config.config_uint(key, u); next = u; break; }
            case MQ_PROP_OPTION: {
                std::vector<std::string> options;
                for (int k = 0; mq_property_option(i, k); k++) options.push_back(mq_property_option(i, k));
                int sel = (int)cur; config.config_options(key, sel, options); next = sel; break; }
            default: { float f = (float)cur; config.config_float(key, f); next = f; break; }
            }
            if (next != cur) { int r = mq_set_property(ctx_, full.c_str(), next); if (r < 0) check(r); reconnect |= r == 1; }
        }
        return reconnect ? NEEDS_RECONNECT : NONE;
    }
    void check_public(int r) { check(r); }

  private:
    void check(int r) { if (r < 0) throw Error(r, mq_last_error(ctx_)); }
    mq_ctx* ctx_ = nullptr;
    uint32_t width_ = 0, height_ = 0;
    bool scene_dirty_ = false;
};

// "Renderer (ReSTIR)": RendererRESTIR, src/render_restir/renderer_restir.hpp:35-49, renderer_restir.cpp:56-325.  It consumes the
// g-buffer node's outputs, which live on the context of the fused GBuffer + MCPG node: the adapter shares that context.
class RendererRESTIRHIP {
  public:
    explicit RendererRESTIRHIP(RendererMarkovChainHIP& gbuffer_node) : g_(gbuffer_node) {}
    std::vector<std::string> describe_inputs() const { // renderer_restir.cpp:56-62
        return {"vtx", "prev_vtx", "idx", "ext", "gbuffer", "prev_gbuffer", "hits", "textures", "tlas", "resolution", "render_info", "reservoirs", "mv"};
    }
    std::vector<ConnectorDesc> describe_outputs(uint32_t width, uint32_t height) { // renderer_restir.cpp:64-90
        mq_io_desc d;
        g_.check_public(mq_describe(g_.handle(), width, height, &d));
        return {{"irradiance", "R32G32B32A32Sfloat", d.bytes[MQ_OUT_RESTIR_IRRADIANCE]}, {"moments", "R32G32Sfloat", d.bytes[MQ_OUT_RESTIR_MOMENTS]},
                {"reservoirs", "ReSTIRDIReservoir64B", d.bytes[MQ_OUT_RESTIR_RESERVOIRS]}};
    }
    NodeStatusFlags on_connected() { return NONE; } // the shared context allocates the node's buffers with its own (renderer_restir.cpp:92-107)
    void process(const RenderInfo& info, void* hip_stream) { g_.check_public(mq_restir_process(g_.handle(), &info.uniform, info.render ? 1 : 0, hip_stream)); } // :109-251, after the g-buffer node's process
    NodeStatusFlags properties(Properties& config) { return g_.visit_properties(config, "restir: "); }                                                          // :253-325
  private:
    RendererMarkovChainHIP& g_;
};

// The graph's "accum" / "volume accum" (merian Accumulate) nodes, the denoiser's albedo re-modulation and "add"
// (res/default_config.json:21-133,404-435,473-497) as one post step on the same context: mq_post_process.
class PostChainHIP {
  public:
    explicit PostChainHIP(RendererMarkovChainHIP& renderer) : g_(renderer) {}
    std::vector<ConnectorDesc> describe_outputs(uint32_t width, uint32_t height) {
        mq_io_desc d;
        g_.check_public(mq_describe(g_.handle(), width, height, &d));
        return {{"accum.out", "R32G32B32A32Sfloat", d.bytes[MQ_OUT_ACCUM]}, {"accum.history", "R32Sfloat", d.bytes[MQ_OUT_ACCUM_HISTORY]},
                {"volume accum.out", "R32G32B32A32Sfloat", d.bytes[MQ_OUT_VOLUME_ACCUM]}, {"volume accum.history", "R32Sfloat", d.bytes[MQ_OUT_VOLUME_ACCUM_HISTORY]},
                {"add.out", "R32G32B32A32Sfloat", d.bytes[MQ_OUT_FINAL]}};
    }
    void process(void* hip_stream) { g_.check_public(mq_post_process(g_.handle(), hip_stream)); }
    void clear() { g_.check_public(mq_post_clear(g_.handle())); } // the nodes' "clear event pattern"
    NodeStatusFlags properties_accum(Properties& config) { return g_.visit_properties(config, "accum: "); }
    NodeStatusFlags properties_volume_accum(Properties& config) { return g_.visit_properties(config, "volume accum: "); }
  private:
    RendererMarkovChainHIP& g_;
};

} // namespace mq
