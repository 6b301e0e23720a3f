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
    // render_mcpg.cpp:419-578: the same visitor serves UI, JSON load and JSON store
    NodeStatusFlags properties(Properties& config) {
        bool reconnect = false;
        for (int i = 0; i < mq_property_count(); i++) {
            const std::string key = mq_property_name(i);
            double cur = 0;
            check(mq_get_property(ctx_, key.c_str(), &cur));
            double next = cur;
            if (key == "adaptive grid type" || key == "LC grid type") { int s = (int)cur; config.config_options(key, s, {"exponential", "quadratic"}); next = s; }
            else if (key == "debug output") { int s = (int)cur; config.config_options(key, s, {"light cache", "mc weight", "mc mean direction", "mc grid", "irradiance", "moments", "mc cos", "mc N", "mc motion vectors"}); next = s; }
            else if (is_bool(key)) { bool b = cur != 0; config.config_bool(key, b); next = b; }
            else if (is_uint(key)) { uint32_t u = (uint32_t)cur; config.config_uint(key, u); next = u; }
            else if (is_int(key)) { int32_t v = (int32_t)cur; config.config_int(key, v); next = v; }
            else { float f = (float)cur; config.config_float(key, f); next = f; }
            if (next != cur) { int r = mq_set_property(ctx_, key.c_str(), next); if (r < 0) check(r); reconnect |= r == 1; }
        }
        return reconnect ? NEEDS_RECONNECT : NONE;
    }

  private:
    static bool is_bool(const std::string& k) {
        for (const char* b : {"randomize seed", "reference mode", "mc fast recovery", "volume forward project", "surf: use LC", "volume: use LC", "hide sun",
                              "enable albedo mipmap", "enable emission mipmap", "quirk: LC max(wo_p,10)", "quirk: 16-bit N*N"}) if (k == b) return true;
        return false;
    }
    static bool is_uint(const std::string& k) {
        for (const char* b : {"seed", "adaptive grid buf size", "static grid buf size", "LC buf size", "dist mc states per vertex"}) if (k == b) return true;
        return false;
    }
    static bool is_int(const std::string& k) {
        for (const char* b : {"mc samples", "spp", "max path length", "volume spp", "dist mc samples", "dist mc grid width"}) if (k == b) return true;
        return false;
    }
    void check(int r) { if (r < 0) throw Error(r, mq_last_error(ctx_)); }
    mq_ctx* ctx_ = nullptr;
    uint32_t width_ = 0, height_ = 0;
    bool scene_dirty_ = false;
};

} // namespace mq
