// Host-only exercise of include/mq_node.hpp: lifecycle order, property visitor (a JSON-load style
// visitor that overrides some values), NEEDS_RECONNECT, and the loud failure without a HIP device.
#include <cstdio>
#include <cstring>
#include <map>
#include "mq_node.hpp"

struct Loader : mq::Properties { // behaves like merian's JSONLoadProperties: overrides what it knows
    std::map<std::string, double> vals; std::map<std::string, std::string> opts; int visited = 0;
    bool config_bool(const std::string& id, bool& v) override { visited++; auto it = vals.find(id); if (it == vals.end()) return false; v = it->second != 0; return true; }
    bool config_int(const std::string& id, int32_t& v) override { visited++; auto it = vals.find(id); if (it == vals.end()) return false; v = (int32_t)it->second; return true; }
    bool config_uint(const std::string& id, uint32_t& v) override { visited++; auto it = vals.find(id); if (it == vals.end()) return false; v = (uint32_t)it->second; return true; }
    bool config_float(const std::string& id, float& v) override { visited++; auto it = vals.find(id); if (it == vals.end()) return false; v = (float)it->second; return true; }
    bool config_options(const std::string& id, int& sel, const std::vector<std::string>& o) override {
        visited++; auto it = opts.find(id); if (it == opts.end()) return false;
        for (size_t i = 0; i < o.size(); i++) if (o[i] == it->second) { sel = (int)i; return true; }
        return false;
    }
};
#define REQUIRE(c) do { if (!(c)) { printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
    mq::RendererMarkovChainHIP node(-1); // host-only: properties, describe, scene; no device
    REQUIRE(node.describe_inputs().size() == 8);
    auto outs = node.describe_outputs(1920, 1080);
    REQUIRE(outs[0].name == "irradiance" && outs[0].bytes == 1920u * 1080u * 16u);
    auto find = [&](const char* n) -> const mq::ConnectorDesc* { for (auto& o : outs) if (o.name == n) return &o; return nullptr; };
    REQUIRE(find("hits") && find("hits")->bytes == 1920u * 1080u * 40u);
    REQUIRE(find("volume") && find("volume")->bytes == 1920u * 1080u * 16u);   // render_mcpg.cpp:44-52: volume, volume_depth, volume_mv, debug
    REQUIRE(find("volume_depth") && find("volume_depth")->bytes == 1920u * 1080u * 2u);
    REQUIRE(find("volume_mv") && find("debug") && find("debug")->bytes == 1920u * 1080u * 8u);
    REQUIRE(find("markovchain") && find("lightcache") && find("volume_distancemc") && find("update_buffer"));
    Loader l;
    l.vals["spp"] = 2; l.vals["BSDF Prob"] = 0.1; l.vals["reference mode"] = 0; l.vals["seed"] = 1234; l.vals["randomize seed"] = 0;
    REQUIRE(node.properties(l) == mq::NONE);          // only pipeline-refresh class changes
    int own = 0; // the fused node's own keys: those without a "restir: " / "accum: " / "volume accum: " prefix
    for (int i = 0; i < mq_property_count(); i++) { std::string k = mq_property_name(i); if (k.rfind("restir: ", 0) && k.rfind("accum: ", 0) && k.rfind("volume accum: ", 0)) own++; }
    REQUIRE(l.visited == own);
    double v = 0; mq_get_property(node.handle(), "spp", &v); REQUIRE(v == 2);
    mq_get_property(node.handle(), "BSDF Prob", &v); REQUIRE(v > 0.0999 && v < 0.1001);
    Loader l2; l2.opts["LC grid type"] = "quadratic"; l2.vals["LC buf size"] = 4000037;
    REQUIRE(node.properties(l2) == mq::NEEDS_RECONNECT);  // render_mcpg.cpp:567-575
    REQUIRE(node.properties(l2) == mq::NONE);              // idempotent
    // a tiny scene can be handed over and committed on the host ...
    float vtx[9] = {0, 0, 0, 1, 0, 0, 0, 1, 0}; uint32_t idx[3] = {0, 2, 1}; mq_ext ext; memset(&ext, 0, sizeof ext);
    node.set_geometry(0, vtx, nullptr, 3, idx, &ext, 1, MQ_GEO_OPAQUE);
    // ... but connecting / processing without a HIP device fails loudly (no CPU fallback)
    bool threw = false;
    try { node.on_connected(); } catch (const mq::Error& e) { threw = e.code == MQ_ENODEVICE; }
    REQUIRE(threw);
    threw = false;
    mq::RenderInfo info;
    try { node.process(info, nullptr); } catch (const mq::Error& e) { threw = e.code == MQ_ENODEVICE; }
    REQUIRE(threw);
    { // the per-frame uniform producer through the adapter
        mq::RenderInfo ri; memset(&ri.uniform, 0, sizeof ri.uniform);
        mq_frame_state fs; memset(&fs, 0, sizeof fs);
        fs.vieworg[0] = 10; fs.viewangles[1] = 90; fs.cl_time = 2.5; fs.frame = 7; fs.render = 1; fs.has_player = 1; fs.weapon = 1; fs.waterlevel = 3;
        fs.sky_mode = 2; fs.sky[0] = 11; fs.sky[1] = 12; fs.notexture = 5; fs.mu_overwrite = 1; fs.mu_t = 0.01f; fs.mu_s_div_mu_t[0] = 0.5f;
        node.update_uniform(ri, fs);
        REQUIRE(ri.render && ri.uniform.frame == 7 && ri.uniform.player == 3u && ri.uniform.cam_x[0] == 10.0f && ri.uniform.cam_x[3] == 0.01f);
        REQUIRE(ri.uniform.sky_rt_bk == (11u | (12u << 16)) && ri.uniform.sky_lf_ft == (0xffffu | (5u << 16)) && ri.uniform.cam_w[3] == 2.5f);
        REQUIRE(ri.uniform.cam_w[1] > 0.999f && ri.uniform.prev_cam_x[3] == 0.005f);
        fs.cl_time = 2.5; node.update_uniform(ri, fs); // the clock stands: time difference 1, previous camera = last camera
        REQUIRE(ri.uniform.cam_w[3] == 1.0f && ri.uniform.prev_cam_x[0] == 10.0f);
    }
    // the ReSTIR node and the post chain on the same context: the reference's own key strings, prefix stripped
    mq::RendererRESTIRHIP restir(node);
    REQUIRE(restir.describe_inputs().size() == 13);
    auto routs = restir.describe_outputs(1920, 1080);
    REQUIRE(routs.size() == 3 && routs[2].name == "reservoirs" && routs[2].bytes == 1920u * 1080u * 64u);
    Loader l3; l3.vals["spp"] = 3; l3.vals["enable temporal reuse"] = 1; l3.opts["temporal bias correction"] = "raytraced"; l3.vals["spatital radius"] = 12;
    REQUIRE(restir.properties(l3) == mq::NONE);
    REQUIRE(l3.visited == 16);
    mq_get_property(node.handle(), "restir: spp", &v); REQUIRE(v == 3);
    mq_get_property(node.handle(), "restir: temporal bias correction", &v); REQUIRE(v == 2);
    mq_get_property(node.handle(), "spp", &v); REQUIRE(v == 2); // the MCPG node's "spp" is another property
    mq::PostChainHIP post(node);
    Loader l4; l4.vals["alpha"] = 0.5;
    REQUIRE(post.properties_accum(l4) == mq::NONE && l4.visited == 6);
    mq_get_property(node.handle(), "accum: alpha", &v); REQUIRE(v == 0.5);
    mq_get_property(node.handle(), "volume accum: alpha", &v); REQUIRE(v > 0.9 && v < 0.91);
    threw = false;
    try { restir.process(info, nullptr); } catch (const mq::Error& e) { threw = e.code == MQ_ENODEVICE; }
    REQUIRE(threw);
    printf("node adapter ok\n");
    return 0;
}
