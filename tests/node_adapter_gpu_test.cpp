// include/mq_node.hpp ON THE DEVICE: the adapter classes a merian maintainer would subclass drive real frames --
// describe_outputs / properties / on_connected / process of the fused GBuffer + MCPG node, then the ReSTIR node and the
// post chain on the same context -- and the outputs are written to files for tests/test_gpu_multi.py to compare with the
// frames the ctypes binding renders from the same inputs.   usage: node_adapter_gpu_test <out dir>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <hip/hip_runtime_api.h>
#include "mq_node.hpp"

struct Loader : mq::Properties {
    std::map<std::string, double> vals;
    bool config_bool(const std::string& id, bool& v) override { auto it = vals.find(id); if (it == vals.end()) return false; v = it->second != 0; return true; }
    bool config_int(const std::string& id, int32_t& v) override { auto it = vals.find(id); if (it == vals.end()) return false; v = (int32_t)it->second; return true; }
    bool config_uint(const std::string& id, uint32_t& v) override { auto it = vals.find(id); if (it == vals.end()) return false; v = (uint32_t)it->second; return true; }
    bool config_float(const std::string& id, float& v) override { auto it = vals.find(id); if (it == vals.end()) return false; v = (float)it->second; return true; }
    bool config_options(const std::string&, int&, const std::vector<std::string>&) override { return false; }
};
static bool dump(const std::string& path, const void* dev, size_t bytes) {
    std::vector<char> host(bytes);
    if (hipMemcpy(host.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
    FILE* f = fopen(path.c_str(), "wb"); if (!f) return false;
    const bool ok = fwrite(host.data(), 1, bytes, f) == bytes; fclose(f);
    return ok;
}
#define REQUIRE(c) do { if (!(c)) { printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: %s <out dir>\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    const uint32_t W = 64, H = 48;
    try {
        mq::RendererMarkovChainHIP node(0);
        REQUIRE(mq_synth_scene(node.handle(), "synth_tiny", 3) == MQ_OK); // the scene source of the tests (a merian graph would call set_geometry / set_texture)
        Loader l;
        l.vals = {{"reference mode", 1}, {"randomize seed", 0}, {"seed", 0x5EED}, {"spp", 2}, {"max path length", 3},
                  {"adaptive grid buf size", 1 << 16}, {"static grid buf size", 1 << 12}, {"LC buf size", 1 << 14}};
        REQUIRE(node.properties(l) == mq::NEEDS_RECONNECT); // the table sizes changed (render_mcpg.cpp:567-575)
        mq::RendererRESTIRHIP restir(node);
        Loader lr; lr.vals = {{"randomize seed", 0}, {"seed", 77}, {"spp", 2}, {"enable temporal reuse", 1}, {"spatial reuse iterations", 2}};
        REQUIRE(restir.properties(lr) == mq::NONE);
        mq::PostChainHIP post(node);
        auto outs = node.describe_outputs(W, H);
        REQUIRE(outs[0].name == "irradiance" && outs[0].bytes == (size_t)W * H * 16);
        REQUIRE(node.on_connected() == mq::NONE);
        mq::RenderInfo info;
        REQUIRE(mq_get_constants(node.handle(), &info.constant) == MQ_OK); // what the synthetic scene set (sun, fov); a Quake node fills this
        REQUIRE(mq_scene_commit(node.handle()) == MQ_OK);
        for (uint32_t f = 0; f < 3; f++) {
            REQUIRE(mq_synth_camera(node.handle(), 10 + f, &info.uniform) == MQ_OK);
            info.constant_data_update = f == 0;
            node.process(info, nullptr);
            restir.process(info, nullptr);
            post.process(nullptr);
        }
        REQUIRE(mq_sync(node.handle()) == MQ_OK);
        size_t bytes = 0;
        const void* p = node.output(MQ_OUT_IRRADIANCE, &bytes);
        REQUIRE(bytes == (size_t)W * H * 16 && dump(dir + "/irradiance.bin", p, bytes));
        p = node.output(MQ_OUT_HITS, &bytes); REQUIRE(dump(dir + "/hits.bin", p, bytes));
        p = node.output(MQ_OUT_RESTIR_IRRADIANCE, &bytes); REQUIRE(dump(dir + "/restir_irradiance.bin", p, bytes));
        p = node.output(MQ_OUT_FINAL, &bytes); REQUIRE(dump(dir + "/final.bin", p, bytes));
    } catch (const mq::Error& e) { printf("FAILED: mq::Error %d: %s\n", e.code, e.what()); return 1; }
    printf("node adapter gpu ok\n");
    return 0;
}
