// ThreadSanitizer driver for the host side's worker pool (mq_bvh.cpp: TaskPool, mq_parallel_for): a host-only context builds a
// 1.2 M-triangle scene, then produces and commits 20 000 particles five times.  tools/run_tsan.sh builds the host objects with
// -fsanitize=thread, links this file against them and runs it: the expected output is "ok: ..." and no ThreadSanitizer report.
#include "mq.h"
#include <cstdio>
#include <cstring>
#include <vector>
int main() {
    mq_ctx* c = nullptr;
    if (mq_create(&c, -1)) { printf("create failed\n"); return 1; }
    mq_synth_scene(c, "synth_azad", 4);
    if (mq_scene_commit(c)) { printf("commit failed: %s\n", mq_last_error(c)); return 1; }
    mq_view v; memset(&v, 0, sizeof v); v.forward[0] = 1; v.right[1] = -1; v.up[2] = 1;
    std::vector<mq_particle> p(20000);
    for (size_t i = 0; i < p.size(); i++) {
        memset(&p[i], 0, sizeof p[i]);
        p[i].org[0] = (float)(i % 100) * 3; p[i].org[1] = (float)(i / 100) * 3; p[i].org[2] = (float)(i % 7) * 11;
        p[i].prev_org[0] = p[i].org[0] + 1; p[i].prev_org[1] = p[i].org[1]; p[i].prev_org[2] = p[i].org[2];
        p[i].seed = (uint32_t)i + 1; p[i].color_rgba = 0xffffff;
    }
    for (int f = 0; f < 5; f++) {
        mq_dyn_begin(c); mq_dyn_add_particles(c, p.data(), (uint32_t)p.size() - 1000 * f, &v, 1, 2, f / 60.0, (f - 1) / 60.0); mq_dyn_end(c, 2);
        if (mq_scene_commit(c)) { printf("commit failed: %s\n", mq_last_error(c)); return 1; }
    }
    uint64_t nt = 0, nn = 0, bb = 0; float sah = 0;
    mq_scene_stats(c, &nt, &nn, &bb, &sah);
    printf("ok: %llu tris, %llu nodes\n", (unsigned long long)nt, (unsigned long long)nn);
    mq_destroy(c);
    return 0;
}
