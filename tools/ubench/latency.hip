// Micro-benchmark: latency of ONE dependent step of a per-lane gather chain on gfx950 when the chip is nearly empty --
// the regime of a rank of an 8-way tile partition (every launch is a drain: its time is the longest chain, not throughput).
//   hipcc -O3 --offload-arch=gfx950 -o latency latency.hip && ./latency
// Each lane: idx -> LOADS x 16 bytes of record idx (divergent: every lane its own record) -> VALU dependent ops -> next idx.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LOADS, int VALU>
__global__ __launch_bounds__(64) void chain(const uint4* __restrict__ tab, uint32_t n_rec, int iters, uint32_t* out) {
    uint32_t idx = (blockIdx.x * 64u + threadIdx.x) * 2654435761u % n_rec;
    float acc = 1.0f;
    for (int i = 0; i < iters; i++) {
        const uint4* p = tab + (size_t)idx * 5;
        uint32_t h = 0;
#pragma unroll
        for (int k = 0; k < LOADS; k++) { uint4 v = p[k]; h ^= v.x + v.y + v.z + v.w; }
        float x = __uint_as_float((h & 0x007fffffu) | 0x3f800000u);
#pragma unroll
        for (int k = 0; k < VALU; k++) x = __builtin_fmaf(x, 0.999f, acc); // dependent chain: one issue slot each
        acc = x * 1e-9f;
        idx = (h ^ (idx * 747796405u + 2891336453u) ^ (uint32_t)(acc > 2.0f)) % n_rec;
    }
    if (acc == 12345.0f) out[0] = idx;
}

template <int LOADS, int VALU>
void run(const uint4* tab, size_t mb, int waves, uint32_t* out) {
    const uint32_t n_rec = (uint32_t)(mb * 1000000 / 80);
    const int iters = 300;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    chain<LOADS, VALU><<<waves, 64>>>(tab, n_rec, 30, out);
    chain<LOADS, VALU><<<waves, 64>>>(tab, n_rec, 10, out); // launch + 10 steps
    CHECK(hipEventRecord(a));
    chain<LOADS, VALU><<<waves, 64>>>(tab, n_rec, 10 + iters, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    hipEvent_t c, d; CHECK(hipEventCreate(&c)); CHECK(hipEventCreate(&d));
    CHECK(hipEventRecord(c));
    chain<LOADS, VALU><<<waves, 64>>>(tab, n_rec, 10, out);
    CHECK(hipEventRecord(d)); CHECK(hipEventSynchronize(d));
    float ms0; CHECK(hipEventElapsedTime(&ms0, c, d));
    printf("table %6zu MB  %5d waves  %d x 16 B + %3d dependent VALU: %7.1f ns per step  (launch + 10 steps: %.1f us)\n", mb, waves, LOADS, VALU, (ms - ms0) * 1e6 / iters, ms0 * 1e3);
}

int main() {
    const size_t bytes = 2200ull << 20;
    uint4* tab; uint32_t* out;
    CHECK(hipMalloc(&tab, bytes)); CHECK(hipMalloc(&out, 4));
    {
        std::vector<uint32_t> h(64u << 20);
        uint32_t s = 12345; for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s; }
        for (size_t off = 0; off < bytes; off += h.size() * 4) CHECK(hipMemcpy((char*)tab + off, h.data(), std::min(h.size() * 4, bytes - off), hipMemcpyHostToDevice));
    }
    for (size_t mb : {2, 8, 40, 200, 2000})
        for (int waves : {256, 4096}) {
            run<5, 0>(tab, mb, waves, out);
            run<5, 300>(tab, mb, waves, out);
        }
    run<1, 0>(tab, 40, 256, out); run<1, 0>(tab, 2000, 256, out);
    run<8, 0>(tab, 40, 4096, out);
    return 0;
}
