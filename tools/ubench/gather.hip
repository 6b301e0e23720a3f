// Micro-benchmark: per-lane gathers of small records (the access pattern of BVH traversal) on gfx950.
//   hipcc -O3 --offload-arch=gfx950 -o gather gather.hip && ./gather
// Each lane walks a dependent chain: idx -> load REC_LOADS x 16 bytes of record idx -> next idx from
// the loaded data.  Reports time per wave-level load instruction and bytes/s for several active-lane
// counts, record sizes and occupancies.  Used to decide how the traversal loop should issue loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// SHARE: lanes per shared record (1 = every lane its own record, 64 = the whole wave reads one record
// through vector loads); SCALAR: wave-uniform record fetched with scalar loads (s_load) instead.
template <int LOADS, int STRIDE16, int SHARE = 1, bool SCALAR = false>
__global__ __launch_bounds__(256) void walk(const uint4* __restrict__ tab, uint32_t n_rec, int iters, int active, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    uint32_t idx = ((blockIdx.x * 256u + threadIdx.x) / SHARE) * 2654435761u % n_rec;
    if (SCALAR) {
        uint32_t acc = 0, ui = __builtin_amdgcn_readfirstlane(idx);
        for (int i = 0; i < iters; i++) {
            const uint4* p = tab + (size_t)ui * STRIDE16;
            uint32_t h = 0;
#pragma unroll
            for (int k = 0; k < LOADS; k++) { uint4 v = p[k]; h ^= v.x + v.y + v.z + v.w; }
            acc += h + lane;
            ui = (h ^ (ui * 747796405u + 2891336453u)) % n_rec;
        }
        if (acc == 0x12345678u) out[0] = acc;
        return;
    }
    uint32_t acc = 0;
    if (lane < active) {
        for (int i = 0; i < iters; i++) {
            const uint4* p = tab + (size_t)idx * STRIDE16;
            uint4 v[LOADS];
#pragma unroll
            for (int k = 0; k < LOADS; k++) v[k] = p[k];
            uint32_t h = 0;
#pragma unroll
            for (int k = 0; k < LOADS; k++) h ^= v[k].x + v[k].y + v[k].z + v[k].w;
            acc += h;
            idx = (h ^ (idx * 747796405u + 2891336453u)) % n_rec;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int LOADS, int STRIDE16, int SHARE = 1, bool SCALAR = false>
void run(const uint4* tab, uint32_t n_rec, int blocks_per_cu, int active, uint32_t* out, const char* what) {
    int iters = 400;
    int grid = 256 * blocks_per_cu;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    walk<LOADS, STRIDE16, SHARE, SCALAR><<<grid, 256>>>(tab, n_rec, 20, active, out);
    CHECK(hipEventRecord(a));
    walk<LOADS, STRIDE16, SHARE, SCALAR><<<grid, 256>>>(tab, n_rec, iters, active, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    double wave_loads = (double)grid * 4 * iters * LOADS;
    double lane_bytes = (double)grid * 4 * active * iters * LOADS * 16;
    printf("%-28s table %6.1f MB  blocks/CU %d  active %2d  %7.3f ms  %6.2f ns per wave-load per CU  %7.1f GB/s useful\n", what,
           n_rec * STRIDE16 * 16 / 1e6, blocks_per_cu, active, ms, ms * 1e6 / (wave_loads / 256), lane_bytes / ms / 1e6);
}

int main() {
    const size_t bytes = 512ull << 20;
    uint4* tab; uint32_t* out;
    CHECK(hipMalloc(&tab, bytes)); CHECK(hipMalloc(&out, 4));
    std::vector<uint32_t> h(bytes / 4);
    uint32_t s = 12345; for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s; }
    CHECK(hipMemcpy(tab, h.data(), bytes, hipMemcpyHostToDevice));
    for (size_t mb : {16, 48}) {
        uint32_t n80 = (uint32_t)(mb * 1000000 / 80), n48 = (uint32_t)(mb * 1000000 / 48);
        for (int occ : {3, 6}) {
            for (int act : {64, 32, 16}) run<5, 5>(tab, n80, occ, act, out, "80-B record, 5 x dwordx4");
            run<3, 3>(tab, n48, occ, 64, out, "48-B record, 3 x dwordx4");
            run<4, 4>(tab, n80, occ, 64, out, "64-B record, 4 x dwordx4");
            run<1, 5>(tab, n80, occ, 64, out, "80-B stride, 1 x dwordx4");
            run<5, 8>(tab, (uint32_t)(mb * 1000000 / 128), occ, 64, out, "80-B record at 128-B stride");
            run<5, 8>(tab, (uint32_t)(mb * 1000000 / 128), occ, 32, out, "80-B record at 128-B stride");
            run<3, 4>(tab, (uint32_t)(mb * 1000000 / 64), occ, 64, out, "48-B record at 64-B stride");
            run<4, 4>(tab, (uint32_t)(mb * 1000000 / 64), occ, 64, out, "64-B record at 64-B stride");
            run<5, 5, 64>(tab, n80, occ, 64, out, "80-B rec, wave-shared, vector");
            run<5, 5, 8>(tab, n80, occ, 64, out, "80-B rec, 8 lanes share");
            run<5, 5, 64, true>(tab, n80, occ, 64, out, "80-B rec, wave-shared, scalar");
        }
    }
    return 0;
}
