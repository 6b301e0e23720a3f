// Micro-benchmark: how many 64-byte records per second can gfx950 fetch from RANDOM places of a table far larger than
// its caches -- the Markov-chain state lookups of the shading kernels (14.6 M per 1080p frame from a 2.1 GB table).
//   hipcc -O3 --offload-arch=gfx950 -o random_states random_states.hip && ./random_states
// Each lane fetches K records per "vertex" (48 of the 64 bytes, like mc_load), with DEPTH of them in flight:
// DEPTH 1 = the next index needs the previous record (a chain), 2 = one ahead (the shading kernels' rolled loop), K = all at once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int K, int DEPTH>
__global__ __launch_bounds__(256) void fetch(const uint4* __restrict__ tab, uint32_t n_rec, int iters, uint32_t* out) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = 0, carry = 0;
    for (int it = 0; it < iters; it++) {
        if (DEPTH >= K) {
            uint4 v[K][3];
#pragma unroll
            for (int k = 0; k < K; k++) { const uint4* p = tab + (size_t)(mix(tid * 977u + it * 31u + k) % n_rec) * 4; v[k][0] = p[0]; v[k][1] = p[1]; v[k][2] = p[2]; }
#pragma unroll
            for (int k = 0; k < K; k++) acc += v[k][0].x + v[k][1].y + v[k][2].z;
        } else if (DEPTH == 2) {
            const uint4* p = tab + (size_t)(mix(tid * 977u + it * 31u) % n_rec) * 4;
            uint4 a0 = p[0], a1 = p[1], a2 = p[2];
#pragma unroll
            for (int k = 0; k < K; k++) {
                uint4 b0 = a0, b1 = a1, b2 = a2;
                if (k + 1 < K) { const uint4* q = tab + (size_t)(mix(tid * 977u + it * 31u + k + 1) % n_rec) * 4; b0 = q[0]; b1 = q[1]; b2 = q[2]; }
                acc += a0.x + a1.y + a2.z;
                a0 = b0; a1 = b1; a2 = b2;
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint4* p = tab + (size_t)(mix(tid * 977u + it * 31u + k + (carry & 1u)) % n_rec) * 4;
                uint4 a0 = p[0], a1 = p[1], a2 = p[2];
                carry = a0.x + a1.y + a2.z; acc += carry;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int K, int DEPTH>
void run(const uint4* tab, uint32_t n_rec, int blocks_per_cu, uint32_t* out) {
    const int iters = 40, grid = 256 * blocks_per_cu;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    fetch<K, DEPTH><<<grid, 256>>>(tab, n_rec, 4, out);
    CHECK(hipEventRecord(a));
    fetch<K, DEPTH><<<grid, 256>>>(tab, n_rec, iters, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double recs = (double)grid * 256 * iters * K;
    printf("table %7.1f MB  K %d  in flight %d  waves/SIMD %d  %8.3f ms  %6.2f G records/s  %7.1f GB/s of 64-byte records\n", n_rec * 64.0 / 1e6, K, DEPTH, blocks_per_cu, ms, recs / ms / 1e6, recs * 64 / ms / 1e6);
}

int main() {
    const size_t sizes[] = {(size_t)32 << 20, (size_t)2148 << 20};
    uint32_t* out; CHECK(hipMalloc(&out, 4));
    for (size_t bytes : sizes) {
        uint4* tab; CHECK(hipMalloc(&tab, bytes)); CHECK(hipMemset(tab, 1, bytes));
        const uint32_t n = (uint32_t)(bytes / 64);
        for (int occ : {3, 8}) {
            if (occ == 3) { run<5, 1>(tab, n, 3, out); run<5, 2>(tab, n, 3, out); run<5, 5>(tab, n, 3, out); }
            else { run<5, 1>(tab, n, 8, out); run<5, 2>(tab, n, 8, out); run<5, 5>(tab, n, 8, out); }
        }
        CHECK(hipFree(tab));
    }
    return 0;
}
