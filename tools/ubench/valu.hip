// Micro-benchmark: issue cost of the VALU instructions the traversal kernel is made of (gfx950).
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o valu valu.hip && ./valu
// Every kernel runs N_ITER iterations of 32 independent copies of one small operation per wave, at
// 1, 2 and 4 waves per SIMD; reports cycles per operation per SIMD (4.0 = one full-rate wave64 instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define N_ITER 2000
typedef float v2f __attribute__((ext_vector_type(2)));

struct OpFma { typedef float T; static __device__ T init(float a, float b, uint32_t w, int k) { return a + k + threadIdx.x; }
    static __device__ T op(T x, float a, float b, uint32_t w) { return __builtin_fmaf(x, a, b); } static __device__ float fold(T x) { return x; } };
struct OpPkFma { typedef v2f T; static __device__ T init(float a, float b, uint32_t w, int k) { v2f r; r.x = a + k + threadIdx.x; r.y = b + k; return r; }
    static __device__ T op(T x, float a, float b, uint32_t w) { v2f va; va.x = a; va.y = a; v2f vb; vb.x = b; vb.y = b; return __builtin_elementwise_fma(x, va, vb); } static __device__ float fold(T x) { return x.x + x.y; } };
struct OpCvt { typedef uint32_t T; static __device__ T init(float a, float b, uint32_t w, int k) { return w + k * 977u + threadIdx.x; }
    static __device__ T op(T x, float a, float b, uint32_t w) { float f; asm volatile("v_cvt_f32_ubyte1_e32 %0, %1" : "=v"(f) : "v"(x)); return __float_as_uint(f) ; } static __device__ float fold(T x) { return (float)x; } };
struct OpMax { typedef float T; static __device__ T init(float a, float b, uint32_t w, int k) { return a + k + threadIdx.x; }
    static __device__ T op(T x, float a, float b, uint32_t w) { return __builtin_fmaxf(__builtin_fmaxf(x, a), b); } static __device__ float fold(T x) { return x; } };
struct OpCnd { typedef float T; static __device__ T init(float a, float b, uint32_t w, int k) { return a + k + threadIdx.x; }
    static __device__ T op(T x, float a, float b, uint32_t w) { return (x > a) ? b : x; } static __device__ float fold(T x) { return x; } };
struct OpMulLo { typedef uint32_t T; static __device__ T init(float a, float b, uint32_t w, int k) { return w + k + threadIdx.x; }
    static __device__ T op(T x, float a, float b, uint32_t w) { return x * w; } static __device__ float fold(T x) { return (float)x; } };
struct OpBfe { typedef uint32_t T; static __device__ T init(float a, float b, uint32_t w, int k) { return w + k + threadIdx.x; }
    static __device__ T op(T x, float a, float b, uint32_t w) { return __builtin_amdgcn_ubfe(x, 5, 9) ^ w; } static __device__ float fold(T x) { return (float)x; } };
struct OpRcp { typedef float T; static __device__ T init(float a, float b, uint32_t w, int k) { return a + k + threadIdx.x; }
    static __device__ T op(T x, float a, float b, uint32_t w) { return __builtin_amdgcn_rcpf(x); } static __device__ float fold(T x) { return x; } };
struct OpDiv { typedef float T; static __device__ T init(float a, float b, uint32_t w, int k) { return a + k + threadIdx.x; }
    static __device__ T op(T x, float a, float b, uint32_t w) { return a / x; } static __device__ float fold(T x) { return x; } };
struct OpSqrt { typedef float T; static __device__ T init(float a, float b, uint32_t w, int k) { return a + k + threadIdx.x; }
    static __device__ T op(T x, float a, float b, uint32_t w) { return __builtin_sqrtf(x); } static __device__ float fold(T x) { return x; } };

template <typename O>
__global__ __launch_bounds__(256) void kern(float* out, float a, float b, uint32_t w) {
    typename O::T x[32];
#pragma unroll
    for (int k = 0; k < 32; k++) x[k] = O::init(a, b, w, k);
    for (int i = 0; i < N_ITER; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) x[k] = O::op(x[k], a, b, w);
    }
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 32; k++) acc += O::fold(x[k]);
    if (acc == 1234.5f) out[0] = acc;
}

template <typename O> void run(const char* name, float* out) {
    for (int waves = 1; waves <= 4; waves *= 2) {
        int grid = 256 * waves; // 256 CUs x `waves` blocks of 4 waves = `waves` waves per SIMD
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        kern<O><<<grid, 256>>>(out, 1.0001f, 0.5f, 0x01020304u);
        CHECK(hipEventRecord(a));
        kern<O><<<grid, 256>>>(out, 1.0001f, 0.5f, 0x01020304u);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        double ops_per_simd = (double)waves * N_ITER * 32;
        printf("%-14s waves/SIMD %d  %8.3f ms  %6.2f ns per op per SIMD (x2.4 GHz = %5.1f cycles)\n", name, waves, ms, ms * 1e6 / ops_per_simd, ms * 1e6 / ops_per_simd * 2.4);
    }
}

int main() {
    float* out; CHECK(hipMalloc(&out, 4));
    run<OpFma>("v_fma_f32", out);
    run<OpPkFma>("v_pk_fma_f32", out);
    run<OpCvt>("cvt_f32_ubyte", out);
    run<OpMax>("max(max())", out);
    run<OpCnd>("cmp+cndmask", out);
    run<OpMulLo>("v_mul_lo_u32", out);
    run<OpBfe>("bfe+xor", out);
    run<OpRcp>("v_rcp_f32", out);
    run<OpDiv>("ieee div", out);
    run<OpSqrt>("ieee sqrt", out);
    return 0;
}
