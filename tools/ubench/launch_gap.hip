// Micro-benchmark: what a chain of DEPENDENT launches costs per link on gfx950 -- stream launches against one
// hipGraph of the same chain -- for kernels that do (almost) nothing and for kernels of ~50 us.
//   hipcc -O3 --offload-arch=gfx950 -o launch_gap launch_gap.hip && ./launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void work(float* p, int iters) {
    float x = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; i++) x = __builtin_fmaf(x, 0.999f, 0.001f);
    p[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

int main() {
    float* p; CHECK(hipMalloc(&p, 1024 * 256 * 4)); CHECK(hipMemset(p, 0, 1024 * 256 * 4));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int chain = 9, reps = 200;
    for (int iters : {1, 20000}) {
        for (int w = 0; w < 3; w++) work<<<1024, 256, 0, s>>>(p, iters);
        CHECK(hipStreamSynchronize(s));
        // one kernel alone
        CHECK(hipEventRecord(a, s));
        for (int r = 0; r < reps; r++) work<<<1024, 256, 0, s>>>(p, iters);
        CHECK(hipEventRecord(b, s)); CHECK(hipEventSynchronize(b));
        float ms_stream; CHECK(hipEventElapsedTime(&ms_stream, a, b));
        // the same launches as graphs of `chain` kernel nodes
        hipGraph_t g; hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int k = 0; k < chain; k++) work<<<1024, 256, 0, s>>>(p, iters);
        CHECK(hipStreamEndCapture(s, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; w++) CHECK(hipGraphLaunch(ge, s));
        CHECK(hipStreamSynchronize(s));
        CHECK(hipEventRecord(a, s));
        for (int r = 0; r < reps / chain; r++) CHECK(hipGraphLaunch(ge, s));
        CHECK(hipEventRecord(b, s)); CHECK(hipEventSynchronize(b));
        float ms_graph; CHECK(hipEventElapsedTime(&ms_graph, a, b));
        printf("kernel of %5d fma per thread: %7.2f us per dependent launch on a stream, %7.2f us per node of a %d-node graph\n",
               iters, ms_stream * 1e3 / reps, ms_graph * 1e3 / (reps / chain * chain), chain);
    }
    return 0;
}
