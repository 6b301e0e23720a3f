// Do kernels launched on two HIP streams run concurrently on this box?  Each kernel is ONE block that spins for
// ~`us` microseconds (wall clock); two of them on two streams take `us` if they overlap, 2 x `us` if they serialise.
// Also: the same with an event dependency pattern like mq_process uses, and with a low-priority stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void spin(long long cycles, int* sink) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 9999) *sink = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t a, b, lo; int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, least);
    printf("priority range: least %d greatest %d\n", least, greatest);
    const long long cyc = 100000000LL / 1000 * 2; // wall_clock64 ticks at 100 MHz: 2 ms
    spin<<<1, 64, 0, a>>>(1000, nullptr); hipDeviceSynchronize();
    for (int grid : {1, 256, 2048}) {
        double t0 = now(); spin<<<grid, 256, 0, a>>>(cyc, nullptr); hipDeviceSynchronize(); double one = now() - t0;
        t0 = now(); spin<<<grid, 256, 0, a>>>(cyc, nullptr); spin<<<grid, 256, 0, b>>>(cyc, nullptr); hipDeviceSynchronize(); double two = now() - t0;
        t0 = now(); spin<<<grid, 256, 0, a>>>(cyc, nullptr); spin<<<grid, 256, 0, a>>>(cyc, nullptr); hipDeviceSynchronize(); double same = now() - t0;
        t0 = now(); spin<<<grid, 256, 0, nullptr>>>(cyc, nullptr); spin<<<grid, 256, 0, lo>>>(cyc, nullptr); hipDeviceSynchronize(); double nul = now() - t0;
        printf("grid %4d x 256: one kernel %.2f ms; two streams %.2f ms; same stream %.2f ms; null stream + low-priority stream %.2f ms\n", grid, one * 1e3, two * 1e3, same * 1e3, nul * 1e3);
    }
    return 0;
}
