// Probe: can two kernels be co-resident on one MI355X (a) from two streams, (b) from one stream with hipExtAnyOrderLaunch?
// Each kernel spins ~T us in every block (wall clock), so N kernels take N*T if serialized and less if they overlap.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void spin_kernel(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int N = 200;
    int* sink;
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(sink, 0, 4));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    const long long ticks = 2000;  // wall_clock64 runs at 100 MHz: 2000 ticks = 20 us
    for (int grid : {1, 256, 1024}) {
        // warm-up
        hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, a, ticks, sink);
        hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, b, ticks, sink);
        CK(hipDeviceSynchronize());
        double t0 = now_ms();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, a, ticks, sink);
        CK(hipDeviceSynchronize());
        double one = now_ms() - t0;
        t0 = now_ms();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, a, ticks, sink);
            hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, b, ticks, sink);
        }
        CK(hipDeviceSynchronize());
        double two = now_ms() - t0;
        t0 = now_ms();
        for (int i = 0; i < 2 * N; ++i)
            hipExtLaunchKernelGGL(spin_kernel, dim3(grid), dim3(256), 0, a, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink);
        CK(hipDeviceSynchronize());
        double any = now_ms() - t0;
        printf("grid %4d: %d kernels on one stream %.2f ms (%.2f us each); %d+%d on two streams %.2f ms; %d any-order on one stream %.2f ms\n",
               grid, N, one, one * 1e3 / N, N, N, two, 2 * N, any);
    }
    return 0;
}
