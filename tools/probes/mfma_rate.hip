// Probe: what does v_mfma_f32_32x32x2_f32 sustain on MI355X with 1, 2, 3 and 4 waves per SIMD and nothing else in the loop?
// (The fp32 encoder GEMM is priced against 157.3 TFLOP/s = 256 CUs x 256 FLOP/clk x 2.4 GHz; this is the ceiling a kernel can
// actually reach at the clock the part holds under that load.)  Each wave keeps 4 independent 32x32 accumulators, exactly like the
// GEMM's 2x2 register tile.  Prints TFLOP/s, shader clock and shader cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(int iters, float* out, long long* clk) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.f + blockIdx.x * 1e-4f;
    const long long c0 = clock64(), t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[3], 0, 0, 0);
        }
    }
    const long long c1 = clock64(), t1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = t1 - t0; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    float* out;
    long long* clk;
    CK(hipMalloc(&out, 4));
    CK(hipMalloc(&clk, 2 * 1024 * sizeof(long long)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 4000;   // x 32 MFMAs x 64 cycles = 8.2 M cycles per wave
    for (int per_cu = 1; per_cu <= 4; ++per_cu) {
        const int grid = 256 * per_cu;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, iters, out, clk);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        long long h[2 * 1024];
        CK(hipMemcpy(h, clk, 2 * grid * sizeof(long long), hipMemcpyDeviceToHost));
        double cyc = 0, ticks = 0;
        for (int i = 0; i < grid; ++i) { cyc += h[2 * i]; ticks += h[2 * i + 1]; }
        cyc /= grid; ticks /= grid;
        const double flops = (double)grid * 4 * iters * 32 * 4096.0;
        printf("%d waves/SIMD: %8.3f ms  %6.1f TFLOP/s   shader clock %.3f GHz   %.1f shader cycles per MFMA per SIMD\n", per_cu, ms,
               flops / ms * 1e-9, cyc / (ticks * 10.0), cyc / (iters * 32.0 * per_cu));
    }
    return 0;
}
