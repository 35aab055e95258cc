// What the bf16 matrix pipe of THIS device sustains on random operands with no memory traffic at all: the practical ceiling of any
// MFMA-bound kernel under the chip's power management (MI355X_MICROARCH.md, DVFS give-back items 1, 5, 7).  Operands sit in registers,
// every wave runs `iters` rounds of 8 independent MFMAs; random vs all-zero operands, v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16,
// one or two waves per SIMD.  The in-kernel clock is d(s_memtime) / d(s_memrealtime) x 100 MHz (median over workgroups).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip && ./mfma_power
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool BIG>
__global__ __launch_bounds__(512, 1) void mfma_loop(const bf16x8* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ stamps, int iters) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = src[(gid * 6 + i) & 65535];
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = src[(gid * 6 + 4 + i) & 65535];
    f32x16 acc32[8];
    f32x4 acc16[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[i][r] = 0.f;
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (BIG) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc32[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)      // two 16-cycle MFMAs per 32-cycle slot of the big shape: the same flops per round
#pragma unroll
                for (int i = 0; i < 8; ++i) acc16[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc16[i], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc32[i][r];
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc16[i][r];
    }
    sink[gid] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

static unsigned short bf16_of(float x) { unsigned u; __builtin_memcpy(&u, &x, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const bool quick = argc > 1 && std::string(argv[1]) == "--quick";   // bench.py: random operands, 32x32x16, two waves per SIMD only; one JSON line
    const int n_src = 65536 * 8;
    std::vector<unsigned short> h(n_src);
    srand(1);
    void *src, *sink, *stamps;
    hipMalloc(&src, n_src * 2); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&stamps, 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 100000;
    for (int zero = 0; zero < (quick ? 1 : 2); ++zero) {
        for (auto& v : h) v = zero ? 0 : bf16_of((float)rand() / RAND_MAX * 2.f - 1.f);
        hipMemcpy(src, h.data(), n_src * 2, hipMemcpyHostToDevice);
        for (int big = 1; big >= (quick ? 1 : 0); --big)
            for (int threads : {256, 512}) {
                if (quick && threads != 512) continue;
                auto launch = [&]() {
                    if (big) hipLaunchKernelGGL(mfma_loop<true>, dim3(256), dim3(threads), 0, 0, (const bf16x8*)src, (float*)sink, (unsigned long long*)stamps, iters);
                    else hipLaunchKernelGGL(mfma_loop<false>, dim3(256), dim3(threads), 0, 0, (const bf16x8*)src, (float*)sink, (unsigned long long*)stamps, iters);
                };
                for (int w = 0; w < 100; ++w) launch();         // ~2 s of back-to-back launches before the timed ones
                hipEventRecord(e0, 0);
                const int reps = 10;
                for (int w = 0; w < reps; ++w) launch();
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> st(512);
                hipMemcpy(st.data(), stamps, 256 * 16, hipMemcpyDeviceToHost);
                std::vector<double> ghz;
                for (int i = 0; i < 256; ++i) ghz.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 0.1);
                std::sort(ghz.begin(), ghz.end());
                const double flops = 256.0 * (threads / 64) * (double)iters * 8 * 32768.0 * reps;      // 2 * 32 * 32 * 16 per big MFMA
                if (quick) {
                    printf("{\"mfma\": \"v_mfma_f32_32x32x16_bf16\", \"operands\": \"random, in registers\", \"waves_per_simd\": 2, \"tflops\": %.1f, \"clock_ghz\": %.2f}\n",
                           flops / (ms * 1e-3) * 1e-12, ghz[128]);
                    continue;
                }
                printf("%s operands, %s, %d wave(s)/SIMD: %8.1f TFLOP/s  in-kernel clock %.2f GHz (median), %6.1f cycles per 32x32x16-equivalent per SIMD\n",
                       zero ? "zero  " : "random", big ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", threads / 256,
                       flops / (ms * 1e-3) * 1e-12, ghz[128], (ms * 1e-3 / reps) * ghz[128] * 1e9 / ((double)iters * 8 * (threads / 256)));
            }
    }
    return 0;
}
