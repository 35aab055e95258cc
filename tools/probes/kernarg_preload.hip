// Probe: what does the kernel-argument fetch cost a dependent chain of small launches?  A kernel's first instruction is an s_load of its
// arguments from the kernarg segment (a cold miss: every launch has its own), and no global load can be issued before the pointers are
// there.  gfx950 can have the command processor PRELOAD the first arguments into SGPRs (-mllvm -amdgpu-kernarg-preload-count=N; only
// scalar / pointer arguments, not a by-value struct).  Same source built twice:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/kernarg_preload.hip -o tools/probes/kernarg_plain
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 tools/probes/kernarg_preload.hip -o tools/probes/kernarg_pre
// Each prints us per launch of a graph-replayed chain of 240 dependent launches (256 blocks x 256 threads, each lane one 16-byte load
// from a 4 MB buffer rotating over 24 buffers + one store), arguments passed (a) as pointers, (b) inside a by-value struct.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Params { const float4* in; float* out; const float* prev; int n; int pad[40]; };
__global__ __launch_bounds__(256) void k_args(const float4* __restrict__ in, float* __restrict__ out, const float* __restrict__ prev, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float4 v = in[i];
    const float p = prev[threadIdx.x & 7];       // dependence on the previous launch's output
    out[i] = v.x + v.y + v.z + v.w + p;
}
__global__ __launch_bounds__(256) void k_struct(const Params q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float4 v = q.in[i];
    const float p = q.prev[threadIdx.x & 7];
    q.out[i] = v.x + v.y + v.z + v.w + p;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int L = 24, N = 256 * 256, CHAIN = 240;
    std::vector<float4*> in(L);
    float *o0, *o1;
    for (int l = 0; l < L; ++l) { CK(hipMalloc(&in[l], N * sizeof(float4))); CK(hipMemset(in[l], 0, N * sizeof(float4))); }
    CK(hipMalloc(&o0, N * 4)); CK(hipMalloc(&o1, N * 4)); CK(hipMemset(o0, 0, N * 4)); CK(hipMemset(o1, 0, N * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int c = 0; c < CHAIN; ++c) {
            float* out = (c & 1) ? o1 : o0; const float* prev = (c & 1) ? o0 : o1;
            if (mode == 0) hipLaunchKernelGGL(k_args, dim3(256), dim3(256), 0, s, (const float4*)in[c % L], out, prev, N);
            else { Params q = {}; q.in = in[c % L]; q.out = out; q.prev = prev; q.n = N; hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, s, q); }
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        const int REP = 50;
        for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.3f us per launch\n", mode == 0 ? "pointer arguments" : "by-value struct ", ms * 1e3 / (REP * CHAIN));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
