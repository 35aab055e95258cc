// fp16-encoder kernels (engine precision "float16", BASELINE config 4: fp16 encoder + fp32 decoder).
// Weights and GEMM inputs are fp16, every accumulation is fp32 on v_mfma_f32_16x16x32_f16, the residual stream,
// LayerNorm statistics, softmax and the encoder output stay fp32 (the reference forces fp32 scores even in fp16
// builds, model.py:292-295, and marks hidden_states float32, model.py:109).
#include "wt_common.h"

#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <type_traits>

namespace wt {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// GELU of the fp16 epilogues: the branch-free erf form shared with the fp32 encoder (wt_common.h: gelu_erf; was A&S 7.1.26 with an rcp + exp)
__device__ __forceinline__ float gelu_erf_h(float x) { return gelu_erf(x); }

// ------------------------------------------------------------------------------------------------ fp32 -> fp16 helpers
// mel [B][C][F] fp32 -> melT [B][F+2][C] fp16 (row = time+1; rows 0 / F+1 zero) : conv1 as implicit GEMM, K = 3C
__global__ __launch_bounds__(256) void mel_transpose_h_kernel(const float* __restrict__ mel, __half* __restrict__ melT, int C, int F) {
    __shared__ float tile[128][65];
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        int c = i >> 6, tl = i & 63, t = t0 + tl;
        tile[c][tl] = t < F ? mel[((size_t)b * C + c) * F + t] : 0.f;
    }
    __syncthreads();
    __half* dst = melT + (size_t)b * (F + 2) * C;
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        int tl = i / C, c = i - tl * C, t = t0 + tl;
        if (t < F) dst[(size_t)(t + 1) * C + c] = __float2half(tile[c][tl]);
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < C; i += 256) {
            dst[i] = __float2half(0.f);
            dst[(size_t)(F + 1) * C + i] = __float2half(0.f);
        }
}
hipError_t launch_mel_transpose_h(const float* mel, void* melT, int B, int n_mels, int frames, hipStream_t s) {
    if (n_mels > 128) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mel_transpose_h_kernel, dim3((frames + 63) / 64, B), dim3(256), 0, s, mel, (__half*)melT, n_mels, frames);
    return hipGetLastError();
}

// LayerNorm fp32 in -> fp16 out (statistics in fp32), one wave per row
__global__ __launch_bounds__(256) void layernorm_h_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, __half* __restrict__ y, int rows, int d) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * d);
    const int n4 = d >> 2;
    float4 v[5];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int c = lane + 64 * i;
        v[i] = c < n4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
        if (lane + 64 * i < n4) {
            float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
            q += (a * a + bb * bb) + (cc * cc + dd * dd);
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / d + 1e-5f);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    __half2* yr = reinterpret_cast<__half2*>(y + (size_t)row * d);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int c = lane + 64 * i;
        if (c < n4) {
            float4 g = w4[c], be = b4[c];
            yr[2 * c] = __floats2half2_rn((v[i].x - mean) * rstd * g.x + be.x, (v[i].y - mean) * rstd * g.y + be.y);
            yr[2 * c + 1] = __floats2half2_rn((v[i].z - mean) * rstd * g.z + be.z, (v[i].w - mean) * rstd * g.w + be.w);
        }
    }
}
hipError_t launch_layernorm_h(const float* x, const float* w, const float* b, void* y, int rows, int d, hipStream_t s) {
    if (d > 1280 || (d & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_h_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, (__half*)y, rows, d);
    return hipGetLastError();
}

// fp32 -> fp16 elementwise (attention context for the out-projection)
__global__ __launch_bounds__(256) void cast_h_kernel(const float4* __restrict__ x, __half2* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = x[i];
        y[2 * i] = __floats2half2_rn(v.x, v.y);
        y[2 * i + 1] = __floats2half2_rn(v.z, v.w);
    }
}
hipError_t launch_cast_h(const float* x, void* y, size_t n, hipStream_t s) {
    if (n & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cast_h_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<const float4*>(x), (__half2*)y, n >> 2);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ fp16 MFMA GEMM
// C[m][n] = epi(sum_k A[m][k] W[n][k] + bias[n]); A, W fp16, accumulate fp32, C fp16 or fp32.
// 128x128x64 block tile, 4 waves (2x2), each wave 4x4 tiles of v_mfma_f32_16x16x32_f16.  Register-staged,
// double-buffered LDS with 160-byte rows (128 B of data + 32 B pad): for the 16x16x32 fragment read
// (lane -> row l&15, 16-byte k-chunk l>>4) every ds_read_b128 lane group then covers 16 distinct 16-byte slots.
constexpr int HBM_ = 128, HBN_ = 128;
// row stride in halfs: 128 B of data + 32 B pad at BK=64, 64 B + 32 B at BK=32 (both keep every ds_read_b128 lane group on
// 16 distinct 16-byte slots for the 16x16x32 fragment map)
template <int HBK_> constexpr int hgemm_ld() { return HBK_ == 64 ? 80 : 48; }
template <int HBK_> constexpr int hgemm_smem() { return 2 * 2 * HBM_ * hgemm_ld<HBK_>() * 2; }  // 81,920 B / 49,152 B

template <bool OUT_HALF, int HBK_>
__global__ __launch_bounds__(256, HBK_ == 64 ? 2 : 3) void gemm_f16_kernel(const GemmParams p) {
    constexpr int HLD_ = hgemm_ld<HBK_>();
    constexpr int NC8 = HBK_ / 8;        // 16-byte chunks per staged row
    constexpr int RPP = 256 / NC8;       // rows per staging pass (32 or 64)
    constexpr int NPASS = HBM_ / RPP;    // 4 or 2
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __half* smem = reinterpret_cast<__half*>(smem_raw);
    const __half* A = reinterpret_cast<const __half*>(p.A);
    const __half* W = reinterpret_cast<const __half*>(p.W);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;

    const int nbx = (p.N + HBN_ - 1) / HBN_, nby = (p.M + HBM_ - 1) / HBM_, total = nbx * nby;
    int bid = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * nbx, g = bid / per_group;
    const int gm = min(GROUP_M, nby - g * GROUP_M), in_g = bid - g * per_group;
    const int by = g * GROUP_M + in_g % gm, bx = in_g / gm;
    const int m0 = by * HBM_, n0 = bx * HBN_;

    // staging: 128 rows x NC8 sixteen-byte chunks per matrix; thread -> chunk c8, rows r0 + RPP i
    const int c8 = tid % NC8, r0 = tid / NC8;
    auto arow = [&](int i) -> const __half* {
        const int m = min(m0 + r0 + RPP * i, p.M - 1);
        const int bb = m / p.a_rows_per_batch;
        return A + (long long)bb * p.a_batch_stride + (long long)(m - bb * p.a_rows_per_batch) * p.lda + c8 * 8;
    };
    auto wrow = [&](int i) -> const __half* { return W + (long long)min(n0 + r0 + RPP * i, p.N - 1) * p.K + c8 * 8; };
    const __half *ap0 = arow(0), *ap1 = arow(1), *ap2 = arow(2), *ap3 = arow(3);
    const __half *wp0 = wrow(0), *wp1 = wrow(1), *wp2 = wrow(2), *wp3 = wrow(3);
    uint4 ra0, ra1, ra2 = make_uint4(0, 0, 0, 0), ra3 = ra2, rw0, rw1, rw2 = ra2, rw3 = ra2;
    const uint4 z4 = make_uint4(0, 0, 0, 0);
#define HG_LOAD(kt_)                                                                         \
    do {                                                                                     \
        const bool ok_ = (kt_) * HBK_ + c8 * 8 < p.K;                                        \
        ra0 = ok_ ? *reinterpret_cast<const uint4*>(ap0 + (kt_) * HBK_) : z4;            \
        ra1 = ok_ ? *reinterpret_cast<const uint4*>(ap1 + (kt_) * HBK_) : z4;            \
        rw0 = ok_ ? *reinterpret_cast<const uint4*>(wp0 + (kt_) * HBK_) : z4;            \
        rw1 = ok_ ? *reinterpret_cast<const uint4*>(wp1 + (kt_) * HBK_) : z4;            \
        if (NPASS == 4) {                                                                \
            ra2 = ok_ ? *reinterpret_cast<const uint4*>(ap2 + (kt_) * HBK_) : z4;        \
            ra3 = ok_ ? *reinterpret_cast<const uint4*>(ap3 + (kt_) * HBK_) : z4;        \
            rw2 = ok_ ? *reinterpret_cast<const uint4*>(wp2 + (kt_) * HBK_) : z4;        \
            rw3 = ok_ ? *reinterpret_cast<const uint4*>(wp3 + (kt_) * HBK_) : z4;        \
        }                                                                                \
    } while (0)
#define HG_STORE(buf_)                                                                       \
    do {                                                                                     \
        __half* As_ = smem + (buf_) * (2 * HBM_ * HLD_) + r0 * HLD_ + c8 * 8;                \
        __half* Ws_ = As_ + HBM_ * HLD_;                                                     \
        *reinterpret_cast<uint4*>(As_) = ra0; *reinterpret_cast<uint4*>(As_ + RPP * HLD_) = ra1;          \
        *reinterpret_cast<uint4*>(Ws_) = rw0; *reinterpret_cast<uint4*>(Ws_ + RPP * HLD_) = rw1;          \
        if (NPASS == 4) {                                                                                 \
            *reinterpret_cast<uint4*>(As_ + 2 * RPP * HLD_) = ra2; *reinterpret_cast<uint4*>(As_ + 3 * RPP * HLD_) = ra3; \
            *reinterpret_cast<uint4*>(Ws_ + 2 * RPP * HLD_) = rw2; *reinterpret_cast<uint4*>(Ws_ + 3 * RPP * HLD_) = rw3; \
        }                                                                                                 \
    } while (0)

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + HBK_ - 1) / HBK_;
    HG_LOAD(0);
    HG_STORE(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) HG_LOAD(kt + 1);
        const __half* As = smem + cur * (2 * HBM_ * HLD_) + (wr * 64 + l15) * HLD_ + 8 * kq;
        const __half* Ws = smem + cur * (2 * HBM_ * HLD_) + HBM_ * HLD_ + (wc * 64 + l15) * HLD_ + 8 * kq;
#pragma unroll
        for (int s = 0; s < HBK_ / 32; ++s) {
            h8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const h8*>(As + i * 16 * HLD_ + 32 * s);
                bf[i] = *reinterpret_cast<const h8*>(Ws + i * 16 * HLD_ + 32 * s);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) HG_STORE(cur ^ 1);
        __syncthreads();
    }
#undef HG_LOAD
#undef HG_STORE
    // epilogue: 16x16 C/D map: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
        const int n = n0 + wc * 64 + tj * 16 + l15;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 64 + ti * 16 + kq * 4 + r;
                if (m >= p.M) continue;
                float v = acc[ti][tj][r] + bv;
                if (p.act) v = gelu_erf_h(v);
                const int cb = m / p.c_rows_per_batch, cr = m - cb * p.c_rows_per_batch;
                if (p.epi == EPI_KV_HEADS) {   // cross-K/V projection of an fp16 decoder engine: head-split caches [b][h][kv_cap][64]
                    const int dm = p.kv_heads * HEAD_DIM, which = n / dm, nn = n - which * dm;
                    float* base = which ? p.C2 : p.C;
                    const long long at = (((long long)cb * p.kv_heads + (nn >> 6)) * p.kv_cap + p.kv_seq_off + cr) * HEAD_DIM + (nn & 63);
                    if (OUT_HALF) reinterpret_cast<__half*>(base)[at] = __float2half(v);
                    else base[at] = v;
                    continue;
                }
                if (p.pos) v += p.pos[(long long)cr * p.N + n];
                const long long off = (long long)cb * p.c_batch_stride + (long long)cr * p.ldc + n;
                if (p.resid) v += p.resid[off];
                if (OUT_HALF) reinterpret_cast<__half*>(p.C)[off] = __float2half(v);
                else p.C[off] = v;
            }
        }
    }
}

// Epilogue of the LDS-DMA kernel.  The 16x16 accumulator map gives a lane one column and four rows per tile: stored as it
// stands, every store instruction writes 4-byte (fp32) or 2-byte (fp16) elements in 64- / 32-byte pieces, and the bias / residual /
// position reads are gathers of the same shape -- measured, that epilogue was ~60 % of a K = 1024 launch.  Here each wave
// transposes its 64x64 tile through its own 32 x 68-float LDS scratch (two halves of 32 rows; the stage buffers are free by
// now), so that a lane owns FOUR CONSECUTIVE columns of one row: bias / residual / position rows are read as float4, the
// output is written as 16-byte (fp32) or 8-byte (fp16) pieces, whole 256- / 128-byte rows per 16 lanes, and the batch split
// (row -> utterance, row in utterance) is one wave-uniform division plus a carry per row instead of a division per element.
constexpr int EPI_LD = 68;  // scratch row stride in floats: 16-byte aligned rows, kq = 1 rows land 16 banks away from kq = 0
template <bool OUT_HALF>
__device__ __forceinline__ void hgemm_epilogue_lds(const GemmParams& p, const f32x4 (&acc)[4][4], const int m0, const int n0, const int wr,
                                                   const int wc, const int lane, float* __restrict__ ew) {
    const int l15 = lane & 15, kq = lane >> 4;
    const int n = n0 + wc * 64 + 4 * l15;               // first of this lane's four output columns
    const bool vec = (n + 3 < p.N) && ((p.ldc & 3) == 0) && ((p.c_batch_stride & 3) == 0) && ((p.N & 3) == 0) &&
                     ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) && ((reinterpret_cast<uintptr_t>(p.resid) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(p.pos) & 15) == 0) && ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) {
        if (vec) bv = *reinterpret_cast<const float4*>(p.bias + n);
        else {
            bv.x = n < p.N ? p.bias[n] : 0.f; bv.y = n + 1 < p.N ? p.bias[n + 1] : 0.f;
            bv.z = n + 2 < p.N ? p.bias[n + 2] : 0.f; bv.w = n + 3 < p.N ? p.bias[n + 3] : 0.f;
        }
    }
    const int mw = m0 + wr * 64;                          // first row of this wave's tile (wave-uniform)
    const int cb_w = mw / p.c_rows_per_batch, cr_w = mw - cb_w * p.c_rows_per_batch;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) ew[(t2 * 16 + kq * 4 + r) * EPI_LD + tj * 16 + l15] = acc[2 * half + t2][tj][r];
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int lr = pass * 4 + kq;                 // row of this half handled by this lane
            const float4 a = *reinterpret_cast<const float4*>(&ew[lr * EPI_LD + 4 * l15]);
            const int o = half * 32 + lr;
            if (mw + o >= p.M) continue;
            int cb = cb_w, cr = cr_w + o;
            while (cr >= p.c_rows_per_batch) { cr -= p.c_rows_per_batch; ++cb; }
            if (p.epi == EPI_KV_HEADS) {   // head-split K/V caches [b][h][kv_cap][64]: a lane's four columns never leave their head
                if (n >= p.N) continue;    // N = 2 * heads * 64: a multiple of 4, so n < N covers n + 3
                const int dm = p.kv_heads * HEAD_DIM, which = n / dm, nn = n - which * dm;
                float* base = which ? p.C2 : p.C;
                const long long at = (((long long)cb * p.kv_heads + (nn >> 6)) * p.kv_cap + p.kv_seq_off + cr) * HEAD_DIM + (nn & 63);
                if (OUT_HALF) {
                    __half2* dst = reinterpret_cast<__half2*>(reinterpret_cast<__half*>(base) + at);
                    dst[0] = __floats2half2_rn(a.x + bv.x, a.y + bv.y);
                    dst[1] = __floats2half2_rn(a.z + bv.z, a.w + bv.w);
                } else {
                    *reinterpret_cast<float4*>(base + at) = make_float4(a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w);
                }
                continue;
            }
            const long long off = (long long)cb * p.c_batch_stride + (long long)cr * p.ldc + n;
            float v[4] = {a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w};
            if (p.act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf_h(v[e]);
            }
            if (vec) {
                if (p.pos) {
                    const float4 q = *reinterpret_cast<const float4*>(p.pos + (long long)cr * p.N + n);
                    v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
                }
                if (p.resid) {
                    const float4 q = *reinterpret_cast<const float4*>(p.resid + off);
                    v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
                }
                if (OUT_HALF) {
                    __half2* dst = reinterpret_cast<__half2*>(reinterpret_cast<__half*>(p.C) + off);
                    dst[0] = __floats2half2_rn(v[0], v[1]);
                    dst[1] = __floats2half2_rn(v[2], v[3]);
                } else {
                    *reinterpret_cast<float4*>(p.C + off) = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= p.N) continue;
                    float x = v[e];
                    if (p.pos) x += p.pos[(long long)cr * p.N + n + e];
                    if (p.resid) x += p.resid[off + e];
                    if (OUT_HALF) reinterpret_cast<__half*>(p.C)[off + e] = __float2half(x);
                    else p.C[off + e] = x;
                }
            }
        }
    }
}

// The same GEMM with both tiles staged by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), for K % 64 == 0
// (every encoder GEMM but conv1).  128x128x64 tile, two 32 KiB stages, 2 blocks per CU.  One wave instruction writes 1 KiB
// linearly = 8 rows x 128 B, so the LDS image is UNPADDED [row][8 chunks of 8 halfs] and the fragment reads are kept
// conflict-free by an XOR swizzle applied on both sides (cdna guide §5.4 rule 21): LDS chunk position p of row r holds global
// chunk p ^ ((r >> 1) & 7).  A 16-lane ds_read_b128 group (rows r..r+15, one logical chunk) then covers all 16 sixteen-byte
// slots of the 256-byte bank row: rows 2a, 2a+1 share a swizzle and sit 128 B apart, the 8 values of a give 8 distinct positions.
// Pipeline as in gemm_f32_dma_kernel: wait for tile kt, ONE barrier, first fragment reads, DMA of tile kt+1 into the other
// stage (everyone is past its reads of that stage), 32 MFMAs.
template <bool OUT_HALF>
__global__ __launch_bounds__(256, 2) void gemm_f16_dma_kernel(const GemmParams p) {
    constexpr int BK = 64;
    __shared__ __attribute__((aligned(1024))) _Float16 smem[2][2][HBM_ * BK];  // [stage][A | W][row * 64 + pos * 8]
    const __half* A = reinterpret_cast<const __half*>(p.A);
    const __half* W = reinterpret_cast<const __half*>(p.W);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;

    const int nbx = (p.N + HBN_ - 1) / HBN_, nby = (p.M + HBM_ - 1) / HBM_, total = nbx * nby;
    int bid = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * nbx, g = bid / per_group;
    const int gm = min(GROUP_M, nby - g * GROUP_M), in_g = bid - g * per_group;
    const int by = g * GROUP_M + in_g % gm, bx = in_g / gm;
    const int m0 = by * HBM_, n0 = bx * HBN_;

    // DMA map: wave w, pass j fills rows j*32 + w*8 .. +7; lane -> (row lane >> 3, chunk position lane & 7)
    const int r_local = lane >> 3;
    const int csrc = (lane & 7) ^ (((wave * 8 + r_local) >> 1) & 7);   // (row >> 1) & 7 with row = j*32 + w*8 + r_local
    // 32-bit byte offsets + scalar bases, inline-asm requests: see gemm_f16_dma3_kernel
    unsigned aoff[4], woff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = j * 32 + wave * 8 + r_local;
        const int m = min(m0 + row, p.M - 1);
        const int bb = m / p.a_rows_per_batch;
        aoff[j] = (unsigned)(((long long)bb * p.a_batch_stride + (long long)(m - bb * p.a_rows_per_batch) * p.lda + csrc * 8) * 2);
        woff[j] = (unsigned)(((long long)min(n0 + row, p.N - 1) * p.K + csrc * 8) * 2);
    }
    typedef __attribute__((address_space(3))) void* lptr_t;
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)(&smem[0][0][0]) + (unsigned)wave * (8 * BK * 2);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // m0 on the clobber list: it is written and consumed inside the one statement
    auto dma = [&](const int stage, const int kt) {
        const unsigned st = lds_base + (unsigned)(stage * 2 * HBM_ * BK * 2);
        const char* ab = reinterpret_cast<const char*>(A) + (long long)kt * (BK * 2);   // wave-uniform
        const char* wb = reinterpret_cast<const char*>(W) + (long long)kt * (BK * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(st + (unsigned)(j * 32 * BK * 2)), "v"(aoff[j]), "s"(ab) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(st + (unsigned)((HBM_ + j * 32) * BK * 2)), "v"(woff[j]), "s"(wb) : "memory", "m0");
        }
    };
#pragma clang diagnostic pop
    // fragment reads (16x16x32: lane -> row l15 of the 16-row tile, 16-byte k-chunk kq + 4 s): tile i of A = rows wr*64 + 16 i + l15;
    // (row >> 1) & 7 == (l15 >> 1) for every tile (16 i and wr*64 are multiples of 16)
    const int swz = (l15 >> 1) & 7;
    const int ra = (wr * 64 + l15) * BK, rb = (wc * 64 + l15) * BK;
    const int po0 = ((kq + 0) ^ swz) * 8, po1 = ((kq + 4) ^ swz) * 8;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    dma(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of tile kt has landed
        __syncthreads();                                   // ... everyone's has, and nobody still reads the other stage
        const _Float16* As = &smem[cur][0][0];
        const _Float16* Ws = &smem[cur][1][0];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int po = s ? po1 : po0;
            h8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const h8*>(As + ra + i * 16 * BK + po);
                bf[i] = *reinterpret_cast<const h8*>(Ws + rb + i * 16 * BK + po);
            }
            if (s == 0 && kt + 1 < nk) dma(cur ^ 1, kt + 1);  // after the first fragment reads are on their way
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();  // every wave is past its last fragment read: the stage buffers become the epilogue scratch
    hgemm_epilogue_lds<OUT_HALF>(p, acc, m0, n0, wr, wc, lane, reinterpret_cast<float*>(&smem[0][0][0]) + wave * (32 * EPI_LD));
}

// Epilogue of the persistent kernel below: the same transposition through LDS, in EIGHT passes of 8 rows per wave (2 KiB of
// scratch per wave -- all that is left beside three 48 KiB stages in 160 KiB).  Pass (ti, rb): the 8 rows kq*4 + rb + {0, 1} of
// accumulator row-tile ti.  Scratch image [8 rows][64 floats], unpadded; XOR swizzle on the column, (row >> 1 & 1) << 4 ^ (row & 1) << 2:
// the two kq values of a 32-lane ds_write_b32 group land in different 16-bank groups, and the two rows of a 16-lane ds_read_b128
// group in alternating 16-byte slots.  A lane then owns columns 4c..4c+3 and 32+4c..32+4c+3 of one row.
template <bool OUT_HALF>
__device__ __forceinline__ void hgemm_epilogue_lds8(const GemmParams& p, const f32x4 (&acc)[4][4], const int m0, const int n0, const int wr,
                                                    const int wc, const int lane, float* __restrict__ ew) {
    const int l15 = lane & 15, kq = lane >> 4;
    const int lr = lane >> 3, c = lane & 7;                   // read side: row of the pass, column chunk
    const int rswz = (((lr >> 1) & 1) << 4) ^ ((lr & 1) << 2);
    const int nn[2] = {n0 + wc * 64 + 4 * c, n0 + wc * 64 + 32 + 4 * c};
    const bool aligned = ((p.ldc & 3) == 0) && ((p.c_batch_stride & 3) == 0) && ((p.N & 3) == 0) &&
                         ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) && ((reinterpret_cast<uintptr_t>(p.resid) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(p.pos) & 15) == 0) && ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0);
    bool vec[2];
    float4 bv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        vec[h] = aligned && (nn[h] + 3 < p.N);
        bv[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            if (vec[h]) bv[h] = *reinterpret_cast<const float4*>(p.bias + nn[h]);
            else {
                bv[h].x = nn[h] < p.N ? p.bias[nn[h]] : 0.f; bv[h].y = nn[h] + 1 < p.N ? p.bias[nn[h] + 1] : 0.f;
                bv[h].z = nn[h] + 2 < p.N ? p.bias[nn[h] + 2] : 0.f; bv[h].w = nn[h] + 3 < p.N ? p.bias[nn[h] + 3] : 0.f;
            }
        }
    }
    // Pin a USE of the bias registers here: every row of an M-edge tile may `continue` below, and a load whose result is only used on
    // some paths is still pending, as far as hipcc's waitcnt pass can tell, at the K-loop's back edge -- it then drains vmcnt(0)
    // (= every LDS-DMA in flight) at the top of EVERY K-step before it reuses those registers.
    asm volatile("" ::"v"(bv[0].x), "v"(bv[0].y), "v"(bv[0].z), "v"(bv[0].w), "v"(bv[1].x), "v"(bv[1].y), "v"(bv[1].z), "v"(bv[1].w));
    const int mw = m0 + wr * 64;
    const int cb_w = mw / p.c_rows_per_batch, cr_w = mw - cb_w * p.c_rows_per_batch;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
        for (int rb = 0; rb < 4; rb += 2) {
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
                    ew[(kq * 2 + rr) * 64 + ((tj * 16 + l15) ^ ((kq & 1) << 4) ^ (rr << 2))] = acc[ti][tj][rb + rr];
            float4 a[2];
            a[0] = *reinterpret_cast<const float4*>(&ew[lr * 64 + ((4 * c) ^ rswz)]);
            a[1] = *reinterpret_cast<const float4*>(&ew[lr * 64 + ((32 + 4 * c) ^ rswz)]);
            const int o = ti * 16 + (lr >> 1) * 4 + rb + (lr & 1);   // row of the wave tile this lane finishes in this pass
            if (mw + o >= p.M) continue;
            int cb = cb_w, cr = cr_w + o;
            while (cr >= p.c_rows_per_batch) { cr -= p.c_rows_per_batch; ++cb; }
            const long long rowoff = (long long)cb * p.c_batch_stride + (long long)cr * p.ldc;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const long long off = rowoff + nn[h];
                float v[4] = {a[h].x + bv[h].x, a[h].y + bv[h].y, a[h].z + bv[h].z, a[h].w + bv[h].w};
                if (p.act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf_h(v[e]);
                }
                if (vec[h]) {
                    if (p.pos) {
                        const float4 q = *reinterpret_cast<const float4*>(p.pos + (long long)cr * p.N + nn[h]);
                        v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
                    }
                    if (p.resid) {
                        const float4 q = *reinterpret_cast<const float4*>(p.resid + off);
                        v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
                    }
                    if (OUT_HALF) {
                        __half2* dst = reinterpret_cast<__half2*>(reinterpret_cast<__half*>(p.C) + off);
                        dst[0] = __floats2half2_rn(v[0], v[1]);
                        dst[1] = __floats2half2_rn(v[2], v[3]);
                    } else {
                        *reinterpret_cast<float4*>(p.C + off) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (nn[h] + e >= p.N) continue;
                        float x = v[e];
                        if (p.pos) x += p.pos[(long long)cr * p.N + nn[h] + e];
                        if (p.resid) x += p.resid[off + e];
                        if (OUT_HALF) reinterpret_cast<__half*>(p.C)[off + e] = __float2half(x);
                        else p.C[off + e] = x;
                    }
                }
            }
        }
    }
}

// Branch-free form of hgemm_epilogue_lds8 for a wave whose 64x64 sub-tile lies wholly inside C with 16-byte-aligned rows: the optional
// operands are compile-time (KIND), the residual / position rows are requested two passes ahead of their use, and nothing in it is a
// branch -- so hipcc's waitcnt pass emits counted waits for those loads only.  In the generic form every `if (p.resid)` / `continue`
// joins with `s_waitcnt vmcnt(0)`: a round trip of every store issued so far AND of the next tile's LDS-DMA in flight, 16 times a tile.
enum { HEK_PLAIN = 0, HEK_RESID = 1, HEK_ACT = 2, HEK_ACT_POS = 3 };
template <bool OUT_HALF, int KIND>
__device__ __forceinline__ void hgemm_epilogue_lds8_fast(const GemmParams& p, const f32x4 (&acc)[4][4], const int m0, const int n0, const int wr,
                                                         const int wc, const int lane, float* __restrict__ ew) {
    constexpr bool RESID = KIND == HEK_RESID, ACT = KIND == HEK_ACT || KIND == HEK_ACT_POS, POS = KIND == HEK_ACT_POS;
    const int l15 = lane & 15, kq = lane >> 4;
    const int lr = lane >> 3, c = lane & 7;
    const int rswz = (((lr >> 1) & 1) << 4) ^ ((lr & 1) << 2);
    const int nn[2] = {n0 + wc * 64 + 4 * c, n0 + wc * 64 + 32 + 4 * c};
    float4 bv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) bv[h] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nn[h]) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int mw = m0 + wr * 64, rpb = p.c_rows_per_batch;
    const int cb_w = mw / rpb, cr_w = mw - cb_w * rpb;
    auto row_of = [&](const int pass, int& cr) -> long long {        // pass = ti * 2 + rb / 2
        const int o = (pass >> 1) * 16 + (lr >> 1) * 4 + (pass & 1) * 2 + (lr & 1);
        cr = cr_w + o;
        const bool over = cr >= rpb;                                  // rpb >= 64: at most one batch boundary inside the sub-tile
        cr -= over ? rpb : 0;
        return (long long)(cb_w + (over ? 1 : 0)) * p.c_batch_stride + (long long)cr * p.ldc;
    };
    float4 rq[2][2], pq[2][2];
    auto fetch = [&](const int pass) {
        int cr;
        const long long rowoff = row_of(pass, cr);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (RESID) rq[pass & 1][h] = *reinterpret_cast<const float4*>(p.resid + rowoff + nn[h]);
            if (POS) pq[pass & 1][h] = *reinterpret_cast<const float4*>(p.pos + (long long)cr * p.N + nn[h]);
        }
    };
    if (RESID || POS) { fetch(0); fetch(1); }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const int ti = pass >> 1, rb = (pass & 1) * 2;
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
                ew[(kq * 2 + rr) * 64 + ((tj * 16 + l15) ^ ((kq & 1) << 4) ^ (rr << 2))] = acc[ti][tj][rb + rr];
        float4 a[2];
        a[0] = *reinterpret_cast<const float4*>(&ew[lr * 64 + ((4 * c) ^ rswz)]);
        a[1] = *reinterpret_cast<const float4*>(&ew[lr * 64 + ((32 + 4 * c) ^ rswz)]);
        int cr;
        const long long rowoff = row_of(pass, cr);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v[4] = {a[h].x + bv[h].x, a[h].y + bv[h].y, a[h].z + bv[h].z, a[h].w + bv[h].w};
            if (ACT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf_h(v[e]);
            }
            if (POS) { v[0] += pq[pass & 1][h].x; v[1] += pq[pass & 1][h].y; v[2] += pq[pass & 1][h].z; v[3] += pq[pass & 1][h].w; }
            if (RESID) { v[0] += rq[pass & 1][h].x; v[1] += rq[pass & 1][h].y; v[2] += rq[pass & 1][h].z; v[3] += rq[pass & 1][h].w; }
            if (OUT_HALF) {
                __half2* dst = reinterpret_cast<__half2*>(reinterpret_cast<__half*>(p.C) + rowoff + nn[h]);
                dst[0] = __floats2half2_rn(v[0], v[1]);
                dst[1] = __floats2half2_rn(v[2], v[3]);
            } else {
                *reinterpret_cast<float4*>(p.C + rowoff + nn[h]) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        if ((RESID || POS) && pass + 2 < 8) fetch(pass + 2);
    }
}
// dispatch: the fast form for interior, aligned sub-tiles of the operand combinations the engine uses, the generic one otherwise
template <bool OUT_HALF>
__device__ __forceinline__ void hgemm_epilogue_lds8_any(const GemmParams& p, const f32x4 (&acc)[4][4], const int m0, const int n0, const int wr,
                                                        const int wc, const int lane, float* __restrict__ ew) {
    const bool fast = m0 + wr * 64 + 64 <= p.M && n0 + wc * 64 + 64 <= p.N && p.c_rows_per_batch >= 64 && ((p.ldc & 3) == 0) &&
                      ((p.c_batch_stride & 3) == 0) && ((p.N & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(p.resid) & 15) == 0) && ((reinterpret_cast<uintptr_t>(p.pos) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0);
    if (fast && !p.pos) {
        if (!p.resid && !p.act) return hgemm_epilogue_lds8_fast<OUT_HALF, HEK_PLAIN>(p, acc, m0, n0, wr, wc, lane, ew);
        if (p.resid && !p.act) return hgemm_epilogue_lds8_fast<OUT_HALF, HEK_RESID>(p, acc, m0, n0, wr, wc, lane, ew);
        if (!p.resid && p.act) return hgemm_epilogue_lds8_fast<OUT_HALF, HEK_ACT>(p, acc, m0, n0, wr, wc, lane, ew);
    } else if (fast && p.act && !p.resid) {
        return hgemm_epilogue_lds8_fast<OUT_HALF, HEK_ACT_POS>(p, acc, m0, n0, wr, wc, lane, ew);
    }
    hgemm_epilogue_lds8<OUT_HALF>(p, acc, m0, n0, wr, wc, lane, ew);
}

// PERSISTENT 256x128x64 kernel for the big-batch GEMMs: 8 waves (4 x 2, 64x64 per wave), THREE 48 KiB stages + 16 KiB of epilogue
// scratch = the whole 160 KiB LDS, one workgroup per CU looping over its tiles (t = blockIdx, blockIdx + grid, ...: a block keeps
// its XCD label, so the XCD-chunked tile order survives).
// Why this shape: the 128x128 kernel above is bound by L2 -> LDS latency, not by MFMA issue -- per K-tile it needs 32 KiB per block and
// has one tile (2 blocks x 32 KiB per CU) in flight, ~1.1 us per K-tile against 0.43 us of MFMA time.  This one needs 25 % fewer
// operand bytes per flop and keeps 96 KiB in flight per CU; and as ONE stream of K-steps over all its tiles its LDS-DMA pipeline never
// drains: the first K-tiles of tile T+1 are requested while tile T is still multiplying and land under T's epilogue (a one-tile
// kernel at one block per CU exposes ~2 us of first-tile latency plus the epilogue per tile: a third of a K = 1024 tile).
// Synchronisation per K-step: a COUNTED `s_waitcnt vmcnt(6)` (this wave's six DMA instructions of the NEWEST step may stay
// outstanding; everything older has landed), then a raw `s_barrier` -- never __syncthreads(), whose fence waits vmcnt(0) and would
// drain the step in flight (cdna guide §5, "Pipelining across barriers").  The barrier inside step g orders (a) every wave's share
// of step g+1 landed -> reads of stage (g+1) % 3, and (b) every wave done with its reads of stage g % 3 -> the DMA of step g+3 into it.
// Epilogue stores / residual loads are younger vector-memory operations than the DMAs in flight, so the counted wait after an
// epilogue only over-waits (vmcnt retires in order), never under-waits.
constexpr int H3_BM = 256, H3_BN = 128, H3_BK = 64, H3_STAGE = (H3_BM + H3_BN) * H3_BK;   // halfs per stage (48 KiB)
constexpr int H3_SCRATCH = 8 * 8 * 64 * 4;                                                 // 8 waves x [8 rows][64 floats]
constexpr int H3_SMEM = 3 * H3_STAGE * 2 + H3_SCRATCH;                                     // 163,840 B = 160 KiB
template <bool OUT_HALF>
__global__ __launch_bounds__(512, 2) void gemm_f16_dma3_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char h3_raw[];
    _Float16* smem = reinterpret_cast<_Float16*>(h3_raw);   // [stage][A 256 rows | W 128 rows][row * 64 + pos * 8], then the scratch
    const __half* A = reinterpret_cast<const __half*>(p.A);
    const __half* W = reinterpret_cast<const __half*>(p.W);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;                 // 4 x 2 waves, 64 x 64 each
    float* ew = reinterpret_cast<float*>(h3_raw + 3 * H3_STAGE * 2) + wave * (8 * 64);

    const int nbx = (p.N + H3_BN - 1) / H3_BN, nby = (p.M + H3_BM - 1) / H3_BM, total = nbx * nby;
    const int G = gridDim.x;                                   // multiple of 8 or == total
    const int ntiles = ((int)blockIdx.x < total) ? (total - (int)blockIdx.x + G - 1) / G : 0;
    if (ntiles == 0) return;                                   // block-uniform
    auto tile_origin = [&](const int i, int& m0, int& n0) {    // i-th tile of this workgroup
        int bid = (int)blockIdx.x + i * G;
        const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        constexpr int GROUP_M = 4;
        const int per_group = GROUP_M * nbx, g = bid / per_group;
        const int gm = min(GROUP_M, nby - g * GROUP_M), in_g = bid - g * per_group;
        m0 = (g * GROUP_M + in_g % gm) * H3_BM;
        n0 = (in_g / gm) * H3_BN;
    };

    // DMA map: one wave instruction = 8 rows x 128 B.  A: 32 instructions (wave w, pass j < 4: rows j*64 + w*8 ..),
    // W: 16 instructions (pass j < 2).  lane -> (row lane >> 3, chunk position lane & 7); source chunk = pos ^ ((row >> 1) & 7)
    const int r_local = lane >> 3;
    const int csrc = (lane & 7) ^ (((wave * 8 + r_local) >> 1) & 7);
    // per-lane BYTE offsets of the six source rows (32 bits: launch_gemm_f16 checks that the operands span < 4 GiB); the K step's
    // column offset is wave-uniform and goes into the scalar base, the LDS destination is scalar too: a request is
    // `s_mov m0 | global_load_lds v_off, s[base]` -- no vector ALU work between the MFMAs (from the builtin hipcc selects the 64-bit
    // vector-address form and spends a v_lshl_add_u64 per request; every such instruction costs MFMA issue slots)
    unsigned aoff[4], woff[2];
    auto set_tile_ptrs = [&](const int i) {
        int m0, n0;
        tile_origin(i, m0, n0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = min(m0 + j * 64 + wave * 8 + r_local, p.M - 1);
            const int bb = m / p.a_rows_per_batch;
            aoff[j] = (unsigned)(((long long)bb * p.a_batch_stride + (long long)(m - bb * p.a_rows_per_batch) * p.lda + csrc * 8) * 2);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) woff[j] = (unsigned)(((long long)min(n0 + j * 64 + wave * 8 + r_local, p.N - 1) * p.K + csrc * 8) * 2);
    };
    typedef __attribute__((address_space(3))) void* lptr_t;
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem + (unsigned)wave * (8 * H3_BK * 2);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // m0 on the clobber list: it is written and consumed inside the one statement
    auto dma = [&](const int stage, const int kt) {
        const unsigned st = lds_base + (unsigned)(stage * H3_STAGE * 2);
        const char* ab = reinterpret_cast<const char*>(A) + (long long)kt * (H3_BK * 2);   // wave-uniform
        const char* wb = reinterpret_cast<const char*>(W) + (long long)kt * (H3_BK * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(st + (unsigned)(j * 64 * H3_BK * 2)), "v"(aoff[j]), "s"(ab) : "memory", "m0");
#pragma unroll
        for (int j = 0; j < 2; ++j)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(st + (unsigned)((H3_BM + j * 64) * H3_BK * 2)), "v"(woff[j]), "s"(wb) : "memory", "m0");
    };
#pragma clang diagnostic pop
    const int swz = (l15 >> 1) & 7;
    const int ra = (wr * 64 + l15) * H3_BK, rb = H3_BM * H3_BK + (wc * 64 + l15) * H3_BK;
    const int po0 = ((kq + 0) ^ swz) * 8, po1 = ((kq + 4) ^ swz) * 8;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Two-phase software pipeline over the two 32-deep k-steps (p0, p1) of a K-step, fragments double-buffered in registers:
    //   reads(g, p1) | MFMA(g, p0) | wait step g+1 + barrier | DMA(step g+3 -> stage g%3) | reads(g+1, p0) | MFMA(g, p1)
    // so every batch of 16 MFMAs runs under the LDS reads of the NEXT batch, and the counted wait + barrier are reached with 16
    // MFMAs already queued.  g runs over ALL K-steps of all tiles of this workgroup; the DMA cursor (d_i, d_kt) runs three steps ahead.
    const int nk = p.K / H3_BK;
    const int steps = ntiles * nk;
    int d_i = 0, d_kt = 0;
    set_tile_ptrs(0);
    auto dma_next = [&](const int stage) {
        dma(stage, d_kt);
        if (++d_kt == nk) {
            d_kt = 0;
            if (++d_i < ntiles) set_tile_ptrs(d_i);
        }
    };
    dma_next(0);
    if (steps > 1) dma_next(1);
    if (steps > 2) dma_next(2);
    if (steps > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (steps > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    h8 fa0[4], fb0[4], fa1[4], fb1[4];
    auto frag = [&](h8 (&fa)[4], h8 (&fb)[4], const int stage, const int po) {
        const _Float16* st = smem + stage * H3_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i] = *reinterpret_cast<const h8*>(st + ra + i * 16 * H3_BK + po);
            fb[i] = *reinterpret_cast<const h8*>(st + rb + i * 16 * H3_BK + po);
        }
    };
    auto mma = [&](const h8 (&fa)[4], const h8 (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    frag(fa0, fb0, 0, po0);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): both edges into the loop header carry an empty LDS scoreboard (see the loop's end)
    int cur = 0, kt = 0, c_i = 0, m0, n0, skip = 0;
    tile_origin(0, m0, n0);
    for (int g = 0; g < steps; ++g) {
        const int nxt = cur == 2 ? 0 : cur + 1;
        frag(fa1, fb1, cur, po1);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): my reads of stage `cur` are done (16 MFMAs were queued behind them)
        if (g + 1 < steps) {
            if (skip) skip = 0;                                                      // waited for before the last epilogue's stores
            else if (g + 2 < steps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // step g+1 landed (step g+2 may fly)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (g + 3 < steps) dma_next(cur);
            frag(fa0, fb0, nxt, po0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        // The reads of (g+1, p0) have had 16 MFMAs to land.  Retire them HERE with the builtin (which hipcc's waitcnt pass tracks;
        // an asm wait it does not): otherwise the pass, unsure what is pending across the back edge, puts `lgkmcnt(0)` in front of
        // the next iteration's MFMA(g+1, p0) -- behind the reads of (g+1, p1) it was supposed to overlap.
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) alone
        cur = nxt;
        if (++kt == nk) {   // this tile's K is complete: finish it under the flight of the next tile's first K-steps
            // Step g+2 would be waited for in the middle of the next step, i.e. BEHIND this epilogue's stores in the one in-order
            // counter: take that wait now (step g+3 may stay in flight) and skip it there.
            if (g + 3 < steps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            skip = 1;
            hgemm_epilogue_lds8_any<OUT_HALF>(p, acc, m0, n0, wr, wc, lane, ew);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            kt = 0;
            if (++c_i < ntiles) tile_origin(c_i, m0, n0);
        }
    }
}

// PERSISTENT 256x256x32 kernel (round 3): 8 waves as 2 x 4, 128 x 64 per wave, FOUR 32 KiB stages + 16 KiB of epilogue scratch.
// Why this shape: per K-step of 64 the 256x128 kernel above moves 48 KiB INTO a CU's LDS and reads 128 KiB of fragments out of it for
// 0.43 us of MFMA time -- the LDS pipe (about 250 B/clk for b128 reads, far less for writes: MI355X_MICROARCH.md) is the busier unit.
// A 256x256 block writes a third fewer operand bytes per flop, and a 128x64 wave tile reads a quarter fewer.  Everything else is the
// scheme of gemm_f16_dma3_kernel: one workgroup per CU streaming the K-steps of all its tiles, LDS-DMA into unpadded XOR-swizzled
// stages (64-byte rows: position p of row r holds chunk p ^ ((r >> 2) & 3), every 16-lane ds_read_b128 group covers the 16 slots of
// a 256-byte bank row), counted vmcnt waits + raw s_barrier, fragments double-buffered in registers.  One K-step = 32 deep:
//   reads A_hi(g) | MFMA A_lo(g) x B(g) | wait step g+1 + barrier | DMA(step g+4 -> stage g%4) | reads A_lo(g+1), B(g+1) | MFMA A_hi(g) x B(g)
constexpr int H4_BM = 256, H4_BN = 256, H4_BK = 32, H4_NST = 4, H4_STAGE = (H4_BM + H4_BN) * H4_BK;   // halfs per stage (32 KiB)
constexpr int H4_SMEM = H4_NST * H4_STAGE * 2 + H3_SCRATCH;                                             // 147,456 B
template <bool OUT_HALF>
__global__ __launch_bounds__(512, 2) void gemm_f16_dma4_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char h4_raw[];
    _Float16* smem = reinterpret_cast<_Float16*>(h4_raw);   // [stage][A 256 rows | W 256 rows][row * 32 + pos * 8], then the scratch
    const __half* A = reinterpret_cast<const __half*>(p.A);
    const __half* W = reinterpret_cast<const __half*>(p.W);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;                 // 2 x 4 waves, 128 x 64 each
    float* ew = reinterpret_cast<float*>(h4_raw + H4_NST * H4_STAGE * 2) + wave * (8 * 64);

    const int nbx = (p.N + H4_BN - 1) / H4_BN, nby = (p.M + H4_BM - 1) / H4_BM, total = nbx * nby;
    const int G = gridDim.x;
    const int ntiles = ((int)blockIdx.x < total) ? (total - (int)blockIdx.x + G - 1) / G : 0;
    if (ntiles == 0) return;                                   // block-uniform
    auto tile_origin = [&](const int i, int& m0, int& n0) {    // i-th tile of this workgroup (XCD-chunked, GROUP_M row tiles per column sweep)
        int bid = (int)blockIdx.x + i * G;
        const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        constexpr int GROUP_M = 4;
        const int per_group = GROUP_M * nbx, g = bid / per_group;
        const int gm = min(GROUP_M, nby - g * GROUP_M), in_g = bid - g * per_group;
        m0 = (g * GROUP_M + in_g % gm) * H4_BM;
        n0 = (in_g / gm) * H4_BN;
    };
    // DMA map: one wave instruction = 16 rows x 64 B.  A and W: 16 instructions each per K-step (wave w, pass j < 2: rows j*128 + w*16 ..).
    const int r_local = lane >> 2;
    const int csrc = (lane & 3) ^ ((r_local >> 2) & 3);
    // per-lane BYTE offsets of the four source rows (32 bits: launch_gemm_f16 checks the operands span < 4 GiB); the K-step's column
    // offset is wave-uniform and goes into the scalar base, so a request is `global_load_lds v_off, s[base]` and the running pointers
    // cost four VGPRs, not sixteen (the kernel is at the 256-register limit of two waves per SIMD)
    unsigned aoff[2], woff[2];
    auto set_tile_ptrs = [&](const int i) {
        int m0, n0;
        tile_origin(i, m0, n0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = min(m0 + j * 128 + wave * 16 + r_local, p.M - 1);
            const int bb = m / p.a_rows_per_batch;
            aoff[j] = (unsigned)(((long long)bb * p.a_batch_stride + (long long)(m - bb * p.a_rows_per_batch) * p.lda + csrc * 8) * 2);
            woff[j] = (unsigned)(((long long)min(n0 + j * 128 + wave * 16 + r_local, p.N - 1) * p.K + csrc * 8) * 2);
        }
    };
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto dma = [&](const int stage, const int kt) {
        _Float16* st = smem + stage * H4_STAGE;
        const char* ab = reinterpret_cast<const char*>(A) + (long long)kt * (H4_BK * 2);   // wave-uniform
        const char* wb = reinterpret_cast<const char*>(W) + (long long)kt * (H4_BK * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)(ab + aoff[j]), (lptr_t)(st + (j * 128 + wave * 16) * H4_BK), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(wb + woff[j]), (lptr_t)(st + H4_BM * H4_BK + (j * 128 + wave * 16) * H4_BK), 16, 0, 0);
        }
    };
    const int po = ((kq ^ ((l15 >> 2) & 3)) * 8);
    const int ra = (wr * 128 + l15) * H4_BK + po, rb = H4_BM * H4_BK + (wc * 64 + l15) * H4_BK + po;

    f32x4 acc_lo[4][4], acc_hi[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc_lo[i][j] = acc_hi[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / H4_BK;
    const int steps = ntiles * nk;
    int d_i = 0, d_kt = 0;
    set_tile_ptrs(0);
    auto dma_next = [&](const int stage) {
        dma(stage, d_kt);
        if (++d_kt == nk) {
            d_kt = 0;
            if (++d_i < ntiles) set_tile_ptrs(d_i);
        }
    };
    dma_next(0);
    if (steps > 1) dma_next(1);
    if (steps > 2) dma_next(2);
    if (steps > 3) dma_next(3);
    if (steps > 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (steps > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (steps > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    h8 fa0[4], fa1[4], fb[4];   // A fragments double-buffered (rows 0..63 / 64..127 of the wave tile), B single: 256 VGPRs are the limit
    auto frag_a = [&](h8 (&fa)[4], const int stage, const int half) {   // half 0: rows 0..63 of the wave tile, 1: rows 64..127
        const _Float16* st = smem + stage * H4_STAGE + ra + half * 64 * H4_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const h8*>(st + i * 16 * H4_BK);
    };
    auto frag_b2 = [&](const int stage, const int j0) {   // B fragments j0, j0 + 1 (columns 16 j0 .. 16 j0 + 31 of the wave tile)
        const _Float16* st = smem + stage * H4_STAGE + rb;
        fb[j0] = *reinterpret_cast<const h8*>(st + j0 * 16 * H4_BK);
        fb[j0 + 1] = *reinterpret_cast<const h8*>(st + (j0 + 1) * 16 * H4_BK);
    };
    auto mma_cols = [&](f32x4 (&acc)[4][4], const h8 (&fa)[4], const int j0) {   // 8 MFMAs: all four row tiles x column tiles j0, j0 + 1
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    frag_a(fa0, 0, 0);
    frag_b2(0, 0);
    frag_b2(0, 2);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): both edges into the loop header carry an empty LDS scoreboard
    int cur = 0, kt = 0, c_i = 0, m0, n0, skip = 0;
    tile_origin(0, m0, n0);
    for (int g = 0; g < steps; ++g) {
        const int nxt = (cur + 1) & 3;
        const bool more = g + 1 < steps;
        frag_a(fa1, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_cols(acc_lo, fa0, 0);
        mma_cols(acc_lo, fa0, 2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): my reads of stage `cur` are done (16 MFMAs were queued behind them)
        if (more) {
            if (skip) skip = 0;                                                      // waited for before the last epilogue's stores
            else if (g + 3 < steps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // step g+1 landed (steps g+2, g+3 may fly)
            else if (g + 2 < steps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (g + 4 < steps) dma_next(cur);
            frag_a(fa0, nxt, 0);
        }
        // the B fragments of step g+1 replace this step's pair by pair, each pair right behind the eight MFMAs that read it last
        __builtin_amdgcn_sched_barrier(0);
        mma_cols(acc_hi, fa1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) frag_b2(nxt, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_cols(acc_hi, fa1, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (more) frag_b2(nxt, 2);
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) alone (see gemm_f16_dma3_kernel)
        cur = nxt;
        if (++kt == nk) {   // this tile's K is complete: finish it under the flight of the next tile's first K-steps
            if (g + 4 < steps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // step g+2 landed; steps g+3, g+4 may fly
            else if (g + 3 < steps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            skip = 1;
            hgemm_epilogue_lds8_any<OUT_HALF>(p, acc_lo, m0, n0, 2 * wr, wc, lane, ew);
            hgemm_epilogue_lds8_any<OUT_HALF>(p, acc_hi, m0, n0, 2 * wr + 1, wc, lane, ew);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc_lo[i][j] = acc_hi[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            kt = 0;
            if (++c_i < ntiles) tile_origin(c_i, m0, n0);
        }
    }
}

// The same 256x256x32 tiles and stages on the guide's TWO-GROUP schedule ("256^2 8-phase template", cdna_hip_programming.md): waves 0-3
// (rows 0..127 of the block tile) and waves 4-7 (rows 128..255) -- one wave of each group per SIMD -- run the same sequence of blocks
//   R1: read A_lo, B of step g | M1: 16 MFMAs | R2: read A_hi, wait for step g+1's LDS-DMA, request step g+3 | M2: 16 MFMAs
// with a barrier after every block, group 1 ONE barrier behind group 0: in every interval one wave of a SIMD multiplies while the
// other reads and requests.  (gemm_f16_dma4_kernel's ablation: the request -> land -> barrier chain and the read -> MFMA -> barrier
// chain of a K-step are each ~1 us and ran back to back there, both waves of a SIMD wanting the MFMA pipe in the same window.)
// Fragments need no double buffering: a block's reads are consumed right after the next barrier, under the OTHER group's MFMAs.
// Ordering of the LDS-DMA: a wave waits (counted vmcnt) for its share of step g+1 in R2(g), before that block's barrier; the first
// read of step g+1 is group 0's R1(g+1), two barriers after group 0's wait and one after group 1's.  The stage refilled in R2(g) is
// (g+3) % 4 = (g-1) % 4, last read in group 1's R2(g-1), retired (lgkmcnt(0)) at the top of its M2(g-1) -- two barriers before group
// 0's R2(g), three before group 1's.  Every wave executes 4 * steps + 1 barriers (group 1: one before its first block, group 0: one
// after its last); nothing between two barriers depends on a wave-divergent condition.
template <bool OUT_HALF>
__global__ __launch_bounds__(512, 2) void gemm_f16_dma5_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char h4_raw[];
    _Float16* smem = reinterpret_cast<_Float16*>(h4_raw);   // [stage][A 256 rows | W 256 rows][row * 32 + pos * 8], then the scratch
    const __half* A = reinterpret_cast<const __half*>(p.A);
    const __half* W = reinterpret_cast<const __half*>(p.W);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;                 // 2 x 4 waves, 128 x 64 each
    float* ew = reinterpret_cast<float*>(h4_raw + H4_NST * H4_STAGE * 2) + wave * (8 * 64);

    const int nbx = (p.N + H4_BN - 1) / H4_BN, nby = (p.M + H4_BM - 1) / H4_BM, total = nbx * nby;
    const int G = gridDim.x;
    const int ntiles = ((int)blockIdx.x < total) ? (total - (int)blockIdx.x + G - 1) / G : 0;
    if (ntiles == 0) return;                                   // block-uniform
    auto tile_origin = [&](const int i, int& m0, int& n0) {    // i-th tile of this workgroup (XCD-chunked, GROUP_M row tiles per column sweep)
        int bid = (int)blockIdx.x + i * G;
        const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        constexpr int GROUP_M = 4;
        const int per_group = GROUP_M * nbx, g = bid / per_group;
        const int gm = min(GROUP_M, nby - g * GROUP_M), in_g = bid - g * per_group;
        m0 = (g * GROUP_M + in_g % gm) * H4_BM;
        n0 = (in_g / gm) * H4_BN;
    };
    // DMA map: one wave instruction = 16 rows x 64 B.  A and W: 16 instructions each per K-step (wave w, pass j < 2: rows j*128 + w*16 ..).
    const int r_local = lane >> 2;
    const int csrc = (lane & 3) ^ ((r_local >> 2) & 3);
    // per-lane BYTE offsets of the four source rows (32 bits: launch_gemm_f16 checks the operands span < 4 GiB); the K-step's column
    // offset is wave-uniform and goes into the scalar base, so a request is `global_load_lds v_off, s[base]` and the running pointers
    // cost four VGPRs, not sixteen (the kernel is at the 256-register limit of two waves per SIMD)
    unsigned aoff[2], woff[2];
    auto set_tile_ptrs = [&](const int i) {
        int m0, n0;
        tile_origin(i, m0, n0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = min(m0 + j * 128 + wave * 16 + r_local, p.M - 1);
            const int bb = m / p.a_rows_per_batch;
            aoff[j] = (unsigned)(((long long)bb * p.a_batch_stride + (long long)(m - bb * p.a_rows_per_batch) * p.lda + csrc * 8) * 2);
            woff[j] = (unsigned)(((long long)min(n0 + j * 128 + wave * 16 + r_local, p.N - 1) * p.K + csrc * 8) * 2);
        }
    };
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto dma = [&](const int stage, const int kt) {
        _Float16* st = smem + stage * H4_STAGE;
        const char* ab = reinterpret_cast<const char*>(A) + (long long)kt * (H4_BK * 2);   // wave-uniform
        const char* wb = reinterpret_cast<const char*>(W) + (long long)kt * (H4_BK * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((gptr_t)(ab + aoff[j]), (lptr_t)(st + (j * 128 + wave * 16) * H4_BK), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(wb + woff[j]), (lptr_t)(st + H4_BM * H4_BK + (j * 128 + wave * 16) * H4_BK), 16, 0, 0);
        }
    };
    const int po = ((kq ^ ((l15 >> 2) & 3)) * 8);
    const int ra = (wr * 128 + l15) * H4_BK + po, rb = H4_BM * H4_BK + (wc * 64 + l15) * H4_BK + po;

    f32x4 acc_lo[4][4], acc_hi[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc_lo[i][j] = acc_hi[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / H4_BK;
    const int steps = ntiles * nk;
    int d_i = 0, d_kt = 0;
    set_tile_ptrs(0);
    auto dma_next = [&](const int stage) {
        dma(stage, d_kt);
        if (++d_kt == nk) {
            d_kt = 0;
            if (++d_i < ntiles) set_tile_ptrs(d_i);
        }
    };
    dma_next(0);
    if (steps > 1) dma_next(1);
    if (steps > 2) dma_next(2);
    if (steps > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (steps > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // step 0 has landed for everybody
    asm volatile("" ::: "memory");
    if (wr == 1) {                         // group 1 runs one interval behind group 0
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    h8 fa[4], fb[4];
    auto frag_a = [&](const int stage, const int half) {   // half 0: rows 0..63 of the wave tile, 1: rows 64..127
        const _Float16* st = smem + stage * H4_STAGE + ra + half * 64 * H4_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const h8*>(st + i * 16 * H4_BK);
    };
    auto frag_b = [&](const int stage) {
        const _Float16* st = smem + stage * H4_STAGE + rb;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const h8*>(st + j * 16 * H4_BK);
    };
    auto mma = [&](f32x4 (&acc)[4][4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    int cur = 0, kt = 0, c_i = 0, m0, n0, skip = 0;
    tile_origin(0, m0, n0);
    for (int g = 0; g < steps; ++g) {
        // ---- R1: this step's first fragments (stage `cur` = step g landed and was published two barriers ago at the latest)
        frag_b(cur);
        frag_a(cur, 0);
        bar();
        // ---- M1
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        mma(acc_lo);
        bar();
        // ---- R2: second half of A; step g+1 must have landed before the barrier that ends this block; refill stage (g+3) % 4
        frag_a(cur, 1);
        if (g + 1 < steps) {
            if (skip) skip = 0;                                                      // waited for before the last epilogue's stores
            else if (g + 2 < steps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // step g+1 landed (step g+2 may fly)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (g + 3 < steps) dma_next((cur + 3) & 3);
        bar();
        // ---- M2
        __builtin_amdgcn_s_waitcnt(0xC07F);
        mma(acc_hi);
        bar();
        cur = (cur + 1) & 3;
        if (++kt == nk) {   // this tile's K is complete (no barrier in here: the other group runs on by at most one interval)
            // the wait R2(g+1) would take -- step g+2 landed, step g+3 may fly -- is taken now, ahead of the epilogue's stores
            if (g + 3 < steps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            skip = 1;
            hgemm_epilogue_lds8_any<OUT_HALF>(p, acc_lo, m0, n0, 2 * wr, wc, lane, ew);
            hgemm_epilogue_lds8_any<OUT_HALF>(p, acc_hi, m0, n0, 2 * wr + 1, wc, lane, ew);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc_lo[i][j] = acc_hi[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            kt = 0;
            if (++c_i < ntiles) tile_origin(c_i, m0, n0);
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // group 0's last barrier pairs with group 1's last block
}

hipError_t launch_gemm_f16(const GemmParams& p, bool out_half, hipStream_t s, int force_variant) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return hipSuccess;
    if ((p.K & 7) || (p.lda & 7) || (p.a_batch_stride & 7)) return hipErrorInvalidValue;
    if (p.a_rows_per_batch < 1 || p.c_rows_per_batch < 1) return hipErrorInvalidValue;   // divisors of the row maps below (as launch_gemm_f32 checks)
    const bool kv = p.epi == EPI_KV_HEADS;   // cross-K/V projection of an fp16 decoder engine: 128x128 kernels only (generic epilogues)
    if (kv && (p.N != 2 * p.kv_heads * HEAD_DIM || !p.C2 || p.resid || p.pos || p.act || (reinterpret_cast<uintptr_t>(p.C) & 15) ||
               (reinterpret_cast<uintptr_t>(p.C2) & 15) || (reinterpret_cast<uintptr_t>(p.bias) & 15)))
        return hipErrorInvalidValue;
    if (!kv && p.epi != EPI_ROWMAJOR) return hipErrorInvalidValue;
    static PerDeviceFlag attr_set;
    static const int bk = [] { const char* ev = tuning_env("WT_HGEMM_BK"); return (ev && (atoi(ev) == 32 || atoi(ev) == 64)) ? atoi(ev) : 32; }();
    if (!attr_set.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<true, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, hgemm_smem<64>());
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<false, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, hgemm_smem<64>());
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<true, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, hgemm_smem<32>());
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<false, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, hgemm_smem<32>());
        if (e != hipSuccess) return e;
        attr_set.set();
    }
    const int nbx = (p.N + HBN_ - 1) / HBN_, nby = (p.M + HBM_ - 1) / HBM_;
    const dim3 grid(nbx * nby);
    static const bool no_dma = tuning_env("WT_HGEMM_NO_DMA") != nullptr;  // A/B switch: register-staged kernel for every shape
    // A/B override: WT_HGEMM_VARIANT=2 forces the 128x128 two-stage kernel, 3 the persistent 256x128x64 kernel, 4 the persistent
    // 256x256x32 kernel (wherever their shape limits allow).
    // Default (measured in round 3, one box, TFLOP/s: 128x128 | 256x128 | 256x256; N / K / epilogue):
    //   M = 12000: 3072/1024 fp16-out 680 | 769 | 726;  4096/1024 GELU fp16-out 597 | 654 | 688;  2048/1024 fp16-out 716 | 834 | 717;
    //              1024/4096 804 | 787 | 823;  1024/1024 584 | 614 | 604
    //   M = 24000: 3072/1024 fp16-out 722 | 832 | 867;  4096/1024 GELU fp16-out 626 | 667 | 703;  2048/1024 fp16-out 741 | 866 | 905;
    //              1024/4096 823 | 938 | 764;  1024/1024 648 | 761 | 663
    // -> 256x256 for the wide fp16-output GEMMs (q|k|v, fc1) once its tiles fill the chip several times over; 256x128 for the other
    //    large launches (N = 1024 keeps the fp32 residual stream and has 4 column tiles of 256: 1.5 rounds of the chip); 128x128 below.
    static const int env_variant = tuning_env("WT_HGEMM_VARIANT") ? atoi(tuning_env("WT_HGEMM_VARIANT")) : 0;
    const int variant = force_variant ? force_variant : env_variant;   // force_variant: kernel tests reach every kernel at small sizes
    const bool dma_ok = !no_dma && (p.K % 64) == 0 && ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0;
    const bool use3 = kv ? false : variant == 3 ? (force_variant || p.M >= 1024) : variant == 2 || variant == 5 ? false : variant == 4 ? !out_half : (p.M >= 16384 || (p.M >= 8192 && p.N >= 2048));
    const long long tiles256 = (long long)((p.N + H4_BN - 1) / H4_BN) * ((p.M + H4_BM - 1) / H4_BM);
    const bool span32 = (long long)((p.M + p.a_rows_per_batch - 1) / p.a_rows_per_batch) * (p.a_batch_stride > 0 ? p.a_batch_stride : 0) * 2 +
                                (long long)p.a_rows_per_batch * p.lda * 2 < (1ll << 32) && (long long)p.N * p.K * 2 < (1ll << 32);
    const bool use4 = !kv && (p.K % H4_BK) == 0 && span32 &&
                      (variant == 4 || (variant == 0 && out_half && ((p.M >= 16384 && p.N >= 2048) || (p.M >= 8192 && p.N >= 4096)) && tiles256 >= 700));
    // the two-group kernel: at its best on long K in ONE round of tiles (fc2 / conv2 of a batch-8 fp16 encoder: 188 tiles, K = 3072 / 4096:
    // 1018 vs 825-890 TFLOP/s for the other kernels); with K = 1024 the per-tile epilogue and the fill of the last round decide, not the loop
    const bool use5 = !kv && (p.K % H4_BK) == 0 && span32 && (variant == 5 || (variant == 0 && p.K >= 2048 && tiles256 <= 256 && tiles256 >= 160));
    if (dma_ok && use5) {
        static PerDeviceFlag attr5;
        if (!attr5.get()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_dma5_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, H4_SMEM);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_dma5_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, H4_SMEM);
            if (e != hipSuccess) return e;
            attr5.set();
        }
        const dim3 grid5(tiles256 < 256 ? (unsigned)tiles256 : 256u);
        if (out_half) hipLaunchKernelGGL(gemm_f16_dma5_kernel<true>, grid5, dim3(512), H4_SMEM, s, p);
        else hipLaunchKernelGGL(gemm_f16_dma5_kernel<false>, grid5, dim3(512), H4_SMEM, s, p);
        return hipGetLastError();
    }
    if (dma_ok && use4 && out_half) {   // fp16-output form only: the fp32-output instantiation spills (36 bytes in its tile-switch path) and
                                        // no shape rule selects it -- a forced variant 4 with fp32 output takes the 256x128 kernel below
        static PerDeviceFlag attr4;
        if (!attr4.get()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_dma4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, H4_SMEM);
            if (e != hipSuccess) return e;
            attr4.set();
        }
        const dim3 grid4(tiles256 < 256 ? (unsigned)tiles256 : 256u);
        hipLaunchKernelGGL(gemm_f16_dma4_kernel<true>, grid4, dim3(512), H4_SMEM, s, p);
        return hipGetLastError();
    }
    if (dma_ok && use3 && span32) {
        static PerDeviceFlag attr3;
        if (!attr3.get()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_dma3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, H3_SMEM);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_dma3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, H3_SMEM);
            if (e != hipSuccess) return e;
            attr3.set();
        }
        const int tiles3 = ((p.N + H3_BN - 1) / H3_BN) * ((p.M + H3_BM - 1) / H3_BM);
        const dim3 grid3(tiles3 < 256 ? tiles3 : 256);   // persistent: one workgroup per CU, tiles blockIdx, blockIdx + 256, ...
        if (out_half) hipLaunchKernelGGL(gemm_f16_dma3_kernel<true>, grid3, dim3(512), H3_SMEM, s, p);
        else hipLaunchKernelGGL(gemm_f16_dma3_kernel<false>, grid3, dim3(512), H3_SMEM, s, p);
        return hipGetLastError();
    }
    if (dma_ok && span32) {
        if (out_half) hipLaunchKernelGGL(gemm_f16_dma_kernel<true>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(gemm_f16_dma_kernel<false>, grid, dim3(256), 0, s, p);
        return hipGetLastError();
    }
    if (bk == 64) {
        if (out_half) hipLaunchKernelGGL((gemm_f16_kernel<true, 64>), grid, dim3(256), hgemm_smem<64>(), s, p);
        else hipLaunchKernelGGL((gemm_f16_kernel<false, 64>), grid, dim3(256), hgemm_smem<64>(), s, p);
    } else {
        if (out_half) hipLaunchKernelGGL((gemm_f16_kernel<true, 32>), grid, dim3(256), hgemm_smem<32>(), s, p);
        else hipLaunchKernelGGL((gemm_f16_kernel<false, 32>), grid, dim3(256), hgemm_smem<32>(), s, p);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ fp16 encoder attention
// softmax(Q K^T / 8) V with fp16 operands on v_mfma_f32_32x32x16_f16, fp32 scores / softmax / accumulators (the reference
// forces fp32 scores in fp16 builds too, model.py:292-295).  Same transposed formulation as the fp32 kernel:
//   S^T[key][query] = K . Q^T  -> the query sits on the lane, softmax statistics are lane-local;
//   O^T[dv][query] += V^T . P^T -> P goes from the S^T accumulator registers straight into the next MFMA's B operand
//                                  (registers 8t..8t+7 = k-step t; their key order 16t + 8(j>>2) + 4h + (j&3) is matched
//                                  by the V^T fragments' key order).
// qkv fp16 [B*S][3d] -> ctx fp16 [B*S][d].  128 queries per block (4 waves x 32), 64-key tiles, double-buffered LDS:
// K rows padded to 144 B (conflict-free ds_read_b128); V stays ROW-major as it comes from memory (192-byte rows) and the V^T fragments are
// read with ds_read_b64_tr_b16: a 16-lane group takes 4 keys x 16 head dims, a lane receives the four keys of its own head dim; the four
// key rows of a block sit on the four 16-bank quarters (48 q mod 64): conflict-free.  Until late round 4 the tile was stored transposed
// with 16 two-byte stores per thread and tile: SQ_LDS_BANK_CONFLICT = 39 % of the LDS-active cycles (profiles/r04d_enc_attn_pmc_counters.txt).
typedef float f32x16h __attribute__((ext_vector_type(16)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
constexpr int HA_BQ = 128, HA_BKV = 64, HA_KLD = 72, HA_VLD = 96;           // strides in halfs
constexpr int HA_TILE = HA_BKV * HA_KLD + HA_BKV * HA_VLD;                   // halfs per (K, V) stage
constexpr int HA_SMEM = 2 * HA_TILE * 2;                                     // 43,008 B

__global__ __launch_bounds__(256, 2) void enc_attn_f16_kernel(const __half* __restrict__ qkv, __half* __restrict__ ctx, int S, int H) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ha_raw[];
    _Float16* smem = reinterpret_cast<_Float16*>(ha_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    int b, h, q0;
    {   // XCD-chunked head-major order, as in enc_attn_kernel: a head's K / V land in one L2 instead of eight
        const int nqb = (S + HA_BQ - 1) / HA_BQ, total = (int)gridDim.x;
        const int L = blockIdx.x, xcd = L & 7, idx = L >> 3, q = total >> 3, r = total & 7;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int bh = v / nqb;
        b = bh / H; h = bh % H; q0 = (v - bh * nqb) * HA_BQ;
    }
    const int d = H * HEAD_DIM, ld = 3 * d;
    const _Float16* base = reinterpret_cast<const _Float16*>(qkv) + (size_t)b * S * ld + h * HEAD_DIM;

    const int qrow = q0 + wave * 32 + l31;
    h8 qf[4];  // Q[query][16s + 8h .. +7], pre-scaled by 64^-0.5 (exact in fp16)
    {
        const _Float16* qp = base + (size_t)min(qrow, S - 1) * ld + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = *reinterpret_cast<const h8*>(qp + 16 * s);
            qf[s] *= (_Float16)0.125f;
        }
    }
    // staging: 64 keys x 8 sixteen-byte chunks per matrix; thread -> chunk c8 (8 head dims), keys r0 and r0 + 32
    const int c8 = tid & 7, r0 = tid >> 3;
    const _Float16* gbase = base + c8 * 8;
    h8 rk0, rk1, rv0, rv1;
#define HA_GLOAD(kv0_)                                                                         \
    do {                                                                                       \
        const _Float16* rp0 = gbase + (size_t)min((kv0_) + r0, S - 1) * ld;                    \
        const _Float16* rp1 = gbase + (size_t)min((kv0_) + r0 + 32, S - 1) * ld;               \
        rk0 = *reinterpret_cast<const h8*>(rp0 + d); rv0 = *reinterpret_cast<const h8*>(rp0 + 2 * d); \
        rk1 = *reinterpret_cast<const h8*>(rp1 + d); rv1 = *reinterpret_cast<const h8*>(rp1 + 2 * d); \
    } while (0)
#define HA_LSTORE(buf_)                                                                        \
    do {                                                                                       \
        _Float16* Ks_ = smem + (buf_) * HA_TILE;                                               \
        _Float16* Vs_ = Ks_ + HA_BKV * HA_KLD;                                                 \
        *reinterpret_cast<h8*>(Ks_ + r0 * HA_KLD + c8 * 8) = rk0;                              \
        *reinterpret_cast<h8*>(Ks_ + (r0 + 32) * HA_KLD + c8 * 8) = rk1;                       \
        *reinterpret_cast<h8*>(Vs_ + r0 * HA_VLD + c8 * 8) = rv0;                              \
        *reinterpret_cast<h8*>(Vs_ + (r0 + 32) * HA_VLD + c8 * 8) = rv1;                       \
    } while (0)
    // transposed V reads: lane 32 hh + 16 g + 4 q + p supplies row (4 hh + q), head dims 16 g + 4 p .. + 3 of a block and receives the four
    // keys 4 hh .. 4 hh + 3 of head dim 16 g + 4 q + p = l31 (cdna guide T10)
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* tr_ptr_t;
    const int v_tr = (4 * hh + ((lane >> 2) & 3)) * HA_VLD + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    auto tr4 = [&](const _Float16* ptr) { return __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)ptr)); };

    f32x16h o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int ntiles = (S + HA_BKV - 1) / HA_BKV;
    HA_GLOAD(0);
    HA_LSTORE(0);
    __syncthreads();
    // One 64-key tile; LAST = the final (possibly ragged) one, peeled: hipcc if-converts the ragged-tile mask into compares and selects
    // executed on EVERY tile otherwise (as it did in the fp32 kernels)
    auto tile = [&](auto LAST_T, const int t) {
        constexpr bool LAST = decltype(LAST_T)::value;
        const int cur = t & 1, kv0 = t * HA_BKV;
        if (!LAST) HA_GLOAD(kv0 + HA_BKV);
        const _Float16* Ks = smem + cur * HA_TILE;
        const _Float16* Vs = Ks + HA_BKV * HA_KLD;

        f32x16h s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
        const _Float16* kp = Ks + l31 * HA_KLD + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const h8 k0 = *reinterpret_cast<const h8*>(kp + 16 * s);
            const h8 k1 = *reinterpret_cast<const h8*>(kp + 32 * HA_KLD + 16 * s);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[s], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[s], s1, 0, 0, 0);
        }
        if (LAST && kv0 + HA_BKV > S) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (key >= S) s0[r] = -INFINITY;
                if (key + 32 >= S) s1[r] = -INFINITY;
            }
        }
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __expf(s0[r] - m_new);
            s1[r] = __expf(s1[r] - m_new);
            ps += s0[r] + s1[r];
        }
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] *= alpha;
            o1[r] *= alpha;
        }
        // P.V: k-step (kt, t2) covers keys kt*32 + 16*t2 .. +15; B fragment = accumulator registers 8*t2 .. 8*t2+7
        const _Float16* vp = Vs + v_tr;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                h8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (_Float16)(kt == 0 ? s0[8 * t2 + j] : s1[8 * t2 + j]);
                const int kb = kt * 32 + 16 * t2;
                const h4 a00 = tr4(vp + kb * HA_VLD), a01 = tr4(vp + (kb + 8) * HA_VLD);
                const h4 a10 = tr4(vp + kb * HA_VLD + 32), a11 = tr4(vp + (kb + 8) * HA_VLD + 32);
                const h8 va0 = {a00[0], a00[1], a00[2], a00[3], a01[0], a01[1], a01[2], a01[3]};
                const h8 va1 = {a10[0], a10[1], a10[2], a10[3], a11[0], a11[1], a11[2], a11[3]};
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(va0, pf, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(va1, pf, o1, 0, 0, 0);
            }
        }
        if (!LAST) HA_LSTORE(cur ^ 1);
        __syncthreads();
    };
    for (int t = 0; t + 1 < ntiles; ++t) tile(std::false_type{}, t);
    tile(std::true_type{}, ntiles - 1);
#undef HA_GLOAD
#undef HA_LSTORE
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32));
    if (qrow < S) {
        _Float16* op = reinterpret_cast<_Float16*>(ctx) + ((size_t)b * S + qrow) * d + h * HEAD_DIM + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const h4 v0 = {(_Float16)(o0[4 * g] * inv), (_Float16)(o0[4 * g + 1] * inv), (_Float16)(o0[4 * g + 2] * inv), (_Float16)(o0[4 * g + 3] * inv)};
            const h4 v1 = {(_Float16)(o1[4 * g] * inv), (_Float16)(o1[4 * g + 1] * inv), (_Float16)(o1[4 * g + 2] * inv), (_Float16)(o1[4 * g + 3] * inv)};
            *reinterpret_cast<h4*>(op + 8 * g) = v0;
            *reinterpret_cast<h4*>(op + 32 + 8 * g) = v1;
        }
    }
}

hipError_t launch_encoder_attention_f16(const void* qkv, void* ctx, int B, int S, int H, hipStream_t s) {
    dim3 grid(((S + HA_BQ - 1) / HA_BQ) * H * B);
    hipLaunchKernelGGL(enc_attn_f16_kernel, grid, dim3(256), HA_SMEM, s, (const __half*)qkv, (__half*)ctx, S, H);
    return hipGetLastError();
}

}  // namespace wt
