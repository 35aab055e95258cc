// Encoder-side HIP kernels for gfx950 (MI355X): mel transpose, LayerNorm, fp32 MFMA GEMM with fused
// bias/GELU/residual/pos-emb epilogues (also used for the two Conv1d as implicit GEMMs and for the
// cross-attention K/V projection), and a flash-style fp32 MFMA self-attention.
//
// Reference semantics: tensorrt_llm/models/whisper/model.py:90-111 (WhisperEncoder.forward),
// :48-66 (WhisperEncoderLayer), layers/attention.py:216-350; numerics follow the bundled HF oracle
// modeling_whisper.py:569-593, :632-641, :992-1011 (erf GELU, q scaled before QK^T).
#include "wt_common.h"
#include <type_traits>

#include <stdlib.h>

namespace wt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// fp32 -> three bf16 planes: b1 = rn(x), b2 = rn(x - b1), b3 = rn(x - b1 - b2); x = b1 + b2 + b3 to 2^-27 relative (wt_common.h: launch_gemm_x3)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3(const float x, __bf16& b1, __bf16& b2, __bf16& b3) {
    b1 = (__bf16)x;
    const float r1 = x - (float)b1;
    b2 = (__bf16)r1;
    b3 = (__bf16)(r1 - (float)b2);
}
__device__ __forceinline__ void store_split3(float* base, const long long off, const long long plane, const float x) {
    __bf16 b1, b2, b3;
    split3(x, b1, b2, b3);
    __bf16* o = reinterpret_cast<__bf16*>(base) + off;
    o[0] = b1;
    o[plane] = b2;
    o[2 * plane] = b3;
}



// ------------------------------------------------------------------------------------------------ mel transpose
// mel [B][C][F] (time contiguous, the layout of run.py's `input_features`) -> melT [B][F+2][C], row = time+1,
// rows 0 and F+1 are the conv zero padding (written here).  Makes conv1 an implicit GEMM with K = 3*C, lda = C.
// SPLIT: melT goes out as three bf16 planes (conv1's A operand for launch_gemm_x3; plane p at + p * plane elements)
template <bool SPLIT>
__global__ __launch_bounds__(256) void mel_transpose_kernel(const float* __restrict__ mel, float* __restrict__ melT,
                                                            int C, int F, size_t plane) {
    __shared__ float tile[128][65];
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        int c = i >> 6, tl = i & 63, t = t0 + tl;
        tile[c][tl] = t < F ? mel[((size_t)b * C + c) * F + t] : 0.f;
    }
    __syncthreads();
    if (SPLIT) {
        __bf16* dst = reinterpret_cast<__bf16*>(melT) + (size_t)b * (F + 2) * C;
        for (int i = threadIdx.x; i < C * 64; i += 256) {
            int tl = i / C, c = i - tl * C, t = t0 + tl;
            if (t < F) {
                __bf16 b1, b2, b3;
                split3(tile[c][tl], b1, b2, b3);
                __bf16* o = dst + (size_t)(t + 1) * C + c;
                o[0] = b1; o[plane] = b2; o[2 * plane] = b3;
            }
        }
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < 3 * C; i += 256) {
                const int pl = i / C, c = i - pl * C;
                dst[pl * plane + c] = (__bf16)0.f;
                dst[pl * plane + (size_t)(F + 1) * C + c] = (__bf16)0.f;
            }
        return;
    }
    float* dst = melT + (size_t)b * (F + 2) * C;
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        int tl = i / C, c = i - tl * C, t = t0 + tl;
        if (t < F) dst[(size_t)(t + 1) * C + c] = tile[c][tl];
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < C; i += 256) {
            dst[i] = 0.f;
            dst[(size_t)(F + 1) * C + i] = 0.f;
        }
}

hipError_t launch_mel_transpose(const float* mel, float* melT, int B, int n_mels, int frames, hipStream_t s) {
    if (n_mels > 128) return hipErrorInvalidValue;
    dim3 grid((frames + 63) / 64, B);
    hipLaunchKernelGGL(mel_transpose_kernel<false>, grid, dim3(256), 0, s, mel, melT, n_mels, frames, (size_t)0);
    return hipGetLastError();
}
hipError_t launch_mel_transpose_split(const float* mel, void* planes, size_t plane_stride, int B, int n_mels, int frames, hipStream_t s) {
    if (n_mels > 128) return hipErrorInvalidValue;
    dim3 grid((frames + 63) / 64, B);
    hipLaunchKernelGGL(mel_transpose_kernel<true>, grid, dim3(256), 0, s, mel, reinterpret_cast<float*>(planes), n_mels, frames, plane_stride);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// One wave per row, row held in registers (d <= 1280, d % 4 == 0), two-pass mean/variance in fp32, eps 1e-5
// (layers/normalization.py:10; nn.LayerNorm default).
__device__ __forceinline__ float wave_allreduce_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// SPLIT: the normalised row goes out as three bf16 planes (the A operand of launch_gemm_x3; `y` = plane 0, plane p at + p * plane elements)
template <bool SPLIT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y, int rows,
                                                        int d, size_t plane) {
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
    const float mean = wave_allreduce_sum(s) / d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int c = lane + 64 * i;
        if (c < n4) {
            float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
            q += (a * a + bb * bb) + (cc * cc + dd * dd);
        }
    }
    const float rstd = rsqrtf(wave_allreduce_sum(q) / d + 1e-5f);
    float4* yr = reinterpret_cast<float4*>(y + (size_t)row * d);
    const float4* w4 = reinterpret_cast<const float4*>(w);
    const float4* b4 = reinterpret_cast<const float4*>(b);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int c = lane + 64 * i;
        if (c < n4) {
            float4 g = w4[c], be = b4[c], o;
            o.x = (v[i].x - mean) * rstd * g.x + be.x;
            o.y = (v[i].y - mean) * rstd * g.y + be.y;
            o.z = (v[i].z - mean) * rstd * g.z + be.z;
            o.w = (v[i].w - mean) * rstd * g.w + be.w;
            if (SPLIT) {
                __bf16* o0 = reinterpret_cast<__bf16*>(y) + (size_t)row * d + 4 * c;
                bf16x4 p1, p2, p3;
                __bf16 b1, b2, b3;
                split3(o.x, b1, b2, b3); p1[0] = b1; p2[0] = b2; p3[0] = b3;
                split3(o.y, b1, b2, b3); p1[1] = b1; p2[1] = b2; p3[1] = b3;
                split3(o.z, b1, b2, b3); p1[2] = b1; p2[2] = b2; p3[2] = b3;
                split3(o.w, b1, b2, b3); p1[3] = b1; p2[3] = b2; p3[3] = b3;
                *reinterpret_cast<bf16x4*>(o0) = p1;
                *reinterpret_cast<bf16x4*>(o0 + plane) = p2;
                *reinterpret_cast<bf16x4*>(o0 + 2 * plane) = p3;
            } else {
                yr[c] = o;
            }
        }
    }
}

hipError_t launch_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int d, hipStream_t s) {
    if (d > 1280 || (d & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, y, rows, d, (size_t)0);
    return hipGetLastError();
}
hipError_t launch_layernorm_split(const float* x, const float* w, const float* b, void* planes, size_t plane_stride, int rows, int d, hipStream_t s) {
    if (d > 1280 || (d & 3) || (plane_stride & 3) || ((uintptr_t)planes & 7)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, reinterpret_cast<float*>(planes), rows, d, plane_stride);
    return hipGetLastError();
}

// Epilogue shared by the fp32 GEMM kernels (and gemm_x3_kernel, whose 32x32x16 bf16 MFMAs leave the same accumulator layout).  C/D layout
// of the 32x32 MFMA: col = lane & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
template <int TJ, int TI = 2>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const f32x16 (&acc)[TI][TJ], const int m0, const int n0, const int wr,
                                              const int wc, const int l31, const int hh) {
    // Row-dependent addressing (batch split, output row pointer) is computed once per row -- the batch index of the
    // wave's first row by ONE wave-uniform division, the rest by carry -- and the residual / position rows of a group of
    // four rows are loaded together before they are used: in a one-round launch nothing overlaps the epilogue, and the
    // per-element division + load-wait-use chain it replaces cost about a quarter of the K = 1024 launches.
    int nn[2];
    float bv[2];
    bool nok[2];
    int kv_which[2] = {0, 0}, kv_h[2] = {0, 0}, kv_j[2] = {0, 0};
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
        nn[tj] = n0 + wc * (32 * TJ) + tj * 32 + l31;
        nok[tj] = nn[tj] < p.N;
        bv[tj] = (p.bias && nok[tj]) ? p.bias[nn[tj]] : 0.f;
        if (p.epi == EPI_KV_HEADS) {
            const int dkv = p.kv_heads * HEAD_DIM;
            kv_which[tj] = nn[tj] / dkv;
            const int r2 = nn[tj] - kv_which[tj] * dkv;
            kv_h[tj] = r2 / HEAD_DIM;
            kv_j[tj] = r2 - kv_h[tj] * HEAD_DIM;
        }
    }
    const int mw = m0 + wr * 64;                       // first row of this wave (wave-uniform)
    const int cb_w = mw / p.c_rows_per_batch, cr_w = mw - cb_w * p.c_rows_per_batch;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            long long offs[4];
            int crs[4];
            bool mok[4];
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const int o = ti * 32 + ri + 8 * rq + 4 * hh;
                mok[ri] = mw + o < p.M;
                int cb = cb_w, cr = cr_w + o;
                while (cr >= p.c_rows_per_batch) { cr -= p.c_rows_per_batch; ++cb; }
                crs[ri] = cr;
                offs[ri] = p.epi == EPI_ROWMAJOR ? (long long)cb * p.c_batch_stride + (long long)cr * p.ldc
                                                 : (long long)cb * p.kv_heads * p.kv_cap * HEAD_DIM + (long long)(cr + p.kv_seq_off) * HEAD_DIM;
            }
            float extra[4][2];
#pragma unroll
            for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                for (int tj = 0; tj < TJ; ++tj)
                    extra[ri][tj] = (p.resid && mok[ri] && nok[tj]) ? p.resid[offs[ri] + nn[tj]] : 0.f;
            float posv[4][2];
#pragma unroll
            for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                for (int tj = 0; tj < TJ; ++tj)
                    posv[ri][tj] = (p.pos && mok[ri] && nok[tj]) ? p.pos[(long long)crs[ri] * p.N + nn[tj]] : 0.f;
#pragma unroll
            for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                for (int tj = 0; tj < TJ; ++tj) {
                    if (!(mok[ri] && nok[tj])) continue;
                    float v = acc[ti][tj][rq * 4 + ri] + bv[tj];
                    if (p.act) v = gelu_erf(v);
                    v += posv[ri][tj];
                    if (p.epi == EPI_ROWMAJOR) {
                        if (p.out_split) store_split3(p.C, offs[ri] + nn[tj], p.c_plane, v + extra[ri][tj]);
                        else p.C[offs[ri] + nn[tj]] = v + extra[ri][tj];
                    } else {
                        float* base = kv_which[tj] ? p.C2 : p.C;
                        base[offs[ri] + (long long)kv_h[tj] * p.kv_cap * HEAD_DIM + kv_j[tj]] = v;
                    }
                }
        }
    }
}

// Fast epilogue for a wave whose 64x64 sub-tile lies wholly inside C: no bounds masks, the optional operands are compile-time, and the
// residual / position loads run two row groups ahead of their use in straight-line code, so the only waits are counted vmcnt(N) for
// loads -- never for the stores.  (gfx9 has ONE counter for loads and stores: in the generic epilogue above every `p.resid ? load : 0`
// becomes a branch whose join waits vmcnt(0), i.e. for the round trip of all stores issued so far; measured per workgroup, its stores
// took 13 us to ISSUE alone on a CU and 60 us beside other workgroups' K loops -- a third of a K = 1024 tile's life.)
template <bool RESID, bool ACT, bool POS, bool KV, int TJ, bool SPLIT = false, int TI = 2>
__device__ __forceinline__ void gemm_epilogue_fast(const GemmParams& p, const f32x16 (&acc)[TI][TJ], const int m0, const int n0, const int wr,
                                                   const int wc, const int l31, const int hh) {
    const int mw = m0 + wr * 64, rpb = p.c_rows_per_batch;
    const int cb_w = mw / rpb, cr_w = mw - cb_w * rpb;                 // wave-uniform: one division
    int nn[2], col[2];
    float bv[2];
    float* base[2];
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
        nn[tj] = n0 + wc * (32 * TJ) + tj * 32 + l31;
        bv[tj] = p.bias ? p.bias[nn[tj]] : 0.f;
        if (KV) {
            const int dkv = p.kv_heads * HEAD_DIM, which = nn[tj] / dkv, r2 = nn[tj] - which * dkv, h = r2 / HEAD_DIM;
            col[tj] = h * p.kv_cap * HEAD_DIM + (r2 - h * HEAD_DIM);
            base[tj] = which ? p.C2 : p.C;
        } else {
            col[tj] = nn[tj];
            base[tj] = p.C;
        }
    }
    const int row_stride = KV ? HEAD_DIM : p.ldc;
    const int batch_stride = KV ? p.kv_heads * p.kv_cap * HEAD_DIM : (int)p.c_batch_stride;
    const int row_add = KV ? p.kv_seq_off : 0;
    // group g = (ti, rq): rows o = ti*32 + 8*rq + 4*hh + ri, ri < 4
    auto row_of = [&](const int g, const int ri, int& cr) -> int {
        const int o = (g >> 2) * 32 + 8 * (g & 3) + 4 * hh + ri;
        cr = cr_w + o;
        const bool over = cr >= rpb;                                  // a 64-row span crosses at most one batch boundary (rpb >= 64)
        cr -= over ? rpb : 0;
        return (cb_w + (over ? 1 : 0)) * batch_stride + (cr + row_add) * row_stride;
    };
    float rv[2][8], pv[2][8];
    auto fetch = [&](const int g) {
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
            int cr;
            const int off = row_of(g, ri, cr);
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj) {
                if (RESID) rv[g & 1][ri * 2 + tj] = p.resid[off + col[tj]];
                if (POS) pv[g & 1][ri * 2 + tj] = p.pos[cr * p.N + nn[tj]];
            }
        }
    };
    if (RESID || POS) { fetch(0); fetch(1); }
#pragma unroll
    for (int g = 0; g < 4 * TI; ++g) {
        float v[8];
#pragma unroll
        for (int ri = 0; ri < 4; ++ri)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj) {
                float x = acc[g >> 2][tj][(g & 3) * 4 + ri] + bv[tj];
                if (ACT) x = gelu_erf(x);
                if (POS) x += pv[g & 1][ri * 2 + tj];
                if (RESID) x += rv[g & 1][ri * 2 + tj];
                v[ri * 2 + tj] = x;
            }
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
            int cr;
            const int off = row_of(g, ri, cr);
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj) {
                if (SPLIT) store_split3(base[tj], off + col[tj], p.c_plane, v[ri * 2 + tj]);
                else base[tj][off + col[tj]] = v[ri * 2 + tj];
            }
        }
        if ((RESID || POS) && g + 2 < 4 * TI) fetch(g + 2);
    }
}

// Epilogue dispatch: the fast form for interior sub-tiles of the operand combinations the engines use, the generic one otherwise.
template <int TJ, int TI = 2>
__device__ __forceinline__ void gemm_epilogue_any(const GemmParams& p, const f32x16 (&acc)[TI][TJ], const int m0, const int n0, const int wr,
                                                  const int wc, const int l31, const int hh) {
    const bool interior = m0 + wr * 64 + 32 * TI <= p.M && n0 + wc * (32 * TJ) + 32 * TJ <= p.N && p.c_rows_per_batch >= 64 && p.epi_fits32;
    if (p.out_split) {   // three bf16 planes out (the A operand of the next launch_gemm_x3): fc1's GELU output
        if (interior && p.epi == EPI_ROWMAJOR && !p.pos && !p.resid && p.act) return gemm_epilogue_fast<false, true, false, false, TJ, true, TI>(p, acc, m0, n0, wr, wc, l31, hh);
        if (interior && p.epi == EPI_ROWMAJOR && !p.pos && !p.resid && !p.act) return gemm_epilogue_fast<false, false, false, false, TJ, true, TI>(p, acc, m0, n0, wr, wc, l31, hh);
        return gemm_epilogue<TJ, TI>(p, acc, m0, n0, wr, wc, l31, hh);
    }
    if (interior && p.epi == EPI_ROWMAJOR && !p.pos) {
        if (p.resid && !p.act) return gemm_epilogue_fast<true, false, false, false, TJ, false, TI>(p, acc, m0, n0, wr, wc, l31, hh);
        if (!p.resid && p.act) return gemm_epilogue_fast<false, true, false, false, TJ, false, TI>(p, acc, m0, n0, wr, wc, l31, hh);
        if (!p.resid && !p.act) return gemm_epilogue_fast<false, false, false, false, TJ, false, TI>(p, acc, m0, n0, wr, wc, l31, hh);
    } else if (interior && p.epi == EPI_ROWMAJOR && p.pos && p.act && !p.resid) {
        return gemm_epilogue_fast<false, true, true, false, TJ, false, TI>(p, acc, m0, n0, wr, wc, l31, hh);
    } else if (interior && p.epi == EPI_KV_HEADS && !p.pos && !p.act && !p.resid) {
        return gemm_epilogue_fast<false, false, false, true, TJ, false, TI>(p, acc, m0, n0, wr, wc, l31, hh);
    }
    gemm_epilogue<TJ, TI>(p, acc, m0, n0, wr, wc, l31, hh);
}

// ------------------------------------------------------------------------------------------------ fp32 MFMA GEMM
// 128x128 block tile, 4 waves in a 2x2 grid, each wave 2x2 tiles of v_mfma_f32_32x32x2_f32 (exact fp32,
// 64 FLOP/clk/SIMD), K-step 16 (32 selectable).  A and W tiles are staged global -> registers -> LDS (rows padded by 4 floats so the
// ds_read_b128 fragment reads are bank-conflict free), double-buffered, one barrier per K-step.
// Fragment trick: one 32x32x2 MFMA takes k = {k0, k1} from lane halves 0/1.  Each lane reads a float4
// A[row][8q+4h .. +3] and issues 4 MFMAs with element j, so lane half h covers k = 8q+4h+j; A and W use the
// same k assignment, so the sum over k is complete and no repacking is needed.
constexpr int GBM = 128, GBN = 128;
template <int GBK>
constexpr int gemm_smem_bytes() { return 2 * 2 * GBM * (GBK + 4) * (int)sizeof(float); }  // 73,728 B at BK=32, 40,960 B at BK=16

template <int GBK>
__global__ __launch_bounds__(256, GBK == 32 ? 2 : 3) void gemm_f32_kernel(const GemmParams p) {
    constexpr int GLD = GBK + 4;
    constexpr int NC4 = GBK / 4;           // float4 chunks per staged row
    constexpr int RPP = 256 / NC4;         // rows covered per staging pass
    constexpr int NPASS = GBM / RPP;       // passes per matrix
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
    // contiguous chunk of the tile list; inside a chunk walk groups of GROUP_M row-tiles column by column so
    // the co-resident blocks of one XCD reuse a few A row-panels and W column-panels out of its private L2.
    const int nbx = (p.N + GBN - 1) / GBN, nby = (p.M + GBM - 1) / GBM, total = nbx * nby;
    int bid = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * nbx, g = bid / per_group;
    const int gm = min(GROUP_M, nby - g * GROUP_M), in_g = bid - g * per_group;
    const int by = g * GROUP_M + in_g % gm, bx = in_g / gm;
    const int m0 = by * GBM, n0 = bx * GBN;

    // staging map: 128 rows x NC4 float4; thread -> column c4, rows r0 + RPP*i
    const int c4 = tid % NC4, r0 = tid / NC4;
    const float* aptr[NPASS];
    const float* wptr[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        int m = min(m0 + r0 + RPP * i, p.M - 1);
        int bb = m / p.a_rows_per_batch;
        aptr[i] = p.A + (long long)bb * p.a_batch_stride + (long long)(m - bb * p.a_rows_per_batch) * p.lda + c4 * 4;
        int n = min(n0 + r0 + RPP * i, p.N - 1);
        wptr[i] = p.W + (long long)n * p.K + c4 * 4;
    }
    float4 ra[NPASS], rw[NPASS];
    auto gload = [&](int kt) {
        const int k = kt * GBK + c4 * 4;
        const bool ok = k < p.K;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            ra[i] = ok ? *reinterpret_cast<const float4*>(aptr[i] + kt * GBK) : make_float4(0.f, 0.f, 0.f, 0.f);
            rw[i] = ok ? *reinterpret_cast<const float4*>(wptr[i] + kt * GBK) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf) {
        float* As = smem + buf * (2 * GBM * GLD);
        float* Ws = As + GBM * GLD;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            *reinterpret_cast<float4*>(As + (r0 + RPP * i) * GLD + c4 * 4) = ra[i];
            *reinterpret_cast<float4*>(Ws + (r0 + RPP * i) * GLD + c4 * 4) = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + GBK - 1) / GBK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const float* As = smem + cur * (2 * GBM * GLD) + (wr * 64 + l31) * GLD + 4 * hh;
        const float* Ws = smem + cur * (2 * GBM * GLD) + GBM * GLD + (wc * 64 + l31) * GLD + 4 * hh;
#pragma unroll
        for (int q = 0; q < GBK / 8; ++q) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + 8 * q);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + 32 * GLD + 8 * q);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(Ws + 8 * q);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(Ws + 32 * GLD + 8 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    gemm_epilogue_any<2>(p, acc, m0, n0, wr, wc, l31, hh);
}

// The same GEMM with the tiles staged by LDS-DMA (global_load_lds_dwordx4: global -> LDS, no VGPR staging, no ds_write),
// used when K is a multiple of 16.  One wave instruction writes 1 KiB linearly (wave-uniform base + lane * 16 B), so the LDS
// tile is UNPADDED, [row][4 chunks of 4 floats], and bank conflicts of the fragment reads are avoided by an XOR swizzle
// applied on both sides (cdna guide §5.4 rule 21): the lane that fills LDS chunk position `pos` of row r fetches global
// chunk pos ^ ((r >> 2) & 3); a fragment read of chunk c of row r reads position c ^ ((r >> 2) & 3).  Rows 4a+b, a,b < 4,
// then cover all 16 sixteen-byte bank groups for every c.  Three 16 KiB stages (48 KiB -> three workgroups per CU; four waves per
// SIMD sustain LESS MFMA throughput than three, tools/probes/mfma_rate.hip: 125 vs 155 TFLOP/s of pure v_mfma_f32_32x32x2_f32):
// the DMA of tile kt+2 is issued right after the barrier that opens tile kt, and the wait at the top of a step is counted
// (vmcnt(4): tile kt+1's four DMA instructions may still be in flight).
// BN = 64: a 128x64 block tile (wave tile 64x32, W stage half used) for launches whose 128x128 tiling fills the chip's 768 workgroup
// slots badly (small.en's N = 768: 564 tiles) -- half-size tiles fill the last round better; same stages, same LDS footprint
// (three workgroups per CU either way).
// KSUB = 2: a stage holds TWO such 16-wide K tiles (32 KiB), two stages, two workgroups per CU: one wait + barrier per 32 of K instead
// of per 16, the DMA of step kt+1 issued right after the barrier that opens step kt (its stage was read during step kt-1).
// ABLATE (timing probes only, results are garbage; WT_TUNING=1 WT_GEMM_ABLATE=n): bit 0 no LDS-DMA in the loop, bit 1 no fragment
// reads (register constants instead), bit 2 no barrier, bit 3 no MFMA, bit 4 every DMA re-fetches K tiles 0 / 1 (cache-hot sources).
template <bool STAMP, int BN, int KSUB = 1, int ABLATE = 0>
__global__ __launch_bounds__(256, KSUB == 2 ? 2 : 3) void gemm_f32_dma_kernel(const GemmParams p) {
    constexpr int STAGES = KSUB == 2 ? 2 : 3;
    constexpr int AHEAD = STAGES - 1;      // K steps in flight beyond the one being multiplied
    constexpr int BK = 16;
    constexpr int TJ = BN / 64;            // 32-column MFMA tiles per wave
    long long t_start = 0, t_first = 0, t_loop = 0;   // probe build: wall-clock stamps (100 MHz)
    if (STAMP) t_start = wall_clock64();
    __shared__ __attribute__((aligned(1024))) float smem[STAGES][KSUB][2][GBM * BK];  // [stage][K sub-tile][A | W][row * 16 + pos * 4]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware tile order (see gemm_f32_kernel)
    const int nbx = (p.N + BN - 1) / BN, nby = (p.M + GBM - 1) / GBM, total = nbx * nby;
    int bid = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * nbx, g = bid / per_group;
    const int gm = min(GROUP_M, nby - g * GROUP_M), in_g = bid - g * per_group;
    const int by = g * GROUP_M + in_g % gm, bx = in_g / gm;
    const int m0 = by * GBM, n0 = bx * BN;

    // DMA map: wave w, pass j fills rows j*64 + w*16 .. +15; lane -> (row lane >> 2, chunk position lane & 3)
    const int r_local = lane >> 2, csrc = (lane & 3) ^ ((r_local >> 2) & 3);
    // per-lane BYTE offsets of the source rows (32 bits: launch_gemm_f32 checks that the operands span < 4 GiB); the K step's column
    // offset is wave-uniform and goes into the scalar base, and the LDS destination is scalar as well (`wave` is read through
    // readfirstlane), so a request is `s_add m0 | global_load_lds v_off, s[base]` with no vector ALU work between the MFMAs
    unsigned aoff[2], woff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = j * 64 + wave * 16 + r_local;
        const int m = min(m0 + row, p.M - 1);
        const int bb = m / p.a_rows_per_batch;
        aoff[j] = (unsigned)(((long long)bb * p.a_batch_stride + (long long)(m - bb * p.a_rows_per_batch) * p.lda + csrc * 4) * 4);
        const int n = min(n0 + row, p.N - 1);   // (BN = 64: only j = 0 is used)
        woff[j] = (unsigned)(((long long)n * p.K + csrc * 4) * 4);
    }
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // (inline asm: from the builtin hipcc selects the 64-bit VECTOR address form and spends a v_lshl_add_u64 per request, and when the
    //  LDS address is not provably scalar also a v_readfirstlane -- vector ALU work in front of every tile's MFMAs)
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)(&smem[0][0][0][0]) + (unsigned)wave * (16 * BK * 4);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // m0 on the clobber list: it is written and consumed inside the one statement
    auto dma = [&](const int stage, const int kt) {
#pragma unroll
        for (int ks = 0; ks < KSUB; ++ks) {
            const char* ab = reinterpret_cast<const char*>(p.A) + (long long)(kt * KSUB + ks) * (BK * 4);   // wave-uniform
            const char* wb = reinterpret_cast<const char*>(p.W) + (long long)(kt * KSUB + ks) * (BK * 4);
            const unsigned st = lds_base + (unsigned)((stage * KSUB + ks) * (2 * GBM * BK * 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             :: "s"(st + (unsigned)(j * 64 * BK * 4)), "v"(aoff[j]), "s"(ab) : "memory", "m0");
                if (j < TJ)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                                 :: "s"(st + (unsigned)(GBM * BK * 4 + j * 64 * BK * 4)), "v"(woff[j]), "s"(wb) : "memory", "m0");
            }
        }
    };
#pragma clang diagnostic pop

    // fragment reads: rows wr*64 + l31 (+32) of A, wc*64 + l31 (+32) of W; chunk 2q + hh at position (2q + hh) ^ swz
    const int swz = (l31 >> 2) & 3;
    const int ra = (wr * 64 + l31) * BK, rb = (wc * (32 * TJ) + l31) * BK;
    const int po0 = ((0 + hh) ^ swz) * 4, po1 = ((2 + hh) ^ swz) * 4;

    f32x16 acc[2][TJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    const int nk = p.K / (BK * KSUB);
    dma(0, 0);
    if (AHEAD > 1 && nk > 1) dma(1, 1);
    // One K step on the compile-time stage CUR (the loop below is unrolled over the stages, so every LDS address of a step is a per-lane
    // base register + an immediate: no vector ALU instruction sits between the MFMAs -- each one there cost MFMA issue slots, the
    // scalar-base DMA form and this unrolling together took the steady-state loop from 0.80 to 0.87 of the MFMA peak).
    auto step = [&](auto CUR, const int kt) {
        constexpr int cur = decltype(CUR)::value;
        constexpr int fill = cur + AHEAD >= STAGES ? cur + AHEAD - STAGES : cur + AHEAD;   // the stage read during step kt-1
        // this wave's share of step kt has landed (with three stages step kt+1's four DMA instructions may still be in flight)
        if (AHEAD > 1 && kt + 1 < nk) {
            if (TJ == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... everyone's has, and nobody still reads the stage refilled next (the fragment reads of step kt-1 were waited for
        // before its MFMAs).  A raw s_barrier: __syncthreads() would add a vmcnt(0) fence and undo the counted wait.
        if (!(ABLATE & 4)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");   // the builtin is IntrNoMem: without this the LDS fragment reads below may be hoisted above it
        if (STAMP && kt == 0) t_first = wall_clock64();
        // (pinning all eight fragment reads above the first MFMA with sched_barrier measured 7 % SLOWER: the compiler's own
        //  interleaving of the second four reads with the first sixteen MFMAs is the better schedule)
        // order inside a step (the compiler's own choice in the rolled loop, pinned here because it moved the second four reads
        // behind all sixteen MFMAs of the first half once the loop was unrolled): 4 reads | DMA | 14 MFMA | 4 reads | 2 MFMA | 16 MFMA
        f32x4 fa0[2 * KSUB], fa1[2 * KSUB], fb0[2 * KSUB], fb1[2 * KSUB];
        auto read_frags = [&](const int h) {       // h = 2 * ks + q: K columns 8h .. 8h+7 of the step
            const float* As = &smem[cur][h >> 1][0][0];
            const float* Ws = &smem[cur][h >> 1][1][0];
            const int po = (h & 1) ? po1 : po0;
            if (ABLATE & 2) {
                const float v = (float)(lane + kt);
                fa0[h] = f32x4{v, v + 1.f, v + 2.f, v + 3.f}; fa1[h] = fa0[h] + 1.f; fb0[h] = fa0[h] * 0.5f; fb1[h] = fb0[h] + 1.f;
            } else {
                fa0[h] = *reinterpret_cast<const f32x4*>(As + ra + po);
                fa1[h] = *reinterpret_cast<const f32x4*>(As + ra + 32 * BK + po);
                fb0[h] = *reinterpret_cast<const f32x4*>(Ws + rb + po);
                fb1[h] = fb0[h];
                if (TJ == 2) fb1[h] = *reinterpret_cast<const f32x4*>(Ws + rb + 32 * BK + po);
            }
        };
        auto mfma4 = [&](const int h, const int jj) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[h][jj], fb0[h][jj], acc[0][0], 0, 0, 0);
            if (TJ == 2) acc[0][TJ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[h][jj], fb1[h][jj], acc[0][TJ - 1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[h][jj], fb0[h][jj], acc[1][0], 0, 0, 0);
            if (TJ == 2) acc[1][TJ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[h][jj], fb1[h][jj], acc[1][TJ - 1], 0, 0, 0);
        };
        read_frags(0);
        if (!(ABLATE & 1) && kt + AHEAD < nk) dma(fill, (ABLATE & 16) ? (kt & 1) : kt + AHEAD);  // after the first fragment reads are on their way
#pragma unroll
        for (int h = 0; h < 2 * KSUB; ++h) {
            if (ABLATE & 8) {       // keep the fragments alive without multiplying
                if (h + 1 < 2 * KSUB) read_frags(h + 1);
                acc[0][0][h & 1] += fa0[h][0] + fa1[h][1] + fb0[h][2] + fb1[h][3];
                continue;
            }
            mfma4(h, 0); mfma4(h, 1); mfma4(h, 2);
            // jj = 3: two MFMAs, the next half's reads, the other two
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[h][3], fb0[h][3], acc[0][0], 0, 0, 0);
            if (TJ == 2) acc[0][TJ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[h][3], fb1[h][3], acc[0][TJ - 1], 0, 0, 0);
            if (h + 1 < 2 * KSUB) {
                __builtin_amdgcn_sched_barrier(0);
                read_frags(h + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[h][3], fb0[h][3], acc[1][0], 0, 0, 0);
            if (TJ == 2) acc[1][TJ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[h][3], fb1[h][3], acc[1][TJ - 1], 0, 0, 0);
        }
    };
    for (int kt = 0; kt < nk; kt += STAGES) {
        step(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < nk) step(std::integral_constant<int, 1>{}, kt + 1);
        if (STAGES == 3 && kt + 2 < nk) step(std::integral_constant<int, STAGES - 1>{}, kt + 2);
    }
    if (STAMP) t_loop = wall_clock64();
    gemm_epilogue_any<TJ>(p, acc, m0, n0, wr, wc, l31, hh);
    if (STAMP) {
        const long long t_issued = wall_clock64();           // every store of this wave issued (the product kernel ends here)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            long long* o = p.dbg_stamps + (long long)blockIdx.x * 8;
            o[0] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));   // HW_REG_HW_ID
            o[1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));   // HW_REG_XCC_ID
            o[2] = t_start; o[3] = t_first; o[4] = t_loop; o[5] = wall_clock64(); o[6] = t_issued; o[7] = 0;
        }
    }
}

hipError_t launch_gemm_f32(const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return hipSuccess;
    {
        // the fast epilogue indexes C / resid / pos with 32-bit element offsets
        const long long batches = (p.M + (long long)p.c_rows_per_batch - 1) / (p.c_rows_per_batch > 0 ? p.c_rows_per_batch : 1) + 1;
        const long long reach = p.epi == EPI_KV_HEADS
                                    ? batches * p.kv_heads * p.kv_cap * HEAD_DIM + ((long long)p.c_rows_per_batch + p.kv_seq_off) * HEAD_DIM
                                    : batches * (p.c_batch_stride > 0 ? p.c_batch_stride : 0) + ((long long)p.c_rows_per_batch + 1) * p.ldc + p.N;
        const long long pos_reach = ((long long)p.c_rows_per_batch + 1) * p.N;
        p.epi_fits32 = reach < (1ll << 31) && pos_reach < (1ll << 31) && p.c_batch_stride >= 0 && p.ldc >= 0;
    }
    if ((p.K & 3) || (p.lda & 3) || (p.a_batch_stride & 3) || p.c_rows_per_batch < 1 || p.a_rows_per_batch < 1) return hipErrorInvalidValue;
    static PerDeviceFlag attr_set;
    static const int force_bk = [] { const char* ev = tuning_env("WT_GEMM_BK"); return ev ? atoi(ev) : 0; }();
    if (!attr_set.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<32>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, gemm_smem_bytes<32>());
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<16>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, gemm_smem_bytes<16>());
        if (e != hipSuccess) return e;
        attr_set.set();
    }
    const int nbx = (p.N + GBN - 1) / GBN, nby = (p.M + GBM - 1) / GBM;
    // measured (tools/microbench.py gemm, M=12000): BK=16 (41 KB of LDS, 128 VGPRs -> 3-4 resident blocks per CU, smaller idle
    // tail in the last round of tiles) beats BK=32 (2 blocks per CU) on every encoder shape: 97-120 vs 77-108 TFLOP/s
    const int tiles = nbx * nby;
    int bk = 16;
    if (force_bk == 16 || force_bk == 32) bk = force_bk;
    static const bool no_dma = tuning_env("WT_GEMM_NO_DMA") != nullptr;  // A/B switch: register-staged kernel for every shape
    // the LDS-DMA kernel addresses its operands with 32-bit byte offsets
    const long long a_batches = (p.M + (long long)p.a_rows_per_batch - 1) / p.a_rows_per_batch;
    const long long a_span = ((a_batches - 1) * (p.a_batch_stride > 0 ? p.a_batch_stride : 0) + (long long)p.a_rows_per_batch * p.lda + p.K) * 4;
    const bool spans32 = a_span < (1ll << 32) && (long long)p.N * p.K * 4 < (1ll << 32) && p.a_batch_stride >= 0 && p.lda >= 0;
    if (!no_dma && spans32 && force_bk == 0 && (p.K % 16) == 0 && ((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0)
    {
        // 128x64 tiles when they balance the CUs better.  Two or three co-resident workgroups saturate a CU's MFMA pipes alike, so
        // a launch takes about ceil(tiles / 256 CUs) tile times; half-size tiles quantise that in half steps (5 % charged for their
        // higher LDS traffic per flop): small.en's N = 768 launches, 564 tiles -> 3 tile times, 1128 half tiles -> 2.5.
        static const int force_bn = tuning_env("WT_GEMM_BN") ? atoi(tuning_env("WT_GEMM_BN")) : 0;
        const long long tiles64 = (long long)((p.N + 63) / 64) * nby;
        const double cost128 = (double)((tiles + 255) / 256), cost64 = 0.5 * 1.05 * (double)((tiles64 + 255) / 256);
        const bool bn64 = force_bn ? force_bn == 64 : cost64 < cost128;
        static const int force_ksub = tuning_env("WT_GEMM_KSUB") ? atoi(tuning_env("WT_GEMM_KSUB")) : 0;
        // 128x128x32 steps on two 32 KiB stages (two workgroups per CU, half the barriers) for launches of many rounds: measured
        // +6.5 % at 2256 tiles and +3.4 % at 3008 (medium.en q|k|v and fc1), equal at 752 / 1504 where the 3-per-CU form fills the chip
        // in fewer rounds
        const bool ksub2 = (force_ksub ? force_ksub == 2 : tiles >= 2048) && (p.K % 32) == 0 && !bn64 && !p.dbg_stamps;
        static const int ablate = tuning_env("WT_GEMM_ABLATE") ? atoi(tuning_env("WT_GEMM_ABLATE")) : 0;
        if (ablate == 1) hipLaunchKernelGGL((gemm_f32_dma_kernel<false, 128, 1, 1>), dim3(tiles), dim3(256), 0, s, p);
        else if (ablate == 2) hipLaunchKernelGGL((gemm_f32_dma_kernel<false, 128, 1, 2>), dim3(tiles), dim3(256), 0, s, p);
        else if (ablate == 4) hipLaunchKernelGGL((gemm_f32_dma_kernel<false, 128, 1, 4>), dim3(tiles), dim3(256), 0, s, p);
        else if (ablate == 5) hipLaunchKernelGGL((gemm_f32_dma_kernel<false, 128, 1, 5>), dim3(tiles), dim3(256), 0, s, p);
        else if (ablate == 7) hipLaunchKernelGGL((gemm_f32_dma_kernel<false, 128, 1, 7>), dim3(tiles), dim3(256), 0, s, p);
        else if (ablate == 8) hipLaunchKernelGGL((gemm_f32_dma_kernel<false, 128, 1, 8>), dim3(tiles), dim3(256), 0, s, p);
        else if (ablate == 16) hipLaunchKernelGGL((gemm_f32_dma_kernel<false, 128, 1, 16>), dim3(tiles), dim3(256), 0, s, p);
        else if (ksub2) hipLaunchKernelGGL((gemm_f32_dma_kernel<false, 128, 2>), dim3(tiles), dim3(256), 0, s, p);
        else if (p.dbg_stamps) hipLaunchKernelGGL((gemm_f32_dma_kernel<true, 128>), dim3(tiles), dim3(256), 0, s, p);
        else if (bn64) hipLaunchKernelGGL((gemm_f32_dma_kernel<false, 64>), dim3((unsigned)tiles64), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((gemm_f32_dma_kernel<false, 128>), dim3(tiles), dim3(256), 0, s, p);
    }
    else if (bk == 16) hipLaunchKernelGGL(gemm_f32_kernel<16>, dim3(tiles), dim3(256), gemm_smem_bytes<16>(), s, p);
    else hipLaunchKernelGGL(gemm_f32_kernel<32>, dim3(tiles), dim3(256), gemm_smem_bytes<32>(), s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ fp32 GEMM on the bf16 matrix cores
// C = epi(A . W^T) with A and W given as THREE bf16 planes each (wt_common.h: launch_gemm_x3): per 32x32x16 block six
// v_mfma_f32_32x32x16_bf16 -- a3.w1, a1.w3, a2.w2, a2.w1, a1.w2, a1.w1, small terms first -- into ONE fp32 accumulator, which is the
// accumulator (and the accumulator layout) of the fp32 kernel, so the epilogues above are shared.  Structure of gemm_f32_dma_kernel:
// 128x128 tile, 4 waves (2x2) x 2x2 blocks, K step 16; every plane tile is 128 rows x 32 B, staged by LDS-DMA (one 1 KiB wave
// instruction = 32 rows; wave w fills rows 32w..32w+31 of all six plane tiles: 6 requests per wave and step), unpadded, the two 16-byte
// chunks of a row XOR-swizzled by (row >> 3) & 1 on both sides so that a 16-lane ds_read_b128 group (rows r..r+15, one logical chunk)
// covers all 16 sixteen-byte bank groups.  Three 24 KiB stages (72 KiB: two workgroups per CU), counted vmcnt(6), raw s_barrier.
// Per step and wave: 12 ds_read_b128 for 24 MFMAs (each A / W fragment is used three times in registers) -- a third of the LDS bytes
// per MFMA of a one-plane fp16 GEMM, which is what lets this loop run MFMA-bound where the fp16 kernels are LDS-bound.
// WM = 32-row blocks per wave, NWC = wave columns (two wave rows always): <2, 2> is the 128x128 tile above (4 waves of 64x64, three
// 24 KiB stages, two workgroups per CU); <4, 4> a 256x256 tile (8 waves of 128x64, three 48 KiB stages, one workgroup per CU);
// <3, 4> a 192x256 tile (8 waves of 96x64, three 42 KiB stages).  The 128x128 form is bound by the L2 -> LDS feed (0.33 of the bf16
// MFMA peak, the plateau of every 128x128 one-barrier GEMM on this part); the big tiles stage half the bytes per flop, and a step of
// theirs is ~3000 MFMA cycles per SIMD, which also covers the DMA's ~1.5 us latency with two steps in flight.  192x256 exists for the
// chip's 256 workgroup slots: M = 12000 gives 63 row tiles, so N = 1024 / 2048 / 3072 / 4096 make 252 / 504 / 756 / 1008 tiles = 0.98 /
// 1.97 / 2.95 / 3.94 rounds, where 256x256 leaves a quarter of a round empty at N = 1024, 2048 and 3072.
// DMA: a step is 3 x (BM / 32 + BN / 32) wave instructions of 1 KiB (32 rows of one plane tile); unit u goes to wave u % NW, so a
// wave issues NU / NW requests per step -- 6, except 5 for waves 2..7 of the 192x256 form, whose counted waits differ accordingly.
constexpr int X3_BK = 16;
template <int WM, int NWC>
__global__ __launch_bounds__(128 * NWC, WM == 2 ? 2 : 1) void gemm_x3_kernel(const GemmParams p) {
    constexpr int STAGES = 3, AHEAD = 2;
    constexpr int NW = 2 * NWC;                       // waves: 2 wave rows x NWC wave columns
    constexpr int BM = 2 * WM * 32, BN = NWC * 64;
    constexpr int ARB = BM / 32, WRB = BN / 32;       // 32-row blocks (= DMA instructions) per plane tile
    constexpr int NU = 3 * (ARB + WRB);               // DMA units per step
    constexpr int J = (NU + NW - 1) / NW;             // requests per wave and step (the last one only for waves < NU - (J - 1) * NW)
    constexpr int LAST_WAVES = NU - (J - 1) * NW;     // waves that issue J requests (the others J - 1); == NW when NU % NW == 0
    constexpr int A_TILE = BM * X3_BK, W_TILE = BN * X3_BK;          // bf16 elements of one plane tile
    constexpr int STAGE = 3 * (A_TILE + W_TILE);
    __shared__ __attribute__((aligned(1024))) __bf16 smem[STAGES * STAGE];   // [stage][A planes 0..2 | W planes 0..2][row * 16 + pos * 8]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wr = wave / NWC, wc = wave % NWC;
    // XCD-aware tile order (see gemm_f32_kernel)
    const int nbx = (p.N + BN - 1) / BN, nby = (p.M + BM - 1) / BM, total = nbx * nby;
    int bid = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP_M = WM == 2 ? 8 : 4;
    const int per_group = GROUP_M * nbx, g = bid / per_group;
    const int gm = min(GROUP_M, nby - g * GROUP_M), in_g = bid - g * per_group;
    const int by = g * GROUP_M + in_g % gm, bx = in_g / gm;
    const int m0 = by * BM, n0 = bx * BN;

    // DMA units of this wave: u = wave + NW * j; u < 3 * ARB: A plane u / ARB, row block u % ARB; else W plane (u - 3 ARB) / WRB, row block (..) % WRB.
    // lane -> (row lane >> 1 of the 32-row block, chunk position lane & 1), source chunk pos ^ ((row >> 3) & 1)
    const int r_local = lane >> 1, csrc = (lane & 1) ^ ((r_local >> 3) & 1);
    unsigned uoff[J];          // per-lane BYTE offset of the unit's row inside its plane (32 bits: gemm_x3_usable checks the spans)
    unsigned ulds[J];          // scalar: LDS byte offset of the unit inside a stage
    long long ubase[J];        // scalar: byte offset of the unit's plane from p.A / p.W
    bool u_is_a[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int u = wave + NW * j;
        const bool is_a = u < 3 * ARB;
        const int v = is_a ? u : u - 3 * ARB, rbn = is_a ? ARB : WRB;
        const int pl = min(v / rbn, 2), rb = v % rbn;
        const int row = rb * 32 + r_local;
        u_is_a[j] = is_a;
        if (is_a) {
            const int m = min(m0 + row, p.M - 1);
            const int bb = m / p.a_rows_per_batch;
            uoff[j] = (unsigned)(((long long)bb * p.a_batch_stride + (long long)(m - bb * p.a_rows_per_batch) * p.lda + csrc * 8) * 2);
            ulds[j] = (unsigned)((pl * A_TILE + rb * 32 * X3_BK) * 2);
            ubase[j] = (long long)pl * p.a_plane * 2;
        } else {
            uoff[j] = (unsigned)(((long long)min(n0 + row, p.N - 1) * p.K + csrc * 8) * 2);
            ulds[j] = (unsigned)((3 * A_TILE + pl * W_TILE + rb * 32 * X3_BK) * 2);
            ubase[j] = (long long)pl * p.w_plane * 2;
        }
    }
    typedef __attribute__((address_space(3))) void* lptr_t;
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)(&smem[0]);
    const bool last_unit = wave < LAST_WAVES;         // wave-uniform
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // m0 on the clobber list: it is written and consumed inside the one statement
    auto dma = [&](const int stage, const int kt) {
        const unsigned st = lds_base + (unsigned)(stage * STAGE * 2);
        const long long koff = (long long)kt * (X3_BK * 2);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (j == J - 1 && LAST_WAVES != NW && !last_unit) break;
            const char* gb = reinterpret_cast<const char*>(u_is_a[j] ? p.A : p.W) + ubase[j] + koff;   // wave-uniform
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(__builtin_amdgcn_readfirstlane(st + ulds[j])), "v"(uoff[j]), "s"(gb) : "memory", "m0");
        }
    };
#pragma clang diagnostic pop
    // fragment reads (32x32x16: lane (r = l31, h = hh) holds row r, k = 8h .. 8h+7 = chunk h): position h ^ ((row >> 3) & 1); the row
    // blocks start at multiples of 32, so (row >> 3) & 1 == (l31 >> 3) & 1 for every block
    const int pos = (hh ^ ((l31 >> 3) & 1)) * 8;
    const int ra = (wr * (WM * 32) + l31) * X3_BK + pos, rb = (wc * 64 + l31) * X3_BK + pos;

    // accumulators of the wave tile's row blocks as NAMED arrays (one three-level array went to scratch: cdna guide rule 20):
    // acc0 = row blocks 0, 1; acc1 = row blocks 2, 3 (WM == 4); acc2 = row block 2 (WM == 3)
    f32x16 acc0[2][2], acc1[2][2], acc2[1][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[0][jj][r] = acc0[1][jj][r] = acc1[0][jj][r] = acc1[1][jj][r] = acc2[0][jj][r] = 0.f;

    const int nk = p.K / X3_BK;
    dma(0, 0);
    if (nk > 1) dma(1, 1);
    auto step = [&](auto CUR, const int kt) {
        constexpr int cur = decltype(CUR)::value;
        constexpr int fill = cur + AHEAD >= STAGES ? cur + AHEAD - STAGES : cur + AHEAD;   // the stage read during step kt-1
        // this wave's share of step kt has landed (step kt+1's requests may be in flight: J of them, or J - 1)
        if (kt + 1 < nk) {
            if (LAST_WAVES == NW || last_unit) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(J) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(J - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();     // raw: __syncthreads() would add a vmcnt(0) fence and undo the counted wait
        asm volatile("" ::: "memory");
        const __bf16* As = smem + cur * STAGE;
        const __bf16* Ws = As + 3 * A_TILE;
        bf16x8 fa[WM][3], fb[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) fb[jj][pl] = *reinterpret_cast<const bf16x8*>(Ws + pl * W_TILE + rb + jj * 32 * X3_BK);
#pragma unroll
            for (int i = 0; i < WM; ++i) fa[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * A_TILE + ra + i * 32 * X3_BK);
        }
        if (kt + AHEAD < nk) dma(fill, kt + AHEAD);   // after the fragment reads are on their way
        // six partial products per block, the small ones first
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PW[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                acc0[0][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][PA[t]], fb[jj][PW[t]], acc0[0][jj], 0, 0, 0);
                acc0[1][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][PA[t]], fb[jj][PW[t]], acc0[1][jj], 0, 0, 0);
                if (WM == 3) acc2[0][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[WM - 1][PA[t]], fb[jj][PW[t]], acc2[0][jj], 0, 0, 0);
                if (WM == 4) {
                    acc1[0][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[WM - 2][PA[t]], fb[jj][PW[t]], acc1[0][jj], 0, 0, 0);
                    acc1[1][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[WM - 1][PA[t]], fb[jj][PW[t]], acc1[1][jj], 0, 0, 0);
                }
            }
        }
    };
    for (int kt = 0; kt < nk; kt += STAGES) {
        step(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < nk) step(std::integral_constant<int, 1>{}, kt + 1);
        if (kt + 2 < nk) step(std::integral_constant<int, 2>{}, kt + 2);
    }
    // the shared epilogues work on a wave's 64- (or 32-) row x 64-column pieces
    gemm_epilogue_any<2>(p, acc0, m0 + wr * (WM * 32), n0 + wc * 64, 0, 0, l31, hh);
    if (WM == 4) gemm_epilogue_any<2>(p, acc1, m0 + wr * (WM * 32) + 64, n0 + wc * 64, 0, 0, l31, hh);
    if (WM == 3) gemm_epilogue_any<2, 1>(p, acc2, m0 + wr * (WM * 32) + 64, n0 + wc * 64, 0, 0, l31, hh);
}

bool gemm_x3_usable(const GemmParams& p) {
    if (p.M <= 0 || p.N <= 0 || p.K < X3_BK || (p.K % X3_BK) || (p.lda & 7) || (p.a_batch_stride & 7) || p.a_rows_per_batch < 1 || p.c_rows_per_batch < 1) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15) || (p.a_plane & 7) || (p.w_plane & 7) || p.a_plane <= 0 || p.w_plane <= 0) return false;
    const long long a_batches = (p.M + (long long)p.a_rows_per_batch - 1) / p.a_rows_per_batch;
    const long long a_span = ((a_batches - 1) * (p.a_batch_stride > 0 ? p.a_batch_stride : 0) + (long long)p.a_rows_per_batch * p.lda + p.K) * 2;
    return a_span < (1ll << 32) && (long long)p.N * p.K * 2 < (1ll << 32) && p.a_batch_stride >= 0 && p.lda >= 0;
}

hipError_t launch_gemm_x3(const GemmParams& p_in, hipStream_t s) {
    GemmParams p = p_in;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return hipSuccess;
    if (!gemm_x3_usable(p)) return hipErrorInvalidValue;
    if (p.out_split && (p.epi != EPI_ROWMAJOR || p.c_plane <= 0)) return hipErrorInvalidValue;
    {   // the fast epilogue indexes C / resid / pos with 32-bit element offsets (as launch_gemm_f32)
        const long long batches = (p.M + (long long)p.c_rows_per_batch - 1) / p.c_rows_per_batch + 1;
        const long long reach = p.epi == EPI_KV_HEADS
                                    ? batches * p.kv_heads * p.kv_cap * HEAD_DIM + ((long long)p.c_rows_per_batch + p.kv_seq_off) * HEAD_DIM
                                    : batches * (p.c_batch_stride > 0 ? p.c_batch_stride : 0) + ((long long)p.c_rows_per_batch + 1) * p.ldc + p.N;
        const long long pos_reach = ((long long)p.c_rows_per_batch + 1) * p.N;
        p.epi_fits32 = reach < (1ll << 31) && pos_reach < (1ll << 31) && p.c_batch_stride >= 0 && p.ldc >= 0;
    }
    // tile shape.  128x128: two 4-wave workgroups per CU (512 slots), L2-feed-bound; 256x256 and 192x256: one 8-wave workgroup per CU
    // (256 slots), about twice as efficient per flop.  Estimated time = rounds of the chip x work per tile / efficiency; A/B:
    // WT_GEMM_X3_TILE=128|192|256
    static const int force_tile = tuning_env("WT_GEMM_X3_TILE") ? atoi(tuning_env("WT_GEMM_X3_TILE")) : 0;
    const long long t128 = (long long)((p.N + 127) / 128) * ((p.M + 127) / 128), t256 = (long long)((p.N + 255) / 256) * ((p.M + 255) / 256),
                    t192 = (long long)((p.N + 255) / 256) * ((p.M + 191) / 192);
    // measured per round of the chip at K = 1024, M = 12000 (tools/microbench.py gemm_x3): 256x256 157 us, 192x256 142 us, 128x128 109 us
    const double cost256 = (double)((t256 + 255) / 256) * 1.0, cost192 = (double)((t192 + 255) / 256) * 0.905, cost128 = (double)((t128 + 511) / 512) * 0.695;
    int tile = force_tile;
    if (tile != 128 && tile != 192 && tile != 256) tile = (p.M < 192 || p.N < 256) ? 128 : (cost256 <= cost192 && cost256 <= cost128) ? 256 : cost192 <= cost128 ? 192 : 128;
    if (tile == 256) hipLaunchKernelGGL((gemm_x3_kernel<4, 4>), dim3((unsigned)t256), dim3(512), 0, s, p);
    else if (tile == 192) hipLaunchKernelGGL((gemm_x3_kernel<3, 4>), dim3((unsigned)t192), dim3(512), 0, s, p);
    else hipLaunchKernelGGL((gemm_x3_kernel<2, 2>), dim3((unsigned)t128), dim3(256), 0, s, p);
    return hipGetLastError();
}

// fp32 [n] -> three bf16 planes (weights at engine open, the encoder memory in front of the cross-K/V projection)
__global__ __launch_bounds__(256) void split3_kernel(const float4* __restrict__ x, __bf16* __restrict__ out, const size_t n4, const size_t plane) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = x[i];
        bf16x4 o1, o2, o3;
        __bf16 a, b, c;
        split3(v.x, a, b, c); o1[0] = a; o2[0] = b; o3[0] = c;
        split3(v.y, a, b, c); o1[1] = a; o2[1] = b; o3[1] = c;
        split3(v.z, a, b, c); o1[2] = a; o2[2] = b; o3[2] = c;
        split3(v.w, a, b, c); o1[3] = a; o2[3] = b; o3[3] = c;
        *reinterpret_cast<bf16x4*>(out + 4 * i) = o1;
        *reinterpret_cast<bf16x4*>(out + plane + 4 * i) = o2;
        *reinterpret_cast<bf16x4*>(out + 2 * plane + 4 * i) = o3;
    }
}
hipError_t launch_split3(const float* x, void* planes, size_t n, size_t plane_stride, hipStream_t s) {
    if ((n & 3) || (plane_stride & 3) || ((uintptr_t)x & 15) || ((uintptr_t)planes & 7)) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    const size_t n4 = n >> 2;
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(split3_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(x), reinterpret_cast<__bf16*>(planes), n4, plane_stride);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ encoder attention
// softmax(Q K^T / 8) V per (utterance, head), S = 1500 keys, head_dim 64, no mask (layers/attention.py:308,
// 337-345; HF :569-593).  Flash-style: scores never leave registers.  One workgroup = 128 queries (4 waves x 32),
// K/V tiles of 64 keys double-buffered in LDS.  Both products run on v_mfma_f32_32x32x2_f32 with the
// TRANSPOSED orientation so that the query index sits on the lane:
//   S^T[key][query] = K . Q^T     -> lane (query, half h) holds 16 keys per 32-key tile
//   O^T[dv][query] += V^T . P^T   -> P comes straight from the S^T accumulator registers (same lane,
//                                    same k assignment), the row max / sum / rescale are lane-local.
constexpr int FA_BQ = 128, FA_BKV = 64, FA_LD = 68;
constexpr int FA_SMEM = 2 * 2 * FA_BKV * FA_LD * (int)sizeof(float);  // 69,632 B

// Round 4 tried four restructurings of this kernel against it on one box (profiles/r04_enc_attn_variants.txt: all within +-1 % of
// 717 us per launch in the micro-benchmark): the V fragments of P.V requested a whole group of MFMAs ahead under pinned scheduling
// (the compiler issues them one MFMA pair ahead with lgkmcnt(0) waits), the softmax's exponentials issued group by group in the shadow
// of the previous group's MFMAs with the accumulator rescale skipped while no running maximum moves, the 256-VGPR budget two waves per
// SIMD allow, and skipping the MFMAs of waves / half tiles that lie entirely in the 1500 -> 1536 padding.  None moved it: the LDS is
// not the limit (SQ_WAIT_INST_LDS 1.4 % of the wave cycles, no bank conflicts: profiles/r04_enc_attn_pmc_counters.txt), and a wave's
// spare MFMA time goes nowhere while its workgroup waits for its slowest wave at the per-tile barrier.  DESIGN.md section 9 has the reading.
// SPLIT: the context goes out as three bf16 planes (`ctx` = plane 0), the A operand of the out-projection's launch_gemm_x3
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void enc_attn_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, int S,
                                                          int H, size_t plane) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    // XCD-aware order (1-D grid; workgroup L runs on XCD L % 8): the query blocks of one (utterance, head) go to ONE XCD, back to back, so
    // the head's K and V (768 KB at S = 1500) are fetched into one L2 instead of eight.  With the plain (x, y, z) grid the 12 query blocks
    // of a head were dealt round-robin over the XCDs: 847 MB fetched per launch for 147 MB of q|k|v (PMC, profiles/r03b_pmc_fetch_...).
    int b, h, q0;
    {
        // XCD x owns a contiguous chunk of the head-major list of (head, query block) pairs (the chunking of gemm_f32_kernel: a bijection
        // for any total); a head that straddles a chunk boundary is fetched by two XCDs
        const int nqb = (S + FA_BQ - 1) / FA_BQ, total = (int)gridDim.x;
        const int L = blockIdx.x, xcd = L & 7, idx = L >> 3, q = total >> 3, r = total & 7;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int bh = v / nqb;
        b = bh / H; h = bh % H; q0 = (v - bh * nqb) * FA_BQ;
    }
    const int d = H * HEAD_DIM, ld = 3 * d;
    const float* base = qkv + (size_t)b * S * ld + h * HEAD_DIM;

    // Q fragment (B operand of S^T): lane (query, h) holds Q[query][8q+4h .. +3], pre-scaled by 64^-0.5 (exact)
    const int qrow = q0 + wave * 32 + l31;
    f32x4 qf[8];
    {
        const float* qp = base + (size_t)min(qrow, S - 1) * ld + 4 * hh;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            qf[q] = *reinterpret_cast<const f32x4*>(qp + 8 * q);
            qf[q] *= 0.125f;
        }
    }
    // staging map: 64 keys x 16 float4 per matrix; thread -> column c4, rows r0 + 16*i
    const int c4 = tid & 15, r0 = tid >> 4;
    // staging registers as named vectors (an array here is left in scratch by hipcc: 1.2 GB of spill traffic per launch)
    f32x4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;
    const float* gbase = base + c4 * 4;
#define FA_GLOAD(kv0_)                                                                   \
    do {                                                                                 \
        const float* rp0 = gbase + (size_t)min((kv0_) + r0, S - 1) * ld;                 \
        const float* rp1 = gbase + (size_t)min((kv0_) + r0 + 16, S - 1) * ld;            \
        const float* rp2 = gbase + (size_t)min((kv0_) + r0 + 32, S - 1) * ld;            \
        const float* rp3 = gbase + (size_t)min((kv0_) + r0 + 48, S - 1) * ld;            \
        rk0 = *reinterpret_cast<const f32x4*>(rp0 + d); rv0 = *reinterpret_cast<const f32x4*>(rp0 + 2 * d); \
        rk1 = *reinterpret_cast<const f32x4*>(rp1 + d); rv1 = *reinterpret_cast<const f32x4*>(rp1 + 2 * d); \
        rk2 = *reinterpret_cast<const f32x4*>(rp2 + d); rv2 = *reinterpret_cast<const f32x4*>(rp2 + 2 * d); \
        rk3 = *reinterpret_cast<const f32x4*>(rp3 + d); rv3 = *reinterpret_cast<const f32x4*>(rp3 + 2 * d); \
    } while (0)
#define FA_LSTORE(buf_)                                                                  \
    do {                                                                                 \
        float* Ks_ = smem + (buf_) * (2 * FA_BKV * FA_LD) + r0 * FA_LD + c4 * 4;         \
        float* Vs_ = Ks_ + FA_BKV * FA_LD;                                               \
        *reinterpret_cast<f32x4*>(Ks_) = rk0; *reinterpret_cast<f32x4*>(Vs_) = rv0;      \
        *reinterpret_cast<f32x4*>(Ks_ + 16 * FA_LD) = rk1; *reinterpret_cast<f32x4*>(Vs_ + 16 * FA_LD) = rv1; \
        *reinterpret_cast<f32x4*>(Ks_ + 32 * FA_LD) = rk2; *reinterpret_cast<f32x4*>(Vs_ + 32 * FA_LD) = rv2; \
        *reinterpret_cast<f32x4*>(Ks_ + 48 * FA_LD) = rk3; *reinterpret_cast<f32x4*>(Vs_ + 48 * FA_LD) = rv3; \
    } while (0)

    // named accumulators (not arrays of vectors: a runtime-looking index sends those to scratch, cdna guide rule 20)
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (S + FA_BKV - 1) / FA_BKV;
    FA_GLOAD(0);
    FA_LSTORE(0);
    __syncthreads();
    // One 64-key tile; LAST = the final (possibly ragged) one.  Peeled because hipcc if-converts the ragged-tile test into 32 compares +
    // 32 selects executed on EVERY tile -- a fifth of the loop's vector ALU work, which competes with the MFMAs for issue.
    auto tile = [&](auto LAST_T, const int t) {
        constexpr bool LAST = decltype(LAST_T)::value;
        const int cur = t & 1, kv0 = t * FA_BKV;
        if (!LAST) FA_GLOAD(kv0 + FA_BKV);
        const float* Ks = smem + cur * (2 * FA_BKV * FA_LD);
        const float* Vs = Ks + FA_BKV * FA_LD;

        f32x16 s0, s1;
        const float* kp = Ks + l31 * FA_LD + 4 * hh;
        // K fragments are read one 8-wide k-group ahead of the MFMAs that use them
        f32x4 kf0[8], kf1[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            kf0[q] = *reinterpret_cast<const f32x4*>(kp + 8 * q);
            kf1[q] = *reinterpret_cast<const f32x4*>(kp + 32 * FA_LD + 8 * q);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q + 2 < 8) {
                kf0[q + 2] = *reinterpret_cast<const f32x4*>(kp + 8 * (q + 2));
                kf1[q + 2] = *reinterpret_cast<const f32x4*>(kp + 32 * FA_LD + 8 * (q + 2));
            }
            const f32x4 k0 = kf0[q], k1 = kf1[q], qq = qf[q];
            if (q == 0) {   // the tile's first products start from the inline constant 0 (no 32 v_mov per tile to clear the score registers)
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(k0[0], qq[0], zero16, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(k1[0], qq[0], zero16, 0, 0, 0);
            } else {
                s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(k0[0], qq[0], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(k1[0], qq[0], s1, 0, 0, 0);
            }
            s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(k0[1], qq[1], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(k1[1], qq[1], s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(k0[2], qq[2], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(k1[2], qq[2], s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(k0[3], qq[3], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(k1[3], qq[3], s1, 0, 0, 0);
        }
        if (LAST && kv0 + FA_BKV > S) {  // ragged last tile: keys >= S get probability 0
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
        l_run = l_run * alpha + ps;  // per-lane partial (16 of the query's 32 keys per tile); halves merged at the end
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] *= alpha;
            o1[r] *= alpha;
        }
        const float* vbase = Vs + 4 * hh * FA_LD + l31;
        // P.V: V values are fetched four keys ahead of the MFMAs that consume them (counted LDS waits, no per-MFMA stall)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float va[4], vb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                va[i] = vbase[(8 * g + i) * FA_LD];
                vb[i] = vbase[(8 * g + i) * FA_LD + 32];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i], s0[4 * g + i], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[i], s0[4 * g + i], o1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float va[4], vb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                va[i] = vbase[(32 + 8 * g + i) * FA_LD];
                vb[i] = vbase[(32 + 8 * g + i) * FA_LD + 32];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i], s1[4 * g + i], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[i], s1[4 * g + i], o1, 0, 0, 0);
            }
        }
        if (!LAST) FA_LSTORE(cur ^ 1);
        __syncthreads();
    };
    for (int t = 0; t + 1 < ntiles; ++t) tile(std::false_type{}, t);
    tile(std::true_type{}, ntiles - 1);
#undef FA_GLOAD
#undef FA_LSTORE
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32));
    if (qrow < S) {
        const size_t at = ((size_t)b * S + qrow) * d + h * HEAD_DIM + 4 * hh;
        if (SPLIT) {
            __bf16* op = reinterpret_cast<__bf16*>(ctx) + at;
            auto put = [&](__bf16* dst, const float x0, const float x1, const float x2, const float x3) {
                bf16x4 p1, p2, p3;
                __bf16 b1, b2, b3;
                split3(x0, b1, b2, b3); p1[0] = b1; p2[0] = b2; p3[0] = b3;
                split3(x1, b1, b2, b3); p1[1] = b1; p2[1] = b2; p3[1] = b3;
                split3(x2, b1, b2, b3); p1[2] = b1; p2[2] = b2; p3[2] = b3;
                split3(x3, b1, b2, b3); p1[3] = b1; p2[3] = b2; p3[3] = b3;
                *reinterpret_cast<bf16x4*>(dst) = p1;
                *reinterpret_cast<bf16x4*>(dst + plane) = p2;
                *reinterpret_cast<bf16x4*>(dst + 2 * plane) = p3;
            };
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                put(op + 8 * g, o0[4 * g + 0] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
                put(op + 32 + 8 * g, o1[4 * g + 0] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            }
        } else {
            float* op = ctx + at;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<float4*>(op + 8 * g) =
                    make_float4(o0[4 * g + 0] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
                *reinterpret_cast<float4*>(op + 32 + 8 * g) =
                    make_float4(o1[4 * g + 0] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ encoder attention, x3 form
// The same flash-style kernel with BOTH products formed like launch_gemm_x3's: q, k, v arrive as three bf16 planes each (the q|k|v GEMM's
// epilogue writes them), the probabilities are split in registers, every 32x32x16 block is six v_mfma_f32_32x32x16_bf16 into the fp32
// accumulators (a3.b1, a1.b3, a2.b2, a2.b1, a1.b2, a1.b1), scores / softmax / accumulators stay fp32.  96 bf16 MFMAs of 32 cycles per
// 64-key tile and wave instead of 128 fp32 MFMAs of 64 cycles.  Transposed formulation as above:
//   S^T[key][query] = K . Q^T   (A = K rows from LDS, B = the wave's Q rows, held in registers for the whole kernel)
//   O^T[dv][query] += V^T . P^T (B = P straight from the S^T accumulator registers: registers 8t .. 8t+7 are k-step t, whose key order
//                                16t + 8 (j >> 2) + 4 h + (j & 3) is matched by reading V^T with the same permutation from a TRANSPOSED V
//                                tile in LDS -- the scheme of enc_attn_f16_kernel).
// LDS: one 64-key stage of 3 x (K [64][72] + V [64][96]) bf16 = 64,512 B.  K rows are padded to 144 B (conflict-free ds_read_b128).  V stays
// ROW-major as it comes from memory (two 16-byte stores per thread and plane) and the V^T fragments are read with ds_read_b64_tr_b16: each
// 16-lane group takes a block of 4 keys x 16 head dims and every lane receives the four keys of its own head dim.  Rows of 192 B put the four
// key rows of a block on the four 16-bank quarters (48 q mod 64 = 0, 48, 32, 16): conflict-free.  (Until late round 4 the tile was stored
// transposed with 48 two-byte stores per thread and tile, two- to four-way conflicting.)  Single-buffered so that two workgroups share a CU:
// the next tile's global loads are issued before the current tile's products and stored behind a second barrier.
constexpr int XA_BQ = 128, XA_BKV = 64, XA_KLD = 72, XA_VLD = 96;                   // strides in bf16 elements
constexpr int XA_PLANE = XA_BKV * XA_KLD + XA_BKV * XA_VLD;                          // one plane's (K, V) tile
constexpr int XA_SMEM = 3 * XA_PLANE * 2;                                            // 64,512 B
__global__ __launch_bounds__(256, 2) void enc_attn_x3_kernel(const __bf16* __restrict__ qkv, __bf16* __restrict__ ctx, int S, int H,
                                                             size_t in_plane, size_t out_plane) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xa_raw[];
    __bf16* smem = reinterpret_cast<__bf16*>(xa_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    int b, h, q0;
    {   // XCD-chunked head-major order, as in enc_attn_kernel: a head's K / V land in one L2 instead of eight
        const int nqb = (S + XA_BQ - 1) / XA_BQ, total = (int)gridDim.x;
        const int L = blockIdx.x, xcd = L & 7, idx = L >> 3, q = total >> 3, r = total & 7;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int bh = v / nqb;
        b = bh / H; h = bh % H; q0 = (v - bh * nqb) * XA_BQ;
    }
    const int d = H * HEAD_DIM, ld = 3 * d;
    const __bf16* base = qkv + (size_t)b * S * ld + h * HEAD_DIM;

    const int qrow = q0 + wave * 32 + l31;
    bf16x8 qf[3][4];   // plane p of Q[query][16 s + 8 h .. +7], pre-scaled by 64^-0.5 (a power of two: exact in every plane)
    {
        const __bf16* qp = base + (size_t)min(qrow, S - 1) * ld + 8 * hh;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qf[pl][s] = *reinterpret_cast<const bf16x8*>(qp + pl * in_plane + 16 * s);
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[pl][s][e] = (__bf16)((float)qf[pl][s][e] * 0.125f);
            }
    }
    // staging: 64 keys x 8 sixteen-byte chunks per matrix and plane; thread -> chunk c8 (8 head dims), keys r0 and r0 + 32
    const int c8 = tid & 7, r0 = tid >> 3;
    const __bf16* gbase = base + c8 * 8;
    bf16x8 rk[3][2], rv[3][2];
    auto gload = [&](const int kv0) {
        const __bf16* rp0 = gbase + (size_t)min(kv0 + r0, S - 1) * ld;
        const __bf16* rp1 = gbase + (size_t)min(kv0 + r0 + 32, S - 1) * ld;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            rk[pl][0] = *reinterpret_cast<const bf16x8*>(rp0 + pl * in_plane + d);
            rv[pl][0] = *reinterpret_cast<const bf16x8*>(rp0 + pl * in_plane + 2 * d);
            rk[pl][1] = *reinterpret_cast<const bf16x8*>(rp1 + pl * in_plane + d);
            rv[pl][1] = *reinterpret_cast<const bf16x8*>(rp1 + pl * in_plane + 2 * d);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            __bf16* Ks_ = smem + pl * XA_PLANE;
            __bf16* Vs_ = Ks_ + XA_BKV * XA_KLD;
            *reinterpret_cast<bf16x8*>(Ks_ + r0 * XA_KLD + c8 * 8) = rk[pl][0];
            *reinterpret_cast<bf16x8*>(Ks_ + (r0 + 32) * XA_KLD + c8 * 8) = rk[pl][1];
            *reinterpret_cast<bf16x8*>(Vs_ + r0 * XA_VLD + c8 * 8) = rv[pl][0];
            *reinterpret_cast<bf16x8*>(Vs_ + (r0 + 32) * XA_VLD + c8 * 8) = rv[pl][1];
        }
    };
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // the six partial products, the small ones first
    // transposed V reads: lane 32 hh + 16 g + 4 q + p supplies row (4 hh + q), head dims 16 g + 4 p .. + 3 of a block and receives the
    // four keys 4 hh .. 4 hh + 3 of head dim 16 g + 4 q + p = l31 (cdna guide T10)
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* tr_ptr_t;
    const int v_tr = (4 * hh + ((lane >> 2) & 3)) * XA_VLD + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    auto tr4 = [&](const __bf16* ptr) { return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)ptr)); };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int ntiles = (S + XA_BKV - 1) / XA_BKV;
    gload(0);
    lstore();
    __syncthreads();
    // One 64-key tile; LAST = the final (possibly ragged) one, peeled: hipcc if-converts the ragged-tile mask into ~90 compare / select /
    // mask instructions executed on EVERY tile otherwise (as it did in enc_attn_kernel): 532 -> 514 us per launch
    auto tile = [&](auto LAST_T, const int t) {
        constexpr bool LAST = decltype(LAST_T)::value;
        const int kv0 = t * XA_BKV;
        if (!LAST) gload(kv0 + XA_BKV);
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 k0[3], k1[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const __bf16* kp = smem + pl * XA_PLANE + l31 * XA_KLD + 8 * hh + 16 * s;
                k0[pl] = *reinterpret_cast<const bf16x8*>(kp);
                k1[pl] = *reinterpret_cast<const bf16x8*>(kp + 32 * XA_KLD);
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0[PA[c]], qf[PB[c]][s], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1[PA[c]], qf[PB[c]][s], s1, 0, 0, 0);
            }
        }
        if (LAST && kv0 + XA_BKV > S) {
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
        // P.V: k-step (kt, t2) covers keys kt*32 + 16*t2 .. +15; B fragment = accumulator registers 8*t2 .. 8*t2+7, split into three planes
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                bf16x8 pf[3];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    __bf16 b1, b2, b3;
                    split3(kt == 0 ? s0[8 * t2 + j] : s1[8 * t2 + j], b1, b2, b3);
                    pf[0][j] = b1; pf[1][j] = b2; pf[2][j] = b3;
                }
                const int kb = kt * 32 + 16 * t2;
                bf16x8 va0[3], va1[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const __bf16* vp = smem + pl * XA_PLANE + XA_BKV * XA_KLD + kb * XA_VLD + v_tr;
                    const bf16x4 a00 = tr4(vp), a01 = tr4(vp + 8 * XA_VLD);
                    const bf16x4 a10 = tr4(vp + 32), a11 = tr4(vp + 8 * XA_VLD + 32);
                    va0[pl] = bf16x8{a00[0], a00[1], a00[2], a00[3], a01[0], a01[1], a01[2], a01[3]};
                    va1[pl] = bf16x8{a10[0], a10[1], a10[2], a10[3], a11[0], a11[1], a11[2], a11[3]};
                }
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va0[PA[c]], pf[PB[c]], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va1[PA[c]], pf[PB[c]], o1, 0, 0, 0);
                }
            }
        }
        __syncthreads();                 // every wave is done with this tile ...
        if (!LAST) {
            lstore();                    // ... the next one goes in (its loads were issued before the products)
            __syncthreads();
        }
    };
    for (int t = 0; t + 1 < ntiles; ++t) tile(std::false_type{}, t);
    tile(std::true_type{}, ntiles - 1);
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32));
    if (qrow < S) {
        __bf16* op = ctx + ((size_t)b * S + qrow) * d + h * HEAD_DIM + 4 * hh;
        auto put = [&](__bf16* dst, const float x0, const float x1, const float x2, const float x3) {
            bf16x4 p1, p2, p3;
            __bf16 b1, b2, b3;
            split3(x0, b1, b2, b3); p1[0] = b1; p2[0] = b2; p3[0] = b3;
            split3(x1, b1, b2, b3); p1[1] = b1; p2[1] = b2; p3[1] = b3;
            split3(x2, b1, b2, b3); p1[2] = b1; p2[2] = b2; p3[2] = b3;
            split3(x3, b1, b2, b3); p1[3] = b1; p2[3] = b2; p3[3] = b3;
            *reinterpret_cast<bf16x4*>(dst) = p1;
            *reinterpret_cast<bf16x4*>(dst + out_plane) = p2;
            *reinterpret_cast<bf16x4*>(dst + 2 * out_plane) = p3;
        };
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            put(op + 8 * g, o0[4 * g + 0] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
            put(op + 32 + 8 * g, o1[4 * g + 0] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
        }
    }
}
// q|k|v as three bf16 planes [3][B*S][3d] (plane stride in_plane elements) -> context as three bf16 planes [3][B*S][d]
hipError_t launch_encoder_attention_x3(const void* qkv_planes, size_t in_plane, void* ctx_planes, size_t out_plane, int B, int S, int H, hipStream_t s) {
    if ((in_plane & 7) || (out_plane & 3) || ((uintptr_t)qkv_planes & 15) || ((uintptr_t)ctx_planes & 7) || ((H * HEAD_DIM) & 7)) return hipErrorInvalidValue;
    static PerDeviceFlag attr_set;
    if (!attr_set.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attn_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, XA_SMEM);
        if (e != hipSuccess) return e;
        attr_set.set();
    }
    dim3 grid(((S + XA_BQ - 1) / XA_BQ) * H * B);
    hipLaunchKernelGGL(enc_attn_x3_kernel, grid, dim3(256), XA_SMEM, s, reinterpret_cast<const __bf16*>(qkv_planes), reinterpret_cast<__bf16*>(ctx_planes), S, H, in_plane, out_plane);
    return hipGetLastError();
}

// workgroups of the attention kernel the runtime places on one CU (2: LDS admits two; checked on the box in round 4)
int encoder_attention_blocks_per_cu() {
    int n = -1;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, FA_SMEM);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, enc_attn_kernel<false>, 256, FA_SMEM) != hipSuccess) return -1;
    return n;
}

hipError_t launch_encoder_attention(const float* qkv, float* ctx, int B, int S, int H, hipStream_t s, void* ctx_planes, size_t plane_stride) {
    static PerDeviceFlag attr_set;
    if (!attr_set.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, FA_SMEM);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, FA_SMEM);
        if (e != hipSuccess) return e;
        attr_set.set();
    }
    dim3 grid(((S + FA_BQ - 1) / FA_BQ) * H * B);
    if (ctx_planes) {
        if ((plane_stride & 3) || ((uintptr_t)ctx_planes & 7)) return hipErrorInvalidValue;
        hipLaunchKernelGGL(enc_attn_kernel<true>, grid, dim3(256), FA_SMEM, s, qkv, reinterpret_cast<float*>(ctx_planes), S, H, plane_stride);
    } else {
        hipLaunchKernelGGL(enc_attn_kernel<false>, grid, dim3(256), FA_SMEM, s, qkv, ctx, S, H, (size_t)0);
    }
    return hipGetLastError();
}

}  // namespace wt
