// Decoder-step HIP kernels for gfx950 (MI355X).  One greedy step over a batch of B utterances is
//   embed -> per layer { LN+QKV skinny GEMM (+ in-place self-KV append) -> self decode-attention ->
//   out-proj (+residual) -> LN+q skinny GEMM -> cross decode-attention -> out-proj (+residual) ->
//   LN+fc1+GELU -> fc2 (+residual) } -> final LN + vocabulary projection -> logits processors + argmax.
// Every kernel is HBM-bound (weights streamed once per step for the whole batch, K/V streamed once per
// utterance), so the design rules are: 16-byte coalesced loads straight to registers, many loads in flight,
// wave-level reductions with v_permlane32_swap / v_permlane16_swap / DPP, and no step-dependent kernel
// arguments (the step counters live in DecState) so a single hipGraph replays for every token.
//
// Reference semantics: tensorrt_llm/models/whisper/model.py:153-304 (WhisperDecoderAttention),
// :306-369 (WhisperDecoderLayer), :407-470 (WhisperDecoder.forward); greedy loop examples/whisper/run.py:171-227;
// processors HF generation/logits_process.py:1281-1328.  Numerics follow the HF oracle.
#include "wt_common.h"
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <type_traits>

namespace wt {

__device__ __forceinline__ float gelu_erf_d(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// sum over the 16 lanes of a DPP row; every lane of the row ends with the total
__device__ __forceinline__ float row16_allreduce_sum(float v) {
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v;
}
__device__ __forceinline__ float wave_allreduce_sum_d(float v) {
    v = row16_allreduce_sum(v);  // every lane now holds the sum of its 16-lane row; add the four rows via SGPRs
    const float r0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 0));
    const float r1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 16));
    const float r2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32));
    const float r3 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// ------------------------------------------------------------------------------------------------ embed
// x[b] = embed_tokens[ids[b][cur_len-1]] + embed_positions[pos]      (model.py:423-425; HF :1149-1156)
// fp16 decoder engines keep embed_tokens (tied to the vocabulary projection) in fp16; embed_positions stays fp32.
__device__ __forceinline__ float4 load_emb4(const float* tab, size_t elem, const bool half) {
    if (!half) return *reinterpret_cast<const float4*>(tab + elem);
    const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(tab) + elem);
    const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&r.x)), b = __half22float2(*reinterpret_cast<const __half2*>(&r.y));
    return make_float4(a.x, a.y, b.x, b.y);
}
__global__ __launch_bounds__(256) void dec_embed_kernel(const int* __restrict__ ids, int ids_ld,
                                                        const float* __restrict__ tok_emb,
                                                        const float* __restrict__ pos_emb, float* __restrict__ x, int d,
                                                        const DecState* __restrict__ st, int emb_half) {
    const int b = blockIdx.x;
    const int tok = ids[(size_t)b * ids_ld + st->cur_len[b] - 1];
    const float4* pe = reinterpret_cast<const float4*>(pos_emb + (size_t)st->pos[b] * d);
    float4* xo = reinterpret_cast<float4*>(x + (size_t)b * d);
    for (int i = threadIdx.x; i < (d >> 2); i += 256) {
        const float4 a = load_emb4(tok_emb, (size_t)tok * d + 4 * i, emb_half != 0), c = pe[i];
        xo[i] = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    }
}
hipError_t launch_dec_embed(const int* ids, int ids_ld, const float* tok_emb, const float* pos_emb, float* x, int B,
                            int d, const DecState* st, hipStream_t s, int emb_half) {
    hipLaunchKernelGGL(dec_embed_kernel, dim3(B), dim3(256), 0, s, ids, ids_ld, tok_emb, pos_emb, x, d, st, emb_half);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ skinny GEMM
// y[b][n] = epi(sum_k X(b)[k] W[n][k] + bias[n]) for NB <= 16 batch rows: W is streamed from HBM exactly once.
// A wave owns a K-slice of <= 256*V columns: its slice of all NB activation rows stays in registers for the life of the
// kernel (NB x V float4; V = 4 for NB <= 8, V = 2 for NB = 16), W rows arrive as coalesced 1 KiB wave loads (float4
// per lane), two rows per iteration with the next pair prefetched.  The 2*NB per-lane partial sums are reduced
// across the wave by a halving butterfly (v_permlane32_swap, v_permlane16_swap, then DPP row rotations), leaving
// each (row, batch) total in one lane, which applies the epilogue.  When K exceeds a slice, the NW waves of a block
// split K (nsplit = 2, 4 or 8) and combine through LDS.  Whole activation rows (K <= 1024) are staged -- and
// LayerNorm-ed -- once per block through LDS.
// WH (fp16 decoder engines, build_decoder.py --engine_precision float16): W is IEEE half, everything else (activations, accumulation,
// LayerNorm, epilogue) stays fp32.  A lane's 16-byte load then carries EIGHT columns, so the slice chunk map becomes
// column(v) = ks0 + 8*lane + 512*(v>>1) + 4*(v&1): the same 256*V columns per wave, the same activation registers and the same K-split
// plan as the fp32 form, half the weight bytes per row (V/2 loads of 1 KiB per wave and row); the halves are widened with
// v_cvt_f32_f16 right before the packed FMAs.
// GELU_IN: PROBE ONLY (wt_dbg_skinny_gelu_in): the consumer side of "fold the cross out-projection through LN3 into
// fc1" -- fc2 would have to finish x = gelu((u - mean.r).rstd + t) on its own activation slice.  Timing only (statistics are stand-ins).
template <int NB, int V, int NW, bool W_NT, bool HALF = false, bool ARGMAX = false, bool WH = false, bool GELU_IN = false>
__device__ __forceinline__ void skinny_body(const SkinnyParams& p, const int nsplit, const int KS, const int rows_per_group,
                                            const int block) {
    extern __shared__ __attribute__((aligned(16))) float sk_smem[];
    float(*xs)[1024] = reinterpret_cast<float(*)[1024]>(sk_smem);                       // [NB][1024]
    float(*comb)[NW][2 * NB] = reinterpret_cast<float(*)[NW][2 * NB]>(sk_smem + NB * 1024);  // [2][NW][2*NB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = NW / nsplit;
    const int sp = wave % nsplit, grp = wave / nsplit;
    const int ks0 = sp * KS;                       // first column of this wave's slice
    const int kend = min(p.K, ks0 + KS);           // one past its last column
    const long long group_id = (long long)block * G + grp;
    const int row_begin = (int)min((long long)p.N, group_id * rows_per_group);
    const int row_end = min(p.N, row_begin + rows_per_group);

    static_assert(!WH || (V % 2) == 0, "half weights: two activation chunks per 16-byte load");
    constexpr int VW = WH ? V / 2 : V;             // 16-byte weight loads per row and lane
    constexpr int WSZ = WH ? 2 : 4;                // bytes per weight element
    auto colu = [&](const int v) { return WH ? ks0 + 8 * lane + 512 * (v >> 1) + 4 * (v & 1) : ks0 + 4 * lane + 256 * v; };
    float4 xr[NB][V];
    bool kok[V];
#pragma unroll
    for (int v = 0; v < V; ++v) kok[v] = colu(v) < kend;
    // ---- stream W ----------------------------------------------------------------------------------------------
    // Every load below is UNCONDITIONAL from a clamped address, with out-of-range lanes zeroed afterwards: a predicated
    // load (`ok ? *p : 0`) compiles to an exec-masked branch whose merge copy makes the compiler wait for vmcnt(0) in the
    // middle of the request burst -- one full HBM round trip before the remaining loads are even issued.
    int kcol[V];   // this lane's column of slice chunk v, clamped into the slice (half weights: the 8-column load is clamped as a whole)
#pragma unroll
    for (int v = 0; v < V; ++v) kcol[v] = WH ? min(ks0 + 8 * lane + 512 * (v >> 1), kend - 8) + 4 * (v & 1) : min(colu(v), kend - 4);
    typedef float f4v __attribute__((ext_vector_type(4)));   // 16 raw bytes: four floats, or eight halves
    f4v wbuf[2][2][VW];
    auto wload = [&](int buf, int row) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const char* wp = reinterpret_cast<const char*>(p.W) + (size_t)min(row + r, p.N - 1) * p.K * WSZ;
#pragma unroll
            for (int c = 0; c < VW; ++c) {
                const f4v* src = reinterpret_cast<const f4v*>(wp + (size_t)kcol[WH ? 2 * c : c] * WSZ);
                if (W_NT) wbuf[buf][r][c] = __builtin_nontemporal_load(src);  // weights are read exactly once per step: streaming loads
                else wbuf[buf][r][c] = *src;
            }
        }
    };
    auto wget = [&](const int buf, const int r, const int v) -> float4 {   // the four weights that meet activation chunk v
        if constexpr (WH) {
            typedef _Float16 h2v __attribute__((ext_vector_type(2)));
            const f4v raw = wbuf[buf][r][v >> 1];
            // (copies first: __builtin_bit_cast applied directly to a vector-element expression `raw[i]` reads element 0 with this clang)
            const float f0 = raw[2 * (v & 1)], f1 = raw[2 * (v & 1) + 1];
            const h2v lo = __builtin_bit_cast(h2v, f0), hi = __builtin_bit_cast(h2v, f1);
            return make_float4((float)lo[0], (float)lo[1], (float)hi[0], (float)hi[1]);
        } else {
            const f4v raw = wbuf[buf][r][v];
            return make_float4(raw[0], raw[1], raw[2], raw[3]);
        }
    };
    // W values of clamped (out-of-slice) lanes are multiplied by activations that ARE zeroed below, so they need no fix-up
    // which (row, batch) total this lane ends up holding after the butterfly
    const int rho = lane >> 4, li = lane & 15;
    const int my_r = rho >> 1;
    const int my_b = li + (rho & 1) * (NB / 2);
    const bool is_out_lane = li < NB / 2;
    // The epilogue operands of the FIRST row pair (bias, residual element, append position) are requested before anything
    // else: loaded where they are used they were two dependent L2 round trips at the very end of every launch
    // (in-kernel timestamps: ~0.8 us between "weights landed" and "stores issued" for 128 packed FMAs and a butterfly).
    float bias_pf = 0.f, resid_pf = 0.f;
    int self_len_pf = 0;
    {
        const int n_pf = min(row_begin + my_r, p.N - 1), b_pf = min(my_b, p.B - 1);
        if (p.bias) bias_pf = p.bias[n_pf];
        if (p.ymode == YMODE_PLAIN) {
            if (p.resid) resid_pf = p.resid[(size_t)b_pf * p.N + n_pf];
        } else if (!ARGMAX && p.ymode == YMODE_QKV_APPEND) {
            self_len_pf = p.st->self_len[b_pf];
        }
    }

    // ---- activation slice -> registers -------------------------------------------------------------------
    // vmcnt retires in issue order, so whatever is requested first is waited for first: with LDS staging the activation
    // rows go out before the first W rows.  (Measured neutral either way: the rows were written by the previous kernel on
    // other XCDs, so they arrive from the memory side together with the first weights, ~1.6-2.2 us after the start;
    // the LayerNorm + staging + barrier that follow are ~0.4-1 us of the launch -- DESIGN.md, timestamp study.)
    // `half_staged`: y = W . [a ; X2] where `a` is the merge of attention split partials (K/2 columns, staged + merged through LDS
    // like the plain `parts` case) and X2 is read directly: the folded cross-attention query behind a two-split self-attention.
    // (a compile-time switch: as a run-time branch it pushed every GEMV of the step into register spills -- 0.6-1 us each, 10 us on
    //  the vocabulary GEMV)
    constexpr bool half_staged = HALF;
    const int Ks = half_staged ? (p.K >> 1) : p.K;   // columns that go through the LDS staging
    const bool staged = half_staged || (p.K <= 1024 && !(p.xmode == XMODE_PLAIN && (p.x_direct || p.X2) && !p.parts));
    if (!staged && row_begin < row_end) wload(0, row_begin);
    if (staged) {
        // every wave of the block needs (a slice of) the same NB whole rows: stage them once per block through LDS.
        // Wave w loads (and LayerNorm-s) rows w and w + NW; after the barrier each wave pulls its slice of all rows.
        bool kfull[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) kfull[v] = (4 * lane + 256 * v) < Ks;
        // gamma/beta are requested together with the rows (not after the statistics) to keep them off the critical path
        float4 g[4], be[4];
        int fcol[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) fcol[v] = min(4 * lane + 256 * v, Ks - 4);
        if (p.xmode == XMODE_LAYERNORM) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                g[v] = *reinterpret_cast<const float4*>(p.ln_w + fcol[v]);
                be[v] = *reinterpret_cast<const float4*>(p.ln_b + fcol[v]);
            }
        }
        float4 xv[2][4];
        if (p.parts == nullptr) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int b = min(wave + NW * j, p.B - 1);
#pragma unroll
                for (int v = 0; v < 4; ++v) xv[j][v] = *reinterpret_cast<const float4*>(p.X + (size_t)b * Ks + fcol[v]);
            }
            if (row_begin < row_end) wload(0, row_begin);
            __builtin_amdgcn_sched_barrier(0);  // all requests are in flight before the first fix-up below
        } else {
            // the activation row is the merge of the attention kernel's split partials (deferred from its tail: no ticket, no
            // write-through stores there).  Same operations in the same order as dec_attn_kernel's own merge -> same bits.
            // exactly two splits (launch_skinny checks): both partials of all 8 chunks are requested before the first use
            float4 o0[2][4], o1[2][4];
            float m0[2][4], m1[2][4], l0[2][4], l1[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int b = min(wave + NW * j, p.B - 1);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float* base = p.parts + ((size_t)b * p.parts_H + (fcol[v] >> 6)) * 2 * PART_STRIDE;
                    o0[j][v] = *reinterpret_cast<const float4*>(base + (fcol[v] & 63));
                    o1[j][v] = *reinterpret_cast<const float4*>(base + PART_STRIDE + (fcol[v] & 63));
                    m0[j][v] = base[64]; l0[j][v] = base[65];
                    m1[j][v] = base[PART_STRIDE + 64]; l1[j][v] = base[PART_STRIDE + 65];
                }
            }
            if (row_begin < row_end) wload(0, row_begin);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float Mg = fmaxf(fmaxf(-INFINITY, m0[j][v]), m1[j][v]);
                    const float w0 = __expf(m0[j][v] - Mg), w1 = __expf(m1[j][v] - Mg);
                    const float Lg = fmaf(w1, l1[j][v], fmaf(w0, l0[j][v], 0.f));
                    xv[j][v].x = fmaf(w1, o1[j][v].x, fmaf(w0, o0[j][v].x, 0.f)) / Lg;
                    xv[j][v].y = fmaf(w1, o1[j][v].y, fmaf(w0, o0[j][v].y, 0.f)) / Lg;
                    xv[j][v].z = fmaf(w1, o1[j][v].z, fmaf(w0, o0[j][v].z, 0.f)) / Lg;
                    xv[j][v].w = fmaf(w1, o1[j][v].w, fmaf(w0, o0[j][v].w, 0.f)) / Lg;
                }
        }
        if (p.xmode == XMODE_LAYERNORM) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (!kfull[v]) g[v] = be[v] = make_float4(0.f, 0.f, 0.f, 0.f);  // gamma = beta = 0 outside the row: padding lanes stay 0
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (!kfull[v] || wave + NW * j >= p.B) xv[j][v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int b = wave + NW * j;
            if (b >= NB) break;
            if (p.xmode == XMODE_LAYERNORM) {
                float sum = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v) sum += (xv[j][v].x + xv[j][v].y) + (xv[j][v].z + xv[j][v].w);
                const float mean = wave_allreduce_sum_d(sum) / Ks;
                float q = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (kfull[v]) {
                        float a = xv[j][v].x - mean, c = xv[j][v].y - mean, e = xv[j][v].z - mean, f = xv[j][v].w - mean;
                        q += (a * a + c * c) + (e * e + f * f);
                    }
                const float rstd = rsqrtf(wave_allreduce_sum_d(q) / Ks + 1e-5f);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    xv[j][v].x = (xv[j][v].x - mean) * rstd * g[v].x + be[v].x;
                    xv[j][v].y = (xv[j][v].y - mean) * rstd * g[v].y + be[v].y;
                    xv[j][v].z = (xv[j][v].z - mean) * rstd * g[v].z + be[v].z;
                    xv[j][v].w = (xv[j][v].w - mean) * rstd * g[v].w + be[v].w;
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) *reinterpret_cast<float4*>(&xs[b][4 * lane + 256 * v]) = xv[j][v];
        }
        __syncthreads();
        if (half_staged && ks0 >= Ks) {  // this wave's K-slice lies in the X2 half (slices never straddle: KS divides K/2): straight from L2
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int v = 0; v < V; ++v) xr[b][v] = *reinterpret_cast<const float4*>(p.X2 + (size_t)min(b, p.B - 1) * Ks + (kcol[v] - Ks));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int v = 0; v < V; ++v)
                    if (!kok[v] || b >= p.B) xr[b][v] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int v = 0; v < V; ++v)
                    xr[b][v] = kok[v] ? *reinterpret_cast<const float4*>(&xs[b][colu(v)]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        if (p.X2 == nullptr) {  // rows longer than the staging buffer (or x_direct): each wave loads its own K-slice straight from L2
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int v = 0; v < V; ++v)
                    xr[b][v] = *reinterpret_cast<const float4*>(p.X + (size_t)min(b, p.B - 1) * p.K + kcol[v]);
        } else {  // concatenated activations [X ; X2], K/2 columns each (K/2 is a multiple of 4: a float4 never straddles)
            const int kh = p.K >> 1;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const int col = kcol[v];
                    const float* src = col < kh ? p.X + (size_t)min(b, p.B - 1) * kh + col : p.X2 + (size_t)min(b, p.B - 1) * kh + (col - kh);
                    xr[b][v] = *reinterpret_cast<const float4*>(src);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < V; ++v)
                if (!kok[v] || b >= p.B) xr[b][v] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (GELU_IN) {   // probe: what the folded-fc1 consumer would add in front of its products (128 GELUs per lane at K = 4096)
            float4 rr[V], tt[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                rr[v] = *reinterpret_cast<const float4*>(p.ln_w + kcol[v]);
                tt[v] = *reinterpret_cast<const float4*>(p.ln_b + kcol[v]);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float mean = p.bias[min(b, p.N - 1)] * 1e-3f, rstd = 1.0f + mean;   // stand-ins for the LayerNorm statistics of row b
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    xr[b][v].x = gelu_erf((xr[b][v].x - mean * rr[v].x) * rstd + tt[v].x);
                    xr[b][v].y = gelu_erf((xr[b][v].y - mean * rr[v].y) * rstd + tt[v].y);
                    xr[b][v].z = gelu_erf((xr[b][v].z - mean * rr[v].z) * rstd + tt[v].z);
                    xr[b][v].w = gelu_erf((xr[b][v].w - mean * rr[v].w) * rstd + tt[v].w);
                }
            }
        }
    }

    // ARGMAX (a compile-time mode: its state would push the plain GEMVs of the step into register spills): running masked maximum
    // of the rows this lane finishes (rows ascend, so the lowest index wins ties)
    float am_best = -INFINITY;
    int am_bidx = 0x7fffffff;
    uint8_t am_mk[2] = {0, 0};  // mask bytes of this lane's rows, requested one iteration ahead with the weights (a load in the
                                // epilogue would be an exposed L2 round trip per row pair: measured +10 us on the 51864-row GEMV)
    bool am_at_begin = false;
    float* am_tr = nullptr;
    if constexpr (ARGMAX) {
        am_mk[0] = p.am_mask[min(row_begin + my_r, p.N - 1)];
        am_at_begin = p.st->cur_len[min(my_b, p.B - 1)] == p.am_begin_index;
        // raw-logits trace row of this step; steps enqueued past the stop test (st->done) or past the trace's last row must not write
        if (p.am_trace && !p.st->done && p.st->step < p.am_trace_steps)
            am_tr = p.am_trace + ((size_t)min(my_b, p.B - 1) * p.am_trace_steps + p.st->step) * p.N;
    }
    const int niter = (rows_per_group + 1) / 2;  // identical for every wave of the block (barriers below)
    // one iteration = two W rows; `cur` (static) is the register buffer holding them, the other one is prefetched
    auto body = [&](auto cur_c, const int it) {
        constexpr int cur = decltype(cur_c)::value;
        const int row = row_begin + 2 * it;
        const bool active = row < row_end;
        if (active && row + 2 < row_end) {
            wload(cur ^ 1, row + 2);
            if constexpr (ARGMAX) am_mk[cur ^ 1] = p.am_mask[min(row + 2 + my_r, p.N - 1)];
        }
        float out = 0.f;
        if (active) {
            float acc[2][NB];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    // packed fp32 FMA (v_pk_fma_f32): even and odd columns accumulate in the two halves of a register pair
                    typedef float f2v __attribute__((ext_vector_type(2)));
                    f2v a2 = f2v{0.f, 0.f};
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const float4 w = wget(cur, r, v);
                        a2 = __builtin_elementwise_fma(f2v{w.x, w.y}, f2v{xr[b][v].x, xr[b][v].y}, a2);
                        a2 = __builtin_elementwise_fma(f2v{w.z, w.w}, f2v{xr[b][v].z, xr[b][v].w}, a2);
                    }
                    acc[r][b] = a2[0] + a2[1];
                }
            // butterfly stage 1 (lanes l <-> l^32): lower half-wave keeps row 0, upper half-wave row 1
            float s1[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0][b]), __float_as_uint(acc[1][b]), false, false);
                s1[b] = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
            }
            // stage 2 (l <-> l^16): even 16-lane rows keep batch b, odd rows batch b + NB/2
            float s2[NB / 2];
#pragma unroll
            for (int b = 0; b < NB / 2; ++b) {
                auto r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(s1[b]), __float_as_uint(s1[b + NB / 2]), false, false);
                s2[b] = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
            }
            // stage 3: total over the 16 lanes of the row; lane li < NB/2 keeps value li
#pragma unroll
            for (int b = 0; b < NB / 2; ++b) {
                const float t = row16_allreduce_sum(s2[b]);
                if (li == b) out = t;
            }
        }
        float total = out;
        bool do_epi = active && is_out_lane;
        if (nsplit > 1) {
            if (do_epi) comb[cur][wave][my_r * NB + my_b] = out;
            __syncthreads();
            do_epi = do_epi && sp == 0;  // the first wave of each K-split group finishes its rows
            if (do_epi) {
                total = 0.f;
                for (int s = 0; s < nsplit; ++s) total += comb[cur][grp * nsplit + s][my_r * NB + my_b];
            }
        }
        const int n = row + my_r;
        if (do_epi && n < row_end && my_b < p.B) {
            float v = total + (it == 0 ? bias_pf : (p.bias ? p.bias[n] : 0.f));
            if constexpr (ARGMAX) {  // logits processors (Suppress -> SuppressAtBegin) + running argmax, logits_process.py:1281-1311
                v *= p.q_scale;
                if (am_tr) am_tr[n] = v;
                const uint8_t mk = am_mk[cur];
                if ((mk & 1) || ((mk & 2) && am_at_begin)) v = -INFINITY;
                if (argmax_better(v, n, am_best, am_bidx)) {
                    am_best = v;
                    am_bidx = n;
                }
            } else if (p.ymode == YMODE_PLAIN) {
                v *= p.q_scale;
                if (p.act) v = gelu_erf_d(v);
                const size_t off = (size_t)my_b * p.N + n;
                if (p.resid) v += it == 0 ? resid_pf : p.resid[off];
                p.Y[off] = v;
            } else {  // fused q|k|v projection: q (scaled) -> Y, k/v rows appended in place at index self_len
                const int third = n / p.d_model, nn = n - third * p.d_model;
                if (third == 0) {
                    p.Y[(size_t)my_b * p.d_model + nn] = v * p.q_scale;
                } else {
                    const int h = nn >> 6, j = nn & 63, H = p.d_model >> 6;
                    float* cache = third == 1 ? p.kcache : p.vcache;
                    const size_t at = (((size_t)my_b * H + h) * p.s_cap + self_len_pf) * HEAD_DIM + j;
                    if (p.kv_half) reinterpret_cast<__half*>(cache)[at] = __float2half(v);   // fp16 engines: resident caches in fp16
                    else cache[at] = v;
                }
            }
        }
    };
    for (int it = 0; it < niter; it += 2) {
        body(std::integral_constant<int, 0>{}, it);
        if (it + 1 < niter) body(std::integral_constant<int, 1>{}, it + 1);
    }
    if constexpr (ARGMAX) {
        // one (max, argmax) per batch row and workgroup: output lanes -> LDS (the activation staging area is free by now) ->
        // thread b scans the NW x 2 candidates of batch row b in a fixed order
        __syncthreads();
        float* cv = sk_smem;                                 // [NW][2][NB]
        int* ci = reinterpret_cast<int*>(sk_smem + NW * 2 * NB);
        if (is_out_lane) {
            cv[(wave * 2 + my_r) * NB + my_b] = am_best;
            ci[(wave * 2 + my_r) * NB + my_b] = am_bidx;
        }
        __syncthreads();
        if (tid < p.B) {
            float best = -INFINITY;
            int bidx = 0x7fffffff;
            for (int i = 0; i < NW * 2; ++i) {
                const float v = cv[i * NB + tid];
                const int ix = ci[i * NB + tid];
                if (argmax_better(v, ix, best, bidx)) {
                    best = v;
                    bidx = ix;
                }
            }
            p.am_val[(size_t)tid * p.am_ld + block] = best;
            p.am_idx[(size_t)tid * p.am_ld + block] = bidx;
        }
    }
}

struct SkinnyPlan {
    int nsplit, KS, rows_per_group, grid;
};

template <int NB, int V, int NW, bool W_NT, bool ARGMAX = false, bool WH = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void skinny_gemm_kernel(const SkinnyParams p, const int nsplit, const int KS,
                                                                             const int rows_per_group) {
    skinny_body<NB, V, NW, W_NT, false, ARGMAX, WH>(p, nsplit, KS, rows_per_group, blockIdx.x);
}

// probe (see GELU_IN above): the K > 1024 direct-activation GEMV with the GELU / LayerNorm finish in its prologue
__global__ __launch_bounds__(256, 2) void skinny_gelu_in_probe_kernel(const SkinnyParams p, const int nsplit, const int KS, const int rows_per_group) {
    skinny_body<8, 4, 4, true, false, false, false, true>(p, nsplit, KS, rows_per_group, blockIdx.x);
}

// Two GEMMs that depend on the same predecessor share ONE launch (the self-attention out-projection and the folded
// cross-attention query): blocks [0, pa.grid) run `a`, the rest run `b`.  The branch is block-uniform.
template <int NB, int V, int NW, bool HALF_B, bool WH = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void skinny_pair_kernel(const SkinnyParams a, const SkinnyPlan pa,
                                                                             const SkinnyParams b, const SkinnyPlan pb) {
    if ((int)blockIdx.x < pa.grid) skinny_body<NB, V, NW, true, false, false, WH>(a, pa.nsplit, pa.KS, pa.rows_per_group, blockIdx.x);
    else skinny_body<NB, V, NW, true, HALF_B, false, WH>(b, pb.nsplit, pb.KS, pb.rows_per_group, blockIdx.x - pa.grid);
}

template <int NB, int V, int NW>
static hipError_t skinny_plan(const SkinnyParams& p, SkinnyPlan* out, int tg_override = 0) {
    if (p.B < 1 || p.B > NB || (p.K & 3)) return hipErrorInvalidValue;
    int nsplit = (p.parts && p.X2) ? 2 : 1;   // half-staged: a wave's K-slice must not straddle the [a ; X2] seam
    while (p.K / nsplit > 256 * V || (p.K % nsplit)) {
        nsplit *= 2;
        if (nsplit > NW) return hipErrorInvalidValue;
    }
    const int KS = p.K / nsplit;
    if (KS & 3) return hipErrorInvalidValue;
    if (p.w_half && ((KS & 7) || (reinterpret_cast<uintptr_t>(p.W) & 15))) return hipErrorInvalidValue;   // eight halves per 16-byte load
    if (p.K > 1024 && p.xmode != XMODE_PLAIN) return hipErrorInvalidValue;  // LayerNorm needs whole rows staged in LDS
    if (p.X2 && (p.xmode != XMODE_PLAIN || (p.K & 7))) return hipErrorInvalidValue;
    if (p.parts && !p.X2 && (p.K > 1024 || p.K != p.parts_H * 64 || p.parts_nsplit != 2)) return hipErrorInvalidValue;
    if (p.parts && p.X2 && (p.K > 2048 || (p.K >> 1) != p.parts_H * 64 || p.parts_nsplit != 2 || (KS > (p.K >> 1)) || ((p.K >> 1) % KS))) return hipErrorInvalidValue;
    const int G = NW / nsplit;
    // row groups (waves x K-splits) per launch: measured per decode step at medium.en B = 8 -- 512: 1.64 ms, 768: 1.53, 1024: 1.475,
    // 1280: 1.478, 1536: 1.49, 2048: 1.51, 3072: 1.51.  1024 = one 4-wave block per CU, each wave streaming two row pairs with
    // the second pair prefetched behind the first (A/B knob: WT_SKINNY_TARGET)
    // (only the 8-row template: batch 16 measured 424 audio-s/s with 2048 and 398 with 1024; batch 1 / 2 / 4: 1.07 / 1.11 / 1.24 ms
    //  per step with 2048 and 1.09 / 1.12 / 1.27 with 1024; batch 6: 1.44 vs 1.41)
    static const int tg_total = tuning_env("WT_SKINNY_TARGET") ? atoi(tuning_env("WT_SKINNY_TARGET")) : (NB == 8 ? 1024 : 2048);
    const int target_groups = (tg_override > 0 ? tg_override : tg_total) / nsplit;
    int rows_per_group = 2 * ((p.N + 2 * target_groups - 1) / (2 * target_groups));
    if (rows_per_group < 2) rows_per_group = 2;
    const int groups = (p.N + rows_per_group - 1) / rows_per_group;
    out->nsplit = nsplit; out->KS = KS; out->rows_per_group = rows_per_group; out->grid = (groups + G - 1) / G;
    return hipSuccess;
}

template <int NB, int V, int NW>
static hipError_t skinny_smem_attr() {
    constexpr int smem = (NB * 1024 + 2 * NW * 2 * NB) * (int)sizeof(float);
    static PerDeviceFlag attr_set;
    if (!attr_set.get() && smem > 48 * 1024) {
        for (const void* f : {reinterpret_cast<const void*>(skinny_gemm_kernel<NB, V, NW, true>),
                              reinterpret_cast<const void*>(skinny_gemm_kernel<NB, V, NW, false>),
                              reinterpret_cast<const void*>(skinny_gemm_kernel<NB, V, NW, true, true>),
                              reinterpret_cast<const void*>(skinny_gemm_kernel<NB, V, NW, false, true>),
                              reinterpret_cast<const void*>(skinny_pair_kernel<NB, V, NW, false>),
                              reinterpret_cast<const void*>(skinny_pair_kernel<NB, V, NW, true>),
                              reinterpret_cast<const void*>(skinny_gemm_kernel<NB, V, NW, true, false, true>),
                              reinterpret_cast<const void*>(skinny_gemm_kernel<NB, V, NW, false, false, true>),
                              reinterpret_cast<const void*>(skinny_gemm_kernel<NB, V, NW, true, true, true>),
                              reinterpret_cast<const void*>(skinny_gemm_kernel<NB, V, NW, false, true, true>),
                              reinterpret_cast<const void*>(skinny_pair_kernel<NB, V, NW, false, true>),
                              reinterpret_cast<const void*>(skinny_pair_kernel<NB, V, NW, true, true>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) return e;
        }
        attr_set.set();
    }
    return hipSuccess;
}

template <int NB, int V, int NW>
static hipError_t skinny_launch_cfg(const SkinnyParams& p, hipStream_t s) {
    SkinnyPlan pl;
    hipError_t e = skinny_plan<NB, V, NW>(p, &pl);
    if (e == hipSuccess) e = skinny_smem_attr<NB, V, NW>();
    if (e != hipSuccess) return e;
    constexpr int smem = (NB * 1024 + 2 * NW * 2 * NB) * (int)sizeof(float);
#define WT_SK_LAUNCH(NT_, AM_, WH_) \
    hipLaunchKernelGGL((skinny_gemm_kernel<NB, V, NW, NT_, AM_, WH_>), dim3(pl.grid), dim3(64 * NW), smem, s, p, pl.nsplit, pl.KS, pl.rows_per_group)
    const int sel = (p.ymode == YMODE_ARGMAX ? 4 : 0) | (p.w_nt ? 2 : 0) | (p.w_half ? 1 : 0);
    switch (sel) {
    case 0: WT_SK_LAUNCH(false, false, false); break;
    case 1: WT_SK_LAUNCH(false, false, true); break;
    case 2: WT_SK_LAUNCH(true, false, false); break;
    case 3: WT_SK_LAUNCH(true, false, true); break;
    case 4: WT_SK_LAUNCH(false, true, false); break;
    case 5: WT_SK_LAUNCH(false, true, true); break;
    case 6: WT_SK_LAUNCH(true, true, false); break;
    default: WT_SK_LAUNCH(true, true, true); break;
    }
#undef WT_SK_LAUNCH
    return hipGetLastError();
}

template <int NB, int V, int NW>
static hipError_t skinny_pair_cfg(const SkinnyParams& a, const SkinnyParams& b, hipStream_t s) {
    SkinnyPlan pa, pb;
    hipError_t e = skinny_plan<NB, V, NW>(a, &pa);
    if (e == hipSuccess) e = skinny_plan<NB, V, NW>(b, &pb);
    // 8-wave workgroups (batch 9..16) are resident one per CU: a pair of more than 256 of them runs in two rounds, the second
    // half empty (measured 12.5 us per launch at medium.en batch 16 with 128 + 256 workgroups).  Re-plan both halves with fewer,
    // fatter row groups until the pair fits one round.
    if (NW == 8)
        for (int tg = 1024; e == hipSuccess && pa.grid + pb.grid > 256 && tg >= 256; tg >>= 1) {
            e = skinny_plan<NB, V, NW>(a, &pa, tg);
            if (e == hipSuccess) e = skinny_plan<NB, V, NW>(b, &pb, tg);
        }
    if (e == hipSuccess) e = skinny_smem_attr<NB, V, NW>();
    if (e != hipSuccess) return e;
    constexpr int smem = (NB * 1024 + 2 * NW * 2 * NB) * (int)sizeof(float);
    if (a.parts && a.X2) return hipErrorInvalidValue;   // the half-staged form exists for the second GEMV of a pair only
    if (a.w_half != b.w_half) return hipErrorInvalidValue;
    const dim3 grid(pa.grid + pb.grid), blk(64 * NW);
    if (a.w_half) {
        if (b.parts && b.X2) hipLaunchKernelGGL((skinny_pair_kernel<NB, V, NW, true, true>), grid, blk, smem, s, a, pa, b, pb);
        else hipLaunchKernelGGL((skinny_pair_kernel<NB, V, NW, false, true>), grid, blk, smem, s, a, pa, b, pb);
    } else {
        if (b.parts && b.X2) hipLaunchKernelGGL((skinny_pair_kernel<NB, V, NW, true>), grid, blk, smem, s, a, pa, b, pb);
        else hipLaunchKernelGGL((skinny_pair_kernel<NB, V, NW, false>), grid, blk, smem, s, a, pa, b, pb);
    }
    return hipGetLastError();
}

hipError_t launch_skinny_gelu_in_probe(const SkinnyParams& p, hipStream_t s) {
    if (p.B < 1 || p.B > 8 || p.K <= 1024 || p.xmode != XMODE_PLAIN || !p.ln_w || !p.ln_b || !p.bias) return hipErrorInvalidValue;
    SkinnyPlan pl;
    hipError_t e = skinny_plan<8, 4, 4>(p, &pl);
    if (e != hipSuccess) return e;
    constexpr int smem = (8 * 1024 + 2 * 4 * 2 * 8) * (int)sizeof(float);
    hipLaunchKernelGGL(skinny_gelu_in_probe_kernel, dim3(pl.grid), dim3(256), smem, s, p, pl.nsplit, pl.KS, pl.rows_per_group);
    return hipGetLastError();
}

int skinny_grid(const SkinnyParams& p) {  // workgroups launch_skinny will use for p (the column count of am_val / am_idx)
    SkinnyPlan pl;
    hipError_t e = p.B <= 2 ? skinny_plan<2, 4, 4>(p, &pl) : p.B <= 4 ? skinny_plan<4, 4, 4>(p, &pl) : p.B <= 8 ? skinny_plan<8, 4, 4>(p, &pl)
                                                                                                   : skinny_plan<16, 2, 8>(p, &pl);
    return e == hipSuccess ? pl.grid : -1;
}

hipError_t launch_skinny(const SkinnyParams& p, hipStream_t s) {
    if (p.B < 1 || p.B > 16 || (p.K & 3)) return hipErrorInvalidValue;
    if (p.parts && p.X2) return hipErrorInvalidValue;   // half-staged: pair launches only
    if (p.ymode == YMODE_ARGMAX && (!p.am_mask || !p.am_val || !p.am_idx || p.am_ld < skinny_grid(p) || p.resid || p.act)) return hipErrorInvalidValue;
    if (p.B <= 2) return skinny_launch_cfg<2, 4, 4>(p, s);
    if (p.B <= 4) return skinny_launch_cfg<4, 4, 4>(p, s);
    if (p.B <= 8) return skinny_launch_cfg<8, 4, 4>(p, s);
    return skinny_launch_cfg<16, 2, 8>(p, s);  // 16 rows: half-width K-slices, 8 waves per block
}

hipError_t launch_skinny_pair(const SkinnyParams& a, const SkinnyParams& b, hipStream_t s) {
    if (a.B != b.B || a.B < 1 || a.B > 16) return hipErrorInvalidValue;
    if (a.B <= 2) return skinny_pair_cfg<2, 4, 4>(a, b, s);
    if (a.B <= 4) return skinny_pair_cfg<4, 4, 4>(a, b, s);
    if (a.B <= 8) return skinny_pair_cfg<8, 4, 4>(a, b, s);
    return skinny_pair_cfg<16, 2, 8>(a, b, s);
}

// ------------------------------------------------------------------------------------------------ decode attention
// One query row per (utterance, head); keys/values [S][64] streamed once.  Grid (n_split, H, B).  A key is read by a group of LPK
// lanes with one 16-byte load each: fp32 caches 16 lanes x 4 dims (a block's 4 waves form 16 independent online-softmax streams),
// fp16 caches (fp16 decoder engines) 8 lanes x 8 dims (32 streams) -- either way one wave instruction reads 1 KiB of consecutive
// keys.  Streams take interleaved keys; their states are merged through LDS into one partial (o[64] unnormalised, m, l) per split.
// With n_split > 1 the splits of a (utterance, head) are merged by whichever block arrives last (arrival ticket; partials stored
// write-through and re-read with sc1 loads, cdna guide §6 G16 R1); the merge walks the partials in split order, so the result is
// bitwise reproducible whatever the arrival order.  Scores, softmax and accumulators are fp32 in both forms (the reference forces
// fp32 scores in fp16 builds too, model.py:292-295).  Output: normalised context rows out[b][h*64 .. +63].
__device__ __forceinline__ float row8_allreduce_sum(float v) {   // sum over each aligned group of 8 lanes, total in every lane
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x141, 0xf, 0xf, false));  // row_half_mirror: i <-> 7-i
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    return v;
}
// PIPE: the U keys a stream takes per iteration are handled as two half-tiles with their own registers; the loads of one half are in
// flight while the other is consumed (the same number of loads in flight as the plain form, but the exp / FMA work of a half-tile --
// a fifth of an iteration with fp16 caches, a tenth with fp32 -- no longer sits between one tile's arrival and the next request).
// NTHR = 512: eight waves per block = twice the streams, half the dependent load -> consume round trips per stream and twice the bytes
// in flight per CU (the kernel is bound by round trips x bytes in flight, not by the HBM rate: it takes 17.1 us with a layer's K/V
// resident in the Infinity Cache and 17.8 from HBM).
// SLOT (continuous mode, cross attention): slot b's keys / values live in cache row slot_row[b] of a pool, not in row b.
template <int U, bool NT, bool KVH, bool PIPE = false, bool ALIVE = false, int NTHR = 256, bool SLOT = false>
__global__ __launch_bounds__(NTHR) void dec_attn_kernel(const DecAttnParams p) {
    constexpr int LPK = KVH ? 8 : 16;          // lanes per key
    constexpr int DPL = HEAD_DIM / LPK;        // head dims per lane: 8 or 4 (16 bytes of K or V either way)
    constexpr int NSTR = NTHR / LPK;           // online-softmax streams per block
    constexpr int KSZ = KVH ? 2 : 4;
    __shared__ float sm_o[NSTR][HEAD_DIM];
    __shared__ float sm_m[NSTR], sm_l[NSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & (LPK - 1), sid = tid / LPK;
    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = p.fixed_len > 0 ? p.fixed_len : p.st->self_len[b] + 1;
    const int chunk = (len + p.n_split - 1) / p.n_split;
    const int s_begin = split * chunk, s_end = min(len, s_begin + chunk);
    const int d = p.H * HEAD_DIM;
    // streams in use: all of them, except that an eight-wave block over a short key range (self-attention early in a transcript) works
    // as a four-wave one -- waves 4-7 leave at once (a finished wave does not hold a barrier), and the merge below walks half as many
    // partials: at 32 keys the full eight-wave form measured 3.8 us against 3.3
    static_assert(NTHR == 256 || NTHR == 512, "four- or eight-wave blocks");
    const int nkeys = s_end - s_begin;
    const int nstr = (NTHR == 256 || nkeys >= 128) ? NSTR : NSTR / 2;
    if (NTHR != 256 && sid >= nstr) return;   // wave-uniform (LPK divides 64)

    const int crow = SLOT ? max(p.slot_row[b], 0) : b;   // (an idle slot reads row 0: finite values nobody uses)
    const char* kb = reinterpret_cast<const char*>(p.kcache) + (((size_t)crow * p.H + h) * p.s_cap * HEAD_DIM + DPL * c) * KSZ;
    const char* vb = reinterpret_cast<const char*>(p.vcache) + (((size_t)crow * p.H + h) * p.s_cap * HEAD_DIM + DPL * c) * KSZ;
    typedef float f4v __attribute__((ext_vector_type(4)));   // 16 raw bytes
    auto load_tile = [&](f4v (&kk)[U], f4v (&vv)[U], const int s0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t key = (size_t)min(s0 + nstr * u, s_end - 1) * (HEAD_DIM * KSZ);
            if (NT) {  // K/V are read exactly once per step: non-temporal loads keep them from displacing weights in L2
                kk[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(kb + key));
                vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(vb + key));
            } else {
                kk[u] = *reinterpret_cast<const f4v*>(kb + key);
                vv[u] = *reinterpret_cast<const f4v*>(vb + key);
            }
        }
    };
    auto widen = [&](const f4v raw, float (&x)[DPL]) {
        if constexpr (KVH) {
            typedef _Float16 h2v __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float f = raw[i];   // (a copy first: __builtin_bit_cast of the element expression itself reads element 0)
                const h2v t = __builtin_bit_cast(h2v, f);
                x[2 * i] = (float)t[0];
                x[2 * i + 1] = (float)t[1];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = raw[i];
        }
    };
    // the first K/V tile is requested before the query prologue, whose loads and reductions then hide under its latency
    const int s_first = s_begin + sid;
    f4v kk[U], vv[U];
    constexpr int UH = PIPE ? U / 2 : U;       // keys per stream and half-tile
    auto load_half = [&](const int h, const int s0) {   // half-tile h = registers [h * UH, (h + 1) * UH); unconditional clamped loads
#pragma unroll
        for (int u = 0; u < UH; ++u) {
            const size_t key = (size_t)min(s0 + nstr * u, s_end - 1) * (HEAD_DIM * KSZ);
            if (NT) {
                kk[h * UH + u] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(kb + key));
                vv[h * UH + u] = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(vb + key));
            } else {
                kk[h * UH + u] = *reinterpret_cast<const f4v*>(kb + key);
                vv[h * UH + u] = *reinterpret_cast<const f4v*>(vb + key);
            }
        }
    };
    if constexpr (PIPE) {
        if (s_begin < s_end) {   // block-uniform; clamping keeps every address inside this block's keys
            load_half(0, s_first);
            load_half(1, s_first + nstr * UH);
        }
    } else {
        if (s_first < s_end) load_tile(kk, vv, s_first);
    }

    float q[DPL];
#pragma unroll
    for (int i = 0; i < DPL; i += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p.q + (size_t)b * d + h * HEAD_DIM + DPL * c + i);
        q[i] = t.x; q[i + 1] = t.y; q[i + 2] = t.z; q[i + 3] = t.w;
    }
    if (p.ln_h) {
        // folded query: finish the LayerNorm of the residual row here (every wave for itself: d <= 1024 floats, no barrier),
        // two-pass statistics like the LayerNorm kernels: q = (u - mean . r) * rstd + t
        float r[DPL], t[DPL];
#pragma unroll
        for (int i = 0; i < DPL; i += 4) {
            const float4 r4 = *reinterpret_cast<const float4*>(p.ln_r + h * HEAD_DIM + DPL * c + i);
            const float4 t4 = *reinterpret_cast<const float4*>(p.ln_t + h * HEAD_DIM + DPL * c + i);
            r[i] = r4.x; r[i + 1] = r4.y; r[i + 2] = r4.z; r[i + 3] = r4.w;
            t[i] = t4.x; t[i + 1] = t4.y; t[i + 2] = t4.z; t[i + 3] = t4.w;
        }
        const float* hr = p.ln_h + (size_t)b * d;
        float4 hv[4];
#pragma unroll
        for (int v = 0; v < 4; ++v)  // unconditional loads from clamped addresses, all four in flight before the first use
            hv[v] = *reinterpret_cast<const float4*>(hr + min(4 * lane + 256 * v, d - 4));
        __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise serialises them: load, wait, select, next load)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (4 * lane + 256 * v >= d) hv[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        float sum = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) sum += (hv[v].x + hv[v].y) + (hv[v].z + hv[v].w);
        const float mean = wave_allreduce_sum_d(sum) / d;
        float sq = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if ((4 * lane + 256 * v) < d) {
                const float a0 = hv[v].x - mean, a1 = hv[v].y - mean, a2 = hv[v].z - mean, a3 = hv[v].w - mean;
                sq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        const float rstd = rsqrtf(wave_allreduce_sum_d(sq) / d + 1e-5f);
#pragma unroll
        for (int i = 0; i < DPL; ++i) q[i] = (q[i] - mean * r[i]) * rstd + t[i];
    }
    // A finished row pads whatever it computes (run.py:219-226 stops a batch-1 decode at its own EOS): in the step graph that
    // wt_decoder_run switches to once a row has finished, such a row leaves here and streams no K/V.  (The test costs the dominant
    // kernel 0.3-1.8 % when it sits in the all-rows-alive graph -- measured -- which is why that graph is built without it.)
    if constexpr (ALIVE) {   // a compile-time form: the all-rows-alive graph runs the kernel exactly as it was
        if (p.alive[b] == 0) return;   // block-uniform
    }
    float m = -INFINITY, l = 0.f;
    float acc[DPL];
#pragma unroll
    for (int i = 0; i < DPL; ++i) acc[i] = 0.f;
    auto consume = [&](const int h, const int s0) {   // online-softmax update with the UH keys of half-tile h (first key s0 < s_end)
        float sc[UH];
        float mx = m;
#pragma unroll
        for (int u = 0; u < UH; ++u) {
            float kf[DPL];
            widen(kk[h * UH + u], kf);
            float dot = q[DPL - 1] * kf[DPL - 1];
#pragma unroll
            for (int i = DPL - 2; i >= 0; --i) dot = fmaf(q[i], kf[i], dot);
            sc[u] = KVH ? row8_allreduce_sum(dot) : row16_allreduce_sum(dot);
            if (s0 + nstr * u >= s_end) sc[u] = -INFINITY;
            mx = fmaxf(mx, sc[u]);
        }
        // plain form: s0 < s_end, so u = 0 is a real key and mx is finite.  PIPE consumes branch-free (a skipped consumer would make
        // the waitcnt pass merge "loads pending" with "loads waited for" and drain vmcnt(0) in front of every request): a stream
        // past its last key has every score at -inf, keeps m, and rescales by exp(0) -- or by 0 while it has seen no key at all.
        const float mxs = PIPE && mx == -INFINITY ? 0.f : mx;
        const float alpha = __expf(m - mxs);
        l *= alpha;
#pragma unroll
        for (int i = 0; i < DPL; ++i) acc[i] *= alpha;
#pragma unroll
        for (int u = 0; u < UH; ++u) {
            const float pr = __expf(sc[u] - mxs);
            l += pr;
            float vf[DPL];
            widen(vv[h * UH + u], vf);
#pragma unroll
            for (int i = 0; i < DPL; ++i) acc[i] = fmaf(pr, vf[i], acc[i]);
        }
        m = mx;
    };
    if constexpr (PIPE) {
        const int stride = nstr * U;
        const int n_it = (s_end - s_begin + stride - 1) / stride;   // block-uniform; a stream's own keys end up to one iteration earlier
        int sA = s_first, sB = s_first + nstr * UH;
        // every load of the prologue (query, LayerNorm operands) has landed before the loop: the waitcnt pass merges the loop-entry
        // state into every iteration, so a register still pending at entry costs a near-full drain per iteration (seen in the ISA
        // as vmcnt(1) / vmcnt(0) in front of the first FMA that reads q).  vmcnt(0), lgkmcnt / expcnt untouched:
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (int it = 0; it < n_it; ++it) {
            // (scheduling fences: without them the consumer of half B is hoisted above the requests of half A.  No branch in the loop
            //  body: requests are unconditional and clamped -- the last iteration re-reads this block's last key -- so that the
            //  waits stay counted, vmcnt(2 * UH), instead of vmcnt(0))
            consume(0, sA);
            sA += stride;
            __builtin_amdgcn_sched_barrier(0);
            load_half(0, sA);
            __builtin_amdgcn_sched_barrier(0);
            consume(1, sB);
            sB += stride;
            __builtin_amdgcn_sched_barrier(0);
            load_half(1, sB);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        for (int s0 = s_first; s0 < s_end; s0 += nstr * U) {  // (a register double buffer of WHOLE tiles measured 0.5-1 us SLOWER per launch)
            if (s0 != s_first) load_tile(kk, vv, s0);
            consume(0, s0);
        }
    }
#pragma unroll
    for (int i = 0; i < DPL; i += 4) *reinterpret_cast<float4*>(&sm_o[sid][DPL * c + i]) = make_float4(acc[i], acc[i + 1], acc[i + 2], acc[i + 3]);
    if (c == 0) {
        sm_m[sid] = m;
        sm_l[sid] = l;
    }
    __syncthreads();
    if (wave != 0) return;  // wave-uniform: only wave 0 publishes / merges
    // merge of the streams' partials by wave 0, with a COMPILE-TIME trip count in either case (a run-time bound keeps hipcc from issuing
    // the LDS reads of all partials ahead of the first exp: the loop then pays one LDS round trip per stream)
    float M, o = 0.f, L = 0.f;
    auto merge = [&](auto COUNT) {
        constexpr int n = decltype(COUNT)::value;
        M = sm_m[0];
#pragma unroll
        for (int i = 1; i < n; ++i) M = fmaxf(M, sm_m[i]);
        if (M > -INFINITY) {
#pragma unroll
            for (int i = 0; i < n; ++i) {
                const float w = __expf(sm_m[i] - M);  // streams with no key have m = -inf -> weight 0
                o = fmaf(w, sm_o[i][lane], o);
                L = fmaf(w, sm_l[i], L);
            }
        }
    };
    if (NTHR == 256 || nstr == NSTR) merge(std::integral_constant<int, NSTR>{});
    else merge(std::integral_constant<int, NSTR / 2>{});
    float* outp = p.out + (size_t)b * d + h * HEAD_DIM;
    if (p.n_split == 1) {
        outp[lane] = o / L;
        return;
    }
    float* pbase = p.part + ((size_t)b * p.H + h) * p.n_split * PART_STRIDE;
    float* pp = pbase + split * PART_STRIDE;
    if (p.defer_merge) {  // the consumer GEMV merges the splits while it stages its activations (kernel boundary = visibility)
        pp[lane] = o;
        if (lane == 0) {
            pp[64] = M;
            pp[65] = L;
        }
        return;
    }
    // publish (cdna guide §6 G16, write-through form): every payload word is stored sc1 by THIS wave, the wave
    // drains its stores, then one lane takes an arrival ticket; the last arriver reads every word with sc1 loads.
    __hip_atomic_store(pp + lane, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) {
        __hip_atomic_store(pp + 64, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 65, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last = 0;
    if (lane == 0) {
        const int ticket = __hip_atomic_fetch_add(p.cnt + b * p.H + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = ticket == p.n_split - 1;
    }
    last = __shfl(last, 0);
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler-only: keep the loads below the ticket
    // merge in split order; EVERY load of a published word is an agent-scope relaxed load (sc1, vector path, L1 bypass)
    float Mg = -INFINITY;
    for (int s = 0; s < p.n_split; ++s)
        Mg = fmaxf(Mg, __hip_atomic_load(pbase + s * PART_STRIDE + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    float og = 0.f, Lg = 0.f;
    for (int s = 0; s < p.n_split; ++s) {
        const float ms = __hip_atomic_load(pbase + s * PART_STRIDE + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float ls = __hip_atomic_load(pbase + s * PART_STRIDE + 65, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float os = __hip_atomic_load(pbase + s * PART_STRIDE + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float w = __expf(ms - Mg);  // a split with no keys has m = -inf, l = 0, o = 0 -> weight 0
        og = fmaf(w, os, og);
        Lg = fmaf(w, ls, Lg);
    }
    outp[lane] = og / Lg;
    if (lane == 0) p.cnt[b * p.H + h] = 0;  // re-arm the ticket for the next launch (kernel boundary orders it)
}
hipError_t launch_dec_attn(const DecAttnParams& p, hipStream_t s) {
    // A/B override: WT_ATTN_VARIANT=0 forces default-policy loads, 1 forces non-temporal loads (2: as the engine asks).  A function-local
    // static with an initialiser: latched once, thread-safe (handles of different host threads launch concurrently).
    static const int variant = [] {
        const char* e = tuning_env("WT_ATTN_VARIANT");
        return e ? atoi(e) : 2;
    }();
    const dim3 grid(p.n_split, p.H, p.B);
    const bool nt = variant == 2 ? p.nt != 0 : variant != 0;
    // non-temporal K/V loads: 18.6 vs 20.4 us per medium.en cross-attention launch; default policy only when a whole decode
    // step fits the Infinity Cache (engine.hip: wt_decoder_begin)
    // half-tile software pipeline (PIPE): measured per cross-attention launch, fp16 caches: batch 16 19.04 -> 18.69 us, batch 8
    // 13.35 -> 12.84 (two splits) / 15.28 -> 13.49 (one); fp32 caches 19.87 vs 19.98 (no gain: a tenth of an iteration is arithmetic
    // there, a fifth with fp16 caches) -- so fp16 caches only.  A/B: WT_ATTN_PIPE=0|1 forces it off / on for both.
    static const int pipe = tuning_env("WT_ATTN_PIPE") ? atoi(tuning_env("WT_ATTN_PIPE")) : -1;
    // fp32 caches: eight-wave blocks (measured per launch, 256 -> 512 threads: cross-attention, two splits 19.5 -> 18.8 us, one split
    // (batch 16) 23.9 -> 19.9; self-attention at 224 / 447 keys 6.6 -> 5.7 / 9.4 -> 8.3; below 128 keys the block works as four waves).
    // A/B: WT_ATTN_THREADS=256|512.
    static const int threads = tuning_env("WT_ATTN_THREADS") ? atoi(tuning_env("WT_ATTN_THREADS")) : 512;
    // (sixteen-wave blocks measured SLOWER than eight: cross-attention 19.2 -> 20.6 us, self-attention at 224 keys 5.9 -> 6.4)
    // (and four keys per stream and iteration stay best in the eight-wave form: U = 2 / 4 / 8 -> 19.9 / 19.8 / 20.1 us cross-attention,
    //  6.3 / 5.8 / 6.0 us self-attention at 224 keys)
    if (p.slot_row) {   // continuous mode: the cross-attention launches of the step graph (the engine's own plans only)
        if (!p.kv_half) {
            if (p.alive && nt) hipLaunchKernelGGL((dec_attn_kernel<4, true, false, false, true, 512, true>), grid, dim3(512), 0, s, p);
            else if (nt) hipLaunchKernelGGL((dec_attn_kernel<4, true, false, false, false, 512, true>), grid, dim3(512), 0, s, p);
            else hipLaunchKernelGGL((dec_attn_kernel<4, false, false, false, false, 512, true>), grid, dim3(512), 0, s, p);
        } else {
            if (p.alive && nt) hipLaunchKernelGGL((dec_attn_kernel<4, true, true, true, true, 256, true>), grid, dim3(256), 0, s, p);
            else if (nt) hipLaunchKernelGGL((dec_attn_kernel<4, true, true, true, false, 256, true>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((dec_attn_kernel<4, false, true, false, false, 256, true>), grid, dim3(256), 0, s, p);
        }
        return hipGetLastError();
    }
    if (threads == 512 && !p.kv_half && pipe != 1) {
        if (p.alive && nt) hipLaunchKernelGGL((dec_attn_kernel<4, true, false, false, true, 512>), grid, dim3(512), 0, s, p);
        else if (nt) hipLaunchKernelGGL((dec_attn_kernel<4, true, false, false, false, 512>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((dec_attn_kernel<4, false, false, false, false, 512>), grid, dim3(512), 0, s, p);
        return hipGetLastError();
    }
    // (fp16 caches stay on four-wave blocks -- they already are 32 streams, 8 lanes per key: eight waves = 64 streams measured 18.65 vs 18.6 us
    //  at batch 16 and 13.1 vs 13.1 at batch 8 with two splits)
    if (p.alive) {   // the skip-finished-rows step graph (non-temporal streaming forms only; others fall through to the plain kernels)
        if (p.kv_half && nt && pipe != 0) { hipLaunchKernelGGL((dec_attn_kernel<4, true, true, true, true>), grid, dim3(256), 0, s, p); return hipGetLastError(); }
        if (!p.kv_half && nt && pipe != 1) { hipLaunchKernelGGL((dec_attn_kernel<4, true, false, false, true>), grid, dim3(256), 0, s, p); return hipGetLastError(); }
    }
    if (p.kv_half) {
        if (nt && pipe != 0) hipLaunchKernelGGL((dec_attn_kernel<4, true, true, true>), grid, dim3(256), 0, s, p);
        else if (nt) hipLaunchKernelGGL((dec_attn_kernel<4, true, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((dec_attn_kernel<4, false, true>), grid, dim3(256), 0, s, p);
    } else {
        if (nt && pipe == 1) hipLaunchKernelGGL((dec_attn_kernel<4, true, false, true>), grid, dim3(256), 0, s, p);
        else if (nt) hipLaunchKernelGGL((dec_attn_kernel<4, true, false>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((dec_attn_kernel<4, false, false>), grid, dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ greedy select
// logits processors + argmax + pad/EOS bookkeeping + append, on device (run.py:199-226; HF utils.py:1502-1526;
// processors logits_process.py:1281-1328 in the order Suppress -> SuppressAtBegin -> Force).  Two launches:
//   greedy_select_kernel  grid (SELECT_CHUNKS, B): masked partial argmax of one vocabulary chunk (lowest index on ties);
//   greedy_finish_kernel  one block: final argmax per row, forced / pad / EOS rules, append, advance the step counters,
//                         stop test, and the NEXT step's input embedding (embed_tokens[tok] + embed_positions[pos]).
constexpr int SELECT_CHUNKS = 8;

__global__ __launch_bounds__(256) void greedy_select_kernel(const SelectParams p) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    const DecState* st = p.st;
    if (st->done) return;  // steps enqueued past the stop test are no-ops
    const int tid = threadIdx.x, chunk = blockIdx.x, b = blockIdx.y, cur_len = st->cur_len[b], step = st->step;
    const bool at_begin = cur_len == p.begin_index;
    const float* lg = p.logits + (size_t)b * p.V;
    float* tr = p.trace ? p.trace + ((size_t)b * (p.max_length - 1) + step) * p.V : nullptr;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    auto consider = [&](float x, uint8_t mk, int v) {
        if ((mk & 1) || ((mk & 2) && at_begin)) x = -INFINITY;
        if (argmax_better(x, v, best, bidx)) {
            best = x;
            bidx = v;
        }
    };
    if ((p.V & 3) == 0) {  // 16-byte path, several independent loads in flight per thread
        const int n4 = p.V >> 2, per = (n4 + SELECT_CHUNKS - 1) / SELECT_CHUNKS;
        const int i0 = chunk * per, i1 = min(n4, i0 + per);
        const float4* lg4 = reinterpret_cast<const float4*>(lg);
        const uchar4* mk4 = reinterpret_cast<const uchar4*>(p.mask);
        float4* tr4 = reinterpret_cast<float4*>(tr);
#pragma unroll 4
        for (int i = i0 + tid; i < i1; i += 256) {
            const float4 x = lg4[i];
            const uchar4 m = mk4[i];
            if (tr) tr4[i] = x;
            consider(x.x, m.x, 4 * i);
            consider(x.y, m.y, 4 * i + 1);
            consider(x.z, m.z, 4 * i + 2);
            consider(x.w, m.w, 4 * i + 3);
        }
    } else {
        const int per = (p.V + SELECT_CHUNKS - 1) / SELECT_CHUNKS;
        const int v0 = chunk * per, v1 = min(p.V, v0 + per);
        for (int v = v0 + tid; v < v1; v += 256) {
            const float x = lg[v];
            if (tr) tr[v] = x;
            consider(x, p.mask[v], v);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bidx, o);
        if (argmax_better(ov, oi, best, bidx)) {
            best = ov;
            bidx = oi;
        }
    }
    if ((tid & 63) == 0) {
        s_val[tid >> 6] = best;
        s_idx[tid >> 6] = bidx;
    }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 4; ++i)
            if (argmax_better(s_val[i], s_idx[i], best, bidx)) {
                best = s_val[i];
                bidx = s_idx[i];
            }
        p.part_val[b * SELECT_CHUNKS + chunk] = best;
        p.part_idx[b * SELECT_CHUNKS + chunk] = bidx;
    }
}

// Final masked argmax of every row over its n_parts partial results (lowest index on ties), shared by the two finish kernels.  All rows
// at once: a group of `lpb` lanes per row (64 at B <= 4, 32 at B <= 8, 16 at B <= 16), every load of the block in flight together -- the
// partials were written by other XCDs a kernel ago, so this is one memory round trip; row by row it was B of them (measured 14 us per
// step at B = 8).  The candidates do not depend on DecState: the first 8 x lpb of a row are requested BEFORE it is read (one dependent
// round trip fewer), by unconditional clamped loads -- `finish_request` / `finish_argmax` are the two halves.
struct FinishCand {
    float pv[8];
    int pi[8];
};
__device__ __forceinline__ void finish_request(const SelectParams& p, const int b, const int l, const int lpb, FinishCand& c) {
    const size_t pbase = (size_t)min(b, p.B - 1) * p.n_parts;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = min(l + k * lpb, p.n_parts - 1);
        c.pv[k] = p.part_val[pbase + i];
        c.pi[k] = p.part_idx[pbase + i];
    }
}
__device__ __forceinline__ int finish_argmax(const SelectParams& p, const int b, const int l, const int lpb, const FinishCand& c) {
    const size_t pbase = (size_t)min(b, p.B - 1) * p.n_parts;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    if (b < p.B) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (l + k * lpb < p.n_parts && argmax_better(c.pv[k], c.pi[k], best, bidx)) {
                best = c.pv[k];
                bidx = c.pi[k];
            }
        for (int i = l + 8 * lpb; i < p.n_parts; i += lpb) {
            const float v = p.part_val[pbase + i];
            const int ix = p.part_idx[pbase + i];
            if (argmax_better(v, ix, best, bidx)) {
                best = v;
                bidx = ix;
            }
        }
    }
    for (int o = lpb >> 1; o >= 1; o >>= 1) {   // lpb <= 64 and a power of two: a group never straddles a wave
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bidx, o);
        if (argmax_better(ov, oi, best, bidx)) {
            best = ov;
            bidx = oi;
        }
    }
    if ((unsigned)bidx >= (unsigned)p.V) bidx = p.eos;   // unreachable with a NaN-aware argmax; never index the embedding out of range
    return bidx;
}
// input of the next step for every row: x[b] = embed_tokens[tok_b] + embed_positions[pos_b]   (model.py:423-425)
__device__ __forceinline__ void finish_embed(const SelectParams& p, const int* s_tok, const int* s_pos, const int tid) {
    const int d4 = p.d_model >> 2;
    for (int i = tid; i < p.B * d4; i += 256) {
        const int b = i / d4, c = i - b * d4;
        const float4 a = load_emb4(p.tok_emb, (size_t)s_tok[b] * p.d_model + 4 * c, p.emb_half != 0);
        const float4 q = reinterpret_cast<const float4*>(p.pos_emb + (size_t)s_pos[b] * p.d_model)[c];
        reinterpret_cast<float4*>(p.next_x + (size_t)b * p.d_model)[c] = make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w);
    }
}

__global__ __launch_bounds__(256) void greedy_finish_kernel(const SelectParams p) {
    __shared__ int s_tok[MAX_ROWS];
    __shared__ int s_pos[MAX_ROWS];
    __shared__ int s_unf[MAX_ROWS];
    DecState* st = p.st;
    const int tid = threadIdx.x;
    const int lpb = p.B <= 4 ? 64 : p.B <= 8 ? 32 : 16;
    const int b = tid / lpb, l = tid % lpb, bc = min(b, p.B - 1);
    FinishCand cand;
    finish_request(p, b, l, lpb, cand);
    const int unf = p.unfinished[bc];
    // the rows of a wt_decoder_begin batch stay aligned: row 0's counters are everybody's
    const int done = st->done, cur_len = st->cur_len[0], step = st->step, pos = st->pos[0], self_len = st->self_len[0], seq = st->seq + 1,
              epoch = st->epoch;
    if (done) {         // block-uniform: steps enqueued past the stop test are no-ops (the host still sees them retire)
        if (tid == 0) {
            st->seq = seq;
            unsigned unf_mask = 0;   // unchanged since the stop: the host may read this word instead of the stopping step's
            for (int r = 0; r < p.B; ++r) unf_mask |= p.unfinished[r] ? 1u << r : 0u;
            if (p.mailbox) __hip_atomic_store(p.mailbox, mailbox_word(epoch, seq, 1, cur_len, unf_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int forced = p.forced[cur_len];                          // ForceTokensLogitsProcessor
    const int row_eos = p.force_eos_rows ? p.force_eos_rows[bc] : -1;
    int tok = finish_argmax(p, b, l, lpb, cand);
    if (b < p.B && l == 0) {
        if (forced >= 0) tok = forced;
        if (p.force_eos_step >= 0 && step == p.force_eos_step) tok = p.eos;  // bench-only transcript length
        if (row_eos >= 0 && step == row_eos) tok = p.eos;                    // ... per row (variable-length workload)
        if (!unf) tok = p.pad;                                 // finished rows keep emitting pad
        p.ids[(size_t)b * p.max_length + cur_len] = tok;
        if (tok == p.eos) p.unfinished[b] = 0;
        s_tok[b] = tok;
        s_pos[b] = pos + 1;
        s_unf[b] = (tok == p.eos) ? 0 : unf;
    }
    __syncthreads();   // (every thread has read the counters above before any of them is advanced)
    if (tid < p.B) {
        st->cur_len[tid] = cur_len + 1;
        st->pos[tid] = pos + 1;
        st->self_len[tid] = self_len + 1;
    }
    if (tid == 0) {
        int nu = 0;
        unsigned unf_mask = 0;
        for (int r = 0; r < p.B; ++r)
            if (s_unf[r]) {
                ++nu;
                unf_mask |= 1u << r;
            }
        st->n_unfinished = nu;
        st->step = step + 1;
        const int now_done = nu == 0 || cur_len + 1 >= p.max_length;   // run.py:219-226
        if (now_done) st->done = 1;
        st->seq = seq;
        // progress report for wt_decoder_run: one self-contained 64-bit word, written through to the pinned host page (the ids
        // themselves are fetched later by a stream-ordered copy, so no release fence -- and no L2 write-back -- is needed here)
        if (p.mailbox) __hip_atomic_store(p.mailbox, mailbox_word(epoch, seq, now_done, cur_len + 1, unf_mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!p.next_x || cur_len + 1 >= p.max_length) return;
    finish_embed(p, s_tok, s_pos, tid);
}

// ---- continuous mode (wt_decoder_stream_*): every slot at its own position; a row that stops is replaced AT ONCE, by this kernel, with
// the next utterance of the waiting queue (its cross K/V were projected into a pool row beforehand): no step is spent on a finished
// row and no host round trip sits between an EOS and the next utterance's first token.  The reference gets per-utterance stopping from
// decoding one clip at a time (run.py:219-226); here the batch stays full instead.
// Host visibility: every token of an active slot also goes to the pinned id row of its pool row, and a stopping row publishes its final
// length there behind a system-scope fence -- the host never copies ids from the device in this mode.
__device__ __forceinline__ bool stream_admit(const SelectParams& p, DecState* st, const int b, int* s_tok, int* s_pos) {
    const int head = st->q_head;
    if (head >= st->q_tail) {   // nothing waiting: the slot idles (pad at its frozen position: finite, unused)
        st->slot_row[b] = -1;
        p.unfinished[b] = 0;
        s_tok[b] = p.pad;
        s_pos[b] = 0;
        return false;
    }
    const int r = st->queue[head % STREAM_QCAP];
    st->q_head = head + 1;
    st->slot_row[b] = r;
    st->cur_len[b] = 1;
    st->pos[b] = 0;
    st->self_len[b] = 0;
    st->row_force[b] = st->rec_force[r];
    p.unfinished[b] = 1;
    p.ids[(size_t)b * p.max_length] = p.start_token;
    __hip_atomic_store(p.host_ids + (size_t)r * p.max_length, p.start_token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    s_tok[b] = p.start_token;
    s_pos[b] = 0;
    return true;
}
__device__ __forceinline__ void stream_report(const SelectParams& p, DecState* st, const int seq) {   // thread 0, after the admissions
    int na = 0;
    unsigned mask = 0;
    for (int r = 0; r < p.B; ++r)
        if (st->slot_row[r] >= 0) {
            ++na;
            mask |= 1u << r;
        }
    st->n_unfinished = na;
    st->done = na == 0;
    st->seq = seq;
    if (p.mailbox) __hip_atomic_store(p.mailbox, mailbox_word(st->epoch, seq, na == 0, st->q_head & 0x7fff, mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void stream_finish_kernel(const SelectParams p) {
    __shared__ int s_tok[MAX_ROWS];
    __shared__ int s_pos[MAX_ROWS];
    __shared__ int s_fin[MAX_ROWS];
    DecState* st = p.st;
    const int tid = threadIdx.x;
    const int lpb = p.B <= 4 ? 64 : p.B <= 8 ? 32 : 16;
    const int b = tid / lpb, l = tid % lpb, bc = min(b, p.B - 1);
    FinishCand cand;
    finish_request(p, b, l, lpb, cand);
    const int row = st->slot_row[bc], cur_len = st->cur_len[bc], pos = st->pos[bc], eos_at = st->row_force[bc];
    const int done = st->done, seq = st->seq + 1;
    if (done) {   // block-uniform: nothing active, nothing waiting -- a surplus step behind the last utterance
        if (tid == 0) {
            st->seq = seq;
            if (p.mailbox) __hip_atomic_store(p.mailbox, mailbox_word(st->epoch, seq, 1, st->q_head & 0x7fff, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int forced = p.forced[min(cur_len, p.max_length)];
    int tok = finish_argmax(p, b, l, lpb, cand);
    if (b < p.B && l == 0) {
        int fin = 0;
        if (row >= 0) {
            if (forced >= 0) tok = forced;
            if (eos_at >= 0 && cur_len - 1 == eos_at) tok = p.eos;   // bench only: this utterance's transcript length
            p.ids[(size_t)b * p.max_length + cur_len] = tok;
            __hip_atomic_store(p.host_ids + (size_t)row * p.max_length + cur_len, tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            fin = tok == p.eos || cur_len + 1 >= p.max_length;        // this utterance's own stop test (run.py:219-226 at batch 1)
            if (fin) {   // release: every id of the row (this thread's own stores, this step's and the earlier kernels') is in host memory
                         // before its length says so
                __hip_atomic_store(p.host_len + row, cur_len + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                st->cur_len[b] = cur_len + 1;
                st->pos[b] = pos + 1;
                st->self_len[b] = pos + 1;
                s_tok[b] = tok;
                s_pos[b] = pos + 1;
            }
        }
        s_fin[b] = fin || row < 0;   // stopped just now, or idle: both take the next waiting utterance if there is one
    }
    __syncthreads();
    if (tid == 0) {
        for (int r = 0; r < p.B; ++r)
            if (s_fin[r]) stream_admit(p, st, r, s_tok, s_pos);
        st->step += 1;
        stream_report(p, st, seq);
    }
    __syncthreads();
    if (!p.next_x) return;
    finish_embed(p, s_tok, s_pos, tid);
}

// host -> device hand-over of prepared utterances (stream-ordered behind their cross-K/V projection): append to the waiting queue,
// then fill idle slots directly -- so a decode that had drained (or has not started) begins without a wasted step
__global__ __launch_bounds__(256) void stream_publish_kernel(const SelectParams p, const StreamPublish pub) {
    __shared__ int s_tok[MAX_ROWS];
    __shared__ int s_pos[MAX_ROWS];
    __shared__ int s_new[MAX_ROWS];
    DecState* st = p.st;
    const int tid = threadIdx.x;
    if (tid == 0) {
        int tail = st->q_tail;
        for (int i = 0; i < pub.n; ++i) {
            st->queue[tail % STREAM_QCAP] = pub.rows[i];
            st->rec_force[pub.rows[i]] = pub.force[i];
            ++tail;
        }
        st->q_tail = tail;
        for (int r = 0; r < p.B; ++r) s_new[r] = st->slot_row[r] < 0 ? (stream_admit(p, st, r, s_tok, s_pos) ? 1 : 0) : 0;
        int na = 0;
        for (int r = 0; r < p.B; ++r) na += st->slot_row[r] >= 0;
        st->n_unfinished = na;
        st->done = na == 0;
    }
    __syncthreads();
    // the first input of the admitted slots (the others keep the input their last step left)
    const int d4 = p.d_model >> 2;
    for (int i = tid; i < p.B * d4; i += 256) {
        const int b = i / d4, c = i - b * d4;
        if (!s_new[b]) continue;
        const float4 a = load_emb4(p.tok_emb, (size_t)s_tok[b] * p.d_model + 4 * c, p.emb_half != 0);
        const float4 q = reinterpret_cast<const float4*>(p.pos_emb + (size_t)s_pos[b] * p.d_model)[c];
        reinterpret_cast<float4*>(p.next_x + (size_t)b * p.d_model)[c] = make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w);
    }
}
hipError_t launch_stream_publish(const SelectParams& p, const StreamPublish& pub, hipStream_t s) {
    if (p.B < 1 || p.B > MAX_ROWS || pub.n < 0 || pub.n > MAX_ROWS || !p.host_ids || !p.host_len || !p.next_x) return hipErrorInvalidValue;
    for (int i = 0; i < pub.n; ++i)
        if (pub.rows[i] < 0 || pub.rows[i] >= STREAM_QCAP) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stream_publish_kernel, dim3(1), dim3(256), 0, s, p, pub);
    return hipGetLastError();
}

hipError_t launch_greedy_select(const SelectParams& p, hipStream_t s) {
    if (p.B > MAX_ROWS || p.n_parts < 1 || (!p.fused && p.n_parts != SELECT_CHUNKS)) return hipErrorInvalidValue;
    if (p.stream && (!p.host_ids || !p.host_len || p.trace)) return hipErrorInvalidValue;
    if (!p.fused) hipLaunchKernelGGL(greedy_select_kernel, dim3(SELECT_CHUNKS, p.B), dim3(256), 0, s, p);  // else: done by the vocabulary GEMV
    if (p.stream) hipLaunchKernelGGL(stream_finish_kernel, dim3(1), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(greedy_finish_kernel, dim3(1), dim3(256), 0, s, p);
    return hipGetLastError();
}

__global__ void dec_init_kernel(DecState* st, int* ids, int* unfinished, int B, int max_length, int start_token, int epoch) {
    const int t = threadIdx.x;
    if (t == 0) {
        st->done = max_length <= 1 ? 1 : 0;
        st->n_unfinished = B;
        st->step = 0;
        st->seq = 0;
        st->epoch = epoch;
        st->q_head = st->q_tail = 0;
    }
    if (t < MAX_ROWS) {
        st->cur_len[t] = 1;
        st->pos[t] = 0;
        st->self_len[t] = 0;
        st->slot_row[t] = t;
        st->row_force[t] = -1;
    }
    for (int b = t; b < B; b += blockDim.x) {
        ids[(size_t)b * max_length] = start_token;
        unfinished[b] = 1;
    }
}
hipError_t launch_dec_init(DecState* st, int* ids, int* unfinished, int B, int max_length, int start_token, int epoch, hipStream_t s) {
    hipLaunchKernelGGL(dec_init_kernel, dim3(1), dim3(64), 0, s, st, ids, unfinished, B, max_length, start_token, epoch);
    return hipGetLastError();
}

__global__ void stream_init_kernel(DecState* st, int* unfinished, int B, int epoch) {
    const int t = threadIdx.x;
    if (t == 0) {
        st->done = 1;            // drained until the first publish
        st->n_unfinished = 0;
        st->step = 0;
        st->seq = 0;
        st->epoch = epoch;
        st->q_head = st->q_tail = 0;
    }
    if (t < MAX_ROWS) {
        st->cur_len[t] = 1;
        st->pos[t] = 0;
        st->self_len[t] = 0;
        st->slot_row[t] = -1;
        st->row_force[t] = -1;
        if (t < B) unfinished[t] = 0;
    }
}
hipError_t launch_stream_init(DecState* st, int* unfinished, int B, int epoch, hipStream_t s) {
    hipLaunchKernelGGL(stream_init_kernel, dim3(1), dim3(64), 0, s, st, unfinished, B, epoch);
    return hipGetLastError();
}

__global__ void set_state_kernel(DecState* st, int cur_len, int pos, int self_len) {
    st->cur_len[0] = cur_len;
    st->pos[0] = pos;
    st->self_len[0] = self_len;
    st->slot_row[0] = 0;
    st->done = 0;
    st->n_unfinished = 1;
    st->step = 0;
    st->seq = 0;
}
hipError_t launch_set_state(DecState* st, int cur_len, int pos, int self_len, hipStream_t s) {
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, s, st, cur_len, pos, self_len);
    return hipGetLastError();
}

// src [LH][src_rows][64] -> dst [LH][dst_rows][64], first n_rows rows of every (layer, head) slab
__global__ __launch_bounds__(256) void copy_cache_rows_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                              int LH, int src_rows, int dst_rows, int n_rows) {
    const long long per = (long long)n_rows * 16, total = per * LH;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long lh = i / per, r = i - lh * per;
        dst[lh * dst_rows * 16 + r] = src[lh * src_rows * 16 + r];
    }
}
hipError_t launch_copy_cache_rows(const float* src, float* dst, int LH, int src_rows, int dst_rows, int n_rows,
                                  hipStream_t s) {
    if (n_rows <= 0) return hipSuccess;
    const long long total = (long long)LH * n_rows * 16;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(copy_cache_rows_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(src),
                       reinterpret_cast<float4*>(dst), LH, src_rows, dst_rows, n_rows);
    return hipGetLastError();
}

}  // namespace wt
