// XCD-fused decoder layer for gfx950 (MI355X): 3 launches per decoder layer instead of 7.
//
// The classic decode step (kernels_decoder.hip) is a chain of 7 dependent launches per layer because every GEMV needs the
// WHOLE output of its predecessor (an all-to-all exchange = a kernel boundary, ~3.6 us of boundary + first-byte latency each).
// MI355X is 8 XCDs x 32 CUs with one L2 per XCD, and the decoder layer factors along exactly that shape:
//   * attention is head-parallel: XCD g owns heads [g*H/8, (g+1)*H/8) -- their q|k|v rows, their attention, and the COLUMNS
//     of the out-projection that multiply their context.  Wo.ctx = sum over XCDs of Wo[:, cols_g].ctx_g: each XCD emits a partial
//     [B][d] and never needs another XCD's context;
//   * the FFN is column-parallel the same way: XCD g owns fc1 rows [g*F/8, ..) and the matching fc2 columns.
// Inside a launch the 32 workgroups of an XCD exchange q / attention partials / GELU(fc1) slices through THEIR L2 (plain stores,
// an XCD-local flag barrier, L1-bypassing `sc1` loads): ~0.5 us per exchange instead of a kernel boundary.  The only chip-wide
// exchanges left are the three residual-stream reductions per layer (sum of 8 partials), done at kernel boundaries: every
// XCD re-reduces the 8 partials redundantly in its prologue (1 element per thread), so the consumer needs no further hop.
//
//   K1  prologue(h = hbase + b + sum parts) -> LN1 -> q|k|v rows of my heads (+append) -> [xcd] -> self-attention partials ->
//       [xcd] -> merge + Wo[:, my cols] . ctx                                                              => parts (8 x [B][d])
//   K2  prologue -> LN2 -> cross q rows of my heads -> [xcd] -> cross-attention partials -> [xcd] -> Wco[:, my cols] . ctx => parts
//   K3  prologue -> LN3 -> fc1 rows of my slice + GELU -> [xcd] -> W2[:, my slice] . f_slice                             => parts
//
// Placement: hardware deals workgroups round-robin over the XCDs (blocks b and b+8 share one -- observed, not promised).  The
// kernels use group = blockIdx % 8 as the XCD label, but NEVER trust it for correctness: every barrier flag carries the
// writer's HW_REG_XCC_ID and the poller compares it with its own; a mismatch or a timeout sets DecState::xcd_err, the launch
// runs to completion (every spin is bounded), and the host replays the decode on the classic 7-launch path (engine.hip).
// Within one XCD the protocol is: payload by plain stores (they stay in that XCD's L2), every storing wave drains
// (`s_waitcnt vmcnt(0)`), workgroup barrier, one flag store per workgroup; consumers poll the 32 flags of their XCD with ONE
// `sc1` load instruction (L1-bypassing, L2-served), then read the payload with `sc1` / non-temporal loads only.
//
// Reference semantics: tensorrt_llm/models/whisper/model.py:153-304, 306-369 (decoder attention / layer); numerics follow the
// HF oracle (modeling_whisper.py:468-526, 710-751).  Summation order differs from the classic path (column-sliced partial sums),
// both are within 1e-3 of the oracle's logits and bit-reproducible run to run (fixed reduction order, no atomics on data).
#include "wt_common.h"

#include <stdlib.h>

namespace wt {

namespace {

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float gelu_erf_x(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float row16_allreduce(float v) {
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v;
}
__device__ __forceinline__ float wave_allreduce(float v) {
    v = row16_allreduce(v);
    const float r0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 0));
    const float r1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 16));
    const float r2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32));
    const float r3 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float dot4f(const float4& a, const float4& b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
__device__ __forceinline__ float4 ld_nt(const float* p) {  // streamed-once data: non-temporal, L1-bypassing
    const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
}
// sc1 loads of data another workgroup OF THIS XCD stored in this launch: bypass this CU's L1, served by the shared L2
__device__ __forceinline__ float4 ld_sc1_f4(const float* base, unsigned byte_off) {
    const f4v t = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000), byte_off, 0, 16));
    return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ float ld_sc1_f(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------- XCD-local barrier
struct XcdCtx {
    int g, j;          // group label (blockIdx % 8) and slot within the group (blockIdx / 8)
    unsigned xcc;      // HW_REG_XCC_ID of the XCD this workgroup really runs on
    unsigned epoch;    // st->step + 1: flags are zeroed by wt_decoder_begin and written once per step
    unsigned* sync;    // this kernel instance's flag lines: [barrier][group][32]
    int* err;          // &DecState::xcd_err
    int* s_dead;       // LDS word: this workgroup has given up waiting (bounded spin expired or placement mismatch)
};

__device__ __forceinline__ unsigned long long realtime() { return __builtin_amdgcn_s_memrealtime(); }  // 100 MHz

// All 32 workgroups of an XCD group meet here.  Precondition: nothing.  Postcondition: every plain store any of them issued
// before the barrier is in the XCD's L2, i.e. visible to sc1 / nt loads of every other one.
//
// vmcnt counts loads and stores of a wave IN ORDER, so draining a wave's stores also waits for every load it issued earlier:
// a wave that keeps weight loads in flight across the barrier must not store; the role-split kernels give all global stores and
// waits to helper waves that never stream weights (the compute waves only execute the two workgroup barriers of this function).
__device__ __forceinline__ void xcd_barrier(const XcdCtx& c, const int which) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave: its stores have been acknowledged by L2
    __syncthreads();
    unsigned* line = c.sync + (which * XCD_GROUPS + c.g) * XCD_SLOTS;
    if (threadIdx.x == 0)
        __hip_atomic_store(line + c.j, (c.epoch << 4) | c.xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // stays in L2
    if (threadIdx.x < 64 && *c.s_dead == 0) {  // wave 0 polls all 32 flags with one load instruction
        const int lane = threadIdx.x;
        const unsigned want = c.epoch;
        const unsigned long long t0 = realtime();
        int fail = 0;
        for (;;) {
            const unsigned v = lane < XCD_SLOTS ? __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                : ((want << 4) | c.xcc);
            const bool here = (v >> 4) == want;
            if (__any(here && (v & 15u) != c.xcc)) { fail = XCD_ERR_PLACEMENT; break; }  // a peer runs on another XCD
            if (__all(here)) break;
            if (realtime() - t0 > XCD_SPIN_TICKS) { fail = XCD_ERR_TIMEOUT; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (fail && lane == 0) {
            *c.s_dead = fail;
            __hip_atomic_store(c.err, fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------- GEMV building blocks
// 2-row x NB-batch butterfly over a full wave (kernels_decoder.hip: skinny_body): a0/a1 = this lane's partial dot products of
// rows A and B with the NB activation rows.  Returns the total of (row, batch) = (lane>>5, (lane&15) + ((lane>>4)&1)*NB/2)
// in lanes with (lane & 15) < NB/2 (other lanes: garbage).
template <int NB>
__device__ __forceinline__ float reduce_pair64(const float (&a0)[NB], const float (&a1)[NB]) {
    float s1[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a0[b]), __float_as_uint(a1[b]), false, false);
        s1[b] = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
    }
    float s2[NB / 2];
#pragma unroll
    for (int b = 0; b < NB / 2; ++b) {
        auto r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(s1[b]), __float_as_uint(s1[b + NB / 2]), false, false);
        s2[b] = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
    }
    float out = 0.f;
    const int li = threadIdx.x & 15;
#pragma unroll
    for (int b = 0; b < NB / 2; ++b) {
        const float t = row16_allreduce(s2[b]);
        if (li == b) out = t;
    }
    return out;
}
// each 32-lane half holds its own row: total of batch (lane&15) + ((lane>>4)&1)*NB/2 in lanes with (lane&15) < NB/2
template <int NB>
__device__ __forceinline__ float reduce_half32(const float (&a)[NB]) {
    float s2[NB / 2];
#pragma unroll
    for (int b = 0; b < NB / 2; ++b) {
        auto r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[b]), __float_as_uint(a[b + NB / 2]), false, false);
        s2[b] = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
    }
    float out = 0.f;
    const int li = threadIdx.x & 15;
#pragma unroll
    for (int b = 0; b < NB / 2; ++b) {
        const float t = row16_allreduce(s2[b]);
        if (li == b) out = t;
    }
    return out;
}
// each 16-lane row holds its own W row: total of batch (lane & 15) in lanes with (lane & 15) < NB
template <int NB>
__device__ __forceinline__ float reduce_row16(const float (&a)[NB]) {
    float out = 0.f;
    const int li = threadIdx.x & 15;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float t = row16_allreduce(a[b]);
        if (li == b) out = t;
    }
    return out;
}

constexpr int NBX = 8;   // batch rows per launch (B <= 8; missing rows are computed on clamped data and dropped)
constexpr int RA = 4;    // phase-A rows per wave (<= 4: 16 rows per workgroup = ffn_dim 4096 / 256 workgroups)

// Residual prologue shared by the three kernels (XcdPrologue, wt_common.h): every XCD group rebuilds the complete residual stream
//   h[b][c] = hbase[b][c] + prev_bias[c] + sum_{p < n_parts} parts_in[p][b][c]          (fixed order: bitwise reproducible)
// into ITS copy hx[g] -- one element per thread and slot --, the group meets at barrier 0, then every workgroup LayerNorms
// all B rows into LDS (two rows per wave, statistics two-pass in fp32 like kernels_decoder.hip).
__device__ __forceinline__ void prologue_reduce(const XcdPrologue& p, const XcdCtx& c, const int B, const int d) {
    const int total = B * d;
    const int chunk = ((total + XCD_SLOTS - 1) / XCD_SLOTS + 3) & ~3;
    const int e0 = c.j * chunk, e1 = min(total, e0 + chunk);
    const float* hb = p.hbase + (size_t)c.g * p.hbase_gstride;
    float* hx = p.hx + (size_t)c.g * total;
    for (int e = e0 + (int)threadIdx.x; e < e1; e += 256) {
        float v = hb[e];
        if (p.prev_bias) v += p.prev_bias[e % d];
        float part[XCD_GROUPS];
#pragma unroll
        for (int q = 0; q < XCD_GROUPS; ++q) part[q] = q < p.n_parts ? p.parts_in[(size_t)q * total + e] : 0.f;
#pragma unroll
        for (int q = 0; q < XCD_GROUPS; ++q) v += part[q];
        hx[e] = v;
    }
}

// after barrier 0: xs[b][0..d) = LayerNorm(hx[g][b]) for b < NBX (rows >= B: zeros); d <= 1024, d % 4 == 0
__device__ __forceinline__ void prologue_layernorm(const XcdPrologue& p, const XcdCtx& c, const int B, const int d, float (*xs)[1024]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* hx = p.hx + (size_t)c.g * B * d;
    float4 g[4], be[4], xv[2][4];
    int fcol[4];
    bool kfull[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        fcol[v] = min(4 * lane + 256 * v, d - 4);
        kfull[v] = (4 * lane + 256 * v) < d;
        g[v] = *reinterpret_cast<const float4*>(p.ln_w + fcol[v]);
        be[v] = *reinterpret_cast<const float4*>(p.ln_b + fcol[v]);
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int b = min(wave + 4 * jj, B - 1);
#pragma unroll
        for (int v = 0; v < 4; ++v) xv[jj][v] = ld_sc1_f4(hx, (unsigned)(((size_t)b * d + fcol[v]) * 4));
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int b = wave + 4 * jj;
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (!kfull[v] || b >= B) xv[jj][v] = make_float4(0.f, 0.f, 0.f, 0.f);
        float sum = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) sum += (xv[jj][v].x + xv[jj][v].y) + (xv[jj][v].z + xv[jj][v].w);
        const float mean = wave_allreduce(sum) / d;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (kfull[v]) {
                const float a0 = xv[jj][v].x - mean, a1 = xv[jj][v].y - mean, a2 = xv[jj][v].z - mean, a3 = xv[jj][v].w - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        const float rstd = rsqrtf(wave_allreduce(q) / d + 1e-5f);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float4 y;
            y.x = (xv[jj][v].x - mean) * rstd * g[v].x + be[v].x;
            y.y = (xv[jj][v].y - mean) * rstd * g[v].y + be[v].y;
            y.z = (xv[jj][v].z - mean) * rstd * g[v].z + be[v].z;
            y.w = (xv[jj][v].w - mean) * rstd * g[v].w + be[v].w;
            if (!kfull[v] || b >= B) y = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&xs[b][4 * lane + 256 * v]) = y;
        }
    }
    __syncthreads();
}

// Phase-A weight tile of one wave: up to RA rows x full K = d (<= 1024: 4 float4 per lane and row), requested at kernel start
struct WTileA {
    float4 w[RA][4];
};
__device__ __forceinline__ void load_tile_a(WTileA& t, const float* W, const int d, const int row0, const int nrows) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        if (r < nrows) {  // wave-uniform
            const float* wp = W + (size_t)(row0 + r) * d;
#pragma unroll
            for (int v = 0; v < 4; ++v) t.w[r][v] = ld_nt(wp + min(4 * lane + 256 * v, d - 4));
        } else {
#pragma unroll
            for (int v = 0; v < 4; ++v) t.w[r][v] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}
// acc[r][b] = partial dot products of tile row r with LDS activation row b (this lane's columns).  One 256-column chunk at a
// time with a compiler barrier in between: left alone, the scheduler hoists all 32 LDS reads (128 VGPRs) above the FMAs and
// spills -- and a scratch reload is a vector-memory load that queues behind every weight load still in flight.
__device__ __forceinline__ void tile_a_dot(const WTileA& t, const float (*xs)[1024], float (&acc)[RA][NBX]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < RA; ++r)
#pragma unroll
        for (int b = 0; b < NBX; ++b) acc[r][b] = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        float4 x[NBX];
#pragma unroll
        for (int b = 0; b < NBX; ++b) x[b] = *reinterpret_cast<const float4*>(&xs[b][4 * lane + 256 * v]);  // zero beyond d
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const float4 w = t.w[r][v];
#pragma unroll
            for (int b = 0; b < NBX; ++b)
                acc[r][b] = fmaf(w.x, x[b].x, fmaf(w.y, x[b].y, fmaf(w.z, x[b].z, fmaf(w.w, x[b].w, acc[r][b]))));
        }
        asm volatile("" ::: "memory");
    }
}

}  // namespace

// =================================================================================================== K3: FFN
// group g: fc1 rows [g*Fg, (g+1)*Fg) (Fg = F/8), slot j: ra = Fg/32 of them; then fc2 output rows [j*rb, (j+1)*rb) (rb = d/32)
// over the K-slice [g*Fg, (g+1)*Fg) -> parts_out[g][b][m].
#define XCD_STAMP(i)                                                                                 \
    do {                                                                                             \
        if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + (i)] = (long long)realtime(); \
    } while (0)

template <bool EARLY_W2>
__global__ __launch_bounds__(256, 1) void xcd_ffn_kernel(const XcdFfnParams p) {
    extern __shared__ __attribute__((aligned(16))) float xsm[];
    float(*xs)[1024] = reinterpret_cast<float(*)[1024]>(xsm);  // [NBX][1024]
    int* s_dead = reinterpret_cast<int*>(xsm + NBX * 1024);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = p.B, d = p.d, F = p.F;
    XcdCtx c;
    c.g = blockIdx.x % XCD_GROUPS;
    c.j = blockIdx.x / XCD_GROUPS;
    c.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));  // HW_REG_XCC_ID[3:0]
    c.epoch = (unsigned)p.st->step + 1u;
    c.sync = p.sync;
    c.err = &p.st->xcd_err;
    c.s_dead = s_dead;
    const int Fg = F / XCD_GROUPS, ra = Fg / XCD_SLOTS, rb = d / XCD_SLOTS;
    // ---- request the fc1 rows of this wave first: the whole prologue runs under their HBM latency
    const int rwa = (ra + 3) >> 2;                                  // rows per wave (last waves may have fewer / none)
    const int a_row0 = c.g * Fg + c.j * ra + wave * rwa;
    const int a_n = max(0, min(rwa, ra - wave * rwa));
    XCD_STAMP(0);
    WTileA ta;
    load_tile_a(ta, p.W1, d, a_row0, a_n);
    // ---- fc2 tile: rows [j*rb + wave*rwb, ..) x columns [g*Fg, +Fg)
    const int rwb = (rb + 3) >> 2;
    const int b_row0 = c.j * rb + wave * rwb;
    const int b_n = max(0, min(rwb, rb - wave * rwb));
    constexpr int RB = 8, VB = 2;                                    // rb <= 32 -> <= 8 rows per wave; Fg <= 512 -> 2 float4 per lane
    float4 tb[RB][VB];
    auto load_tb = [&]() {
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int v = 0; v < VB; ++v) {
                tb[r][v] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < b_n) tb[r][v] = ld_nt(p.W2 + (size_t)(b_row0 + r) * F + c.g * Fg + min(4 * lane + 256 * v, Fg - 4));
            }
    };
    if (EARLY_W2) load_tb();   // all 128 KB of this workgroup's weights are in flight before anything else happens
    if (tid == 0) *s_dead = __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // an earlier launch gave up: do not wait
    prologue_reduce(p.pro, c, B, d);
    XCD_STAMP(1);
    xcd_barrier(c, 0);
    XCD_STAMP(2);
    prologue_layernorm(p.pro, c, B, d, xs);
    XCD_STAMP(3);
    // ---- phase A: f[b][n] = GELU(W1[n] . LN(h[b]) + b1[n]) for my rows -> fx[g][b][n - g*Fg]
    float* fx = p.fx + (size_t)c.g * B * Fg;
    {
        float acc[RA][NBX];
        tile_a_dot(ta, xs, acc);
        const int rho = lane >> 4, li = lane & 15;
        const int my_r = rho >> 1, my_b = li + (rho & 1) * (NBX / 2);
#pragma unroll
        for (int pr = 0; pr < RA / 2; ++pr) {
            const float tot = reduce_pair64<NBX>(acc[2 * pr], acc[2 * pr + 1]);
            const int r = 2 * pr + my_r;
            if (li < NBX / 2 && r < a_n && my_b < B) {
                const int n = a_row0 + r;
                fx[(size_t)my_b * Fg + (n - c.g * Fg)] = gelu_erf_x(tot + p.b1[n]);
            }
        }
    }
    XCD_STAMP(4);
    if (!EARLY_W2) load_tb();  // requested before the barrier, lands while we wait
    xcd_barrier(c, 1);
    XCD_STAMP(5);
    // ---- phase B: parts_out[g][b][m] = W2[m][slice] . f[b][slice]
    {
        float4 xf[NBX][VB];
#pragma unroll
        for (int b = 0; b < NBX; ++b)
#pragma unroll
            for (int v = 0; v < VB; ++v)
                xf[b][v] = ld_sc1_f4(fx, (unsigned)(((size_t)min(b, B - 1) * Fg + min(4 * lane + 256 * v, Fg - 4)) * 4));
#pragma unroll
        for (int b = 0; b < NBX; ++b)
#pragma unroll
            for (int v = 0; v < VB; ++v)
                if (4 * lane + 256 * v >= Fg || b >= B) xf[b][v] = make_float4(0.f, 0.f, 0.f, 0.f);
        XCD_STAMP(6);
        float* po = p.parts_out + (size_t)c.g * B * d;
        const int rho = lane >> 4, li = lane & 15;
        const int my_r = rho >> 1, my_b = li + (rho & 1) * (NBX / 2);
#pragma unroll
        for (int pr = 0; pr < RB / 2; ++pr) {
            float a0[NBX], a1[NBX];
#pragma unroll
            for (int b = 0; b < NBX; ++b) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int v = 0; v < VB; ++v) {
                    s0 += dot4f(tb[2 * pr][v], xf[b][v]);
                    s1 += dot4f(tb[2 * pr + 1][v], xf[b][v]);
                }
                a0[b] = s0;
                a1[b] = s1;
            }
            const float tot = reduce_pair64<NBX>(a0, a1);
            const int r = 2 * pr + my_r;
            if (li < NBX / 2 && r < b_n && my_b < B) po[(size_t)my_b * d + b_row0 + r] = tot;
        }
    }
    XCD_STAMP(7);
}

// Role-split form (the product): 8 waves.  Waves 0-3 = helpers: residual prologue, LayerNorm, every global store, every XCD
// wait -- they never have weight loads outstanding, so their `vmcnt(0)` drains cost one L2 round trip.  Waves 4-7 = compute: they
// request ALL their fc1 and fc2 weights (128 KB per workgroup at medium.en) in the first microsecond, then only consume LDS:
// HBM streams from the first cycle of the launch to the last FMA while the helpers run the dependency chain beside it.
// The two roles are separate code paths (registers are allocated per path) that execute the SAME number of workgroup barriers:
//   S1 S2 = XCD barrier 0 | S3 = LayerNorm-ed rows in LDS | S4 = fc1 outputs in LDS | S5 S6 = XCD barrier 1 | S7 = f slice in LDS
#define XCD_STAMP2(i)                                                                                                     \
    do {                                                                                                                  \
        if (p.stamps && (lane == 0) && (wave == 0 || wave == 4)) p.stamps[(size_t)blockIdx.x * 16 + (wave >> 2) * 8 + (i)] = (long long)realtime(); \
    } while (0)

__device__ __forceinline__ void ffn_helper_role(const XcdFfnParams& p, const XcdCtx& c, float (*xs)[1024], float (*fs2)[512], float* fs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = p.B, d = p.d, Fg = p.F / XCD_GROUPS, ra = Fg / XCD_SLOTS;
    if (tid == 0) *c.s_dead = __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // an earlier launch gave up: do not wait
    prologue_reduce(p.pro, c, B, d);
    XCD_STAMP2(1);
    xcd_barrier(c, 0);                       // S1, S2
    XCD_STAMP2(2);
    prologue_layernorm(p.pro, c, B, d, xs);  // rows wave, wave+4 -> LDS; S3
    XCD_STAMP2(3);
    __syncthreads();                         // S4: the compute waves have left this workgroup's fc1 outputs in `fs`
    XCD_STAMP2(4);
    float* fx = p.fx + (size_t)c.g * B * Fg;
    {
        const int r = tid / NBX, b = tid % NBX;
        if (r < ra && b < B) fx[(size_t)b * Fg + c.j * ra + r] = fs[r * NBX + b];
    }
    xcd_barrier(c, 1);                       // S5, S6
    XCD_STAMP2(5);
    for (int i = tid; i < NBX * (Fg >> 2); i += 256) {  // the slice f[b][g*Fg .. +Fg) of every batch row -> LDS
        const int b = i / (Fg >> 2), c4 = i - b * (Fg >> 2);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < B) v = ld_sc1_f4(fx, (unsigned)(((size_t)b * Fg + 4 * c4) * 4));
        *reinterpret_cast<float4*>(&fs2[b][4 * c4]) = v;
    }
    __syncthreads();                         // S7
    XCD_STAMP2(6);
}

__device__ __forceinline__ void ffn_compute_role(const XcdFfnParams& p, const XcdCtx& c, float (*xs)[1024], float (*fs2)[512], float* fs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), cw = wave - 4;  // wave-uniform: SGPRs
    const int B = p.B, d = p.d, F = p.F;
    const int Fg = F / XCD_GROUPS, ra = Fg / XCD_SLOTS, rb = d / XCD_SLOTS;
    const int rwa = (ra + 3) >> 2, rwb = (rb + 3) >> 2;
    const int a_row0 = c.g * Fg + c.j * ra + cw * rwa, a_n = max(0, min(rwa, ra - cw * rwa));
    const int b_row0 = c.j * rb + cw * rwb, b_n = max(0, min(rwb, rb - cw * rwb));
    constexpr int RB = 8, VB = 2;   // rb <= 32 -> <= 8 fc2 rows per wave; Fg <= 512 -> 2 float4 per lane and row
    // ---- every weight byte of this workgroup is requested now
    WTileA ta;
    load_tile_a(ta, p.W1, d, a_row0, a_n);
    float4 tb[RB][VB];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int v = 0; v < VB; ++v) {
            tb[r][v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < b_n) tb[r][v] = ld_nt(p.W2 + (size_t)(b_row0 + r) * F + c.g * Fg + min(4 * lane + 256 * v, Fg - 4));
        }
    XCD_STAMP2(1);
    __syncthreads();  // S1
    __syncthreads();  // S2
    __syncthreads();  // S3: LayerNorm-ed rows are in LDS
    XCD_STAMP2(3);
    // ---- phase A: f[b][n] = GELU(W1[n] . LN(h[b]) + b1[n]) -> LDS
    const int rho = lane >> 4, li = lane & 15;
    const int my_r = rho >> 1, my_b = li + (rho & 1) * (NBX / 2);
    {
        float acc[RA][NBX];
        tile_a_dot(ta, xs, acc);
#pragma unroll
        for (int pr = 0; pr < RA / 2; ++pr) {
            const float tot = reduce_pair64<NBX>(acc[2 * pr], acc[2 * pr + 1]);
            const int r = 2 * pr + my_r;
            if (li < NBX / 2 && r < a_n) fs[(cw * rwa + r) * NBX + my_b] = gelu_erf_x(tot + p.b1[a_row0 + r]);
        }
    }
    XCD_STAMP2(4);
    __syncthreads();  // S4
    __syncthreads();  // S5
    __syncthreads();  // S6
    __syncthreads();  // S7: this XCD's whole f slice is in LDS
    XCD_STAMP2(6);
    // ---- phase B: parts_out[g][b][m] = W2[m][slice] . f[b][slice]
    float4 xf[NBX][VB];
#pragma unroll
    for (int b = 0; b < NBX; ++b)
#pragma unroll
        for (int v = 0; v < VB; ++v) {
            const int col = 4 * lane + 256 * v;
            xf[b][v] = col < Fg ? *reinterpret_cast<const float4*>(&fs2[b][col]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    float* po = p.parts_out + (size_t)c.g * B * d;
#pragma unroll
    for (int pr = 0; pr < RB / 2; ++pr) {
        float a0[NBX], a1[NBX];
#pragma unroll
        for (int b = 0; b < NBX; ++b) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int v = 0; v < VB; ++v) {
                s0 += dot4f(tb[2 * pr][v], xf[b][v]);
                s1 += dot4f(tb[2 * pr + 1][v], xf[b][v]);
            }
            a0[b] = s0;
            a1[b] = s1;
        }
        const float tot = reduce_pair64<NBX>(a0, a1);
        const int r = 2 * pr + my_r;
        if (li < NBX / 2 && r < b_n && my_b < B) po[(size_t)my_b * d + b_row0 + r] = tot;
    }
    XCD_STAMP2(7);
}

__global__ __launch_bounds__(512, 2) void xcd_ffn_rs_kernel(const XcdFfnParams p) {
    extern __shared__ __attribute__((aligned(16))) float xsm[];
    float(*xs)[1024] = reinterpret_cast<float(*)[1024]>(xsm);                    // [NBX][1024] LayerNorm-ed residual rows
    float(*fs2)[512] = reinterpret_cast<float(*)[512]>(xsm + NBX * 1024);        // [NBX][512]  this XCD's GELU(fc1) slice
    float* fs = xsm + NBX * 1024 + NBX * 512;                                     // [16][NBX]   this workgroup's fc1 outputs
    XcdCtx c;
    c.g = blockIdx.x % XCD_GROUPS;
    c.j = blockIdx.x / XCD_GROUPS;
    c.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));  // HW_REG_XCC_ID[3:0]
    c.epoch = (unsigned)p.st->step + 1u;
    c.sync = p.sync;
    c.err = &p.st->xcd_err;
    c.s_dead = reinterpret_cast<int*>(fs + 16 * NBX);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    XCD_STAMP2(0);
    if (wave < 4) ffn_helper_role(p, c, xs, fs2, fs);
    else ffn_compute_role(p, c, xs, fs2, fs);
}

hipError_t launch_xcd_ffn(const XcdFfnParams& p, hipStream_t s) {
    if (p.B < 1 || p.B > NBX || p.d > 1024 || (p.d % 128) || p.F > 4096 || (p.F % 256) || p.pro.n_parts < 0 || p.pro.n_parts > XCD_GROUPS)
        return hipErrorInvalidValue;
    constexpr int smem = (NBX * 1024 + 4) * (int)sizeof(float);
    static PerDeviceFlag attr_set;
    if (!attr_set.get()) {
        for (const void* f : {reinterpret_cast<const void*>(xcd_ffn_kernel<true>), reinterpret_cast<const void*>(xcd_ffn_kernel<false>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) return e;
        }
        attr_set.set();
    }
    constexpr int smem_rs = (NBX * 1024 + NBX * 512 + 16 * NBX + 4) * (int)sizeof(float);
    static PerDeviceFlag attr_rs;
    if (!attr_rs.get()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(xcd_ffn_rs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem_rs);
        if (e != hipSuccess) return e;
        attr_rs.set();
    }
    static const int variant = getenv("WT_XCD_FFN_VARIANT") ? atoi(getenv("WT_XCD_FFN_VARIANT")) : 2;
    if (variant == 0) hipLaunchKernelGGL(xcd_ffn_kernel<false>, dim3(XCD_BLOCKS), dim3(256), smem, s, p);
    else if (variant == 1) hipLaunchKernelGGL(xcd_ffn_kernel<true>, dim3(XCD_BLOCKS), dim3(256), smem, s, p);
    else hipLaunchKernelGGL(xcd_ffn_rs_kernel, dim3(XCD_BLOCKS), dim3(512), smem_rs, s, p);
    return hipGetLastError();
}

// =================================================================================================== residual finish
// h[b][c] = hbase[g=0][b][c] + prev_bias[c] + sum of the 8 partials: the complete residual stream for a consumer that is not
// XCD-partitioned (final LayerNorm + vocabulary projection).  One element per thread.
__global__ __launch_bounds__(256) void xcd_finish_kernel(const XcdPrologue p, float* __restrict__ out, const int total, const int d) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float v = p.hbase[e];
    if (p.prev_bias) v += p.prev_bias[e % d];
    for (int q = 0; q < p.n_parts; ++q) v += p.parts_in[(size_t)q * total + e];
    out[e] = v;
}
hipError_t launch_xcd_finish(const XcdPrologue& p, float* out, int B, int d, hipStream_t s) {
    const int total = B * d;
    hipLaunchKernelGGL(xcd_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, s, p, out, total, d);
    return hipGetLastError();
}

// =================================================================================================== census (tests / probes)
// out[blockIdx] = HW_REG_XCC_ID: lets a test check the "blocks b and b+8 share an XCD" observation on the box it runs on
__global__ void xcd_census_kernel(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
}
hipError_t launch_xcd_census(int* out, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(xcd_census_kernel, dim3(blocks), dim3(256), 0, s, out);
    return hipGetLastError();
}

}  // namespace wt
