// Shared declarations of the MI355X Whisper engine (internal; the public ABI is include/whisper_trtllm_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

#include "host_logic.h"  // HEAD_DIM, BlobHeader / BlobTensor, EngineDims (host-only part, also built under ASan/UBSan)

namespace wt {

// ---- device-resident decode state: every per-step quantity kernels need lives here so that one captured
// hipGraph replays for every step (nothing step-dependent is baked into kernel arguments) ------------------
// The per-ROW quantities are arrays over the decode slots: a batch started by wt_decoder_begin keeps its rows aligned (all entries
// equal), the continuous mode (wt_decoder_stream_*) refills a slot with the next waiting utterance the moment its row stops, so every
// slot runs at its own position -- what the reference's one-clip-at-a-time loop gives for free (run.py:219-226, cal_wer.py:249-287).
constexpr int MAX_ROWS = 16;       // decode slots per engine call
constexpr int STREAM_QCAP = 256;   // continuous mode: ring of waiting utterances, and the largest cross-cache pool (rows)
struct DecState {
    int done;          // stop test fired (all rows EOS, or cur_len >= max_length); continuous mode: no slot active and nothing waiting
    int n_unfinished;  // rows that have not produced EOS yet (continuous mode: active slots)
    int step;          // 0-based count of executed steps
    int seq;           // greedy_finish launches since wt_decoder_begin, no-op ones included (the host mailbox's progress counter)
    int epoch;         // which wt_decoder_begin this state belongs to (low 16 bits travel in the mailbox word)
    int q_head;        // continuous mode: utterances admitted to a slot so far (device-owned) ...
    int q_tail;        // ... of those published by the host (stream_publish_kernel)
    int pad0;
    int cur_len[MAX_ROWS];    // tokens in row b of `ids` so far (prompt included)
    int pos[MAX_ROWS];        // decoder position of the token being fed to row b
    int self_len[MAX_ROWS];   // valid keys already in row b's self cache; the new key/value row is written at this index
    int slot_row[MAX_ROWS];   // continuous mode: cross-cache pool row of the utterance in slot b, -1 = idle slot
    int row_force[MAX_ROWS];  // continuous mode, bench only: row-local 0-based step at which slot b's utterance emits EOS, -1 = never
    int queue[STREAM_QCAP];       // continuous mode: pool rows waiting for a slot, entry i at queue[i % STREAM_QCAP]
    int rec_force[STREAM_QCAP];   // continuous mode: row_force of the utterance in pool row r
};

// Host mailbox word (64 bits, written by greedy_finish_kernel with ONE system-scope store into pinned host memory after every
// step, read by wt_decoder_run without touching the stream): [63:48] epoch, [47:32] seq, [31] done, [30:16] cur_len,
// [15:0] bit b = row b unfinished.
__host__ __device__ inline unsigned long long mailbox_word(int epoch, int seq, int done, int cur_len, unsigned unfinished_mask) {
    return ((unsigned long long)(epoch & 0xffff) << 48) | ((unsigned long long)(seq & 0xffff) << 32) |
           ((unsigned long long)(done ? 1u : 0u) << 31) | ((unsigned long long)(cur_len & 0x7fff) << 16) | (unfinished_mask & 0xffffu);
}

// ---- fp32 GEMM (encoder / cross-KV):  C[m][n] = epi( sum_k A[m][k] * W[n][k] + bias[n] ) ------------------
struct GemmParams {
    const float* A;       // row m at A + (m / a_rows_per_batch) * a_batch_stride + (m % a_rows_per_batch) * lda
    const float* W;       // [N][K] row-major (nn.Linear layout, x @ W^T)
    const float* bias;    // [N] or nullptr
    const float* resid;   // same addressing as C, or nullptr (may alias C)
    const float* pos;     // [c_rows_per_batch][N] added after the activation (encoder embed_positions) or nullptr
    float* C;             // row m at C + (m / c_rows_per_batch) * c_batch_stride + (m % c_rows_per_batch) * ldc
    float* C2;            // EPI_KV_HEADS: value cache base (C is the key cache base)
    long long a_batch_stride, c_batch_stride;
    int M, N, K, lda, ldc;
    int a_rows_per_batch, c_rows_per_batch;
    int act;              // 0 none, 1 exact-erf GELU
    int epi;              // 0 row-major, 1 split into K/V head caches [b][h][s_cap][64]
    int kv_heads, kv_cap, kv_seq_off;  // EPI_KV_HEADS: heads per row, cache capacity (rows), first row to write
    int epi_fits32;                    // set by launch_gemm_f32: every output / residual / position offset fits an int (fast epilogue)
    // launch_gemm_x3 (fp32 product on the bf16 matrix cores, exactly split operands): A and W point at plane 0 of THREE bf16 planes with
    // the fp32 operand's row layout (lda / a_batch_stride / K in ELEMENTS as usual), plane p at + p * a_plane / w_plane elements
    long long a_plane, w_plane;
    int out_split;                     // the epilogue writes C as three bf16 planes (row layout as C, plane p at + p * c_plane elements)
    long long c_plane;
    long long* dbg_stamps;             // probe builds only (wt_dbg_gemm_stamps): [tiles][8] placement + wall-clock stamps per workgroup
};
enum { EPI_ROWMAJOR = 0, EPI_KV_HEADS = 1 };

constexpr int PART_STRIDE = 68;  // attention split partial: o[64], m, l, 2 pad floats

// ---- decode-step skinny GEMM:  y[b][n] = epi( sum_k X(b)[k] * W[n][k] + bias[n] ),  b < NB <= 8 -----------
enum { XMODE_PLAIN = 0, XMODE_LAYERNORM = 1 };
enum { YMODE_PLAIN = 0, YMODE_QKV_APPEND = 1, YMODE_ARGMAX = 2 };
struct SkinnyParams {
    const float* X;        // [B][K]  (with X2: [B][K/2], columns [0, K/2))
    const float* parts;    // optional: X is the MERGE of attention split partials [B][parts_H][parts_nsplit][PART_STRIDE] (o[64], m, l)
                           //           written by dec_attn_kernel with defer_merge (K = parts_H * 64, LDS-staged path only)
    const float* X2;       // optional second activation [B][K/2] for columns [K/2, K): y = W . [X ; X2]  (x_direct path)
    const float* ln_w;     // LAYERNORM gamma/beta [K]
    const float* ln_b;
    const float* W;        // [N][K]
    const float* bias;     // [N] or nullptr
    const float* resid;    // [B][N] or nullptr (may alias Y)
    float* Y;              // PLAIN: [B][N]
    float* kcache;         // QKV_APPEND: self key/value cache of this layer [B][H][s_cap][64]; q goes to Y [B][d]
    float* vcache;
    const DecState* st;
    int B, N, K;
    int xmode;             // XMODE_*
    int w_nt;              // stream W with non-temporal loads
    int x_direct;          // PLAIN only: every wave loads its activations straight from L2 (no LDS staging)
    int act;               // 0 none, 1 GELU
    int ymode;
    int d_model, s_cap;    // QKV_APPEND
    float q_scale;         // QKV_APPEND: multiply the q third by this (head_dim^-0.5)
    int parts_nsplit, parts_H;
    int w_half;            // W (and only W) is IEEE half [N][K]: fp16 decoder engines; K-slices must be multiples of 8
    int kv_half;           // QKV_APPEND: the self caches are IEEE half (resident caches of an fp16 engine's fast path)
    // YMODE_ARGMAX (vocabulary projection of the greedy fast path): the logits never reach HBM unless a trace is requested.
    // Every workgroup reduces its rows to one masked (max, argmax) per batch row: am_val / am_idx [B][am_ld], column = workgroup.
    const uint8_t* am_mask;  // [N] bit0: always suppressed, bit1: suppressed when cur_len == am_begin_index
    float* am_val;
    int* am_idx;
    float* am_trace;       // optional raw-logits trace [B][am_trace_steps][N] (row st->step), or nullptr
    int am_ld, am_begin_index, am_trace_steps;
};

// ---- decode attention (query length 1), split over the key axis ---------------------------------------------
struct DecAttnParams {
    const float* q;        // [B][d] (already scaled)
    const float* kcache;   // [B][H][s_cap][64]
    const float* vcache;
    float* part;           // scratch [B][H][n_split][PART_STRIDE] (n_split > 1)
    int* cnt;              // arrival tickets [B][H], zero between launches (n_split > 1)
    float* out;            // [B][d] normalised attention output
    const DecState* st;
    int B, H, s_cap;
    int n_split;
    int fixed_len;         // >0: number of keys (cross attention); 0: use st->self_len + 1
    int nt;                // stream K/V with non-temporal loads (0: default cache policy)
    int defer_merge;       // n_split > 1: leave the split partials in `part` (plain stores, no ticket); the consumer GEMV merges them
    // folded cross-attention query: q holds u = s.Wq.diag(gamma).h1 (+ const) and the kernel finishes the
    // LayerNorm per row: q = (u - mean(h1) . ln_r) * rstd(h1) + ln_t.  ln_h == nullptr: q is used as it is.
    const float* ln_h;     // [B][d] residual stream whose LayerNorm statistics normalise q
    const float* ln_r;     // [d] row sums of s.Wq.diag(gamma)
    const float* ln_t;     // [d] s.(Wq.beta + bq)
    int kv_half;           // kcache / vcache are IEEE half (fp16 decoder engines keep their RESIDENT caches in fp16; fp32 arithmetic)
    const int* alive;      // optional [B]: rows with alive[b] == 0 (finished: they emit pad whatever their logits) stream no K/V; their
                           // context / partials keep the previous step's (finite) values.  nullptr: every row attends.
    const int* slot_row;   // optional [B] (continuous mode, cross attention): slot b reads cache row max(slot_row[b], 0) instead of row b
};

// A/B tuning switches (WT_NSPLIT_CROSS, WT_GEMM_NO_DMA, ...; DESIGN.md "Tuning knobs") are lab tools, not part of the C-ABI's
// contract: they are read ONLY when the process also sets WT_TUNING=1, so a stray variable in a production environment cannot
// change what the library behind the ABI does.
inline const char* tuning_env(const char* name) {
    static const bool on = [] { const char* t = getenv("WT_TUNING"); return t && t[0] == '1'; }();
    return on ? getenv(name) : nullptr;
}

// torch.argmax semantics (the reference's greedy step, run.py:205): NaN counts as the maximum, the lowest index wins ties.
// True when candidate (v, i) beats the running (best, bidx).
__device__ __forceinline__ bool argmax_better(float v, int i, float best, int bidx) {
    const bool vn = v != v, bn = best != best;
    return vn ? (!bn || i < bidx) : (!bn && (v > best || (v == best && i < bidx)));
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE property: a launcher keeps one "done" flag per device
// (`static PerDeviceFlag f; if (!f.get()) { ...set...; f.set(); }`), so a second device opened in the same process gets
// its dynamic-LDS limit raised too.
// Distinct handles may be driven from different host threads at once (runtime.WhisperPipeline does): the flag is atomic, and two
// threads that both find it clear simply both set the (idempotent) attribute.
struct PerDeviceFlag {
    std::atomic<bool> done[64] = {};
    static int cur() { int dev = 0; (void)hipGetDevice(&dev); return dev & 63; }
    bool get() const { return done[cur()].load(std::memory_order_acquire); }
    void set() { done[cur()].store(true, std::memory_order_release); }
};

// Every ABI entry point runs on its handle's device and leaves the CALLER's current device (torch's) as it found it.
struct DeviceGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev); else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// launchers (kernels_*.hip)
hipError_t launch_mel_transpose(const float* mel, float* melT, int B, int n_mels, int frames, hipStream_t s);
hipError_t launch_mel_transpose_split(const float* mel, void* planes, size_t plane_stride, int B, int n_mels, int frames, hipStream_t s);   // three bf16 planes
hipError_t launch_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int d, hipStream_t s);
hipError_t launch_gemm_f32(const GemmParams& p, hipStream_t s);
// ---- fp32 GEMM on the bf16 matrix cores.  x = b1 + b2 + b3 with bf16 b1 = rn(x), b2 = rn(x - b1), b3 = rn(x - b1 - b2) represents an fp32
// value to 2^-27 relative (three 8-bit significands cover fp32's 24 bits; bf16 has fp32's exponent range), and a product a.w is the sum of
// the six partial products a_i.w_j with i + j <= 4 up to 3 x 2^-27 relative -- below the 2^-24 rounding of an fp32 FMA.  Every partial
// product of two bf16 values is exact in the MFMA's fp32 accumulator, which accumulates as the fp32 MFMA does.  Six
// v_mfma_f32_32x32x16_bf16 replace sixteen v_mfma_f32_32x32x2_f32 per 32x32x16 block: 16 / 6 = 2.7x the fp32 MFMA peak.
hipError_t launch_gemm_x3(const GemmParams& p, hipStream_t s);
bool gemm_x3_usable(const GemmParams& p);   // shape / alignment limits of launch_gemm_x3 (else: launch_gemm_f32 on the fp32 operands)
hipError_t launch_split3(const float* x, void* planes, size_t n, size_t plane_stride, hipStream_t s);   // fp32 [n] -> three bf16 planes
hipError_t launch_layernorm_split(const float* x, const float* w, const float* b, void* planes, size_t plane_stride, int rows, int d, hipStream_t s);
hipError_t launch_encoder_attention(const float* qkv, float* ctx, int B, int S, int H, hipStream_t s, void* ctx_planes = nullptr, size_t plane_stride = 0);
    // ctx_planes != nullptr: the context goes out as three bf16 planes (the A operand of launch_gemm_x3) instead of fp32 `ctx`
int encoder_attention_blocks_per_cu();
// x3 form (both products from bf16 MFMAs of exactly split operands, fp32 scores / softmax / accumulators): q|k|v in, context out as planes
hipError_t launch_encoder_attention_x3(const void* qkv_planes, size_t in_plane, void* ctx_planes, size_t out_plane, int B, int S, int H, hipStream_t s);
// fp16-encoder path (kernels_encoder_f16.hip): `void*` operands are __half buffers
hipError_t launch_mel_transpose_h(const float* mel, void* melT, int B, int n_mels, int frames, hipStream_t s);
hipError_t launch_layernorm_h(const float* x, const float* w, const float* b, void* y, int rows, int d, hipStream_t s);
hipError_t launch_cast_h(const float* x, void* y, size_t n, hipStream_t s);
hipError_t launch_gemm_f16(const GemmParams& p, bool out_half, hipStream_t s, int force_variant = 0);  // 0: by shape; 2 / 3: see kernels_encoder_f16.hip
hipError_t launch_encoder_attention_f16(const void* qkv, void* ctx, int B, int S, int H, hipStream_t s);  // half in, half out  // p.A / p.W point at __half data

hipError_t launch_dec_embed(const int* ids, int ids_ld, const float* tok_emb, const float* pos_emb, float* x, int B,
                            int d, const DecState* st, hipStream_t s, int emb_half = 0);   // emb_half: tok_emb is IEEE half
hipError_t launch_skinny(const SkinnyParams& p, hipStream_t s);
hipError_t launch_skinny_gelu_in_probe(const SkinnyParams& p, hipStream_t s);   // probe only (tools/microbench.py fold_fc1)
int skinny_grid(const SkinnyParams& p);  // workgroups launch_skinny uses for p (column count of am_val / am_idx), -1 if p is invalid
// two independent skinny GEMMs (same batch) in ONE launch: blocks [0, grid_a) run `a`, the rest run `b`
hipError_t launch_skinny_pair(const SkinnyParams& a, const SkinnyParams& b, hipStream_t s);
hipError_t launch_dec_attn(const DecAttnParams& p, hipStream_t s);

struct SelectParams {
    const float* logits;   // [B][V]
    const uint8_t* mask;   // [V] bit0: always suppressed, bit1: suppressed when cur_len == begin_index
    const int* forced;     // [max_length] forced token for generation index i, or -1
    int* ids;              // [B][max_length]
    int* unfinished;       // [B]
    DecState* st;
    float* trace;          // optional [B][max_length-1][V]
    float* part_val;       // scratch [B][n_parts] partial maxima (n_parts = 8 chunks of greedy_select_kernel, or the workgroups
    int* part_idx;         //   of the vocabulary GEMV when the argmax is fused into its epilogue: fused != 0)
    int n_parts, fused;
    const float* tok_emb;  // next-step embedding: embed_tokens [V][d], embed_positions [T][d]
    const float* pos_emb;
    float* next_x;         // [B][d] decoder input of the next step, or nullptr
    int d_model;
    int B, V, max_length, begin_index, eos, pad, force_eos_step;
    int emb_half;          // tok_emb is IEEE half (fp16 decoder engines)
    const int* force_eos_rows;       // optional [B]: row b emits EOS at 0-based step force_eos_rows[b] (< 0: never) -- bench-only transcript lengths
    unsigned long long* mailbox;     // optional: device address of the engine's pinned host mailbox (mailbox_word above)
    // continuous mode (stream != 0): finished utterances report to the host through pinned memory, indexed by cross-cache pool row
    int stream, start_token;
    int* host_ids;                   // device address of the pinned id rows [pool][max_length]
    int* host_len;                   // device address of the pinned final lengths [pool] (0 = still decoding)
};
hipError_t launch_greedy_select(const SelectParams& p, hipStream_t s);
hipError_t launch_dec_init(DecState* st, int* ids, int* unfinished, int B, int max_length, int start_token, int epoch, hipStream_t s);
// continuous mode: empty slots, empty queue
hipError_t launch_stream_init(DecState* st, int* unfinished, int B, int epoch, hipStream_t s);
// continuous mode: append `n` prepared utterances (cross-cache pool rows, forced-EOS steps) to the waiting queue and admit into idle slots
struct StreamPublish {
    int n;
    int rows[MAX_ROWS];
    int force[MAX_ROWS];
};
hipError_t launch_stream_publish(const SelectParams& p, const StreamPublish& pub, hipStream_t s);
hipError_t launch_copy_cache_rows(const float* src, float* dst, int LH, int src_rows, int dst_rows, int n_rows,
                                  hipStream_t s);
hipError_t launch_set_state(DecState* st, int cur_len, int pos, int self_len, hipStream_t s);

// GELU (erf form) of the encoder epilogues.  erff() is ~38 vector ALU instructions in two divergent branches, executed beside the other
// workgroups' MFMAs: 10 % of a small.en fc1 launch, 4 % of medium.en's.  Branch-free form: erf(|x| / sqrt 2) = 1 - 2^Q(|x|), Q of degree 8
// without constant term (coefficients and error check: tools/fit_gelu.py), and
//   gelu(x) = x - h for x >= 0,  h for x < 0,  h = 0.5 x 2^Q(|x|)       (no 1 + erf cancellation in the negative tail)
// 13 instructions + one v_exp_f32.  |gelu_erf - exact| <= 5.1e-7 on [-12, 12] in float32 arithmetic, <= 1.3e-7 max(|x|, 1): one ulp of the
// result, the size of the rounding of x - h itself.
__device__ __forceinline__ float gelu_erf(float x) {
    const float t = fabsf(x);   // no clamp: beyond the fitted range Q(t) t keeps falling monotonically to -inf (checked in fit_gelu.py), 2^Q -> 0
    float q = -2.835055965988431e-06f;
    q = fmaf(q, t, 3.937911969842389e-05f);
    q = fmaf(q, t, -0.00018618583271745592f);
    q = fmaf(q, t, -0.00013692546053789556f);
    q = fmaf(q, t, 0.007063408847898245f);
    q = fmaf(q, t, -0.05249617248773575f);
    q = fmaf(q, t, -0.4592081904411316f);
    q = fmaf(q, t, -1.1511051654815674f);
    const float h = (0.5f * x) * __builtin_amdgcn_exp2f(q * t);
    // (fmaxf: gelu(x) >= x / 2 for x >= 0, so the max is an identity for finite x; for x = +inf, where h = inf * 0 = NaN, it returns +inf
    //  like the erf form.  x = -inf gives NaN in both forms, NaN stays NaN.)
    return x >= 0.f ? fmaxf(x - h, 0.5f * x) : h;
}

}  // namespace wt
