/*
 * whisper_trtllm_amd_debug.h — kernel-level test hooks of libwhisper_trtllm_amd.so.
 *
 * NOT part of the drop-in boundary (that is whisper_trtllm_amd.h).  These launch one HIP kernel each on
 * caller-provided device buffers so tests/test_gpu_kernels.py can check every kernel against a plain
 * fp32 torch reference at edge shapes (ragged tiles, K tails, every batch-width instantiation).
 * All pointers are device pointers; calls are asynchronous on `stream`; 0 = ok, else see wt_last_error().
 */
#ifndef WHISPER_TRTLLM_AMD_DEBUG_H
#define WHISPER_TRTLLM_AMD_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif

/* C[M][N] = act(A[M][K(lda)] W[N][K]^T + bias) (+ resid); act: 0 none, 1 erf-GELU */
int wt_dbg_gemm(const float* A, int lda, const float* W, const float* bias, const float* resid, float* C, int M, int N,
                int K, int act, void* stream);
/* the same launch from a probe build of the LDS-DMA kernel (K % 16 == 0): every workgroup leaves {HW_ID, XCC_ID, wall clock (100 MHz) at
 * entry, first tile landed, K loop done, stores drained, last store issued, 0} in stamps[workgroup][8] (tools/gemm_stamps.py) */
int wt_dbg_gemm_stamps(const float* A, int lda, const float* W, const float* bias, const float* resid, float* C, int M, int N, int K,
                       int act, long long* stamps, void* stream);
/* fp16 operands (A [M][K(lda)], W [N][K] as IEEE half), fp32 accumulate; C is half when out_half else float */
int wt_dbg_gemm_f16(const void* A, int lda, const void* W, const float* bias, const float* resid, void* C, int M, int N, int K,
                    int act, int out_half, void* stream);
/* the same with the LDS-DMA kernel forced (K % 64 == 0): variant 2 = 128x128 two-stage, 3 = 256x128 three-stage; 0 = by shape */
int wt_dbg_gemm_f16_variant(const void* A, int lda, const void* W, const float* bias, const float* resid, void* C, int M, int N, int K,
                            int act, int out_half, int variant, void* stream);
int wt_dbg_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int d, void* stream);
/* qkv [B*S][3*H*64] -> ctx [B*S][H*64], softmax(QK^T/8)V per head */
int wt_dbg_encoder_attention(const float* qkv, float* ctx, int B, int S, int H, void* stream);
/* same with fp16 qkv / ctx (fp32 scores, softmax and accumulators) */
int wt_dbg_encoder_attention_f16(const void* qkv, void* ctx, int B, int S, int H, void* stream);
/* Y[B][N] = act((X' W^T + bias) * scale) (+ resid); xmode 0: X'=X[B][K], 1: X'=LayerNorm(X) */
int wt_dbg_skinny(const float* X, const float* ln_w, const float* ln_b, const float* W, const float* bias,
                  const float* resid, float* Y, int B, int N, int K, int xmode, int act, float scale, void* stream);
/* q [B][H*64] (pre-scaled), k/v cache [B][H][s_cap][64] with the first `len` rows valid -> out [B][H*64];
 * part: scratch [B][H][n_split][68]; cnt: int [B][H], must be zero on entry and is left zero */
int wt_dbg_encoder_attention_occupancy(void);   /* workgroups of enc_attn_kernel per CU as the runtime computes it */
int wt_dbg_gemm_x3(const float* A, const float* W, const float* bias, const float* resid, void* C, int M, int N, int K, int act,
                   void* a_planes, void* w_planes, int flags, void* stream);   /* fp32 GEMM from bf16 MFMAs of exactly split operands; flags: 1 three-plane output, 2 planes already split */
int wt_dbg_encoder_attention_split(const float* qkv, void* ctx_planes, int B, int S, int H, void* stream);   /* context as three bf16 planes [3][B*S][H*64] */
int wt_dbg_encoder_attention_x3(const float* qkv, void* qkv_planes, void* ctx_planes, int B, int S, int H, int skip_split, void* stream);
int wt_dbg_skinny_gelu_in(const float* X, const float* r, const float* t, const float* W, const float* bias, const float* resid, float* Y,
                          int B, int N, int K, void* stream);   /* probe: timing only */
int wt_dbg_decode_attention(const float* q, const float* kcache, const float* vcache, float* part, int* cnt, float* out,
                            int B, int H, int s_cap, int len, int n_split, void* stream);
/* the same with the folded query: `u` [B][H*64] is finished per row as (u - mean(ln_h[b]) * ln_r) * rstd(ln_h[b]) + ln_t
 * (LayerNorm statistics of ln_h [B][H*64], eps 1e-5) before the attention — DESIGN.md §4, builder.py:_fold_cross_query */
int wt_dbg_decode_attention_folded(const float* u, const float* kcache, const float* vcache, float* part, int* cnt, float* out,
                                   const float* ln_h, const float* ln_r, const float* ln_t, int B, int H, int s_cap, int len,
                                   int n_split, void* stream);
/* decode attention with the split merge DEFERRED into the consumer: dec_attn_kernel leaves its n_split >= 2 partials in
 * `part` [B][H][n_split][68] and the out-projection Y = merge(part) . W^T + bias + resid ([B][H*64]) merges them while it
 * stages its activation rows (the cross-attention -> out-projection pair of the decode step) */
int wt_dbg_attention_then_projection(const float* q, const float* kcache, const float* vcache, float* part, const float* W,
                                     const float* bias, const float* resid, float* Y, int B, int H, int s_cap, int len, int n_split,
                                     void* stream);
/* the decode step's self-attention -> pair launch with TWO key splits and the merge deferred into BOTH halves of the pair:
 * a = softmax(q K^T) V over the first `len` cache rows (partials in `part` [B][H][2][68]); h1 = h + Wo.a + bo (Wo [d][d]);
 * u = Wf.[a ; h] + c (Wf [d][2d]) */
int wt_dbg_self_attention_then_pair(const float* q, const float* kcache, const float* vcache, float* part, const float* Wo, const float* bo,
                                    const float* h, float* h1, const float* Wf, const float* c, float* u, int B, int H, int s_cap, int len,
                                    void* stream);
/* two skinny GEMMs in one launch: Ya = Xa . Wa^T + bias_a + resid_a ([B][Na], K = Ka) and
 * Yb = [Xb ; Xb2] . Wb^T + bias_b ([B][Nb], K = Kb = 2 * columns of Xb) */
int wt_dbg_skinny_pair(const float* Xa, const float* Wa, const float* bias_a, const float* resid_a, float* Ya, int Na, int Ka,
                       const float* Xb, const float* Xb2, const float* Wb, const float* bias_b, float* Yb, int Nb, int Kb, int B,
                       void* stream);

/* ---- fp16 decoder engines: IEEE-half weights (`W`, `Wa`, `Wb`) and / or IEEE-half K/V caches; activations, accumulation, LayerNorm,
 * softmax and outputs are fp32.  Arguments as in the fp32 hooks of the same name. */
int wt_dbg_skinny_f16(const float* X, const float* ln_w, const float* ln_b, const void* W, const float* bias, const float* resid, float* Y,
                      int B, int N, int K, int xmode, int act, float scale, void* stream);
/* ln_h / ln_r / ln_t: all null (plain query) or all set (folded query) */
int wt_dbg_decode_attention_f16(const float* q, const void* kcache, const void* vcache, float* part, int* cnt, float* out, const float* ln_h,
                                const float* ln_r, const float* ln_t, int B, int H, int s_cap, int len, int n_split, void* stream);
/* two key splits, merge deferred into the out-projection (half caches, half W) */
int wt_dbg_attention_then_projection_f16(const float* q, const void* kcache, const void* vcache, float* part, const void* W, const float* bias,
                                         const float* resid, float* Y, int B, int H, int s_cap, int len, void* stream);
int wt_dbg_skinny_pair_f16(const float* Xa, const void* Wa, const float* bias_a, const float* resid_a, float* Ya, int Na, int Ka, const float* Xb,
                           const float* Xb2, const void* Wb, const float* bias_b, float* Yb, int Nb, int Kb, int B, void* stream);
/* cross-K/V projection: A half [B][rows_total][H*64] (the first `rows` rows of every utterance), W half [2*H*64][H*64], bias f32 ->
 * K / V caches [B][H][kv_cap][64], rows [seq_off, seq_off + rows), stored as half (out_half) or f32 */
int wt_dbg_gemm_f16_kv(const void* A, int rows_total, const void* W, const float* bias, void* kcache, void* vcache, int B, int rows, int H,
                       int kv_cap, int seq_off, int out_half, void* stream);

#ifdef __cplusplus
}
#endif
#endif
