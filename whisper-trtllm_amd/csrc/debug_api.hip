// Kernel-level test hooks (include/whisper_trtllm_amd_debug.h); thin wrappers over the launchers.
#include "../../include/whisper_trtllm_amd_debug.h"
#include "wt_common.h"

#include <string.h>

using namespace wt;

static int rc_of(hipError_t e) { return e == hipSuccess ? 0 : -5; }

extern "C" int wt_dbg_gemm(const float* A, int lda, const float* W, const float* bias, const float* resid, float* C, int M,
                           int N, int K, int act, void* stream) {
    GemmParams g;
    memset(&g, 0, sizeof g);
    g.A = A; g.lda = lda; g.a_rows_per_batch = M; g.W = W; g.bias = bias; g.resid = resid; g.C = C; g.ldc = N;
    g.c_rows_per_batch = M; g.M = M; g.N = N; g.K = K; g.act = act;
    return rc_of(launch_gemm_f32(g, (hipStream_t)stream));
}
extern "C" int wt_dbg_gemm_stamps(const float* A, int lda, const float* W, const float* bias, const float* resid, float* C, int M,
                                  int N, int K, int act, long long* stamps, void* stream) {
    GemmParams g;
    memset(&g, 0, sizeof g);
    g.A = A; g.lda = lda; g.a_rows_per_batch = M; g.W = W; g.bias = bias; g.resid = resid; g.C = C; g.ldc = N;
    g.c_rows_per_batch = M; g.M = M; g.N = N; g.K = K; g.act = act; g.dbg_stamps = stamps;
    return rc_of(launch_gemm_f32(g, (hipStream_t)stream));
}
// fp32 GEMM on the bf16 matrix cores (launch_gemm_x3): the hook splits A and W into the caller's scratch planes (3 * M * K and 3 * N * K
// bf16) and multiplies; flags bit 0: C receives three bf16 planes of [M][N] instead of fp32; bit 1: skip the splitting (planes in place)
extern "C" int wt_dbg_gemm_x3(const float* A, const float* W, const float* bias, const float* resid, void* C, int M, int N, int K, int act,
                              void* a_planes, void* w_planes, int flags, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int out_split = flags & 1;
    int rc = 0;
    if (!(flags & 2)) {   // bit 1: the planes are in place already (timing of the product alone)
        if ((rc = rc_of(launch_split3(A, a_planes, (size_t)M * K, (size_t)M * K, s)))) return rc;
        if ((rc = rc_of(launch_split3(W, w_planes, (size_t)N * K, (size_t)N * K, s)))) return rc;
    }
    GemmParams g;
    memset(&g, 0, sizeof g);
    g.A = (const float*)a_planes; g.lda = K; g.a_rows_per_batch = M; g.a_plane = (long long)M * K; g.W = (const float*)w_planes; g.w_plane = (long long)N * K;
    g.bias = bias; g.resid = resid; g.C = (float*)C; g.ldc = N; g.c_rows_per_batch = M; g.M = M; g.N = N; g.K = K; g.act = act;
    g.out_split = out_split; g.c_plane = (long long)M * N;
    return rc_of(launch_gemm_x3(g, s));
}
extern "C" int wt_dbg_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int d, void* stream) {
    return rc_of(launch_layernorm(x, w, b, y, rows, d, (hipStream_t)stream));
}
extern "C" int wt_dbg_encoder_attention(const float* qkv, float* ctx, int B, int S, int H, void* stream) {
    return rc_of(launch_encoder_attention(qkv, ctx, B, S, H, (hipStream_t)stream));
}
extern "C" int wt_dbg_encoder_attention_occupancy(void) { return encoder_attention_blocks_per_cu(); }
extern "C" int wt_dbg_encoder_attention_split(const float* qkv, void* ctx_planes, int B, int S, int H, void* stream) {
    return rc_of(launch_encoder_attention(qkv, nullptr, B, S, H, (hipStream_t)stream, ctx_planes, (size_t)B * S * H * 64));
}
// x3 attention: the hook splits fp32 q|k|v [B*S][3d] into the caller's scratch planes (3 * B*S * 3d bf16) and attends; context out as planes
extern "C" int wt_dbg_encoder_attention_x3(const float* qkv, void* qkv_planes, void* ctx_planes, int B, int S, int H, int skip_split, void* stream) {
    const size_t n = (size_t)B * S * 3 * H * 64;
    int rc = skip_split ? 0 : rc_of(launch_split3(qkv, qkv_planes, n, n, (hipStream_t)stream));
    if (rc) return rc;
    return rc_of(launch_encoder_attention_x3(qkv_planes, n, ctx_planes, (size_t)B * S * H * 64, B, S, H, (hipStream_t)stream));
}
extern "C" int wt_dbg_skinny(const float* X, const float* ln_w, const float* ln_b, const float* W, const float* bias,
                             const float* resid, float* Y, int B, int N, int K, int xmode, int act, float scale,
                             void* stream) {
    SkinnyParams k;
    memset(&k, 0, sizeof k);
    k.X = X; k.ln_w = ln_w; k.ln_b = ln_b; k.W = W; k.bias = bias; k.resid = resid; k.Y = Y; k.B = B; k.N = N; k.K = K;
    k.xmode = xmode & 1; k.x_direct = (xmode >> 1) & 1; k.w_nt = (xmode >> 2) & 1; k.act = act; k.q_scale = scale; k.ymode = YMODE_PLAIN;
    return rc_of(launch_skinny(k, (hipStream_t)stream));
}
// probe (tools/microbench.py fold_fc1): fc2 with the GELU / LayerNorm finish of a folded fc1 in its prologue; timing only
extern "C" int wt_dbg_skinny_gelu_in(const float* X, const float* r, const float* t, const float* W, const float* bias, const float* resid,
                                     float* Y, int B, int N, int K, void* stream) {
    SkinnyParams k;
    memset(&k, 0, sizeof k);
    k.X = X; k.ln_w = r; k.ln_b = t; k.W = W; k.bias = bias; k.resid = resid; k.Y = Y; k.B = B; k.N = N; k.K = K; k.w_nt = 1; k.q_scale = 1.f;
    return rc_of(launch_skinny_gelu_in_probe(k, (hipStream_t)stream));
}
extern "C" int wt_dbg_decode_attention(const float* q, const float* kcache, const float* vcache, float* part, int* cnt,
                                       float* out, int B, int H, int s_cap, int len, int n_split, void* stream) {
    if (len < 1 || len > s_cap || n_split < 1 || n_split > 16) return -22;
    DecAttnParams a;
    memset(&a, 0, sizeof a);
    a.q = q; a.kcache = kcache; a.vcache = vcache; a.part = part; a.cnt = cnt; a.out = out; a.B = B; a.H = H; a.s_cap = s_cap; a.n_split = n_split;
    a.fixed_len = len; a.nt = 1;
    return rc_of(launch_dec_attn(a, (hipStream_t)stream));
}
extern "C" int wt_dbg_decode_attention_folded(const float* u, const float* kcache, const float* vcache, float* part, int* cnt,
                                              float* out, const float* ln_h, const float* ln_r, const float* ln_t, int B, int H,
                                              int s_cap, int len, int n_split, void* stream) {
    if (len < 1 || len > s_cap || n_split < 1 || n_split > 16 || !ln_h || !ln_r || !ln_t || H * 64 > 1024) return -22;
    DecAttnParams a;
    memset(&a, 0, sizeof a);
    a.q = u; a.kcache = kcache; a.vcache = vcache; a.part = part; a.cnt = cnt; a.out = out; a.B = B; a.H = H; a.s_cap = s_cap; a.n_split = n_split;
    a.fixed_len = len; a.nt = 1; a.ln_h = ln_h; a.ln_r = ln_r; a.ln_t = ln_t;
    return rc_of(launch_dec_attn(a, (hipStream_t)stream));
}
extern "C" int wt_dbg_attention_then_projection(const float* q, const float* kcache, const float* vcache, float* part, const float* W,
                                               const float* bias, const float* resid, float* Y, int B, int H, int s_cap, int len,
                                               int n_split, void* stream) {
    if (len < 1 || len > s_cap || n_split != 2 || H * 64 > 1024) return -22;
    DecAttnParams a;
    memset(&a, 0, sizeof a);
    a.q = q; a.kcache = kcache; a.vcache = vcache; a.part = part; a.B = B; a.H = H; a.s_cap = s_cap; a.n_split = n_split;
    a.fixed_len = len; a.nt = 1; a.defer_merge = 1;
    int rc = rc_of(launch_dec_attn(a, (hipStream_t)stream));
    if (rc) return rc;
    SkinnyParams k;
    memset(&k, 0, sizeof k);
    k.parts = part; k.parts_nsplit = n_split; k.parts_H = H; k.W = W; k.bias = bias; k.resid = resid; k.Y = Y; k.B = B; k.N = H * 64;
    k.K = H * 64; k.q_scale = 1.f; k.w_nt = 1;
    return rc_of(launch_skinny(k, (hipStream_t)stream));
}
extern "C" int wt_dbg_skinny_pair(const float* Xa, const float* Wa, const float* bias_a, const float* resid_a, float* Ya, int Na, int Ka,
                                  const float* Xb, const float* Xb2, const float* Wb, const float* bias_b, float* Yb, int Nb, int Kb,
                                  int B, void* stream) {
    SkinnyParams a, b;
    memset(&a, 0, sizeof a);
    memset(&b, 0, sizeof b);
    a.X = Xa; a.W = Wa; a.bias = bias_a; a.resid = resid_a; a.Y = Ya; a.B = B; a.N = Na; a.K = Ka; a.q_scale = 1.f; a.w_nt = 1;
    b.X = Xb; b.X2 = Xb2; b.x_direct = 1; b.W = Wb; b.bias = bias_b; b.Y = Yb; b.B = B; b.N = Nb; b.K = Kb; b.q_scale = 1.f; b.w_nt = 1;
    return rc_of(launch_skinny_pair(a, b, (hipStream_t)stream));
}

extern "C" int wt_dbg_self_attention_then_pair(const float* q, const float* kcache, const float* vcache, float* part, const float* Wo,
                                              const float* bo, const float* h, float* h1, const float* Wf, const float* c, float* u, int B, int H,
                                              int s_cap, int len, void* stream) {
    if (len < 1 || len > s_cap || H * 64 > 1024) return -22;
    const int d = H * 64;
    DecAttnParams a;
    memset(&a, 0, sizeof a);
    a.q = q; a.kcache = kcache; a.vcache = vcache; a.part = part; a.B = B; a.H = H; a.s_cap = s_cap; a.n_split = 2;
    a.fixed_len = len; a.nt = 1; a.defer_merge = 1;
    int rc = rc_of(launch_dec_attn(a, (hipStream_t)stream));
    if (rc) return rc;
    SkinnyParams k, k2;
    memset(&k, 0, sizeof k);
    memset(&k2, 0, sizeof k2);
    k.parts = part; k.parts_nsplit = 2; k.parts_H = H; k.W = Wo; k.bias = bo; k.resid = h; k.Y = h1; k.B = B; k.N = d; k.K = d; k.q_scale = 1.f; k.w_nt = 1;
    k2.parts = part; k2.parts_nsplit = 2; k2.parts_H = H; k2.X2 = h; k2.x_direct = 1; k2.W = Wf; k2.bias = c; k2.Y = u; k2.B = B; k2.N = d; k2.K = 2 * d;
    k2.q_scale = 1.f; k2.w_nt = 1;
    return rc_of(launch_skinny_pair(k, k2, (hipStream_t)stream));
}

extern "C" int wt_dbg_gemm_f16(const void* A, int lda, const void* W, const float* bias, const float* resid, void* C, int M, int N,
                               int K, int act, int out_half, void* stream) {
    GemmParams g;
    memset(&g, 0, sizeof g);
    g.A = (const float*)A; g.lda = lda; g.a_rows_per_batch = M; g.W = (const float*)W; g.bias = bias; g.resid = resid;
    g.C = (float*)C; g.ldc = N; g.c_rows_per_batch = M; g.M = M; g.N = N; g.K = K; g.act = act;
    return rc_of(launch_gemm_f16(g, out_half != 0, (hipStream_t)stream));
}

extern "C" int wt_dbg_gemm_f16_variant(const void* A, int lda, const void* W, const float* bias, const float* resid, void* C, int M, int N,
                                       int K, int act, int out_half, int variant, void* stream) {
    GemmParams g;
    memset(&g, 0, sizeof g);
    g.A = (const float*)A; g.lda = lda; g.a_rows_per_batch = M; g.W = (const float*)W; g.bias = bias; g.resid = resid;
    g.C = (float*)C; g.ldc = N; g.c_rows_per_batch = M; g.M = M; g.N = N; g.K = K; g.act = act;
    return rc_of(launch_gemm_f16(g, out_half != 0, (hipStream_t)stream, variant));
}

extern "C" int wt_dbg_encoder_attention_f16(const void* qkv, void* ctx, int B, int S, int H, void* stream) {
    return rc_of(launch_encoder_attention_f16(qkv, ctx, B, S, H, (hipStream_t)stream));
}


// ---- fp16 decoder engines: half weights (skinny GEMVs) and half K/V caches (decode attention); everything else fp32 ----
extern "C" int wt_dbg_skinny_f16(const float* X, const float* ln_w, const float* ln_b, const void* W, const float* bias,
                                 const float* resid, float* Y, int B, int N, int K, int xmode, int act, float scale, void* stream) {
    SkinnyParams k;
    memset(&k, 0, sizeof k);
    k.X = X; k.ln_w = ln_w; k.ln_b = ln_b; k.W = (const float*)W; k.w_half = 1; k.bias = bias; k.resid = resid; k.Y = Y; k.B = B; k.N = N; k.K = K;
    k.xmode = xmode & 1; k.x_direct = (xmode >> 1) & 1; k.w_nt = (xmode >> 2) & 1; k.act = act; k.q_scale = scale; k.ymode = YMODE_PLAIN;
    return rc_of(launch_skinny(k, (hipStream_t)stream));
}
extern "C" int wt_dbg_decode_attention_f16(const float* q, const void* kcache, const void* vcache, float* part, int* cnt, float* out,
                                           const float* ln_h, const float* ln_r, const float* ln_t, int B, int H, int s_cap, int len,
                                           int n_split, void* stream) {
    if (len < 1 || len > s_cap || n_split < 1 || n_split > 16 || H * 64 > 1024) return -22;
    if ((ln_h || ln_r || ln_t) && !(ln_h && ln_r && ln_t)) return -22;
    DecAttnParams a;
    memset(&a, 0, sizeof a);
    a.q = q; a.kcache = (const float*)kcache; a.vcache = (const float*)vcache; a.kv_half = 1; a.part = part; a.cnt = cnt; a.out = out;
    a.B = B; a.H = H; a.s_cap = s_cap; a.n_split = n_split; a.fixed_len = len; a.nt = 1; a.ln_h = ln_h; a.ln_r = ln_r; a.ln_t = ln_t;
    return rc_of(launch_dec_attn(a, (hipStream_t)stream));
}
extern "C" int wt_dbg_attention_then_projection_f16(const float* q, const void* kcache, const void* vcache, float* part, const void* W,
                                                   const float* bias, const float* resid, float* Y, int B, int H, int s_cap, int len,
                                                   void* stream) {
    if (len < 1 || len > s_cap || H * 64 > 1024) return -22;
    DecAttnParams a;
    memset(&a, 0, sizeof a);
    a.q = q; a.kcache = (const float*)kcache; a.vcache = (const float*)vcache; a.kv_half = 1; a.part = part; a.B = B; a.H = H; a.s_cap = s_cap;
    a.n_split = 2; a.fixed_len = len; a.nt = 1; a.defer_merge = 1;
    int rc = rc_of(launch_dec_attn(a, (hipStream_t)stream));
    if (rc) return rc;
    SkinnyParams k;
    memset(&k, 0, sizeof k);
    k.parts = part; k.parts_nsplit = 2; k.parts_H = H; k.W = (const float*)W; k.w_half = 1; k.bias = bias; k.resid = resid; k.Y = Y; k.B = B;
    k.N = H * 64; k.K = H * 64; k.q_scale = 1.f; k.w_nt = 1;
    return rc_of(launch_skinny(k, (hipStream_t)stream));
}
extern "C" int wt_dbg_skinny_pair_f16(const float* Xa, const void* Wa, const float* bias_a, const float* resid_a, float* Ya, int Na, int Ka,
                                      const float* Xb, const float* Xb2, const void* Wb, const float* bias_b, float* Yb, int Nb, int Kb,
                                      int B, void* stream) {
    SkinnyParams a, b;
    memset(&a, 0, sizeof a);
    memset(&b, 0, sizeof b);
    a.X = Xa; a.W = (const float*)Wa; a.w_half = 1; a.bias = bias_a; a.resid = resid_a; a.Y = Ya; a.B = B; a.N = Na; a.K = Ka; a.q_scale = 1.f; a.w_nt = 1;
    b.X = Xb; b.X2 = Xb2; b.x_direct = 1; b.W = (const float*)Wb; b.w_half = 1; b.bias = bias_b; b.Y = Yb; b.B = B; b.N = Nb; b.K = Kb; b.q_scale = 1.f; b.w_nt = 1;
    return rc_of(launch_skinny_pair(a, b, (hipStream_t)stream));
}
// cross-K/V projection of an fp16 decoder engine: A half [B][rows_total][d] (first `rows` rows of every utterance are projected),
// W half [2d][d], bias f32 [2d] -> K / V caches [B][H][kv_cap][64] rows [seq_off, seq_off + rows), half (out_half) or f32
extern "C" int wt_dbg_gemm_f16_kv(const void* A, int rows_total, const void* W, const float* bias, void* kcache, void* vcache, int B, int rows,
                                  int H, int kv_cap, int seq_off, int out_half, void* stream) {
    const int d = H * 64;
    if (rows < 1 || rows > rows_total || seq_off < 0 || seq_off + rows > kv_cap) return -22;
    GemmParams g;
    memset(&g, 0, sizeof g);
    g.A = (const float*)A; g.lda = d; g.a_rows_per_batch = rows; g.a_batch_stride = (long long)rows_total * d;
    g.W = (const float*)W; g.bias = bias; g.M = B * rows; g.N = 2 * d; g.K = d;
    g.epi = EPI_KV_HEADS; g.C = (float*)kcache; g.C2 = (float*)vcache; g.c_rows_per_batch = rows; g.kv_heads = H; g.kv_cap = kv_cap; g.kv_seq_off = seq_off;
    return rc_of(launch_gemm_f16(g, out_half != 0, (hipStream_t)stream));
}
