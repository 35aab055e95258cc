// Host-only logic of the C-ABI (see host_logic.h).  Plain C++17, no HIP: also built with g++ -fsanitize=address,undefined
// by the CPU test-suite and driven with mutated blobs.
#include "host_logic.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <initializer_list>

namespace wt {

static int fail(char* err, size_t errlen, int code, const char* fmt, ...) {
    if (err && errlen) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, errlen, fmt, ap);
        va_end(ap);
    }
    return code;
}

int parse_blob(const void* blob, size_t nbytes, ParsedBlob* out, char* err, size_t errlen) {
    if (!blob || !out) return fail(err, errlen, WT_E_INVALID, "parse_blob: null argument");
    out->tensors.clear();
    if (nbytes < sizeof(BlobHeader)) return fail(err, errlen, WT_E_INVALID, "engine blob too small (%zu bytes)", nbytes);
    BlobHeader& hd = out->hd;
    memcpy(&hd, blob, sizeof hd);
    if (memcmp(hd.magic, "WTENGINE", 8) != 0) return fail(err, errlen, WT_E_INVALID, "engine blob has bad magic");
    if (hd.version != BLOB_VERSION)
        return fail(err, errlen, WT_E_UNSUPPORTED, "engine blob version %u not supported (this library reads version %u; rebuild the engine)",
                    hd.version, BLOB_VERSION);
    // bounds: never add two file-controlled 64-bit numbers; compare against what is left instead
    const uint64_t n64 = (uint64_t)nbytes;
    bool ok = hd.total_bytes == n64 && hd.table_off >= sizeof(BlobHeader) && hd.table_off <= n64 &&
              (uint64_t)hd.n_tensors <= (n64 - hd.table_off) / sizeof(BlobTensor);
    uint64_t table_end = 0;
    if (ok) {
        table_end = hd.table_off + (uint64_t)hd.n_tensors * sizeof(BlobTensor);  // <= n64 by the check above
        ok = hd.data_off >= table_end && hd.data_off <= n64 && (hd.data_off & 15) == 0;
    }
    if (!ok)
        return fail(err, errlen, WT_E_INVALID, "engine blob is truncated or corrupt (header says %llu bytes, got %zu)",
                    (unsigned long long)hd.total_bytes, nbytes);
    if (hd.kind != WT_KIND_ENCODER && hd.kind != WT_KIND_DECODER) return fail(err, errlen, WT_E_INVALID, "unknown engine kind %u", hd.kind);
    if (hd.precision != WT_F32 && hd.precision != WT_F16)
        return fail(err, errlen, WT_E_UNSUPPORTED, "engine precision %u: float32 or float16 (builder.py:55)", hd.precision);

    EngineDims& e = out->dims;
    e.kind = (int)hd.kind;
    e.precision = (int)hd.precision;
    e.d = hd.cfg[CFG_D_MODEL]; e.H = hd.cfg[CFG_HEADS]; e.L = hd.cfg[CFG_LAYERS]; e.F = hd.cfg[CFG_FFN];
    e.C = hd.cfg[CFG_MELS]; e.S = hd.cfg[CFG_SRC_POS]; e.T = hd.cfg[CFG_TGT_POS]; e.V = hd.cfg[CFG_VOCAB];
    e.tied = hd.cfg[CFG_TIED] != 0;
    auto bad = [&](const char* why) {
        return fail(err, errlen, WT_E_INVALID, "engine config invalid: %s (d=%d H=%d L=%d F=%d C=%d S=%d T=%d V=%d)", why, e.d, e.H,
                    e.L, e.F, e.C, e.S, e.T, e.V);
    };
    if (e.d <= 0 || e.H <= 0 || e.H > 16 || e.d != e.H * HEAD_DIM) return bad("head_dim must be 64, at most 16 heads");
    if (e.d > 1024 || (e.d & 3)) return bad("d_model must be a multiple of 4 and <= 1024");
    if (e.L <= 0 || e.L > 64 || e.F <= 0 || (e.F & 3) || e.F > 4096) return bad("ffn_dim must be a multiple of 4 and <= 4096, 1..64 layers");
    if (e.S <= 0 || e.S > 4096 || e.C <= 0 || e.C > 128 || (e.C & 3)) return bad("num_mel_bins must be a multiple of 4 and <= 128, max_source_positions <= 4096");
    if (e.kind == WT_KIND_DECODER && (e.T <= 1 || e.T > 4096 || e.V <= 0 || e.V > (1 << 20))) return bad("decoder needs max_target_positions in 2..4096 and a vocabulary");
    if (e.precision == WT_F16 && e.kind == WT_KIND_ENCODER && ((e.C & 7) || (e.d & 7) || (e.F & 7) || e.F < e.d))
        return bad("float16 encoder needs num_mel_bins, d_model, ffn_dim multiples of 8 and ffn_dim >= d_model (workspace layout)");
    if (e.precision == WT_F16 && e.kind == WT_KIND_DECODER && ((e.d & 7) || (e.F & 7)))
        return bad("float16 decoder needs d_model and ffn_dim to be multiples of 8 (eight halves per 16-byte weight load)");

    const char* base = (const char*)blob;
    out->tensors.reserve(hd.n_tensors);
    for (uint32_t i = 0; i < hd.n_tensors; ++i) {
        BlobTensor bt;
        memcpy(&bt, base + hd.table_off + (uint64_t)i * sizeof(BlobTensor), sizeof bt);
        bt.name[sizeof(bt.name) - 1] = 0;
        bool tok = bt.offset >= hd.data_off && bt.offset <= n64 && bt.nbytes <= n64 - bt.offset && (bt.offset & 15) == 0 &&
                   (bt.dtype == WT_F32 || bt.dtype == WT_F16) && bt.ndim <= 4;
        if (!tok) return fail(err, errlen, WT_E_INVALID, "tensor '%s' has a bad table entry", bt.name);
        // element count with an overflow guard: no dimension (and no partial product) may exceed the blob size
        uint64_t n = 1;
        for (uint32_t k = 0; k < bt.ndim && tok; ++k) {
            if (bt.shape[k] < 0 || (uint64_t)bt.shape[k] > n64) { tok = false; break; }
            if (bt.shape[k] != 0 && n > n64 / (uint64_t)bt.shape[k]) { tok = false; break; }
            n *= (uint64_t)bt.shape[k];
        }
        const uint64_t esz = bt.dtype == WT_F16 ? 2 : 4;
        if (!tok || n > n64 / esz || n * esz != bt.nbytes)
            return fail(err, errlen, WT_E_INVALID, "tensor '%s': shape and byte count disagree", bt.name);
        out->tensors.push_back(bt);
    }
    return WT_OK;
}

// --------------------------------------------------------------------------------------------- shape inference
static void set_desc(wt_tensor_desc* t, const char* name, int dtype, std::initializer_list<int64_t> shape) {
    memset(t, 0, sizeof *t);
    snprintf(t->name, sizeof t->name, "%s", name);
    t->dtype = dtype;
    t->ndim = (int)shape.size();
    int k = 0;
    for (int64_t s : shape) t->shape[k++] = s;
}

int infer_shapes(const EngineDims& e, const wt_tensor_desc* in, int n_in, wt_tensor_desc* out, int* n_out, ShapeState* st, char* err,
                 size_t errlen) {
    if ((!in && n_in) || n_in < 0 || !out || !n_out || !st) return fail(err, errlen, WT_E_INVALID, "wt_engine_infer_shapes: null argument");
    st->ok = false;
    auto find = [&](const char* name) -> const wt_tensor_desc* {
        for (int i = 0; i < n_in; ++i)
            if (strncmp(in[i].name, name, WT_NAME_LEN) == 0) return &in[i];
        return nullptr;
    };
    struct Spec { const char* name; int dtype; };
    static const Spec enc_in[] = {{"data", WT_F32}, {"length", WT_F32}};
    static const Spec dec_in[] = {{"data", WT_I32}, {"length", WT_I32}, {"encoder_hidden_states", WT_F32},
                                  {"self_past_key", WT_F32}, {"self_past_value", WT_F32}, {"cross_past_key", WT_F32},
                                  {"cross_past_value", WT_F32}, {"past_self_cache_mask", WT_F32}, {"past_cross_cache_mask", WT_F32}};
    const Spec* specs = e.kind == WT_KIND_ENCODER ? enc_in : dec_in;
    const int nspec = e.kind == WT_KIND_ENCODER ? 2 : 9;
    for (int i = 0; i < n_in; ++i) {  // session.py:128-136: unknown name / wrong dtype -> error
        const Spec* sp = nullptr;
        for (int k = 0; k < nspec; ++k)
            if (strncmp(in[i].name, specs[k].name, WT_NAME_LEN) == 0) sp = &specs[k];
        if (!sp) return fail(err, errlen, WT_E_NOTFOUND, "Tensor:%.*s is not an input tensor", WT_NAME_LEN, in[i].name);
        if (sp->dtype != in[i].dtype) return fail(err, errlen, WT_E_INVALID, "Tensor:%.*s has wrong dtype", WT_NAME_LEN, in[i].name);
        if (in[i].ndim < 0 || in[i].ndim > WT_MAX_DIMS) return fail(err, errlen, WT_E_INVALID, "Tensor:%.*s has a bad rank", WT_NAME_LEN, in[i].name);
    }
    auto shape_is = [](const wt_tensor_desc* t, std::initializer_list<int64_t> want) {
        if (!t || t->ndim != (int)want.size()) return false;
        int k = 0;
        for (int64_t s : want) {
            if (s >= 0 && t->shape[k] != s) return false;
            ++k;
        }
        return true;
    };
    if (e.kind == WT_KIND_ENCODER) {
        const wt_tensor_desc* data = find("data");
        if (!shape_is(data, {-1, e.C, 2 * e.S}) || data->shape[0] < 1 || data->shape[0] > (1 << 16))
            return fail(err, errlen, WT_E_INVALID, "encoder input 'data' must be f32 [B,%d,%d]", e.C, 2 * e.S);
        if (*n_out < 1) return fail(err, errlen, WT_E_INVALID, "output descriptor capacity too small");
        st->c_B = (int)data->shape[0];
        set_desc(&out[0], "hidden_states", WT_F32, {data->shape[0], e.S, e.d});
        *n_out = 1;
        st->ok = true;
        return WT_OK;
    }
    const int64_t L = e.L, H = e.H, S = e.S;
    if (!shape_is(find("data"), {1, 1}))
        return fail(err, errlen, WT_E_INVALID, "decoder input 'data' must be i32 [1,1] (batch_size and id_len are fixed to 1, model.py:474-477)");
    if (!shape_is(find("encoder_hidden_states"), {1, S, e.d}))
        return fail(err, errlen, WT_E_INVALID, "'encoder_hidden_states' must be f32 [1,%d,%d]", (int)S, e.d);
    const wt_tensor_desc *spk = find("self_past_key"), *spv = find("self_past_value");
    if (!shape_is(spk, {L, H, -1, HEAD_DIM}) || !shape_is(spv, {L, H, -1, HEAD_DIM}) || spk->shape[2] != spv->shape[2] ||
        spk->shape[2] < 1 || spk->shape[2] > e.T + 1)
        return fail(err, errlen, WT_E_INVALID, "'self_past_key/value' must be f32 [%d,%d,s,64] with 1 <= s <= %d", (int)L, (int)H, e.T + 1);
    if (!shape_is(find("cross_past_key"), {L, H, S, HEAD_DIM}) || !shape_is(find("cross_past_value"), {L, H, S, HEAD_DIM}))
        return fail(err, errlen, WT_E_INVALID, "'cross_past_key/value' must be f32 [%d,%d,%d,64]", (int)L, (int)H, (int)S);
    const wt_tensor_desc *ms = find("past_self_cache_mask"), *mc = find("past_cross_cache_mask");
    if (!ms || ms->ndim != 1 || ms->shape[0] < 1 || ms->shape[0] > e.T + 1)
        return fail(err, errlen, WT_E_INVALID, "'past_self_cache_mask' must be f32 [m_s], 1 <= m_s <= %d", e.T + 1);
    if (!mc || mc->ndim != 1 || mc->shape[0] < 1 || mc->shape[0] > S + 1)
        return fail(err, errlen, WT_E_INVALID, "'past_cross_cache_mask' must be f32 [m_c], 1 <= m_c <= %d", (int)S + 1);
    if (ms->shape[0] - 1 >= e.T) return fail(err, errlen, WT_E_INVALID, "position %d exceeds max_target_positions %d", (int)ms->shape[0] - 1, e.T);
    if (*n_out < 5) return fail(err, errlen, WT_E_INVALID, "output descriptor capacity too small");
    st->c_s = (int)spk->shape[2];
    st->c_ms = (int)ms->shape[0];
    st->c_mc = (int)mc->shape[0];
    const int64_t cache_len = st->c_ms - 1 < st->c_s ? st->c_ms - 1 : st->c_s;  // model.py:278
    set_desc(&out[0], "hidden_states", WT_F32, {1, 1, e.V});
    set_desc(&out[1], "next_self_keys", WT_F32, {L, H, cache_len + 1, HEAD_DIM});
    set_desc(&out[2], "next_self_values", WT_F32, {L, H, cache_len + 1, HEAD_DIM});
    set_desc(&out[3], "next_cross_keys", WT_F32, {L, H, S, HEAD_DIM});
    set_desc(&out[4], "next_cross_values", WT_F32, {L, H, S, HEAD_DIM});
    *n_out = 5;
    st->ok = true;
    return WT_OK;
}

}  // namespace wt
