// C-ABI of the MI355X Whisper engine (include/whisper_trtllm_amd.h): blob parsing, weight upload, workspace,
// encoder forward, the Session-compatible by-value decoder step, and the resident-KV greedy fast path with
// hipGraph replay.  Reference call sites replaced: tensorrt_llm/runtime/session.py:54-178 (Session) and
// examples/whisper/run.py:57-227 (engine wrappers + greedy_search).
#include "../../include/whisper_trtllm_amd.h"
#include "wt_common.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using namespace wt;

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
int wt_set_error(int code, const char* fmt, ...) {  // shared with frontend.hip
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess) return fail(WT_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                          __FILE__, __LINE__);                                            \
    } while (0)

struct DevTensor {
    float* ptr = nullptr;
    int dtype = WT_F32;
    int ndim = 0;
    int64_t shape[4] = {0, 0, 0, 0};
};

struct EncLayerW {
    const float *ln1_w, *ln1_b, *qkv_w, *qkv_b, *o_w, *o_b, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    const void *qkv_w3 = nullptr, *o_w3 = nullptr, *fc1_w3 = nullptr, *fc2_w3 = nullptr;   // the same matrices as three bf16 planes (launch_gemm_x3)
};
struct DecLayerW {
    const float *ln1_w, *ln1_b, *qkv_w, *qkv_b, *o_w, *o_b;
    const float *fold_w, *fold_c, *fold_r, *fold_t;  // folded LN + cross-attention query (builder.py:_pack_decoder)
    const float *ckv_w, *ckv_b, *co_w, *co_b;
    const float *ln3_w, *ln3_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    const void* ckv_w3 = nullptr;   // cross k|v weights as three bf16 planes (launch_gemm_x3)
};

struct EvTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    double ms = 0.0;
    long long launches = 0;
};

// The weight payload of an engine in device memory: ONE allocation, read-only after wt_engine_open, shared (reference-counted) by the
// handles wt_engine_clone makes from it -- N workers per GPU (runtime.WhisperPipeline) hold one copy of the weights and N workspaces.
struct WeightStore {
    char* base = nullptr;
    char* x3 = nullptr;    // fp32 engines: the big GEMMs' weights once more as three bf16 planes each (launch_gemm_x3), shared like `base`
    int device = 0;
    ~WeightStore() {
        if (!base && !x3) return;
        DeviceGuard guard(device);
        if (base) hipFree(base);
        if (x3) hipFree(x3);
    }
};

struct wt_engine {
    const void *conv1_w3 = nullptr, *conv2_w3 = nullptr;   // conv weights as three bf16 planes (launch_gemm_x3)
    void *melT3 = nullptr, *c1_3 = nullptr;                 // x3: the convs' A operands as three bf16 planes (time-major, zero pad rows)
    size_t melT3_plane = 0, c1_3_plane = 0;                 // their plane strides: fixed by the workspace CAPACITY, so that a smaller batch finds
                                                            // utterance b's rows -- and the zero padding rows nobody ever writes -- where a larger one left them
    int kind = 0, device = 0, precision = WT_F32;
    int d = 0, H = 0, L = 0, F = 0, C = 0, S = 0, T = 0, V = 0;
    std::shared_ptr<WeightStore> weights;
    char* weights_base = nullptr;   // == weights->base
    std::map<std::string, DevTensor> w;
    // encoder
    std::vector<EncLayerW> enc_layers;
    const float *conv1_w = nullptr, *conv1_b = nullptr, *conv2_w = nullptr, *conv2_b = nullptr, *enc_pos = nullptr,
                *enc_ln_w = nullptr, *enc_ln_b = nullptr;
    int enc_cap = 0;
    char* enc_ws = nullptr;
    float *melT = nullptr, *c1 = nullptr, *hbuf = nullptr, *xbuf = nullptr, *qkv = nullptr, *ctx = nullptr, *ffn = nullptr;
    bool use_x3 = false;          // fp32 engine: the layer GEMMs / the cross-K/V projection run launch_gemm_x3 (fp32 product from bf16 MFMAs)
    void *xs3 = nullptr, *ctx3 = nullptr, *ffn3 = nullptr, *qkv3 = nullptr;   // x3 activations: three bf16 planes of [M][d] / [M][d] / [M][F] / [M][3d]
    void* enc3 = nullptr;         // decoder: three bf16 planes of the encoder memory in front of the cross-K/V projection
    void *melT_h = nullptr, *c1_h = nullptr, *x_h = nullptr, *ctx_h = nullptr, *ffn_h = nullptr;  // fp16 engines
    // decoder
    std::vector<DecLayerW> dec_layers;
    const float *tok_emb = nullptr, *pos_emb = nullptr, *proj_w = nullptr, *dec_ln_w = nullptr, *dec_ln_b = nullptr;
    int dec_cap = 0, dec_maxlen_cap = 0;
    char* dec_ws = nullptr;
    float *self_k = nullptr, *self_v = nullptr, *cross_k = nullptr, *cross_v = nullptr;  // resident caches: fp32, or IEEE half in fp16 engines
    bool w_half = false;   // fp16 decoder engine: every GEMV weight matrix (incl. the tied vocabulary / token table) is IEEE half
    int kv_esz = 4;        // bytes per element of the RESIDENT caches (2 in fp16 engines; Session-path caches are the caller's f32)
    void* enc_h = nullptr; // fp16 engines: half copy of the encoder memory [B][S][d], the A operand of the cross-K/V projection
    float *dh = nullptr, *dh2 = nullptr, *dq = nullptr, *datt = nullptr, *dffn = nullptr, *part = nullptr, *logits = nullptr;
    int* att_cnt = nullptr;
    float* sel_val = nullptr;
    int* sel_idx = nullptr;
    DecState* st = nullptr;
    int *ids = nullptr, *unfinished = nullptr, *forced = nullptr;
    uint8_t* mask = nullptr;
    std::vector<uint8_t> h_mask;   // host images of the device rule tables (wt_decoder_begin uploads only on change)
    std::vector<int> h_forced;
    bool tables_valid = false;
    DecState* h_state = nullptr;  // pinned
    unsigned long long* mailbox = nullptr;      // pinned, coherent host word greedy_finish_kernel reports every step into (wt_decoder_run)
    unsigned long long* mailbox_dev = nullptr;  // its device address
    int epoch = 0;                // wt_decoder_begin count (tags the mailbox word: late no-op steps of the previous decode are ignored)
    int issued = 0;               // decoder steps enqueued since wt_decoder_begin
    // continuous mode (wt_decoder_stream_*): the cross caches are a POOL of rows, `B` decode slots take utterances from a device queue
    int cross_rows = 0;           // rows per layer of the resident cross caches (>= dec_cap; the pool of the continuous mode)
    bool stream_mode = false;
    int pool_rows = 0;            // pool rows the current stream may use (<= cross_rows)
    int* h_ids = nullptr;         // pinned, coherent [h_pool_cap][h_ml_cap]: ids of the utterance in pool row r (written by stream_finish_kernel)
    int* h_len = nullptr;         // pinned, coherent [h_pool_cap]: its final length, 0 = still waiting / decoding
    int *h_ids_dev = nullptr, *h_len_dev = nullptr;
    int h_pool_cap = 0, h_ml_cap = 0;
    std::vector<char> pool_state; // host allocator of pool rows: 0 free, 1 submitted (waiting or decoding), 2 finished but not collected
    int submitted = 0, finished_seen = 0, admitted_seen = 0;
    int* force_rows = nullptr;    // device [16]: per-row forced-EOS steps (bench-only variable-length workload)
    std::vector<int> h_force_rows;
    bool force_rows_on = false;
    // greedy session
    bool begun = false;
    int B = 0, max_length = 0, begin_index = 0, eos = 0, pad = 0, force_eos_step = -1, nsplit_self = 1, nsplit_cross = 4, start_token = 0;
    float* trace = nullptr;
    hipStream_t own_stream = nullptr;
    // step graphs: [0] every row attends (the only one a decode without early finishers ever replays), [1] finished rows stream no K/V,
    // [2] / [3] the same two for the continuous mode (per-slot cross-cache rows, stream_finish_kernel)
    static constexpr int N_GRAPHS = 4;
    hipGraph_t graph[N_GRAPHS] = {nullptr, nullptr, nullptr, nullptr};
    hipGraphExec_t graph_exec[N_GRAPHS] = {nullptr, nullptr, nullptr, nullptr};
    bool graph_ok[N_GRAPHS] = {false, false, false, false};
    bool graph_valid = false, use_graph = true;
    int nt_loads = 1;  // stream weights and K/V with non-temporal loads (set per decode in wt_decoder_begin)
    // Session-compat shapes (set by infer_shapes)
    bool shapes_ok = false;
    int c_B = 1, c_s = 0, c_ms = 0, c_mc = 0;
    // profiling
    bool profiling = false;
    EvTimer t_cross, t_gemm, t_enc_attn, t_skinny;
};

// ------------------------------------------------------------------------------------------------- helpers
static void timer_begin(wt_engine* e, EvTimer& t, hipStream_t s, hipEvent_t* a, hipEvent_t* b) {
    *a = *b = nullptr;
    if (!e->profiling) return;
    if (t.pool.empty()) {
        hipEvent_t x, y;
        if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) return;
        t.pool.push_back({x, y});
    }
    auto pr = t.pool.back();
    t.pool.pop_back();
    *a = pr.first;
    *b = pr.second;
    hipEventRecord(*a, s);
}
static void timer_end(wt_engine* e, EvTimer& t, hipStream_t s, hipEvent_t a, hipEvent_t b) {
    if (!a) return;
    hipEventRecord(b, s);
    t.pending.push_back({a, b});
}
static void timer_collect(EvTimer& t) {
    for (auto& pr : t.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            t.ms += ms;
            t.launches += 1;
        }
        t.pool.push_back(pr);
    }
    t.pending.clear();
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
constexpr int SEL_PARTS_CAP = 4096;  // partial (max, argmax) slots per batch row: >= workgroups of the vocabulary GEMV

// ------------------------------------------------------------------------------------------------- open / close
extern "C" int wt_abi_version(void) { return WT_ABI_VERSION; }
extern "C" const char* wt_last_error(void) { return g_err; }

extern "C" void wt_engine_close(wt_engine* e) {
    if (!e) return;
    DeviceGuard guard(e->device);
    for (int v = 0; v < wt_engine::N_GRAPHS; ++v) {
        if (e->graph_exec[v]) hipGraphExecDestroy(e->graph_exec[v]);
        if (e->graph[v]) hipGraphDestroy(e->graph[v]);
    }
    for (EvTimer* t : {&e->t_cross, &e->t_gemm, &e->t_enc_attn, &e->t_skinny}) {
        timer_collect(*t);
        for (auto& pr : t->pool) {
            hipEventDestroy(pr.first);
            hipEventDestroy(pr.second);
        }
    }
    if (e->own_stream) hipStreamDestroy(e->own_stream);
    if (e->h_state) hipHostFree(e->h_state);
    if (e->mailbox) hipHostFree((void*)e->mailbox);
    if (e->h_ids) hipHostFree(e->h_ids);
    if (e->h_len) hipHostFree(e->h_len);
    if (e->enc_ws) hipFree(e->enc_ws);
    if (e->dec_ws) hipFree(e->dec_ws);
    e->weights.reset();   // the last handle sharing the payload frees it
    delete e;
}

// Split the weight matrices of the big GEMMs into three bf16 planes each (one extra allocation, shared by clones like the payload).
static int build_x3_weights(wt_engine* e) {
    const size_t d = e->d, F = e->F;
    struct Job { const float* src; size_t n; const void** dst; };
    std::vector<Job> jobs;
    if (e->kind == WT_KIND_ENCODER && (e->C % 8) == 0 && ((3 * e->C) % 16) == 0) {
        jobs.push_back({e->conv1_w, d * 3 * (size_t)e->C, &e->conv1_w3});
        jobs.push_back({e->conv2_w, d * 3 * d, &e->conv2_w3});
    }
    if (e->kind == WT_KIND_ENCODER)
        for (EncLayerW& l : e->enc_layers) {
            jobs.push_back({l.qkv_w, 3 * d * d, &l.qkv_w3}); jobs.push_back({l.o_w, d * d, &l.o_w3});
            jobs.push_back({l.fc1_w, F * d, &l.fc1_w3}); jobs.push_back({l.fc2_w, d * F, &l.fc2_w3});
        }
    else
        for (DecLayerW& l : e->dec_layers) jobs.push_back({l.ckv_w, 2 * d * d, &l.ckv_w3});
    size_t total = 0;
    for (const Job& j : jobs) total += align_up(j.n * 6, 256);
    if (!total) return WT_OK;
    hipError_t he = hipMalloc((void**)&e->weights->x3, total);
    if (he != hipSuccess) return fail(WT_E_NOMEM, "hipMalloc(%zu) for the split weights failed: %s", total, hipGetErrorString(he));
    hipStream_t up = nullptr;
    he = hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
    size_t off = 0;
    for (const Job& j : jobs) {
        if (he != hipSuccess) break;
        void* dst = e->weights->x3 + off;
        he = launch_split3(j.src, dst, j.n, j.n, up);
        *j.dst = dst;
        off += align_up(j.n * 6, 256);
    }
    if (he == hipSuccess) he = hipStreamSynchronize(up);
    if (up) hipStreamDestroy(up);
    if (he != hipSuccess) return fail(WT_E_HIP, "splitting the weights failed: %s", hipGetErrorString(he));
    e->use_x3 = true;
    return WT_OK;
}

extern "C" int wt_engine_open(const void* blob, size_t nbytes, int device, wt_engine** out) {
    if (!blob || !out) return fail(WT_E_INVALID, "wt_engine_open: null argument");
    *out = nullptr;
    ParsedBlob pb;
    {
        char msg[400] = "";
        const int prc = parse_blob(blob, nbytes, &pb, msg, sizeof msg);  // host_logic.cpp: overflow-safe bounds, config limits
        if (prc) return fail(prc, "%s", msg);
    }
    const BlobHeader& hd = pb.hd;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(WT_E_INVALID, "device %d out of range (%d visible)", device, ndev);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail(WT_E_HIP, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(guard.err));

    wt_engine* e = new wt_engine();
    e->kind = pb.dims.kind;
    e->precision = pb.dims.precision;
    e->device = device;
    e->d = pb.dims.d; e->H = pb.dims.H; e->L = pb.dims.L; e->F = pb.dims.F;
    e->C = pb.dims.C; e->S = pb.dims.S; e->T = pb.dims.T; e->V = pb.dims.V;

    // upload the tensor payload in one allocation
    const size_t payload = nbytes - hd.data_off;
    e->weights = std::make_shared<WeightStore>();
    e->weights->device = device;
    hipError_t he = hipMalloc((void**)&e->weights->base, payload ? payload : 256);
    e->weights_base = e->weights->base;
    if (he != hipSuccess) {
        int rc = fail(WT_E_NOMEM, "hipMalloc(%zu) for weights failed: %s", payload, hipGetErrorString(he));
        wt_engine_close(e);
        return rc;
    }
    {   // upload on a private non-blocking stream, not the legacy stream: a synchronous hipMemcpy is refused by the runtime while another
        // host thread (another handle's worker) is capturing its step graph
        hipStream_t up = nullptr;
        he = hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
        if (he == hipSuccess) he = hipMemcpyAsync(e->weights_base, (const char*)blob + hd.data_off, payload, hipMemcpyHostToDevice, up);
        if (he == hipSuccess) he = hipStreamSynchronize(up);
        if (up) hipStreamDestroy(up);
    }
    if (he != hipSuccess) {
        int rc = fail(WT_E_HIP, "weight upload failed: %s", hipGetErrorString(he));
        wt_engine_close(e);
        return rc;
    }
    for (const BlobTensor& bt : pb.tensors) {  // entries were validated by parse_blob
        DevTensor t;
        t.ptr = (float*)(e->weights_base + (bt.offset - hd.data_off));
        t.ndim = (int)bt.ndim;
        t.dtype = (int)bt.dtype;
        for (uint32_t k = 0; k < bt.ndim; ++k) t.shape[k] = bt.shape[k];
        e->w[bt.name] = t;
    }
    // resolve the parameter tree; every lookup failure is reported by name
    g_err[0] = 0;
    auto need = [&](const std::string& name, std::initializer_list<int64_t> shape) -> const float* {
        auto it = e->w.find(name);
        if (it == e->w.end()) { if (!g_err[0]) fail(WT_E_NOTFOUND, "engine blob has no tensor '%s'", name.c_str()); return nullptr; }
        int k = 0;
        bool ok = it->second.ndim == (int)shape.size();
        for (int64_t s : shape) { ok = ok && it->second.shape[k] == s; ++k; }
        if (!ok && !g_err[0]) fail(WT_E_INVALID, "tensor '%s' has the wrong shape for this config", name.c_str());
        // fp16 engines carry their GEMM / GEMV operands (2-D *.weight, incl. the token table tied to the vocabulary projection) as
        // fp16, everything else (biases, LayerNorm parameters, position tables, the folded query's vectors) as fp32
        const bool is_gemm_w = shape.size() == 2 && name.size() > 7 && name.compare(name.size() - 7, 7, ".weight") == 0 &&
                               name != "embed_positions.weight";
        const int want = (e->precision == WT_F16 && is_gemm_w) ? WT_F16 : WT_F32;
        if (ok && it->second.dtype != want) {
            ok = false;
            if (!g_err[0]) fail(WT_E_INVALID, "tensor '%s' has the wrong element type for this engine precision", name.c_str());
        }
        return ok ? it->second.ptr : nullptr;
    };
    const int64_t d = e->d, F = e->F;
    if (e->kind == WT_KIND_ENCODER) {
        e->conv1_w = need("conv1.weight", {d, 3 * e->C});
        e->conv1_b = need("conv1.bias", {d});
        e->conv2_w = need("conv2.weight", {d, 3 * d});
        e->conv2_b = need("conv2.bias", {d});
        e->enc_pos = need("embed_positions", {e->S, d});
        e->enc_ln_w = need("layer_norm.weight", {d});
        e->enc_ln_b = need("layer_norm.bias", {d});
        for (int i = 0; i < e->L; ++i) {
            std::string p = "layers." + std::to_string(i) + ".";
            EncLayerW l;
            l.ln1_w = need(p + "self_attn_layer_norm.weight", {d}); l.ln1_b = need(p + "self_attn_layer_norm.bias", {d});
            l.qkv_w = need(p + "self_attn.qkv.weight", {3 * d, d}); l.qkv_b = need(p + "self_attn.qkv.bias", {3 * d});
            l.o_w = need(p + "self_attn.dense.weight", {d, d}); l.o_b = need(p + "self_attn.dense.bias", {d});
            l.ln2_w = need(p + "final_layer_norm.weight", {d}); l.ln2_b = need(p + "final_layer_norm.bias", {d});
            l.fc1_w = need(p + "fc1.weight", {F, d}); l.fc1_b = need(p + "fc1.bias", {F});
            l.fc2_w = need(p + "fc2.weight", {d, F}); l.fc2_b = need(p + "fc2.bias", {d});
            e->enc_layers.push_back(l);
        }
    } else {
        e->w_half = e->precision == WT_F16;
        e->kv_esz = e->w_half ? 2 : 4;
        e->tok_emb = need("embed_tokens.weight", {e->V, d});
        e->pos_emb = need("embed_positions.weight", {e->T, d});
        e->proj_w = hd.cfg[CFG_TIED] ? e->tok_emb : need("proj_out.weight", {e->V, d});
        e->dec_ln_w = need("layer_norm.weight", {d});
        e->dec_ln_b = need("layer_norm.bias", {d});
        for (int i = 0; i < e->L; ++i) {
            std::string p = "layers." + std::to_string(i) + ".";
            DecLayerW l;
            l.ln1_w = need(p + "self_attn_layer_norm.weight", {d}); l.ln1_b = need(p + "self_attn_layer_norm.bias", {d});
            l.qkv_w = need(p + "self_attn.qkv.weight", {3 * d, d}); l.qkv_b = need(p + "self_attn.qkv.bias", {3 * d});
            l.o_w = need(p + "self_attn.dense.weight", {d, d}); l.o_b = need(p + "self_attn.dense.bias", {d});
            l.fold_w = need(p + "encoder_attn.q_fold.weight", {d, 2 * d}); l.fold_c = need(p + "encoder_attn.q_fold.bias", {d});
            l.fold_r = need(p + "encoder_attn.q_fold.rowsum", {d}); l.fold_t = need(p + "encoder_attn.q_fold.shift", {d});
            l.ckv_w = need(p + "encoder_attn.kv.weight", {2 * d, d}); l.ckv_b = need(p + "encoder_attn.kv.bias", {2 * d});
            l.co_w = need(p + "encoder_attn.dense.weight", {d, d}); l.co_b = need(p + "encoder_attn.dense.bias", {d});
            l.ln3_w = need(p + "final_layer_norm.weight", {d}); l.ln3_b = need(p + "final_layer_norm.bias", {d});
            l.fc1_w = need(p + "fc1.weight", {F, d}); l.fc1_b = need(p + "fc1.bias", {F});
            l.fc2_w = need(p + "fc2.weight", {d, F}); l.fc2_b = need(p + "fc2.bias", {d});
            e->dec_layers.push_back(l);
        }
    }
    if (g_err[0]) {
        std::string keep = g_err;
        wt_engine_close(e);
        snprintf(g_err, sizeof g_err, "%s", keep.c_str());
        return WT_E_INVALID;
    }
    {   // fp32 engines: the weights of the big GEMMs once more as three bf16 planes (wt_common.h: launch_gemm_x3)
        static const bool x3_on = !(tuning_env("WT_GEMM_X3") && atoi(tuning_env("WT_GEMM_X3")) == 0);   // A/B switch: native fp32 MFMA GEMMs
        const int rc = (x3_on && e->precision == WT_F32 && (e->d % 16) == 0 && (e->F % 16) == 0) ? build_x3_weights(e) : WT_OK;
        if (rc) {
            std::string keep = g_err;
            wt_engine_close(e);
            snprintf(g_err, sizeof g_err, "%s", keep.c_str());
            return rc;
        }
    }
    *out = e;
    return WT_OK;
}

extern "C" int wt_engine_clone(const wt_engine* src, wt_engine** out) {
    if (!src || !out) return fail(WT_E_INVALID, "wt_engine_clone: null argument");
    *out = nullptr;
    wt_engine* e = new wt_engine();
    e->kind = src->kind; e->device = src->device; e->precision = src->precision;
    e->d = src->d; e->H = src->H; e->L = src->L; e->F = src->F; e->C = src->C; e->S = src->S; e->T = src->T; e->V = src->V;
    e->weights = src->weights;            // shared, read-only
    e->weights_base = src->weights_base;
    e->w = src->w;
    e->enc_layers = src->enc_layers;
    e->conv1_w = src->conv1_w; e->conv1_b = src->conv1_b; e->conv2_w = src->conv2_w; e->conv2_b = src->conv2_b;
    e->enc_pos = src->enc_pos; e->enc_ln_w = src->enc_ln_w; e->enc_ln_b = src->enc_ln_b;
    e->dec_layers = src->dec_layers;
    e->tok_emb = src->tok_emb; e->pos_emb = src->pos_emb; e->proj_w = src->proj_w; e->dec_ln_w = src->dec_ln_w; e->dec_ln_b = src->dec_ln_b;
    e->w_half = src->w_half; e->kv_esz = src->kv_esz; e->use_x3 = src->use_x3; e->conv1_w3 = src->conv1_w3; e->conv2_w3 = src->conv2_w3;
    // workspace, resident caches, step graphs, streams, mailbox, timers: this handle's own, allocated on first use like a fresh engine's
    *out = e;
    return WT_OK;
}

extern "C" int wt_engine_get_info(const wt_engine* e, wt_engine_info* out) {
    if (!e || !out) return fail(WT_E_INVALID, "wt_engine_get_info: null argument");
    out->kind = e->kind; out->precision = e->precision;
    out->d_model = e->d; out->n_heads = e->H; out->n_layers = e->L; out->ffn_dim = e->F;
    out->n_mels = e->C; out->max_source_positions = e->S; out->max_target_positions = e->T; out->vocab_size = e->V;
    return WT_OK;
}

extern "C" int wt_engine_gemm_mode(const wt_engine* e) { return e && e->use_x3 ? 1 : 0; }

// ------------------------------------------------------------------------------------------------- encoder
// (workspace clears are stream-ordered on the CALLER's stream: a synchronous hipMemset runs on the legacy stream, which the runtime
// refuses while ANOTHER host thread -- another worker's handle -- is capturing its step graph)
static int enc_reserve(wt_engine* e, int B, hipStream_t s) {
    if (B <= e->enc_cap) return WT_OK;
    if (e->enc_ws) { hipFree(e->enc_ws); e->enc_ws = nullptr; e->enc_cap = 0; }
    const size_t Fr = 2 * (size_t)e->S, M = (size_t)B * e->S, d = e->d;
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off = align_up(off + floats * 4, 256); return o; };
    const size_t o_melT = take((size_t)B * (Fr + 2) * e->C + 4 * e->C);
    const size_t o_c1 = take((size_t)B * (Fr + 2) * d + 4 * d);
    const size_t o_h = take(M * d), o_x = take(M * d), o_qkv = take(M * 3 * d), o_ctx = take(M * d), o_ffn = take(M * e->F);
    // x3 GEMMs: their A operands as three bf16 planes (6 bytes per element = 1.5 floats)
    const size_t o_xs3 = take(e->use_x3 ? (M * d * 3 + 1) / 2 : 0), o_ctx3 = take(e->use_x3 ? (M * d * 3 + 1) / 2 : 0), o_ffn3 = take(e->use_x3 ? (M * e->F * 3 + 1) / 2 : 0);
    const size_t o_qkv3 = take(e->use_x3 ? (M * 3 * d * 3 + 1) / 2 : 0);
    const size_t n_melT = (size_t)B * (Fr + 2) * e->C + 4 * e->C, n_c1 = (size_t)B * (Fr + 2) * d + 4 * d;   // elements per plane
    const size_t o_melT3 = take(e->conv1_w3 ? (n_melT * 3 + 1) / 2 : 0), o_c1_3 = take(e->conv1_w3 ? (n_c1 * 3 + 1) / 2 : 0);
    // fp16 engines reuse the fp32-sized regions for their half-precision activations (half the bytes)
    hipError_t he = hipMalloc((void**)&e->enc_ws, off);
    if (he != hipSuccess) return fail(WT_E_NOMEM, "hipMalloc(%zu) for encoder workspace (batch %d) failed: %s", off, B, hipGetErrorString(he));
    e->melT = (float*)(e->enc_ws + o_melT); e->c1 = (float*)(e->enc_ws + o_c1); e->hbuf = (float*)(e->enc_ws + o_h);
    e->xbuf = (float*)(e->enc_ws + o_x); e->qkv = (float*)(e->enc_ws + o_qkv); e->ctx = (float*)(e->enc_ws + o_ctx);
    e->ffn = (float*)(e->enc_ws + o_ffn);
    e->xs3 = e->enc_ws + o_xs3; e->ctx3 = e->enc_ws + o_ctx3; e->ffn3 = e->enc_ws + o_ffn3; e->qkv3 = e->enc_ws + o_qkv3;
    e->melT3 = e->enc_ws + o_melT3; e->c1_3 = e->enc_ws + o_c1_3;
    e->melT3_plane = n_melT; e->c1_3_plane = n_c1;
    if (e->conv1_w3) {   // zero padding rows of the plane forms (a bf16 zero is all-zero bits)
        HIPCHK(hipMemsetAsync(e->melT3, 0, n_melT * 6, s));
        HIPCHK(hipMemsetAsync(e->c1_3, 0, n_c1 * 6, s));
    }
    e->melT_h = e->melT; e->c1_h = e->c1; e->x_h = e->xbuf; e->ffn_h = e->ffn;
    e->ctx_h = (char*)e->ffn + M * e->F * 2;  // second half of the ffn region (M*F*2 bytes >= M*d*2)
    // conv zero-padding rows (row 0 / row F+1 of every utterance) are never written by the kernels below
    HIPCHK(hipMemsetAsync(e->melT, 0, ((size_t)B * (Fr + 2) * e->C + 4 * e->C) * 4, s));
    HIPCHK(hipMemsetAsync(e->c1, 0, ((size_t)B * (Fr + 2) * d + 4 * d) * 4, s));
    // Growth is rare.  The ABI takes a stream per call: finish the clears before returning, so that a following call on this handle
    // from a DIFFERENT stream cannot overtake them (allowed while other threads capture: thread-local capture mode)
    HIPCHK(hipStreamSynchronize(s));
    e->enc_cap = B;
    return WT_OK;
}

#define LAUNCH(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) return fail(WT_E_HIP, "kernel launch %s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

static int timed_gemm(wt_engine* e, const GemmParams& p, hipStream_t s) {
    hipEvent_t a, b;
    timer_begin(e, e->t_gemm, s, &a, &b);
    LAUNCH(launch_gemm_f32(p, s));
    timer_end(e, e->t_gemm, s, a, b);
    return WT_OK;
}

// fp16 encoder: fp16 GEMM operands, fp32 accumulation / residual stream / LayerNorm statistics / attention.
static int encoder_forward_f16(wt_engine* e, const float* mel, int B, float* out, hipStream_t s) {
    const int S = e->S, Fr = 2 * S, d = e->d, C = e->C, M = B * S;
    int rc;
    auto hgemm = [&](const GemmParams& q, bool out_half) {
        hipEvent_t a, b;
        timer_begin(e, e->t_gemm, s, &a, &b);
        hipError_t le = launch_gemm_f16(q, out_half, s);
        timer_end(e, e->t_gemm, s, a, b);
        return le == hipSuccess ? WT_OK : fail(WT_E_HIP, "fp16 GEMM launch failed: %s", hipGetErrorString(le));
    };
    LAUNCH(launch_mel_transpose_h(mel, e->melT_h, B, C, Fr, s));
    GemmParams g;
    memset(&g, 0, sizeof g);
    g.A = (const float*)e->melT_h; g.lda = C; g.a_rows_per_batch = Fr; g.a_batch_stride = (long long)(Fr + 2) * C;
    g.W = e->conv1_w; g.bias = e->conv1_b; g.M = B * Fr; g.N = d; g.K = 3 * C; g.act = 1;
    g.C = (float*)((char*)e->c1_h + (size_t)d * 2); g.ldc = d; g.c_rows_per_batch = Fr; g.c_batch_stride = (long long)(Fr + 2) * d;
    if ((rc = hgemm(g, true))) return rc;
    memset(&g, 0, sizeof g);
    g.A = (const float*)e->c1_h; g.lda = 2 * d; g.a_rows_per_batch = S; g.a_batch_stride = (long long)(Fr + 2) * d;
    g.W = e->conv2_w; g.bias = e->conv2_b; g.M = M; g.N = d; g.K = 3 * d; g.act = 1; g.pos = e->enc_pos;
    g.C = e->hbuf; g.ldc = d; g.c_rows_per_batch = S; g.c_batch_stride = (long long)S * d;
    if ((rc = hgemm(g, false))) return rc;
    auto dense = [&](const void* A, int K, const float* Wt, const float* bias, int N, void* Cout, int act, const float* resid,
                     bool out_half) {
        GemmParams q;
        memset(&q, 0, sizeof q);
        q.A = (const float*)A; q.lda = K; q.a_rows_per_batch = M; q.W = Wt; q.bias = bias; q.M = M; q.N = N; q.K = K; q.act = act;
        q.C = (float*)Cout; q.ldc = N; q.c_rows_per_batch = M; q.resid = resid;
        return hgemm(q, out_half);
    };
    for (int i = 0; i < e->L; ++i) {
        const EncLayerW& l = e->enc_layers[i];
        LAUNCH(launch_layernorm_h(e->hbuf, l.ln1_w, l.ln1_b, e->x_h, M, d, s));
        if ((rc = dense(e->x_h, d, l.qkv_w, l.qkv_b, 3 * d, e->qkv, 0, nullptr, true))) return rc;   // q|k|v as fp16
        {
            hipEvent_t a, b;
            timer_begin(e, e->t_enc_attn, s, &a, &b);
            LAUNCH(launch_encoder_attention_f16(e->qkv, e->ctx_h, B, S, e->H, s));                      // fp16 in/out, fp32 softmax
            timer_end(e, e->t_enc_attn, s, a, b);
        }
        if ((rc = dense(e->ctx_h, d, l.o_w, l.o_b, d, e->hbuf, 0, e->hbuf, false))) return rc;
        LAUNCH(launch_layernorm_h(e->hbuf, l.ln2_w, l.ln2_b, e->x_h, M, d, s));
        if ((rc = dense(e->x_h, d, l.fc1_w, l.fc1_b, e->F, e->ffn_h, 1, nullptr, true))) return rc;
        if ((rc = dense(e->ffn_h, e->F, l.fc2_w, l.fc2_b, d, e->hbuf, 0, e->hbuf, false))) return rc;
    }
    LAUNCH(launch_layernorm(e->hbuf, e->enc_ln_w, e->enc_ln_b, out, M, d, s));
    return WT_OK;
}

extern "C" int wt_encoder_forward(wt_engine* e, const float* mel, int B, float* out, void* stream) {
    if (!e || e->kind != WT_KIND_ENCODER) return fail(WT_E_INVALID, "wt_encoder_forward: not an encoder engine");
    if (!mel || !out || B < 1) return fail(WT_E_INVALID, "wt_encoder_forward: bad arguments (batch %d)", B);
    DeviceGuard guard(e->device);
    HIPCHK(guard.err);
    hipStream_t s = (hipStream_t)stream;
    int rc = enc_reserve(e, B, s);
    if (rc) return rc;
    const int S = e->S, Fr = 2 * S, d = e->d, C = e->C, M = B * S;
    if (e->precision == WT_F16) return encoder_forward_f16(e, mel, B, out, s);
    GemmParams g;
    bool convs_x3 = false;
    if (e->use_x3 && e->conv1_w3) {   // both convolutions as launch_gemm_x3 (operand planes; conv1's GELU output goes out as planes for conv2)
        const size_t n_melT = e->melT3_plane, n_c1 = e->c1_3_plane;
        GemmParams c1g, c2g;
        memset(&c1g, 0, sizeof c1g);
        c1g.A = (const float*)e->melT3; c1g.lda = C; c1g.a_rows_per_batch = Fr; c1g.a_batch_stride = (long long)(Fr + 2) * C; c1g.a_plane = (long long)n_melT;
        c1g.W = (const float*)e->conv1_w3; c1g.w_plane = (long long)d * 3 * C; c1g.bias = e->conv1_b; c1g.M = B * Fr; c1g.N = d; c1g.K = 3 * C; c1g.act = 1;
        c1g.C = (float*)((char*)e->c1_3 + (size_t)d * 2); c1g.ldc = d; c1g.c_rows_per_batch = Fr; c1g.c_batch_stride = (long long)(Fr + 2) * d;
        c1g.out_split = 1; c1g.c_plane = (long long)n_c1;
        memset(&c2g, 0, sizeof c2g);
        c2g.A = (const float*)e->c1_3; c2g.lda = 2 * d; c2g.a_rows_per_batch = S; c2g.a_batch_stride = (long long)(Fr + 2) * d; c2g.a_plane = (long long)n_c1;
        c2g.W = (const float*)e->conv2_w3; c2g.w_plane = (long long)d * 3 * d; c2g.bias = e->conv2_b; c2g.M = M; c2g.N = d; c2g.K = 3 * d; c2g.act = 1; c2g.pos = e->enc_pos;
        c2g.C = e->hbuf; c2g.ldc = d; c2g.c_rows_per_batch = S; c2g.c_batch_stride = (long long)S * d;
        if (gemm_x3_usable(c1g) && gemm_x3_usable(c2g)) {
            convs_x3 = true;
            LAUNCH(launch_mel_transpose_split(mel, e->melT3, n_melT, B, C, Fr, s));
            for (const GemmParams* q : {&c1g, &c2g}) {
                hipEvent_t a, b;
                timer_begin(e, e->t_gemm, s, &a, &b);
                LAUNCH(launch_gemm_x3(*q, s));
                timer_end(e, e->t_gemm, s, a, b);
            }
        }
    }
    if (!convs_x3) {
    LAUNCH(launch_mel_transpose(mel, e->melT, B, C, Fr, s));
    memset(&g, 0, sizeof g);
    // conv1 (k=3, pad 1) + GELU as an implicit GEMM over the time-major padded mel (model.py:78,97)
    g.A = e->melT; g.lda = C; g.a_rows_per_batch = Fr; g.a_batch_stride = (long long)(Fr + 2) * C;
    g.W = e->conv1_w; g.bias = e->conv1_b; g.M = B * Fr; g.N = d; g.K = 3 * C; g.act = 1;
    g.C = e->c1 + d; g.ldc = d; g.c_rows_per_batch = Fr; g.c_batch_stride = (long long)(Fr + 2) * d;
    if ((rc = timed_gemm(e, g, s))) return rc;
    // conv2 (k=3, stride 2, pad 1) + GELU + embed_positions, written as [B*S, d] (model.py:79,98-102)
    memset(&g, 0, sizeof g);
    g.A = e->c1; g.lda = 2 * d; g.a_rows_per_batch = S; g.a_batch_stride = (long long)(Fr + 2) * d;
    g.W = e->conv2_w; g.bias = e->conv2_b; g.M = M; g.N = d; g.K = 3 * d; g.act = 1; g.pos = e->enc_pos;
    g.C = e->hbuf; g.ldc = d; g.c_rows_per_batch = S; g.c_batch_stride = (long long)S * d;
    if ((rc = timed_gemm(e, g, s))) return rc;
    }

    auto dense = [&](const float* A, int K, const float* Wt, const float* bias, int N, float* Cout, int act,
                     const float* resid) {
        GemmParams q;
        memset(&q, 0, sizeof q);
        q.A = A; q.lda = K; q.a_rows_per_batch = M; q.W = Wt; q.bias = bias; q.M = M; q.N = N; q.K = K; q.act = act;
        q.C = Cout; q.ldc = N; q.c_rows_per_batch = M; q.resid = resid;
        return timed_gemm(e, q, s);
    };
    // launch_gemm_x3 form of a layer GEMM: A / W as three bf16 planes, fp32 (or, for fc1, three-plane) output
    auto dense3 = [&](const void* A3, int K, const void* W3, const float* bias, int N, void* Cout, int act, const float* resid, bool out_split) {
        GemmParams q;
        memset(&q, 0, sizeof q);
        q.A = (const float*)A3; q.lda = K; q.a_rows_per_batch = M; q.a_plane = (long long)M * K; q.W = (const float*)W3; q.w_plane = (long long)N * K;
        q.bias = bias; q.M = M; q.N = N; q.K = K; q.act = act; q.C = (float*)Cout; q.ldc = N; q.c_rows_per_batch = M; q.resid = resid;
        q.out_split = out_split ? 1 : 0; q.c_plane = (long long)M * N;
        hipEvent_t a, b;
        timer_begin(e, e->t_gemm, s, &a, &b);
        LAUNCH(launch_gemm_x3(q, s));
        timer_end(e, e->t_gemm, s, a, b);
        return WT_OK;
    };
    if (e->use_x3) {
        // fp32 products formed from six bf16 MFMAs of exactly split operands (wt_common.h: launch_gemm_x3): every producer of a GEMM's
        // A operand -- LayerNorm, attention, fc1's GELU epilogue -- writes the three planes directly, so no extra pass splits anything
        const size_t pl_d = (size_t)M * d;
        for (int i = 0; i < e->L; ++i) {
            const EncLayerW& l = e->enc_layers[i];
            LAUNCH(launch_layernorm_split(e->hbuf, l.ln1_w, l.ln1_b, e->xs3, pl_d, M, d, s));
            static const bool attn_x3 = !(tuning_env("WT_ATTN_X3") && atoi(tuning_env("WT_ATTN_X3")) == 0);   // A/B switch: fp32-MFMA attention
            if ((rc = dense3(e->xs3, d, l.qkv_w3, l.qkv_b, 3 * d, attn_x3 ? e->qkv3 : (void*)e->qkv, 0, nullptr, attn_x3))) return rc;
            {
                hipEvent_t a, b;
                timer_begin(e, e->t_enc_attn, s, &a, &b);
                if (attn_x3) LAUNCH(launch_encoder_attention_x3(e->qkv3, (size_t)M * 3 * d, e->ctx3, pl_d, B, S, e->H, s));
                else LAUNCH(launch_encoder_attention(e->qkv, nullptr, B, S, e->H, s, e->ctx3, pl_d));
                timer_end(e, e->t_enc_attn, s, a, b);
            }
            if ((rc = dense3(e->ctx3, d, l.o_w3, l.o_b, d, e->hbuf, 0, e->hbuf, false))) return rc;
            LAUNCH(launch_layernorm_split(e->hbuf, l.ln2_w, l.ln2_b, e->xs3, pl_d, M, d, s));
            if ((rc = dense3(e->xs3, d, l.fc1_w3, l.fc1_b, e->F, e->ffn3, 1, nullptr, true))) return rc;
            if ((rc = dense3(e->ffn3, e->F, l.fc2_w3, l.fc2_b, d, e->hbuf, 0, e->hbuf, false))) return rc;
        }
        LAUNCH(launch_layernorm(e->hbuf, e->enc_ln_w, e->enc_ln_b, out, M, d, s));
        return WT_OK;
    }
    for (int i = 0; i < e->L; ++i) {
        const EncLayerW& l = e->enc_layers[i];
        LAUNCH(launch_layernorm(e->hbuf, l.ln1_w, l.ln1_b, e->xbuf, M, d, s));
        if ((rc = dense(e->xbuf, d, l.qkv_w, l.qkv_b, 3 * d, e->qkv, 0, nullptr))) return rc;
        {
            hipEvent_t a, b;
            timer_begin(e, e->t_enc_attn, s, &a, &b);
            LAUNCH(launch_encoder_attention(e->qkv, e->ctx, B, S, e->H, s));
            timer_end(e, e->t_enc_attn, s, a, b);
        }
        if ((rc = dense(e->ctx, d, l.o_w, l.o_b, d, e->hbuf, 0, e->hbuf))) return rc;
        LAUNCH(launch_layernorm(e->hbuf, l.ln2_w, l.ln2_b, e->xbuf, M, d, s));
        if ((rc = dense(e->xbuf, d, l.fc1_w, l.fc1_b, e->F, e->ffn, 1, nullptr))) return rc;
        if ((rc = dense(e->ffn, e->F, l.fc2_w, l.fc2_b, d, e->hbuf, 0, e->hbuf))) return rc;
    }
    LAUNCH(launch_layernorm(e->hbuf, e->enc_ln_w, e->enc_ln_b, out, M, d, s));
    return WT_OK;
}

// ------------------------------------------------------------------------------------------------- decoder
static int dec_reserve(wt_engine* e, int B, int max_length, hipStream_t s, int pool_rows = 0) {
    if (B <= e->dec_cap && max_length <= e->dec_maxlen_cap && pool_rows <= e->cross_rows) return WT_OK;
    if (B < e->dec_cap) B = e->dec_cap;                      // never shrink: a later, smaller call keeps what an earlier one needed
    if (max_length < e->dec_maxlen_cap) max_length = e->dec_maxlen_cap;
    if (pool_rows < e->cross_rows) pool_rows = e->cross_rows;
    if (e->dec_ws) { hipFree(e->dec_ws); e->dec_ws = nullptr; e->dec_cap = 0; e->cross_rows = 0; }
    e->graph_valid = false;
    e->tables_valid = false;
    const int cap_len = max_length > e->T ? max_length : e->T;
    const int cross_rows = pool_rows > B ? pool_rows : B, enc_rows = B > MAX_ROWS ? B : MAX_ROWS;
    const size_t d = e->d, kv_self = (size_t)e->L * B * e->H * e->T * HEAD_DIM, kv_cross = (size_t)e->L * cross_rows * e->H * e->S * HEAD_DIM;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_sk = take(kv_self * e->kv_esz), o_sv = take(kv_self * e->kv_esz), o_ck = take(kv_cross * e->kv_esz), o_cv = take(kv_cross * e->kv_esz);
    const size_t o_ench = take(e->w_half ? (size_t)enc_rows * e->S * d * 2 : 0);
    const size_t o_enc3 = take(e->use_x3 ? (size_t)enc_rows * e->S * d * 6 : 0);
    const size_t o_h = take((size_t)B * d * 4), o_h2 = take((size_t)B * d * 4), o_q = take((size_t)B * d * 4), o_att = take((size_t)B * d * 4), o_f = take((size_t)B * e->F * 4);
    const size_t o_cnt = take((size_t)B * e->H * 4);
    const size_t o_selv = take((size_t)B * SEL_PARTS_CAP * 4), o_seli = take((size_t)B * SEL_PARTS_CAP * 4);
    const size_t o_part = take((size_t)B * e->H * 16 * PART_STRIDE * 4), o_lg = take((size_t)B * e->V * 4);
    const size_t o_st = take(sizeof(DecState)), o_ids = take((size_t)B * cap_len * 4), o_unf = take((size_t)B * 4);
    const size_t o_forced = take((size_t)(cap_len + 1) * 4), o_mask = take((size_t)e->V), o_frows = take(16 * 4);
    hipError_t he = hipMalloc((void**)&e->dec_ws, off);
    if (he != hipSuccess) return fail(WT_E_NOMEM, "hipMalloc(%zu) for decoder workspace (batch %d) failed: %s", off, B, hipGetErrorString(he));
    char* b = e->dec_ws;
    e->self_k = (float*)(b + o_sk); e->self_v = (float*)(b + o_sv); e->cross_k = (float*)(b + o_ck); e->cross_v = (float*)(b + o_cv);
    e->dh = (float*)(b + o_h); e->dh2 = (float*)(b + o_h2); e->dq = (float*)(b + o_q); e->datt = (float*)(b + o_att); e->att_cnt = (int*)(b + o_cnt); e->sel_val = (float*)(b + o_selv); e->sel_idx = (int*)(b + o_seli); e->dffn = (float*)(b + o_f); e->part = (float*)(b + o_part);
    e->logits = (float*)(b + o_lg); e->st = (DecState*)(b + o_st); e->ids = (int*)(b + o_ids); e->unfinished = (int*)(b + o_unf);
    e->forced = (int*)(b + o_forced); e->mask = (uint8_t*)(b + o_mask); e->force_rows = (int*)(b + o_frows);
    e->h_force_rows.clear();
    e->enc_h = e->w_half ? (void*)(b + o_ench) : nullptr;
    e->enc3 = e->use_x3 ? (void*)(b + o_enc3) : nullptr;
    HIPCHK(hipMemsetAsync(e->att_cnt, 0, (size_t)B * e->H * 4, s));  // arrival tickets start (and are left) at zero; stream-ordered (see enc_reserve)
    if (!e->h_state) HIPCHK(hipHostMalloc((void**)&e->h_state, sizeof(DecState), hipHostMallocDefault));
    if (!e->mailbox) {
        HIPCHK(hipHostMalloc((void**)&e->mailbox, 64, hipHostMallocMapped | hipHostMallocCoherent));
        *e->mailbox = 0;
        HIPCHK(hipHostGetDevicePointer((void**)&e->mailbox_dev, (void*)e->mailbox, 0));
    }
    if (!e->own_stream) {
        HIPCHK(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    }
    HIPCHK(hipStreamSynchronize(s));   // as in enc_reserve: the ticket clear above is ordered before any later call on any stream
    e->dec_cap = B;
    e->cross_rows = cross_rows;
    e->dec_maxlen_cap = cap_len;
    return WT_OK;
}

static int pick_splits(int B, int H, int len) {
    // Key splits of the cross-attention launch.  Two splits are merged for free by the consuming GEMV (deferred merge), more
    // are merged by the attention kernel's last-arriving block, which costs ~2 us: measured on medium.en, ms per step at
    // B = 1: 16 splits 1.25, 8: 1.12, 4: 1.09, 2: 1.14;  B = 2: 8: 1.16, 4: 1.12, 2: 1.15;  B = 4: 4: 1.27, 2: 1.26;
    // B = 8: 2: 1.47, 3: 1.76, 4: 1.80.  So: 4 while (utterance, head) pairs are scarce (<= 64), else 2, none from 256 pairs up.
    int n = (256 + B * H - 1) / (B * H);
    n = n >= 4 ? 4 : n >= 2 ? 2 : 1;   // (tiny.en B = 8, 48 pairs: 4 splits 0.218, 2 splits 0.225; small.en B = 8, 96 pairs: 2 splits 0.693, 4: 0.706)
    while (n > 1 && len / n < 48) n >>= 1;
    return n;
}

// one decoder step over the resident (or caller-provided) caches; every step-dependent quantity comes from *st
struct StepIO {
    const int* ids; int ids_ld;            // token fed to row b = ids[b*ids_ld + st->cur_len-1]
    float *self_k, *self_v; int self_cap;  // layer stride = B*H*self_cap*64 elements
    float *cross_k, *cross_v;              // layer stride = cross_rows*H*S*64 elements
    int cross_rows;                        // cache rows per layer (the engine's resident pool; 1 on the Session path)
    const int* slot_row;                   // continuous mode: slot b attends over cache row slot_row[b] (DecAttnParams.slot_row)
    int kv_esz;                            // bytes per cache element: 4 (fp32; always on the Session path), 2 (resident caches of an fp16 engine)
    float* logits;                         // [B][V]
    int B, nsplit_self, nsplit_cross;
    const int* alive;                      // skip-finished-rows graph of the fast path: per-row "unfinished" flags (DecAttnParams.alive)
    bool embed;                            // launch the input-embedding kernel (the fast path gets it from greedy_finish)
    // greedy fast path: masked argmax fused into the vocabulary GEMV (logits == nullptr: they never reach HBM unless traced)
    const SelectParams* argmax;            // or nullptr: write plain logits
    int* argmax_parts;                     // out: partial results per batch row (workgroups of the GEMV)
};

// The seven dependent launches of decoder layer i (model.py:321-369).  h = residual stream in, h1 = the other buffer of the
// two-buffer stream (the layer's output ends up in h1).  `part` selects one launch: the decode step enqueues 0..6 in order,
// wt_decoder_time_kernel replays one kind over all layers.
enum { LP_QKV = 0, LP_SELF_ATTN, LP_PAIR, LP_CROSS_ATTN, LP_CROSS_OUT, LP_FC1, LP_FC2, LP_COUNT };
static int enqueue_layer_part(wt_engine* e, const StepIO& io, int i, float* h, float* h1, int part, hipStream_t s) {
    static const bool defer = tuning_env("WT_NO_DEFER_MERGE") == nullptr;  // A/B switch
    const int d = e->d, B = io.B, H = e->H;
    const DecLayerW& l = e->dec_layers[i];
    const size_t self_layer = (size_t)i * B * H * io.self_cap * HEAD_DIM * io.kv_esz, cross_layer = (size_t)i * io.cross_rows * H * e->S * HEAD_DIM * io.kv_esz;
    float* sk = (float*)((char*)io.self_k + self_layer);
    float* sv = (float*)((char*)io.self_v + self_layer);
    const float* ck = (const float*)((const char*)io.cross_k + cross_layer);
    const float* cv = (const float*)((const char*)io.cross_v + cross_layer);
    const int wh = e->w_half ? 1 : 0, kvh = io.kv_esz == 2 ? 1 : 0;
    SkinnyParams k, k2;
    DecAttnParams a;
    memset(&k, 0, sizeof k);
    memset(&k2, 0, sizeof k2);
    memset(&a, 0, sizeof a);
    switch (part) {
    case LP_QKV:  // self attention (model.py:273-281, 283-304): LN -> q|k|v, append k/v row in place
        k.X = h; k.ln_w = l.ln1_w; k.ln_b = l.ln1_b; k.xmode = XMODE_LAYERNORM; k.W = l.qkv_w; k.bias = l.qkv_b;
        k.Y = e->dq; k.kcache = sk; k.vcache = sv; k.st = e->st; k.B = B; k.N = 3 * d; k.K = d; k.ymode = YMODE_QKV_APPEND;
        k.d_model = d; k.s_cap = io.self_cap; k.q_scale = 0.125f; k.w_nt = e->nt_loads; k.w_half = wh; k.kv_half = kvh;
        LAUNCH(launch_skinny(k, s));
        break;
    case LP_SELF_ATTN:
        a.q = e->dq; a.kcache = sk; a.vcache = sv; a.part = e->part; a.cnt = e->att_cnt; a.out = e->datt; a.st = e->st; a.B = B; a.H = H; a.s_cap = io.self_cap;
        a.n_split = io.nsplit_self; a.fixed_len = 0; a.nt = e->nt_loads; a.kv_half = kvh; a.alive = io.alive;
        a.defer_merge = defer && io.nsplit_self == 2;   // both halves of the pair launch below merge the two partials while staging
        LAUNCH(launch_dec_attn(a, s));
        break;
    case LP_PAIR:
        // ONE launch: out-projection + residual (h1 = h + Wo.a + bo) and the folded cross-attention query
        // u = s.Wq.diag(gamma2).(h + Wo.a + bo) = fold_w.[a ; h] + fold_c; LayerNorm statistics of h1 are applied
        // by the cross-attention kernel (model.py:261-272 semantics, one dependent launch fewer per layer)
        k.X = e->datt; k.xmode = XMODE_PLAIN; k.W = l.o_w; k.bias = l.o_b; k.resid = h;
        k.Y = h1; k.st = e->st; k.B = B; k.N = d; k.K = d; k.q_scale = 1.f; k.w_nt = e->nt_loads; k.w_half = wh;
        k2.X = e->datt; k2.X2 = h; k2.xmode = XMODE_PLAIN; k2.x_direct = 1; k2.W = l.fold_w; k2.bias = l.fold_c;
        k2.Y = e->dq; k2.st = e->st; k2.B = B; k2.N = d; k2.K = 2 * d; k2.q_scale = 1.f; k2.w_nt = e->nt_loads; k2.w_half = wh;
        if (defer && io.nsplit_self == 2) {   // the self-attention context `a` arrives as two split partials per (utterance, head)
            k.parts = e->part; k.parts_nsplit = 2; k.parts_H = H;
            k2.parts = e->part; k2.parts_nsplit = 2; k2.parts_H = H;
        }
        LAUNCH(launch_skinny_pair(k, k2, s));
        break;
    case LP_CROSS_ATTN: {  // cross attention over the encoder memory: K/V already resident
        a.q = e->dq; a.kcache = ck; a.vcache = cv; a.part = e->part; a.cnt = e->att_cnt; a.out = e->datt; a.st = e->st; a.B = B; a.H = H; a.s_cap = e->S;
        a.n_split = io.nsplit_cross; a.fixed_len = e->S; a.nt = e->nt_loads; a.ln_h = h1; a.ln_r = l.fold_r; a.ln_t = l.fold_t; a.kv_half = kvh;
        a.alive = io.alive; a.slot_row = io.slot_row;
        a.defer_merge = defer && io.nsplit_cross == 2;  // the out-projection below merges the two split partials while staging them
                                                        // (more splits, i.e. batch < 8: the attention kernel merges its own, by ticket)
        hipEvent_t ta, tb;
        timer_begin(e, e->t_cross, s, &ta, &tb);
        LAUNCH(launch_dec_attn(a, s));
        timer_end(e, e->t_cross, s, ta, tb);
        break;
    }
    case LP_CROSS_OUT:
        k.X = e->datt; k.xmode = XMODE_PLAIN; k.W = l.co_w; k.bias = l.co_b; k.resid = h1;
        k.Y = h1; k.st = e->st; k.B = B; k.N = d; k.K = d; k.q_scale = 1.f; k.w_nt = e->nt_loads; k.w_half = wh;
        if (defer && io.nsplit_cross == 2) { k.parts = e->part; k.parts_nsplit = io.nsplit_cross; k.parts_H = H; }
        LAUNCH(launch_skinny(k, s));
        break;
    case LP_FC1:  // FFN (model.py:363-367)
        k.X = h1; k.ln_w = l.ln3_w; k.ln_b = l.ln3_b; k.xmode = XMODE_LAYERNORM; k.W = l.fc1_w; k.bias = l.fc1_b;
        k.Y = e->dffn; k.st = e->st; k.B = B; k.N = e->F; k.K = d; k.act = 1; k.q_scale = 1.f; k.w_nt = e->nt_loads; k.w_half = wh;
        LAUNCH(launch_skinny(k, s));
        break;
    case LP_FC2:
        k.X = e->dffn; k.xmode = XMODE_PLAIN; k.W = l.fc2_w; k.bias = l.fc2_b; k.resid = h1; k.Y = h1; k.st = e->st;
        k.B = B; k.N = d; k.K = e->F; k.q_scale = 1.f; k.w_nt = e->nt_loads; k.w_half = wh;
        LAUNCH(launch_skinny(k, s));
        break;
    default:
        return fail(WT_E_INVALID, "enqueue_layer_part: bad part %d", part);
    }
    return WT_OK;
}

static int enqueue_step(wt_engine* e, const StepIO& io, hipStream_t s) {
    const int d = e->d, B = io.B;
    if (io.embed) LAUNCH(launch_dec_embed(io.ids, io.ids_ld, e->tok_emb, e->pos_emb, e->dh, B, d, e->st, s, e->w_half));
    SkinnyParams k;
    // The residual stream ping-pongs between two buffers once per layer: the self-attention out-projection (h1 = h + Wo.a)
    // and the folded cross-attention query (which still reads h) share one launch, so h1 cannot overwrite h in place.
    float *h = e->dh, *h1 = e->dh2;
    for (int i = 0; i < e->L; ++i) {
        for (int part = 0; part < LP_COUNT; ++part) {
            const int rc = enqueue_layer_part(e, io, i, h, h1, part, s);
            if (rc) return rc;
        }
        std::swap(h, h1);
    }
    // final LN + vocabulary projection (model.py:455-457; logits are the engine's 'hidden_states' output)
    memset(&k, 0, sizeof k);
    k.X = h; k.ln_w = e->dec_ln_w; k.ln_b = e->dec_ln_b; k.xmode = XMODE_LAYERNORM; k.W = e->proj_w; k.Y = io.logits;
    k.st = e->st; k.B = B; k.N = e->V; k.K = d; k.q_scale = 1.f; k.w_nt = e->nt_loads; k.w_half = e->w_half;
    if (io.argmax) {  // Suppress -> SuppressAtBegin -> argmax in the epilogue (Force / pad / EOS rules: greedy_finish_kernel)
        const SelectParams& sp = *io.argmax;
        k.ymode = YMODE_ARGMAX; k.Y = nullptr; k.am_mask = sp.mask; k.am_val = sp.part_val; k.am_idx = sp.part_idx;
        k.am_begin_index = sp.begin_index; k.am_trace = sp.trace; k.am_trace_steps = sp.max_length - 1;
        const int grid = skinny_grid(k);
        if (grid < 1 || grid > SEL_PARTS_CAP) return fail(WT_E_UNSUPPORTED, "vocabulary GEMV plan has %d workgroups (cap %d)", grid, SEL_PARTS_CAP);
        k.am_ld = grid;
        *io.argmax_parts = grid;
    }
    {
        hipEvent_t ta, tb;
        timer_begin(e, e->t_skinny, s, &ta, &tb);
        LAUNCH(launch_skinny(k, s));
        timer_end(e, e->t_skinny, s, ta, tb);
    }
    return WT_OK;
}

static int cross_kv_project(wt_engine* e, const float* enc_hidden, int B, int rows, int seq_off, float* ck, float* cv, int kv_esz,
                            hipStream_t s, int layer_rows = 0, int row0 = 0) {
    // K/V projection of encoder rows [0, rows) of every utterance into cache rows [seq_off, seq_off+rows).  fp16 engines: the encoder
    // memory is rounded to fp16 once (the MFMA's A operand), the product accumulates in fp32 and lands in the caches as `kv_esz` says.
    // The caches hold `layer_rows` utterance rows per layer (default B); the B utterances land in rows [row0, row0 + B).
    const int d = e->d;
    if (layer_rows <= 0) layer_rows = B;
    if (row0 < 0 || row0 + B > layer_rows) return fail(WT_E_INVALID, "cross_kv_project: rows [%d, %d) outside a cache of %d rows per layer", row0, row0 + B, layer_rows);
    const size_t row_bytes = (size_t)e->H * e->S * HEAD_DIM * kv_esz;
    if (e->w_half) LAUNCH(launch_cast_h(enc_hidden, e->enc_h, (size_t)B * e->S * d, s));
    const size_t enc_plane = (size_t)B * e->S * d;
    bool x3 = e->use_x3 && !e->w_half && B <= (e->dec_cap > MAX_ROWS ? e->dec_cap : MAX_ROWS);
    if (x3) {   // the encoder memory as three bf16 planes (the x3 GEMM's A operand)
        GemmParams t;
        memset(&t, 0, sizeof t);
        t.A = (const float*)e->enc3; t.lda = d; t.a_rows_per_batch = rows; t.a_batch_stride = (long long)e->S * d; t.a_plane = (long long)enc_plane;
        t.W = (const float*)e->dec_layers[0].ckv_w3; t.w_plane = 2ll * d * d; t.M = B * rows; t.N = 2 * d; t.K = d; t.c_rows_per_batch = rows;
        x3 = gemm_x3_usable(t);
        if (x3) LAUNCH(launch_split3(enc_hidden, e->enc3, enc_plane, enc_plane, s));
    }
    for (int i = 0; i < e->L; ++i) {
        const DecLayerW& l = e->dec_layers[i];
        const size_t layer = ((size_t)i * layer_rows + row0) * row_bytes;
        GemmParams g;
        memset(&g, 0, sizeof g);
        g.A = e->w_half ? (const float*)e->enc_h : enc_hidden; g.lda = d; g.a_rows_per_batch = rows; g.a_batch_stride = (long long)e->S * d;
        g.W = l.ckv_w; g.bias = l.ckv_b; g.M = B * rows; g.N = 2 * d; g.K = d;
        g.epi = EPI_KV_HEADS; g.C = (float*)((char*)ck + layer); g.C2 = (float*)((char*)cv + layer);
        g.c_rows_per_batch = rows; g.kv_heads = e->H; g.kv_cap = e->S; g.kv_seq_off = seq_off;
        if (e->w_half) {
            hipEvent_t ta, tb;
            timer_begin(e, e->t_gemm, s, &ta, &tb);
            LAUNCH(launch_gemm_f16(g, kv_esz == 2, s));
            timer_end(e, e->t_gemm, s, ta, tb);
        } else if (x3) {
            g.A = (const float*)e->enc3; g.a_plane = (long long)enc_plane; g.W = (const float*)l.ckv_w3; g.w_plane = 2ll * d * d;
            hipEvent_t ta, tb;
            timer_begin(e, e->t_gemm, s, &ta, &tb);
            LAUNCH(launch_gemm_x3(g, s));
            timer_end(e, e->t_gemm, s, ta, tb);
        } else {
            int rc = timed_gemm(e, g, s);
            if (rc) return rc;
        }
    }
    return WT_OK;
}

// Everything wt_decoder_begin and wt_decoder_stream_begin share: argument checks, workspace, token-rule tables, cache policy and key
// splits of the step graph.  `B` = rows of the batch / decode slots of the stream; `pool_rows` = cross-cache rows (0: B).
static int decode_setup(wt_engine* e, int B, const wt_greedy_params* p, int pool_rows, bool stream_mode, hipStream_t s, const char* who) {
    if (B > MAX_ROWS) return fail(WT_E_UNSUPPORTED, "%s: batch %d > %d per call; shard the batch", who, B, MAX_ROWS);
    if (p->max_length < 2 || p->max_length > e->T) return fail(WT_E_INVALID, "max_length %d outside [2, max_target_positions=%d]", p->max_length, e->T);
    auto tok_ok = [&](int t) { return t >= 0 && t < e->V; };
    if (!tok_ok(p->decoder_start_token_id) || !tok_ok(p->eos_token_id) || !tok_ok(p->pad_token_id))
        return fail(WT_E_INVALID, "start/eos/pad token id outside the vocabulary");
    int rc = dec_reserve(e, B, p->max_length, s, pool_rows);
    if (rc) return rc;
    // token rules -> device tables (SuppressTokens / SuppressTokensAtBegin / ForceTokens).  The host images live in the engine:
    // when the rules are the ones of the previous decode (the usual case: one rule set per checkpoint) nothing is uploaded and
    // the call stays asynchronous; only a CHANGED rule set is uploaded, followed by one stream synchronisation.
    std::vector<uint8_t> mask(e->V, 0);
    for (int i = 0; i < p->n_suppress_tokens; ++i) {
        if (!tok_ok(p->suppress_tokens[i])) return fail(WT_E_INVALID, "suppress token %d outside the vocabulary", p->suppress_tokens[i]);
        mask[p->suppress_tokens[i]] |= 1;
    }
    for (int i = 0; i < p->n_begin_suppress_tokens; ++i) {
        if (!tok_ok(p->begin_suppress_tokens[i])) return fail(WT_E_INVALID, "begin-suppress token %d outside the vocabulary", p->begin_suppress_tokens[i]);
        mask[p->begin_suppress_tokens[i]] |= 2;
    }
    std::vector<int> forced(e->dec_maxlen_cap + 1, -1);
    for (int i = 0; i < p->n_forced; ++i) {
        const int idx = p->forced_decoder_ids[2 * i], tok = p->forced_decoder_ids[2 * i + 1];
        if (!tok_ok(tok)) return fail(WT_E_INVALID, "forced token %d outside the vocabulary", tok);
        if (idx >= 0 && idx <= e->dec_maxlen_cap) forced[idx] = tok;
    }
    if (!e->tables_valid || mask != e->h_mask || forced != e->h_forced) {
        e->tables_valid = false;   // a failed upload below must not leave "valid" tables whose host images already hold the new rules
        e->h_mask.swap(mask);
        e->h_forced.swap(forced);
        HIPCHK(hipMemcpyAsync(e->mask, e->h_mask.data(), e->h_mask.size(), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(e->forced, e->h_forced.data(), e->h_forced.size() * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));  // pageable sources: make sure the copies have left the host images before they can change
        e->tables_valid = true;
    }
    {   // bench-only per-row transcript lengths: uploaded (with one synchronisation) only when they change
        std::vector<int> rows;
        if (p->force_eos_steps && !stream_mode) rows.assign(p->force_eos_steps, p->force_eos_steps + B);
        const bool on = !rows.empty();
        if (on && rows != e->h_force_rows) {
            e->h_force_rows.swap(rows);
            HIPCHK(hipMemcpyAsync(e->force_rows, e->h_force_rows.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
        }
        if (on != e->force_rows_on) e->graph_valid = false;
        e->force_rows_on = on;
    }
    const bool same = e->begun && e->B == B && e->max_length == p->max_length && e->trace == p->logits_trace &&
                      e->begin_index == p->begin_index && e->eos == p->eos_token_id && e->pad == p->pad_token_id &&
                      e->force_eos_step == p->force_eos_step && e->start_token == p->decoder_start_token_id;
    if (!same) e->graph_valid = false;
    e->B = B; e->max_length = p->max_length; e->begin_index = p->begin_index; e->eos = p->eos_token_id;
    e->pad = p->pad_token_id; e->force_eos_step = p->force_eos_step; e->trace = p->logits_trace; e->start_token = p->decoder_start_token_id;
    // measured (tools/microbench.py, medium.en B=8): self attention is fastest unsplit at every length <= 448;
    // cross attention (1500 keys) with ~one block per CU
    {   // Everything a step touches is read exactly once per step: stream it non-temporally -- unless one whole step (weights + this
        // batch's K/V) fits the 256 MB Infinity Cache, where default-policy loads let step t+1 hit what step t brought in.
        // Measured ms per step, non-temporal vs default: tiny.en B=1 (136 MB) 0.171 vs 0.160; tiny.en B=8 (300 MB) 0.224 vs 0.221;
        // base.en B=8 0.324 vs 0.340; small.en B=8 0.688 vs 0.711; medium.en B=8 1.61 vs 1.66.
        const double p_step = (double)e->L * (6.0 * e->d * e->d + 2.0 * e->d * e->F) + (double)e->V * e->d;
        const double step_bytes = (e->w_half ? 2.0 : 4.0) * p_step + (double)B * 2.0 * e->L * e->H * ((double)e->S + p->max_length) * HEAD_DIM * e->kv_esz;
        const int nt = step_bytes > 160e6;
        if (nt != e->nt_loads) e->graph_valid = false;
        e->nt_loads = nt;
    }
    // self attention: unsplit.  Two key splits with the merge deferred into BOTH halves of the pair launch (WT_NSPLIT_SELF=2) are
    // implemented and tested, and measured a wash at medium.en batch 8: self-attention 9.0 -> 7.6 us at 447 keys, the pair launch
    // 8.2 -> 9.7 us (its blocks stage and merge two partial sets), 1.659 vs 1.649 ms per step in one A/B on one box.
    {
        const bool can_defer = tuning_env("WT_NO_DEFER_MERGE") == nullptr && e->d <= 1024;
        const int want = tuning_env("WT_NSPLIT_SELF") ? atoi(tuning_env("WT_NSPLIT_SELF")) : 1;
        const int ns = (want == 2 && can_defer) ? 2 : 1;
        if (ns != e->nsplit_self) e->graph_valid = false;
        e->nsplit_self = ns;
    }
    e->nsplit_cross = tuning_env("WT_NSPLIT_CROSS") ? atoi(tuning_env("WT_NSPLIT_CROSS")) : pick_splits(B, e->H, e->S);
    e->epoch = e->epoch % 0xffff + 1;   // 1..65535: never the value of a freshly zeroed mailbox
    e->issued = 0;
    e->stream_mode = stream_mode;
    return WT_OK;
}

extern "C" int wt_decoder_begin(wt_engine* e, const float* enc_hidden, int B, const wt_greedy_params* p, void* stream) {
    if (!e || e->kind != WT_KIND_DECODER) return fail(WT_E_INVALID, "wt_decoder_begin: not a decoder engine");
    if (!enc_hidden || !p || B < 1) return fail(WT_E_INVALID, "wt_decoder_begin: bad arguments");
    DeviceGuard guard(e->device);
    HIPCHK(guard.err);
    hipStream_t s = (hipStream_t)stream;
    int rc = decode_setup(e, B, p, 0, false, s, "wt_decoder_begin");
    if (rc) return rc;
    LAUNCH(launch_dec_init(e->st, e->ids, e->unfinished, B, p->max_length, p->decoder_start_token_id, e->epoch, s));
    LAUNCH(launch_dec_embed(e->ids, p->max_length, e->tok_emb, e->pos_emb, e->dh, B, e->d, e->st, s, e->w_half));  // input of step 0
    rc = cross_kv_project(e, enc_hidden, B, e->S, 0, e->cross_k, e->cross_v, e->kv_esz, s, e->cross_rows, 0);
    if (rc) return rc;
    e->begun = true;
    return WT_OK;
}

static void fill_select_params(wt_engine* e, SelectParams& sp) {
    memset(&sp, 0, sizeof sp);
    sp.logits = e->logits; sp.mask = e->mask; sp.forced = e->forced; sp.ids = e->ids; sp.unfinished = e->unfinished;
    sp.st = e->st; sp.trace = e->trace; sp.B = e->B; sp.V = e->V; sp.max_length = e->max_length;
    sp.begin_index = e->begin_index; sp.eos = e->eos; sp.pad = e->pad; sp.force_eos_step = e->force_eos_step;
    sp.part_val = e->sel_val; sp.part_idx = e->sel_idx; sp.tok_emb = e->tok_emb; sp.pos_emb = e->pos_emb; sp.next_x = e->dh;
    sp.d_model = e->d; sp.n_parts = 8; sp.fused = 0; sp.emb_half = e->w_half;
    sp.force_eos_rows = e->force_rows_on ? e->force_rows : nullptr; sp.mailbox = e->mailbox_dev;
    sp.start_token = e->start_token;
    if (e->stream_mode) {
        sp.stream = 1; sp.host_ids = e->h_ids_dev; sp.host_len = e->h_len_dev; sp.force_eos_step = -1; sp.force_eos_rows = nullptr; sp.trace = nullptr;
    }
}

// variant bit 0: finished / idle rows stream no K/V (`alive`); bit 1: the continuous mode's graph (per-slot cross-cache rows and state)
static int enqueue_fast_step(wt_engine* e, hipStream_t s, int variant = 0) {
    StepIO io;
    io.alive = (variant & 1) ? e->unfinished : nullptr;
    io.ids = e->ids; io.ids_ld = e->max_length; io.self_k = e->self_k; io.self_v = e->self_v; io.self_cap = e->T;
    io.cross_k = e->cross_k; io.cross_v = e->cross_v; io.cross_rows = e->cross_rows; io.logits = e->logits; io.B = e->B; io.kv_esz = e->kv_esz;
    io.slot_row = (variant & 2) ? e->st->slot_row : nullptr;   // (address arithmetic on a device pointer: nothing is dereferenced here)
    io.nsplit_self = e->nsplit_self; io.nsplit_cross = e->nsplit_cross;
    io.embed = false;  // dh already holds this step's input: written by wt_decoder_begin (step 0) or by the previous greedy_finish
    static const bool fuse = tuning_env("WT_NO_FUSED_ARGMAX") == nullptr;  // A/B switch: logits to HBM + greedy_select_kernel
    SelectParams sp;
    fill_select_params(e, sp);
    int parts = 0;
    io.argmax = fuse ? &sp : nullptr;
    io.argmax_parts = &parts;
    int rc = enqueue_step(e, io, s);
    if (rc) return rc;
    if (fuse) { sp.fused = 1; sp.n_parts = parts; }
    LAUNCH(launch_greedy_select(sp, s));
    return WT_OK;
}

// Captures and instantiations are rare (once per engine and decode configuration) and serialised process-wide: several host threads
// drive their own handles concurrently (runtime.WhisperPipeline), and a multi-worker run under rocprofv3 aborted inside the runtime
// while captures overlapped.  Replays (hipGraphLaunch) stay concurrent.  (What that does and does not explain of the round-3 abort:
// DESIGN.md "The four-worker abort under rocprofv3".)
static std::mutex g_capture_mutex;

// enqueue n_steps decoder steps on `s` (the caller holds the device guard).  variant 1: the step graph in which finished rows stream
// no K/V (wt_decoder_run switches to it once the mailbox reports a finished row; never with a logits trace, which records every
// row's logits at every step as the reference computes them).  The continuous mode replays variants 2 / 3.
static int enqueue_steps(wt_engine* e, int n_steps, hipStream_t s, int variant = 0) {
    if (e->trace) variant = 0;
    if (e->stream_mode) variant |= 2;
    if (e->profiling || !e->use_graph) {  // eager: per-kernel event timers need real launches
        for (int i = 0; i < n_steps; ++i) {
            int rc = enqueue_fast_step(e, s, variant);
            if (rc) return rc;
            e->issued += 1;
        }
        return WT_OK;
    }
    // The step graph is CAPTURED on the engine's own stream (the caller's may be the NULL stream, which cannot be captured;
    // capturing executes nothing) and REPLAYED on the caller's stream: a replay on a second stream measured 5 % slower
    // per step (1.69 vs 1.60 ms, medium.en B=8) than on the stream the rest of the pass already runs on.
    if (!e->graph_valid) {
        for (int v = 0; v < wt_engine::N_GRAPHS; ++v) {
            if (e->graph_exec[v]) { hipGraphExecDestroy(e->graph_exec[v]); e->graph_exec[v] = nullptr; }
            if (e->graph[v]) { hipGraphDestroy(e->graph[v]); e->graph[v] = nullptr; }
            e->graph_ok[v] = false;
        }
        e->graph_valid = true;
    }
    if (!e->graph_ok[variant]) {
        std::lock_guard<std::mutex> lock(g_capture_mutex);
        if (e->graph[variant]) { hipGraphDestroy(e->graph[variant]); e->graph[variant] = nullptr; }   // left by a failed attempt
        HIPCHK(hipStreamBeginCapture(e->own_stream, hipStreamCaptureModeThreadLocal));
        int rc = enqueue_fast_step(e, e->own_stream, variant);
        hipError_t ce = hipStreamEndCapture(e->own_stream, &e->graph[variant]);
        if (rc != WT_OK || ce != hipSuccess) {
            if (e->graph[variant]) { hipGraphDestroy(e->graph[variant]); e->graph[variant] = nullptr; }
            return rc != WT_OK ? rc : fail(WT_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
        }
        const hipError_t ie = hipGraphInstantiate(&e->graph_exec[variant], e->graph[variant], nullptr, nullptr, 0);
        if (ie != hipSuccess) {
            hipGraphDestroy(e->graph[variant]);
            e->graph[variant] = nullptr;
            e->graph_exec[variant] = nullptr;
            return fail(WT_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
        }
        e->graph_ok[variant] = true;
    }
    for (int i = 0; i < n_steps; ++i) {
        HIPCHK(hipGraphLaunch(e->graph_exec[variant], s));
        e->issued += 1;   // counted only once it is really in the queue: wt_decoder_run compares it with the steps the mailbox reports
    }
    return WT_OK;
}

// ------------------------------------------------------------------------------------------------- continuous decoding
// The reference transcribes one clip at a time, so every utterance stops at its own EOS (run.py:219-226; dataset loop cal_wer.py:249-287).
// A batch started by wt_decoder_begin runs to its LONGEST row.  The continuous mode keeps `slots` rows busy instead: utterances are
// SUBMITTED (their cross K/V projected into a free row of a cache pool, any number ahead of time), wait in a device queue, and the
// kernel that sees a row stop puts the next waiting utterance into that slot in the same step -- no finished row is ever stepped, no
// host round trip sits between an EOS and the next utterance's first token.  Finished utterances report through pinned host memory.
extern "C" int wt_decoder_stream_begin(wt_engine* e, int slots, int pool_rows, const wt_greedy_params* p, void* stream) {
    if (!e || e->kind != WT_KIND_DECODER) return fail(WT_E_INVALID, "wt_decoder_stream_begin: not a decoder engine");
    if (!p || slots < 1) return fail(WT_E_INVALID, "wt_decoder_stream_begin: bad arguments");
    if (pool_rows <= 0) pool_rows = 4 * slots;     // slots decoding + 3 x slots prepared ahead
    if (pool_rows < slots + 1 || pool_rows > STREAM_QCAP) return fail(WT_E_INVALID, "wt_decoder_stream_begin: pool of %d rows outside [slots + 1, %d]", pool_rows, STREAM_QCAP);
    if (p->logits_trace || p->force_eos_step >= 0 || p->force_eos_steps)
        return fail(WT_E_UNSUPPORTED, "wt_decoder_stream_begin: logits_trace / force_eos_step(s) belong to wt_decoder_begin (per-utterance lengths: wt_decoder_stream_submit)");
    DeviceGuard guard(e->device);
    HIPCHK(guard.err);
    hipStream_t s = (hipStream_t)stream;
    int rc = decode_setup(e, slots, p, pool_rows, true, s, "wt_decoder_stream_begin");
    if (rc) return rc;
    if (pool_rows > e->h_pool_cap || e->dec_maxlen_cap > e->h_ml_cap) {   // pinned report area: ids and final lengths per pool row
        HIPCHK(hipStreamSynchronize(s));   // no step of an earlier stream may still write the old area
        const int rows = pool_rows > e->h_pool_cap ? pool_rows : e->h_pool_cap;
        if (e->h_ids) { hipHostFree(e->h_ids); e->h_ids = nullptr; }
        if (e->h_len) { hipHostFree(e->h_len); e->h_len = nullptr; }
        e->h_pool_cap = 0;
        HIPCHK(hipHostMalloc((void**)&e->h_ids, (size_t)rows * e->dec_maxlen_cap * 4, hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(hipHostMalloc((void**)&e->h_len, (size_t)rows * 4, hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(hipHostGetDevicePointer((void**)&e->h_ids_dev, e->h_ids, 0));
        HIPCHK(hipHostGetDevicePointer((void**)&e->h_len_dev, e->h_len, 0));
        e->h_pool_cap = rows;
        e->h_ml_cap = e->dec_maxlen_cap;
        e->graph_valid = false;   // the step graph carries these addresses
    }
    e->pool_rows = pool_rows;
    e->pool_state.assign(pool_rows, 0);
    e->submitted = e->finished_seen = e->admitted_seen = 0;
    for (int r = 0; r < pool_rows; ++r) __atomic_store_n(e->h_len + r, 0, __ATOMIC_RELAXED);
    LAUNCH(launch_stream_init(e->st, e->unfinished, slots, e->epoch, s));
    e->begun = true;
    return WT_OK;
}

extern "C" int wt_decoder_stream_submit(wt_engine* e, const float* enc_hidden, int n, const int32_t* force_eos_steps, int32_t* handles,
                                        void* stream) {
    if (!e || e->kind != WT_KIND_DECODER || !e->begun || !e->stream_mode) return fail(WT_E_STATE, "wt_decoder_stream_submit: no stream open (wt_decoder_stream_begin)");
    if (!enc_hidden || !handles || n < 1 || n > MAX_ROWS) return fail(WT_E_INVALID, "wt_decoder_stream_submit: 1 .. %d utterances per call", MAX_ROWS);
    if (e->submitted - e->admitted_seen + n > STREAM_QCAP) return fail(WT_E_STATE, "wt_decoder_stream_submit: waiting queue full");
    DeviceGuard guard(e->device);
    HIPCHK(guard.err);
    hipStream_t s = (hipStream_t)stream;
    // pool rows for the n utterances: as few contiguous runs as the free list allows (one batched K/V projection per run and layer)
    std::vector<int> rows;
    for (int r = 0; r < e->pool_rows && (int)rows.size() < n; ++r)
        if (e->pool_state[r] == 0) rows.push_back(r);
    if ((int)rows.size() < n) return fail(WT_E_STATE, "wt_decoder_stream_submit: %d free cache rows for %d utterances (collect finished ones first)", (int)rows.size(), n);
    StreamPublish pub;
    memset(&pub, 0, sizeof pub);
    pub.n = n;
    for (int i = 0; i < n; ++i) {
        pub.rows[i] = rows[i];
        pub.force[i] = force_eos_steps ? force_eos_steps[i] : -1;
        handles[i] = rows[i];
    }
    for (int i = 0; i < n;) {
        int j = i + 1;
        while (j < n && rows[j] == rows[j - 1] + 1) ++j;
        const int rc = cross_kv_project(e, enc_hidden + (size_t)i * e->S * e->d, j - i, e->S, 0, e->cross_k, e->cross_v, e->kv_esz, s, e->cross_rows, rows[i]);
        if (rc) return rc;
        i = j;
    }
    for (int i = 0; i < n; ++i) {
        __atomic_store_n(e->h_len + rows[i], 0, __ATOMIC_RELAXED);   // (the row is free: no kernel writes it before the publish below)
        e->pool_state[rows[i]] = 1;
    }
    SelectParams sp;
    fill_select_params(e, sp);
    LAUNCH(launch_stream_publish(sp, pub, s));
    e->submitted += n;
    return WT_OK;
}

// newly finished utterances (their pinned length word has become non-zero): state 1 -> 2
static int stream_scan_finished(wt_engine* e) {
    int n = 0;
    for (int r = 0; r < e->pool_rows; ++r)
        if (e->pool_state[r] == 1 && __atomic_load_n(e->h_len + r, __ATOMIC_ACQUIRE) != 0) {
            e->pool_state[r] = 2;
            ++n;
        }
    e->finished_seen += n;
    return n;
}

// Steps until (a) every submitted utterance has finished, or (b) fewer than `min_waiting` submitted utterances are still waiting for a
// slot (min_waiting > 0: the caller has more to submit and wants the queue topped up).  Follows the mailbox like wt_decoder_run.
extern "C" int wt_decoder_stream_run(wt_engine* e, int min_waiting, int lookahead, int* n_finished, int* n_waiting, int* n_steps, void* stream) {
    if (!e || e->kind != WT_KIND_DECODER || !e->begun || !e->stream_mode) return fail(WT_E_STATE, "wt_decoder_stream_run: no stream open (wt_decoder_stream_begin)");
    DeviceGuard guard(e->device);
    HIPCHK(guard.err);
    hipStream_t s = (hipStream_t)stream;
    if (lookahead <= 0) lookahead = e->nt_loads ? 1 : 3;
    if (lookahead > 64) lookahead = 64;
    int spins = 0, slices = 0;
    unsigned long long last = ~0ull;
    for (;;) {
        const unsigned long long mb = __atomic_load_n(e->mailbox, __ATOMIC_ACQUIRE);
        const bool mine = (int)(mb >> 48) == e->epoch;
        const int retired = mine ? e->issued - (int)((unsigned)(e->issued - (int)((mb >> 32) & 0xffff)) & 0xffffu) : 0;
        if (mine) {   // admitted utterances: the word carries the low 15 bits of the device's count
            const int adm = (int)((mb >> 16) & 0x7fff);
            e->admitted_seen += (int)((unsigned)(adm - e->admitted_seen) & 0x7fffu);
            if (e->admitted_seen > e->submitted) e->admitted_seen = e->submitted;
        }
        if (mb != last) {
            last = mb;
            stream_scan_finished(e);
        }
        const int waiting_ub = e->submitted - e->admitted_seen;   // upper bound: admissions by the publish kernel show up with the next step's word
        if (e->finished_seen >= e->submitted) break;
        if (min_waiting > 0 && waiting_ub < min_waiting) break;
        if (e->issued - retired <= lookahead) {
            // idle slots with nothing waiting (the tail of the workload): the graph whose attention skips them
            const unsigned full = (1u << e->B) - 1u;
            const int variant = (mine && retired > 0 && ((unsigned)mb & 0xffffu) != full && waiting_ub == 0) ? 1 : 0;
            const int rc = enqueue_steps(e, 1, s, variant);
            if (rc) return rc;
            spins = slices = 0;
            continue;
        }
        if (spins < 2000) {
            ++spins;
            __builtin_ia32_pause();
            continue;
        }
        struct timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
        if (++slices % 64 == 0) {
            const hipError_t q = hipStreamQuery(s);
            if (q != hipSuccess && q != hipErrorNotReady) return fail(WT_E_HIP, "wt_decoder_stream_run: stream failed: %s", hipGetErrorString(q));
            if (q == hipSuccess && slices > 4096) {
                stream_scan_finished(e);
                if (e->finished_seen >= e->submitted) break;
                const unsigned long long m2 = __atomic_load_n(e->mailbox, __ATOMIC_ACQUIRE);
                if (m2 == mb) return fail(WT_E_STATE, "wt_decoder_stream_run: idle stream, %d of %d utterances finished", e->finished_seen, e->submitted);
            }
        }
    }
    if (n_finished) *n_finished = e->finished_seen;
    if (n_waiting) *n_waiting = e->submitted - e->admitted_seen;
    if (n_steps) *n_steps = e->issued;
    return WT_OK;
}

// Finished utterance `handle` (from wt_decoder_stream_submit): *len = its length (start token and EOS included), 0 while it is still
// waiting or decoding (ids_out untouched).  With ids_out != NULL a finished utterance's ids are copied to HOST memory (capacity `cap`
// >= max_length int32) and its cache row is released for the next submit.
extern "C" int wt_decoder_stream_collect(wt_engine* e, int handle, int32_t* ids_out, int cap, int* len) {
    if (!e || e->kind != WT_KIND_DECODER || !e->stream_mode) return fail(WT_E_STATE, "wt_decoder_stream_collect: no stream open");
    if (handle < 0 || handle >= e->pool_rows || !len) return fail(WT_E_INVALID, "wt_decoder_stream_collect: bad handle %d", handle);
    if (e->pool_state[handle] == 0) return fail(WT_E_STATE, "wt_decoder_stream_collect: row %d holds no utterance", handle);
    const int n = __atomic_load_n(e->h_len + handle, __ATOMIC_ACQUIRE);
    *len = n;
    if (n == 0 || !ids_out) return WT_OK;
    if (cap < n) return fail(WT_E_INVALID, "wt_decoder_stream_collect: %d ids, capacity %d", n, cap);
    if (e->pool_state[handle] == 1) { e->pool_state[handle] = 2; e->finished_seen += 1; }
    memcpy(ids_out, e->h_ids + (size_t)handle * e->max_length, (size_t)n * 4);   // row stride of stream_finish_kernel: this stream's max_length
    e->pool_state[handle] = 0;
    return WT_OK;
}

extern "C" int wt_decoder_steps(wt_engine* e, int n_steps, void* stream) {
    if (!e || e->kind != WT_KIND_DECODER) return fail(WT_E_INVALID, "wt_decoder_steps: not a decoder engine");
    if (!e->begun) return fail(WT_E_STATE, "wt_decoder_steps called before wt_decoder_begin");
    if (n_steps <= 0) return WT_OK;
    DeviceGuard guard(e->device);
    HIPCHK(guard.err);
    return enqueue_steps(e, n_steps, (hipStream_t)stream);
}

// Run the decode in flight to its stop test WITHOUT synchronising the stream per chunk: greedy_finish_kernel reports
// (epoch, steps retired, done, length, unfinished rows) into a pinned host word after every step, and this loop keeps exactly
// `lookahead` steps queued behind the one the GPU is running.  The GPU never waits for the host (the next step is already in its
// queue when one retires) and at most `lookahead` steps are enqueued past the stop -- they are no-ops for the token state but
// stream the weights, which is what the old "8 steps, then hipStreamSynchronize" loop paid up to 7 times per utterance batch.
extern "C" int wt_decoder_run(wt_engine* e, int lookahead, int* cur_len, int* n_unfinished, void* stream) {
    if (!e || e->kind != WT_KIND_DECODER) return fail(WT_E_INVALID, "wt_decoder_run: not a decoder engine");
    if (!e->begun) return fail(WT_E_STATE, "wt_decoder_run called before wt_decoder_begin");
    if (e->stream_mode) return fail(WT_E_STATE, "wt_decoder_run: a continuous decode is open on this handle (wt_decoder_stream_run)");
    DeviceGuard guard(e->device);
    HIPCHK(guard.err);
    hipStream_t s = (hipStream_t)stream;
    if (lookahead <= 0) lookahead = e->nt_loads ? 1 : 3;   // small models: a step is 0.15-0.3 ms, keep more of them queued
    if (lookahead > 64) lookahead = 64;
    const int max_steps = e->max_length - 1;               // MaxLengthCriteria fires at the latest after this many steps
    unsigned long long mb = 0;
    int spins = 0, slices = 0;   // since the last launch: busy-wait iterations, then 20 us sleep slices
    for (;;) {
        mb = __atomic_load_n(e->mailbox, __ATOMIC_ACQUIRE);
        const bool mine = (int)(mb >> 48) == e->epoch;
        // the word carries the low 16 bits of the retired count: compare modulo 2^16 (a caller may have enqueued thousands of no-op
        // steps through wt_decoder_steps before handing over to this loop)
        const int retired = mine ? e->issued - (int)((unsigned)(e->issued - (int)((mb >> 32) & 0xffff)) & 0xffffu) : 0;
        if (mine && ((mb >> 31) & 1)) break;
        if (e->issued < max_steps && e->issued - retired <= lookahead) {
            // a row has finished (the word's unfinished mask lost a bit): from here on replay the graph whose attention skips such rows
            const unsigned full = (1u << e->B) - 1u;
            const int variant = (mine && retired > 0 && ((unsigned)mb & 0xffffu) != full) ? 1 : 0;
            int rc = enqueue_steps(e, 1, s, variant);
            if (rc) return rc;
            spins = slices = 0;
            continue;
        }
        if (e->issued >= max_steps && retired >= e->issued) break;   // every possible step has retired (done is set with the last)
        // Wait for the word: spin briefly (it often lands within microseconds of a check), then sleep in 20 us slices -- a whole step
        // (0.15-1.5 ms) is queued behind the running one, so the host has that long to enqueue the next, and N workers per GPU
        // (runtime.WhisperPipeline) x 8 ranks need not burn N x 8 host cores.
        if (spins < 2000) {
            ++spins;
            __builtin_ia32_pause();
            continue;
        }
        struct timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
        if (++slices % 64 == 0) {   // every few ms of waiting: is the stream still alive?
            const hipError_t q = hipStreamQuery(s);
            if (q != hipSuccess && q != hipErrorNotReady) return fail(WT_E_HIP, "wt_decoder_run: stream failed: %s", hipGetErrorString(q));
            if (q == hipSuccess && slices > 4096) {   // an idle stream for > 0.1 s and the steps still unreported: they died without a word
                mb = __atomic_load_n(e->mailbox, __ATOMIC_ACQUIRE);
                const int r2 = (int)(mb >> 48) == e->epoch ? e->issued - (int)((unsigned)(e->issued - (int)((mb >> 32) & 0xffff)) & 0xffffu) : 0;
                if (r2 < e->issued && !((mb >> 31) & 1)) return fail(WT_E_STATE, "wt_decoder_run: %d steps enqueued, %d reported", e->issued, r2);
            }
        }
    }
    if (cur_len) *cur_len = (int)((mb >> 16) & 0x7fff);
    if (n_unfinished) *n_unfinished = __builtin_popcountll(mb & 0xffff);
    for (EvTimer* t : {&e->t_cross, &e->t_gemm, &e->t_enc_attn, &e->t_skinny}) timer_collect(*t);
    return WT_OK;
}

extern "C" int wt_decoder_poll(wt_engine* e, int* cur_len, int* n_unfinished, int* done, void* stream) {
    if (!e || e->kind != WT_KIND_DECODER || !e->begun) return fail(WT_E_STATE, "wt_decoder_poll: no decode in flight");
    DeviceGuard guard(e->device);
    HIPCHK(guard.err);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(e->h_state, e->st, sizeof(DecState), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (cur_len) *cur_len = e->h_state->cur_len[0];
    if (n_unfinished) *n_unfinished = e->h_state->n_unfinished;
    if (done) *done = e->h_state->done;
    for (EvTimer* t : {&e->t_cross, &e->t_gemm, &e->t_enc_attn, &e->t_skinny}) timer_collect(*t);
    return WT_OK;
}

extern "C" int wt_decoder_read_ids(wt_engine* e, int32_t* ids_out, int ld, void* stream) {
    if (!e || e->kind != WT_KIND_DECODER || !e->begun) return fail(WT_E_STATE, "wt_decoder_read_ids: no decode in flight");
    if (!ids_out || ld < e->max_length) return fail(WT_E_INVALID, "wt_decoder_read_ids: row stride %d < max_length %d", ld, e->max_length);
    DeviceGuard guard(e->device);
    HIPCHK(guard.err);
    HIPCHK(hipMemcpy2DAsync(ids_out, (size_t)ld * 4, e->ids, (size_t)e->max_length * 4, (size_t)e->max_length * 4, e->B,
                            hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return WT_OK;
}

extern "C" int wt_decoder_greedy(wt_engine* e, const float* enc_hidden, int B, const wt_greedy_params* p, int32_t* ids_out,
                                 int* out_len, void* stream) {
    int rc = wt_decoder_begin(e, enc_hidden, B, p, stream);
    if (rc) return rc;
    int cur = 1, nu = B;
    if ((rc = wt_decoder_run(e, 0, &cur, &nu, stream))) return rc;
    if (ids_out && (rc = wt_decoder_read_ids(e, ids_out, p->max_length, stream))) return rc;
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (out_len) *out_len = cur;
    return WT_OK;
}

// ------------------------------------------------------------------------------------------------- Session surface
extern "C" int wt_engine_infer_shapes(wt_engine* e, const wt_tensor_desc* in, int n_in, wt_tensor_desc* out, int* n_out) {
    if (!e) return fail(WT_E_INVALID, "wt_engine_infer_shapes: null argument");
    EngineDims dims;
    dims.kind = e->kind; dims.precision = e->precision; dims.d = e->d; dims.H = e->H; dims.L = e->L; dims.F = e->F;
    dims.C = e->C; dims.S = e->S; dims.T = e->T; dims.V = e->V;
    ShapeState st;
    char msg[400] = "";
    const int rc = infer_shapes(dims, in, n_in, out, n_out, &st, msg, sizeof msg);  // host_logic.cpp
    e->shapes_ok = st.ok;
    if (rc) return fail(rc, "%s", msg);
    e->c_B = st.c_B; e->c_s = st.c_s; e->c_ms = st.c_ms; e->c_mc = st.c_mc;
    return WT_OK;
}

extern "C" int wt_engine_run(wt_engine* e, const wt_binding* in, int n_in, const wt_binding* out, int n_out, void* stream) {
    if (!e) return fail(WT_E_INVALID, "wt_engine_run: null engine");
    if (!e->shapes_ok) return fail(WT_E_STATE, "wt_engine_run: call wt_engine_infer_shapes with the current input shapes first");
    auto get = [&](const wt_binding* arr, int n, const char* name) -> void* {
        for (int i = 0; i < n; ++i)
            if (arr[i].name && strcmp(arr[i].name, name) == 0) return arr[i].ptr;
        return nullptr;
    };
    hipStream_t s = (hipStream_t)stream;
    if (e->kind == WT_KIND_ENCODER) {
        const float* data = (const float*)get(in, n_in, "data");
        float* hs = (float*)get(out, n_out, "hidden_states");
        if (!data || !hs) return fail(WT_E_INVALID, "encoder run needs bindings 'data' and 'hidden_states'");
        return wt_encoder_forward(e, data, e->c_B, hs, stream);
    }
    // ---- decoder, by-value cache protocol of run.py:103-148 / model.py:407-470 (batch 1) ----
    const int* data = (const int*)get(in, n_in, "data");
    const float* enc = (const float*)get(in, n_in, "encoder_hidden_states");
    const float* spk = (const float*)get(in, n_in, "self_past_key");
    const float* spv = (const float*)get(in, n_in, "self_past_value");
    const float* cpk = (const float*)get(in, n_in, "cross_past_key");
    const float* cpv = (const float*)get(in, n_in, "cross_past_value");
    float* logits = (float*)get(out, n_out, "hidden_states");
    float* nsk = (float*)get(out, n_out, "next_self_keys");
    float* nsv = (float*)get(out, n_out, "next_self_values");
    float* nck = (float*)get(out, n_out, "next_cross_keys");
    float* ncv = (float*)get(out, n_out, "next_cross_values");
    if (!data || !enc || !spk || !spv || !cpk || !cpv || !logits || !nsk || !nsv || !nck || !ncv)
        return fail(WT_E_INVALID, "decoder run is missing a binding (need 7 input tensors besides the two masks and 5 outputs)");
    DeviceGuard guard(e->device);
    HIPCHK(guard.err);
    int rc = dec_reserve(e, 1, e->T, (hipStream_t)stream);
    if (rc) return rc;
    const int LH = e->L * e->H, S = e->S;
    const int cache_len = e->c_ms - 1 < e->c_s ? e->c_ms - 1 : e->c_s;  // model.py:278
    const int cross_len = e->c_mc - 1;                                    // model.py:264
    // next_* start as copies of the used part of past_* (the reference's concat), then the step runs in place on them
    LAUNCH(launch_copy_cache_rows(spk, nsk, LH, e->c_s, cache_len + 1, cache_len, s));
    LAUNCH(launch_copy_cache_rows(spv, nsv, LH, e->c_s, cache_len + 1, cache_len, s));
    LAUNCH(launch_copy_cache_rows(cpk, nck, LH, S, S, cross_len, s));
    LAUNCH(launch_copy_cache_rows(cpv, ncv, LH, S, S, cross_len, s));
    if (cross_len < S) {  // cur = proj(enc[0 : S - cross_len]) lands behind the kept rows (model.py:265-272)
        rc = cross_kv_project(e, enc, 1, S - cross_len, cross_len, nck, ncv, 4, s);   // Session I/O caches are f32 in every engine (model.py:464-468)
        if (rc) return rc;
    }
    LAUNCH(launch_set_state(e->st, /*cur_len=*/1, /*pos=*/e->c_ms - 1, /*self_len=*/cache_len, s));
    StepIO io;
    io.ids = data; io.ids_ld = 1; io.self_k = nsk; io.self_v = nsv; io.self_cap = cache_len + 1;
    io.cross_k = nck; io.cross_v = ncv; io.cross_rows = 1; io.slot_row = nullptr; io.logits = logits; io.B = 1; io.kv_esz = 4;
    io.nsplit_self = 1; io.nsplit_cross = pick_splits(1, e->H, S);
    io.alive = nullptr;
    io.embed = true;
    io.argmax = nullptr; io.argmax_parts = nullptr;
    e->begun = false;  // the resident greedy state is clobbered by this call
    return enqueue_step(e, io, s);
}

// ------------------------------------------------------------------------------------------------- profiling
// Replays graph `g` (captured by the caller; consumed here) `iters` times between two hipEvents on `s`; every exit path releases
// the graph, its executable and the events.
static int replay_and_time(hipGraph_t g, int iters, int launches_per_replay, float* avg_us, hipStream_t s) {
    hipGraphExec_t ge = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    float ms = 0.f;
    hipError_t he;
    {
        std::lock_guard<std::mutex> lock(g_capture_mutex);
        he = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    }
    if (he == hipSuccess) he = hipEventCreate(&a);
    if (he == hipSuccess) he = hipEventCreate(&b);
    if (he == hipSuccess) he = hipGraphLaunch(ge, s);  // warm-up replay (replayed on the caller's stream, like the decode step)
    if (he == hipSuccess) he = hipEventRecord(a, s);
    for (int i = 0; i < iters && he == hipSuccess; ++i) he = hipGraphLaunch(ge, s);
    if (he == hipSuccess) he = hipEventRecord(b, s);
    if (he == hipSuccess) he = hipEventSynchronize(b);
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, a, b);
    if (a) hipEventDestroy(a);
    if (b) hipEventDestroy(b);
    if (ge) hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
    if (he != hipSuccess) return fail(WT_E_HIP, "timing replay failed: %s", hipGetErrorString(he));
    *avg_us = ms * 1e3f / ((float)iters * launches_per_replay);
    return WT_OK;
}

extern "C" int wt_decoder_time_cross_attention(wt_engine* e, int iters, float* avg_us, void* stream) {
    return wt_decoder_time_kernel(e, "cross_attn", iters, avg_us, stream);
}

// Average launch time (us) of ONE kind of decode-step launch (LP_* above), measured the way the decode step runs it: the L
// per-layer launches of that kind (each layer's own weights / caches, so nothing is cache-resident that would not be in the real
// step) are captured into a hipGraph and replayed `iters` times between two hipEvents recorded on the launch stream.  (Events
// cannot bracket single kernels inside the step graph; an eager pass adds dispatch gaps.)  The residual stream is not advanced:
// every launch reads the current buffers, which is what its duration depends on.  Only to be called BETWEEN steps of a decode:
// the launches scribble on the step scratch (dq, datt, dh2, dffn, the attention partials and the self-cache row at self_len),
// all of which the next step rewrites before reading.
extern "C" int wt_decoder_time_kernel(wt_engine* e, const char* which, int iters, float* avg_us, void* stream) {
    if (!e || e->kind != WT_KIND_DECODER || !which || !avg_us || iters < 1) return fail(WT_E_INVALID, "wt_decoder_time_kernel: bad arguments");
    if (!e->begun) return fail(WT_E_STATE, "wt_decoder_time_kernel needs a decode in flight (wt_decoder_begin)");
    static const char* names[LP_COUNT] = {"qkv", "self_attn", "pair", "cross_attn", "cross_out", "fc1", "fc2"};
    int part = -1;
    for (int i = 0; i < LP_COUNT; ++i)
        if (!strcmp(which, names[i])) part = i;
    if (part < 0) return fail(WT_E_NOTFOUND, "unknown decode kernel '%s' (qkv, self_attn, pair, cross_attn, cross_out, fc1, fc2)", which);
    DeviceGuard guard(e->device);
    HIPCHK(guard.err);
    hipStream_t s = (hipStream_t)stream;
    StepIO io;
    io.ids = e->ids; io.ids_ld = e->max_length; io.self_k = e->self_k; io.self_v = e->self_v; io.self_cap = e->T;
    io.cross_k = e->cross_k; io.cross_v = e->cross_v; io.cross_rows = e->cross_rows; io.logits = e->logits; io.B = e->B; io.kv_esz = e->kv_esz;
    io.nsplit_self = e->nsplit_self; io.nsplit_cross = e->nsplit_cross; io.embed = false; io.argmax = nullptr; io.argmax_parts = nullptr;
    io.alive = nullptr; io.slot_row = nullptr;
    struct ProfilingOff {   // no event records inside the capture; the caller's setting comes back on every exit path
        wt_engine* e; bool was;
        explicit ProfilingOff(wt_engine* e_) : e(e_), was(e_->profiling) { e->profiling = false; }
        ~ProfilingOff() { e->profiling = was; }
    } profiling_off(e);
    hipGraph_t g = nullptr;
    {   // captures / instantiations are serialised process-wide (enqueue_steps)
        std::lock_guard<std::mutex> lock(g_capture_mutex);
        HIPCHK(hipStreamBeginCapture(e->own_stream, hipStreamCaptureModeThreadLocal));
        int rc = WT_OK;
        for (int i = 0; i < e->L && rc == WT_OK; ++i) rc = enqueue_layer_part(e, io, i, e->dh, e->dh2, part, e->own_stream);
        hipError_t ce = hipStreamEndCapture(e->own_stream, &g);
        if (rc != WT_OK || ce != hipSuccess) {
            if (g) hipGraphDestroy(g);
            return fail(WT_E_HIP, "capturing the timing graph of '%s' failed", which);
        }
    }
    return replay_and_time(g, iters, e->L, avg_us, s);
}

extern "C" int wt_engine_set_profiling(wt_engine* e, int enabled) {
    if (!e) return fail(WT_E_INVALID, "null engine");
    e->profiling = enabled != 0;
    for (EvTimer* t : {&e->t_cross, &e->t_gemm, &e->t_enc_attn, &e->t_skinny}) {
        timer_collect(*t);
        t->ms = 0.0;
        t->launches = 0;
    }
    return WT_OK;
}
extern "C" int wt_engine_get_timer(wt_engine* e, const char* which, wt_kernel_timer* out) {
    if (!e || !which || !out) return fail(WT_E_INVALID, "null argument");
    EvTimer* t = nullptr;
    if (!strcmp(which, "dec_cross_attn")) t = &e->t_cross;
    else if (!strcmp(which, "gemm_f32")) t = &e->t_gemm;
    else if (!strcmp(which, "enc_attn")) t = &e->t_enc_attn;
    else if (!strcmp(which, "vocab_proj")) t = &e->t_skinny;
    else return fail(WT_E_NOTFOUND, "unknown timer '%s'", which);
    timer_collect(*t);
    out->ms_total = (float)t->ms;
    out->launches = t->launches;
    return WT_OK;
}
