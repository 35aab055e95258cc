/*
 * whisper_trtllm_amd.h — C-ABI of the MI355X-native Whisper encoder/decoder engine.
 *
 * This is the drop-in boundary for the hot path of EdVince/whisper-trtllm
 * (tensorrt_llm/models/whisper/model.py + examples/whisper/run.py).  The reference crosses the
 * process→device boundary through TensorRT's execution context, wrapped by
 *   tensorrt_llm/runtime/session.py:54   Session.from_serialized_engine(bytes)
 *   tensorrt_llm/runtime/session.py:116  Session.infer_shapes(List[TensorInfo]) -> List[TensorInfo] | None
 *   tensorrt_llm/runtime/session.py:148  Session.run(inputs, outputs, stream) -> bool   (async enqueue)
 * and its only native C symbol is `initLibNvInferPlugins` (cpp/tensorrt_llm/plugins/api/InferPlugin.cpp:151),
 * loaded with ctypes.CDLL in tensorrt_llm/plugin/plugin.py:10-22.  TensorRT does not exist on ROCm, so the
 * C-ABI below is shaped like Session: an opaque engine handle built from a serialized blob, shape inference
 * by tensor name, and a stream-ordered run with name -> device-pointer bindings.  Entry points (5)-(9) add
 * the batched in-place-KV fast path that replaces run.py's per-token Session.run + clone loop.
 *
 * Conventions
 *   - every function returns 0 on success or a negative WT_E_* code; nothing throws across the ABI;
 *     `wt_last_error()` returns a thread-local human-readable message for the last failure.
 *   - the caller owns every I/O buffer (device pointers, e.g. torch tensors); the engine owns weights,
 *     workspace and the resident KV cache.  No torch types appear in any signature.
 *   - calls are stream-ordered on the `hipStream_t` passed as `void* stream`; only wt_decoder_poll and
 *     wt_decoder_greedy synchronise (documented below).  First use with a new batch size allocates workspace.
 *   - every call runs on its handle's device and restores the caller's current device before returning.
 *   - a handle is NOT thread-safe (matches one IExecutionContext per Session, session.py:48); distinct handles --
 *     on different devices or several on ONE device -- may be driven concurrently from different host threads
 *     (the Python side's WhisperPipeline runs N engine pairs per GPU that way; the library's lazily latched
 *     state is atomic / initialised once).
 */
#ifndef WHISPER_TRTLLM_AMD_H
#define WHISPER_TRTLLM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WT_ABI_VERSION 3

/* error codes */
#define WT_OK 0
#define WT_E_INVALID (-22)   /* bad argument / malformed blob / shape or dtype mismatch */
#define WT_E_NOMEM (-12)     /* device or host allocation failed */
#define WT_E_NOTFOUND (-2)   /* unknown tensor name */
#define WT_E_HIP (-5)        /* a HIP runtime call failed; message carries hipGetErrorString */
#define WT_E_UNSUPPORTED (-38)
#define WT_E_STATE (-1)      /* call sequence violated (e.g. steps before begin) */

/* tensor element types (numbering is ours; the Python shim maps trt.float32 etc. onto it) */
typedef enum { WT_F32 = 0, WT_F16 = 1, WT_I32 = 2, WT_I8 = 3 } wt_dtype;
typedef enum { WT_KIND_ENCODER = 1, WT_KIND_DECODER = 2 } wt_engine_kind;

#define WT_MAX_DIMS 6
#define WT_NAME_LEN 48

/* == TensorInfo(name, dtype, shape) of session.py:28-33 */
typedef struct {
    char name[WT_NAME_LEN];
    int32_t dtype; /* wt_dtype */
    int32_t ndim;
    int64_t shape[WT_MAX_DIMS];
} wt_tensor_desc;

/* one entry of the inputs/outputs dict of Session.run (session.py:166-176): name -> device pointer */
typedef struct {
    const char* name;
    void* ptr;
} wt_binding;

typedef struct wt_engine wt_engine;

/* model hyper-parameters baked into the blob (HF config keys, build_encoder.py:48-56 / build_decoder.py:45-56) */
typedef struct {
    int32_t kind;      /* wt_engine_kind */
    int32_t precision; /* wt_dtype of the GEMM / GEMV weight operands: WT_F32, or WT_F16 (--engine_precision float16, builder.py:55;
                          accumulation, LayerNorm, softmax, residual stream and every Session-visible tensor stay fp32) */
    int32_t d_model, n_heads, n_layers, ffn_dim;
    int32_t n_mels, max_source_positions, max_target_positions, vocab_size;
} wt_engine_info;

/* (1) replaces Session.from_serialized_engine (session.py:54): parse the blob, upload weights to `device`.
 * Blobs are versioned (engine_pack.py VERSION); a blob written by another version is refused with WT_E_UNSUPPORTED. */
int wt_engine_open(const void* blob, size_t nbytes, int device, wt_engine** out);
/* (1b) a second handle on the SAME weights (TensorRT: one ICudaEngine, several IExecutionContexts -- session.py:48 creates one per
 * Session): the read-only weight payload is shared and reference-counted, workspace / resident caches / step graphs are the new
 * handle's own.  Lets N host threads drive N decodes on one device (distinct handles may run concurrently) at the cost of one copy
 * of the weights.  The payload is freed when the last handle sharing it is closed, in any order. */
int wt_engine_clone(const wt_engine* src, wt_engine** out);
/* (2) engine teardown (TensorRT: ICudaEngine/IExecutionContext destructors). */
void wt_engine_close(wt_engine* e);
int wt_engine_get_info(const wt_engine* e, wt_engine_info* out);
/* How an fp32 engine forms its large matrix products: 0 = v_mfma_f32_32x32x2_f32 (the fp32 matrix instruction), 1 = "x3": every
 * fp32 operand exactly split into three bf16 values (x = b1 + b2 + b3 to 2^-27 relative), the product accumulated in fp32 from the six
 * partial products of order <= 2^-18 on v_mfma_f32_32x32x16_bf16 -- as accurate as the fp32 instruction (tests/test_gpu_kernels.py::
 * test_gemm_x3_is_an_fp32_gemm) at 16 / 6 = 2.7x its peak.  The default for fp32 engines; fp16 engines report 0. */
int wt_engine_gemm_mode(const wt_engine* e);

/* (3) replaces Session.infer_shapes (session.py:116-146): check names/dtypes of the inputs, remember the
 * input shapes for the next run, and report the outputs.  `*n_out` is in: capacity of `out`, out: count.
 * Encoder  in : data f32 [B,n_mels,2*S], length f32 [B] (ignored)            out: hidden_states f32 [B,S,d]
 * Decoder  in : data i32 [1,1], length i32 [1] (ignored), encoder_hidden_states f32 [1,S,d],
 *               self_past_key/value f32 [L,H,s,64], cross_past_key/value f32 [L,H,S,64],
 *               past_self_cache_mask f32 [m_s], past_cross_cache_mask f32 [m_c]  (only the LENGTHS are read)
 *          out: hidden_states f32 [1,1,V] (logits), next_self_keys/values f32 [L,H,min(m_s-1,s)+1,64],
 *               next_cross_keys/values f32 [L,H,S,64]            (model.py:474-516, :459-468) */
int wt_engine_infer_shapes(wt_engine* e, const wt_tensor_desc* in, int n_in, wt_tensor_desc* out, int* n_out);

/* (4) replaces Session.run (session.py:148-178) == context.execute_async_v3(stream): enqueue one engine
 * execution with the shapes of the last wt_engine_infer_shapes call.  Asynchronous: 0 != finished. */
int wt_engine_run(wt_engine* e, const wt_binding* in, int n_in, const wt_binding* out, int n_out, void* stream);

/* (5) batched encoder: mel f32 [batch, n_mels, 2*S] -> hidden f32 [batch, S, d].  Asynchronous. */
int wt_encoder_forward(wt_engine* enc, const float* mel, int batch, float* hidden_out, void* stream);

/* greedy-search rules == get_logits_processor / get_stopping_criteria of run.py:150-169 */
typedef struct {
    int32_t decoder_start_token_id;
    int32_t eos_token_id;
    int32_t pad_token_id;
    int32_t max_length;               /* MaxLengthCriteria: stop when len >= max_length */
    int32_t begin_index;              /* SuppressTokensAtBeginLogitsProcessor.begin_index */
    const int32_t* suppress_tokens;   /* host pointers, copied at begin */
    int32_t n_suppress_tokens;
    const int32_t* begin_suppress_tokens;
    int32_t n_begin_suppress_tokens;
    const int32_t* forced_decoder_ids; /* n_forced pairs (generation index, token id) */
    int32_t n_forced;
    int32_t force_eos_step;           /* bench only: emit EOS at this 0-based step for every row; -1 = off */
    float* logits_trace;              /* optional device buffer f32 [batch, max_length-1, V] of raw logits, or NULL */
    const int32_t* force_eos_steps;   /* bench only: host array [batch], row b emits EOS at 0-based step force_eos_steps[b]
                                         (< 0: never) -- the variable-length workload of bench.py; NULL = off */
} wt_greedy_params;

/* (6) start a greedy decode of `batch` utterances (1 <= batch <= 16 per call, WT_E_UNSUPPORTED above; shard larger
 * batches over calls or GPUs): project the encoder memory f32 [batch,S,d] into the
 * resident cross-KV cache, reset the self-KV cache and the id buffer to [[decoder_start_token_id]]*batch.
 * Replaces greedy_search() step 0 (run.py:171-197 with past_key_values=None).  An fp16 engine keeps these RESIDENT caches in fp16
 * (they never leave the engine; the by-value caches of wt_engine_run stay f32 like the reference's, model.py:464-468).
 * Asynchronous when the token rules
 * (suppress / begin-suppress / forced lists) equal those of the previous decode on this handle; a changed rule set is
 * uploaded and the stream synchronised once. */
int wt_decoder_begin(wt_engine* dec, const float* enc_hidden, int batch, const wt_greedy_params* p, void* stream);
/* (7) enqueue `n_steps` decoder steps (token embed -> L layers -> vocab projection -> logits processors ->
 * argmax -> pad/EOS bookkeeping -> append), all on device.  Steps after every row finished are no-ops.
 * Replaces the body of run.py:195-217.  Asynchronous. */
int wt_decoder_steps(wt_engine* dec, int n_steps, void* stream);
/* (8) synchronise `stream` and report progress: current sequence length (prompt included), number of
 * unfinished rows, and whether the stop test of run.py:219-226 has fired. */
int wt_decoder_poll(wt_engine* dec, int* cur_len, int* n_unfinished, int* done, void* stream);
/* (8b) run the decode in flight to its stop test (run.py:195-226): enqueues steps one at a time, keeping `lookahead` steps queued
 * behind the one the GPU is executing (0 = engine default: 1 for the large models, 3 when a step is < ~0.3 ms), and follows the
 * progress through a pinned host word the last kernel of every step writes -- no stream synchronisation, no idle GPU between
 * chunks, and at most `lookahead` steps enqueued past the stop (they are no-ops for the token state).  Returns once the stop test
 * has been OBSERVED; the surplus steps may still be draining, so follow it with stream-ordered calls (wt_decoder_read_ids).
 * `*cur_len` = final sequence length (prompt included), `*n_unfinished` = rows that never produced EOS. */
int wt_decoder_run(wt_engine* dec, int lookahead, int* cur_len, int* n_unfinished, void* stream);
/* copy the generated ids (int32 [batch, cur_len], row-major) into a DEVICE buffer of capacity
 * batch*max_length int32; stream-ordered. */
int wt_decoder_read_ids(wt_engine* dec, int32_t* ids_out, int ld, void* stream);
/* (9) convenience: begin + wt_decoder_run; writes ids int32 [batch, max_length] (row stride
 * max_length) to DEVICE memory and the final length to *out_len.  Synchronises the stream. */
int wt_decoder_greedy(wt_engine* dec, const float* enc_hidden, int batch, const wt_greedy_params* p,
                      int32_t* ids_out, int* out_len, void* stream);

/* ---- (10) continuous decoding: every utterance stops at its own EOS and its decode slot is refilled at once.
 * The reference gets per-utterance stopping from transcribing one clip at a time (examples/whisper/run.py:219-226; dataset loop
 * cal_wer.py:249-287); a batch started by wt_decoder_begin runs to its longest row.  Here `slots` rows (1..16) stay busy: utterances
 * are submitted ahead of time (their cross K/V projected into a free row of a cache POOL of `pool_rows` rows, <= 256; 0 = 4 x slots),
 * wait in a device-side queue, and the kernel that sees a row emit EOS (or reach max_length) puts the next waiting utterance into that
 * slot within the same step.  Ids of every utterance equal those of the same utterance decoded alone.  Finished utterances are
 * reported through pinned host memory: no device-to-host copy, no stream synchronisation.  logits_trace / force_eos_step(s) of `p`
 * must be unset.  One stream per handle; wt_decoder_begin on the handle ends it. */
int wt_decoder_stream_begin(wt_engine* dec, int slots, int pool_rows, const wt_greedy_params* p, void* stream);
/* submit `n` (1..16) more utterances: enc_hidden f32 [n,S,d] on the device (consumed stream-ordered by this call's K/V projection).
 * `force_eos_steps` (bench only, NULL = off): utterance i emits EOS at its own 0-based step force_eos_steps[i].  handles[i] receives
 * the handle to collect utterance i with.  WT_E_STATE when fewer than n cache rows are free (collect finished utterances first).
 * Utterances are admitted to slots in submission order.  Asynchronous. */
int wt_decoder_stream_submit(wt_engine* dec, const float* enc_hidden, int n, const int32_t* force_eos_steps, int32_t* handles, void* stream);
/* enqueue decoder steps (as wt_decoder_run: `lookahead` steps queued behind the running one, progress followed through the pinned
 * mailbox) until every submitted utterance has finished -- or, with min_waiting > 0, until fewer than `min_waiting` submitted
 * utterances are still waiting for a slot, so that the caller can submit more before a slot runs dry.  *n_finished = utterances
 * finished since wt_decoder_stream_begin, *n_waiting = an upper bound of those still waiting, *n_steps = decoder steps enqueued since
 * wt_decoder_stream_begin (each is `slots` row-steps: the denominator of the slot utilisation).  Any of the three may be NULL. */
int wt_decoder_stream_run(wt_engine* dec, int min_waiting, int lookahead, int* n_finished, int* n_waiting, int* n_steps, void* stream);
/* *len = length of utterance `handle` (start token and EOS included) once it has finished, 0 while it waits or decodes.  With ids_out
 * != NULL (HOST memory, capacity `cap` >= *len int32) a finished utterance's ids are copied out and its cache row is released. */
int wt_decoder_stream_collect(wt_engine* dec, int handle, int32_t* ids_out, int cap, int* len);

/* per-phase device timers (hipEvents on the caller's stream) for bench.py's roofline block */
typedef struct {
    float ms_total;    /* sum over launches of the timed kernel since the last reset */
    int64_t launches;
} wt_kernel_timer;
/* enable/disable event timing of the dominant decode kernel (cross-attention) and the encoder GEMM */
int wt_engine_set_profiling(wt_engine* e, int enabled);
int wt_engine_get_timer(wt_engine* e, const char* which, wt_kernel_timer* out);
/* average launch time (us) of the dominant decode kernel, cross-attention, over the engine's resident caches:
 * the L per-layer launches are captured into a hipGraph (as the decode step runs them), replayed `iters` times
 * between two hipEvents on the launch stream.  Needs a decode in flight (wt_decoder_begin).  Synchronises. */
int wt_decoder_time_cross_attention(wt_engine* dec, int iters, float* avg_us, void* stream);
/* the same for any of the seven per-layer launches of a decode step: which = "qkv" (LN + q|k|v GEMV + KV append), "self_attn",
 * "pair" (self out-projection + folded cross query), "cross_attn", "cross_out" (split merge + cross out-projection), "fc1", "fc2".
 * The residual stream is not advanced (every launch reads the buffers as they stand).  Synchronises.  Call it only BETWEEN
 * steps of a decode (after a wt_decoder_poll): the launches overwrite the step scratch (query, attention context, FFN buffer,
 * second residual buffer, split partials and the self-cache row at the current length) -- all of which the next step
 * rewrites before it reads them, so the decode itself is not disturbed. */
int wt_decoder_time_kernel(wt_engine* dec, const char* which, int iters, float* avg_us, void* stream);

/* ---- log-mel front-end: replaces the CPU numpy STFT inside the reference's timed loop,
 * `hf_processor(sample["array"], ...)` run.py:267 == WhisperFeatureExtractor._np_extract_fbank_features
 * (transformers/models/whisper/feature_extraction_whisper.py:94-111).  The host passes the Hann window as fp64 [n_fft] (the
 * reference frames in float64, audio_utils.py:399-401) and the mel filter bank f32 [n_mels][npw] (columns >= n_bins zero, npw =
 * n_bins rounded up to a multiple of 4); the fp64 DFT matrix is built inside.  Windowed DFT and power spectrum run in fp64
 * (v_mfma_f64_16x16x4_f64), like the reference's float64 rfft; features agree with it to ~1e-6. */
typedef struct wt_logmel wt_logmel;
int wt_logmel_create(int device, int n_fft, int hop, int n_mels, int n_frames, const double* window, const float* filters, int npw,
                     wt_logmel** out);
void wt_logmel_destroy(wt_logmel* h);
/* audio f32 [batch][n_in] on the device (n_in samples each; shorter than 30 s = zero-padded, longer = trimmed)
 * -> mel_out f32 [batch][n_mels][n_frames].  Asynchronous on `stream`. */
int wt_logmel_forward(wt_logmel* h, const float* audio, int batch, int n_in, float* mel_out, void* stream);

const char* wt_last_error(void);
int wt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
