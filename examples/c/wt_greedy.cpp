// Plain C/C++ host for the C-ABI (no Python, no torch): engines built by build_encoder.py / build_decoder.py, a float32
// log-mel file, greedy decode, token ids on stdout.  This is what a non-Python embedder of the reference's hot path links:
// only include/whisper_trtllm_amd.h and the HIP runtime for device buffers.
//
//   hipcc -O2 -Iinclude examples/c/wt_greedy.cpp -Lwhisper-trtllm_amd/lib -lwhisper_trtllm_amd -Wl,-rpath,$PWD/whisper-trtllm_amd/lib -o wt_greedy
//   ./wt_greedy WhisperEncoder.engine WhisperDecoder.engine mel.f32 <batch> rules.txt [workers | stream<slots>]
//
// With `stream<slots>` (e.g. stream2) the utterances go through the CONTINUOUS mode of the ABI (wt_decoder_stream_begin / _submit / _run /
// _collect): `slots` decode rows stay busy, every utterance stops at its own EOS (the reference's run.py:219-226 decodes one clip at a time)
// and the device refills the slot it frees; the ids come back through pinned host memory.  Each printed row then ends at its EOS.
//
// With `workers` > 1 the same batch is decoded by that many host threads at once, each on its own clone of the two engines
// (wt_engine_clone: one copy of the weights, own workspace / caches / graphs) and its own stream -- distinct handles may be driven
// concurrently -- and the program fails unless every worker produced the same ids.
//
// rules.txt (whitespace separated integers): decoder_start eos pad max_length begin_index
//   n_suppress s_1 .. s_n   n_begin_suppress b_1 .. b_n   n_forced (index token)_1 .. (index token)_n
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "whisper_trtllm_amd.h"

static std::vector<char> read_file(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)n);
    if (n > 0 && fread(buf.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read on %s\n", path); exit(2); }
    fclose(f);
    return buf;
}

#define WT(call) do { int rc_ = (call); if (rc_ != WT_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, wt_last_error()); return 1; } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

// encoder + greedy decode of one batch on one pair of handles and one stream; ids (batch x max_length) and the length come back
static int decode_once(wt_engine* enc, wt_engine* dec, const wt_engine_info& ei, const std::vector<char>& mel, int batch, const wt_greedy_params& gp,
                       std::vector<int32_t>* ids_out, int* len_out) {
    hipStream_t stream;
    HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    float *d_mel = nullptr, *d_hidden = nullptr;
    int32_t* d_ids = nullptr;
    HIP(hipMalloc((void**)&d_mel, mel.size()));
    HIP(hipMalloc((void**)&d_hidden, (size_t)batch * ei.max_source_positions * ei.d_model * sizeof(float)));
    HIP(hipMalloc((void**)&d_ids, (size_t)batch * gp.max_length * sizeof(int32_t)));
    HIP(hipMemcpyAsync(d_mel, mel.data(), mel.size(), hipMemcpyHostToDevice, stream));
    WT(wt_encoder_forward(enc, d_mel, batch, d_hidden, stream));
    WT(wt_decoder_greedy(dec, d_hidden, batch, &gp, d_ids, len_out, stream));
    ids_out->resize((size_t)batch * gp.max_length);
    HIP(hipMemcpyAsync(ids_out->data(), d_ids, ids_out->size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    HIP(hipStreamSynchronize(stream));
    (void)hipFree(d_mel); (void)hipFree(d_hidden); (void)hipFree(d_ids);
    (void)hipStreamDestroy(stream);
    return 0;
}

// the same utterances through the continuous mode: `slots` rows decode at a time, submitted in chunks of at most `slots`
static int decode_stream(wt_engine* enc, wt_engine* dec, const wt_engine_info& ei, const std::vector<char>& mel, int batch, const wt_greedy_params& gp, int slots) {
    hipStream_t stream;
    HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    float *d_mel = nullptr, *d_hidden = nullptr;
    const size_t hid = (size_t)ei.max_source_positions * ei.d_model;
    HIP(hipMalloc((void**)&d_mel, mel.size()));
    HIP(hipMalloc((void**)&d_hidden, (size_t)batch * hid * sizeof(float)));
    HIP(hipMemcpyAsync(d_mel, mel.data(), mel.size(), hipMemcpyHostToDevice, stream));
    WT(wt_encoder_forward(enc, d_mel, batch, d_hidden, stream));
    WT(wt_decoder_stream_begin(dec, slots, 0, &gp, stream));
    std::vector<int32_t> handles((size_t)batch);
    std::vector<std::vector<int32_t>> rows((size_t)batch);
    int submitted = 0, collected = 0, n_fin = 0, n_wait = 0, n_steps = 0;
    while (collected < batch) {
        while (submitted < batch) {                       // as many chunks as the cache pool takes (WT_E_STATE: collect first)
            const int n = batch - submitted < slots ? batch - submitted : slots;
            const int rc = wt_decoder_stream_submit(dec, d_hidden + (size_t)submitted * hid, n, nullptr, handles.data() + submitted, stream);
            if (rc == WT_E_STATE) break;
            if (rc != WT_OK) { fprintf(stderr, "wt_decoder_stream_submit -> %d: %s\n", rc, wt_last_error()); return 1; }
            submitted += n;
        }
        WT(wt_decoder_stream_run(dec, submitted < batch ? slots : 0, 0, &n_fin, &n_wait, &n_steps, stream));
        for (int u = 0; u < submitted; ++u) {             // harvest what has finished (releases its cache row)
            if (!rows[(size_t)u].empty()) continue;
            std::vector<int32_t> buf((size_t)gp.max_length);
            int len = 0;
            WT(wt_decoder_stream_collect(dec, handles[(size_t)u], buf.data(), gp.max_length, &len));
            if (len > 0) { buf.resize((size_t)len); rows[(size_t)u] = buf; ++collected; }
        }
    }
    HIP(hipStreamSynchronize(stream));
    for (int u = 0; u < batch; ++u) {
        for (size_t t = 0; t < rows[(size_t)u].size(); ++t) printf(t ? " %d" : "%d", rows[(size_t)u][t]);
        printf("\n");
    }
    fprintf(stderr, "continuous mode: %d utterances, %d slots, %d decoder steps\n", batch, slots, n_steps);
    (void)hipFree(d_mel); (void)hipFree(d_hidden);
    (void)hipStreamDestroy(stream);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 6 && argc != 7) { fprintf(stderr, "usage: %s <encoder.engine> <decoder.engine> <mel.f32> <batch> <rules.txt> [workers | stream<slots>]\n", argv[0]); return 2; }
    const int batch = atoi(argv[4]);
    const int stream_slots = (argc == 7 && !strncmp(argv[6], "stream", 6)) ? atoi(argv[6] + 6) : 0;
    const int workers = (argc == 7 && !stream_slots) ? atoi(argv[6]) : 1;
    if (argc == 7 && !strncmp(argv[6], "stream", 6) && (stream_slots < 1 || stream_slots > 16)) { fprintf(stderr, "stream<slots>: 1..16 slots\n"); return 2; }
    if (workers < 1 || workers > 16) { fprintf(stderr, "workers must be 1..16\n"); return 2; }
    std::vector<char> enc_blob = read_file(argv[1]), dec_blob = read_file(argv[2]), mel = read_file(argv[3]);

    std::vector<int> rules;
    { FILE* f = fopen(argv[5], "r"); if (!f) { fprintf(stderr, "cannot open %s\n", argv[5]); return 2; } int v; while (fscanf(f, "%d", &v) == 1) rules.push_back(v); fclose(f); }
    size_t at = 0;
    auto next = [&]() { if (at >= rules.size()) { fprintf(stderr, "rules file too short\n"); exit(2); } return rules[at++]; };
    wt_greedy_params gp = {};
    gp.decoder_start_token_id = next(); gp.eos_token_id = next(); gp.pad_token_id = next(); gp.max_length = next(); gp.begin_index = next();
    std::vector<int32_t> suppress((size_t)next()); for (auto& x : suppress) x = next();
    std::vector<int32_t> begin_suppress((size_t)next()); for (auto& x : begin_suppress) x = next();
    std::vector<int32_t> forced(2 * (size_t)next()); for (auto& x : forced) x = next();
    gp.suppress_tokens = suppress.data(); gp.n_suppress_tokens = (int)suppress.size();
    gp.begin_suppress_tokens = begin_suppress.data(); gp.n_begin_suppress_tokens = (int)begin_suppress.size();
    gp.forced_decoder_ids = forced.data(); gp.n_forced = (int)forced.size() / 2;
    gp.force_eos_step = -1; gp.logits_trace = nullptr; gp.force_eos_steps = nullptr;

    std::vector<wt_engine*> encs((size_t)workers, nullptr), decs((size_t)workers, nullptr);
    WT(wt_engine_open(enc_blob.data(), enc_blob.size(), 0, &encs[0]));
    WT(wt_engine_open(dec_blob.data(), dec_blob.size(), 0, &decs[0]));
    for (int k = 1; k < workers; ++k) {            // further execution contexts on the SAME device weights
        WT(wt_engine_clone(encs[0], &encs[(size_t)k]));
        WT(wt_engine_clone(decs[0], &decs[(size_t)k]));
    }
    wt_engine_info ei;
    WT(wt_engine_get_info(encs[0], &ei));
    const size_t mel_floats = (size_t)batch * ei.n_mels * 2 * ei.max_source_positions;
    if (mel.size() != mel_floats * sizeof(float)) { fprintf(stderr, "mel file holds %zu bytes, expected %zu\n", mel.size(), mel_floats * sizeof(float)); return 2; }

    if (stream_slots) {
        const int rc = decode_stream(encs[0], decs[0], ei, mel, batch, gp, stream_slots);
        wt_engine_close(encs[0]);
        wt_engine_close(decs[0]);
        return rc;
    }
    std::vector<std::vector<int32_t>> ids((size_t)workers);
    std::vector<int> lens((size_t)workers, 0), rcs((size_t)workers, 0);
    std::vector<std::thread> threads;
    for (int k = 1; k < workers; ++k)
        threads.emplace_back([&, k] {
            for (int rep = 0; rep < 3 && rcs[(size_t)k] == 0; ++rep)      // a few decodes per worker: graphs are captured, then replayed
                rcs[(size_t)k] = decode_once(encs[(size_t)k], decs[(size_t)k], ei, mel, batch, gp, &ids[(size_t)k], &lens[(size_t)k]);
        });
    for (int rep = 0; rep < (workers > 1 ? 3 : 1) && rcs[0] == 0; ++rep) rcs[0] = decode_once(encs[0], decs[0], ei, mel, batch, gp, &ids[0], &lens[0]);
    for (auto& t : threads) t.join();
    for (int k = 0; k < workers; ++k) {
        if (rcs[(size_t)k]) return 1;
        if (lens[(size_t)k] != lens[0] || memcmp(ids[(size_t)k].data(), ids[0].data(), ids[0].size() * sizeof(int32_t)) != 0) {
            fprintf(stderr, "worker %d decoded different ids than worker 0\n", k);
            return 3;
        }
    }
    for (int b = 0; b < batch; ++b) {
        for (int t = 0; t < lens[0]; ++t) printf(t ? " %d" : "%d", ids[0][(size_t)b * gp.max_length + t]);
        printf("\n");
    }
    for (int k = workers - 1; k >= 0; --k) {       // any order: the weight payload is freed with the last handle that shares it
        wt_engine_close(encs[(size_t)k]);
        wt_engine_close(decs[(size_t)k]);
    }
    return 0;
}
