// CPU-only mutation test of the C-ABI's host logic (whisper-trtllm_amd/csrc/host_logic.cpp), built by
// tests/test_host_sanitizer.py with  g++ -fsanitize=address,undefined -fno-sanitize-recover=all.
// Usage: fuzz_host_logic <valid.engine> <iterations> <seed>
// It (1) parses the valid blob and re-checks every accepted table entry against the blob bounds, (2) replays a set of
// targeted corruptions (64-bit offsets near 2^64, huge tensor counts, shapes whose product wraps, fp16 engines with
// ffn_dim < d_model), each of which MUST be rejected, (3) runs `iterations` random mutations (byte flips in header / table,
// field overwrites with boundary values, truncations): the parser may accept or reject, but every accepted entry must be
// in bounds and the sanitizers must stay silent, and (4) drives infer_shapes with random descriptors.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <vector>

#include "../../whisper-trtllm_amd/csrc/host_logic.h"

using namespace wt;

static int g_fail = 0;
#define CHECK(c, ...)                                 \
    do {                                              \
        if (!(c)) {                                   \
            fprintf(stderr, "CHECK failed: " __VA_ARGS__); \
            fprintf(stderr, "  (%s:%d)\n", __FILE__, __LINE__); \
            ++g_fail;                                 \
        }                                             \
    } while (0)

// the invariants wt_engine_open relies on after parse_blob returns WT_OK
static void check_accepted(const ParsedBlob& pb, const std::vector<unsigned char>& blob) {
    const uint64_t n = blob.size();
    CHECK(pb.hd.total_bytes == n, "total_bytes");
    CHECK(pb.hd.data_off <= n, "data_off");
    CHECK(pb.tensors.size() == pb.hd.n_tensors, "tensor count");
    unsigned long long touched = 0;
    for (const BlobTensor& t : pb.tensors) {
        CHECK(t.offset >= pb.hd.data_off && t.offset <= n && t.nbytes <= n - t.offset, "tensor '%s' out of bounds", t.name);
        CHECK(memchr(t.name, 0, sizeof t.name) != nullptr, "unterminated name");
        CHECK(t.ndim <= 4, "rank");
        unsigned long long cnt = 1;
        for (uint32_t k = 0; k < t.ndim; ++k) cnt *= (unsigned long long)t.shape[k];
        CHECK(cnt * (t.dtype == WT_F16 ? 2 : 4) == t.nbytes, "shape/bytes");
        if (t.nbytes) {  // what the upload + the kernels would read: first and last byte of the payload (ASan checks the access)
            touched += blob[t.offset];
            touched += blob[t.offset + t.nbytes - 1];
        }
    }
    if (touched == 0xffffffffffffffffULL) puts("");  // keep the reads alive
    const EngineDims& e = pb.dims;
    CHECK(e.d > 0 && e.d <= 1024 && e.d == e.H * 64 && e.F > 0 && e.F <= 4096 && e.L > 0, "dims");
    if (e.precision == WT_F16 && e.kind == WT_KIND_ENCODER) CHECK(e.F >= e.d, "fp16 encoder with ffn_dim < d_model accepted");
    if (e.precision == WT_F16) CHECK((e.d & 7) == 0 && (e.F & 7) == 0, "fp16 engine whose d_model / ffn_dim is not a multiple of 8 accepted");
}

template <class T>
static void put(std::vector<unsigned char>& b, size_t off, T v) {
    if (off + sizeof v <= b.size()) memcpy(&b[off], &v, sizeof v);
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<unsigned char> good;
    unsigned char buf[65536];
    for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;) good.insert(good.end(), buf, buf + n);
    fclose(f);
    const long iters = atol(argv[2]);
    std::mt19937_64 rng(strtoull(argv[3], nullptr, 10));
    char err[400];
    ParsedBlob pb;
    int rc = parse_blob(good.data(), good.size(), &pb, err, sizeof err);
    CHECK(rc == WT_OK, "the valid blob was rejected: %s", err);
    if (rc) return 1;
    check_accepted(pb, good);
    const BlobHeader hd = pb.hd;
    const size_t off_ntens = offsetof(BlobHeader, n_tensors), off_table = offsetof(BlobHeader, table_off),
                 off_data = offsetof(BlobHeader, data_off), off_total = offsetof(BlobHeader, total_bytes),
                 off_cfg = offsetof(BlobHeader, cfg), off_prec = offsetof(BlobHeader, precision);
    const size_t t0 = hd.table_off;  // first table entry
    const size_t e_shape = offsetof(BlobTensor, shape), e_off = offsetof(BlobTensor, offset), e_nb = offsetof(BlobTensor, nbytes),
                 e_ndim = offsetof(BlobTensor, ndim), e_dtype = offsetof(BlobTensor, dtype);

    // ---- (2) targeted corruptions: each MUST be rejected
    struct Case { const char* what; std::vector<unsigned char> b; };
    std::vector<Case> must_reject;
    auto mk = [&](const char* what) -> std::vector<unsigned char>& { must_reject.push_back({what, good}); return must_reject.back().b; };
    put<uint64_t>(mk("table_off near 2^64 (sum wraps)"), off_table, ~0ULL - 100);
    put<uint64_t>(mk("table_off = 2^64 - table bytes"), off_table, 0ULL - (uint64_t)hd.n_tensors * sizeof(BlobTensor));
    put<uint32_t>(mk("n_tensors = 2^32-1"), off_ntens, 0xffffffffu);
    put<uint64_t>(mk("data_off inside the table"), off_data, hd.table_off + 16);
    put<uint64_t>(mk("data_off beyond the blob"), off_data, good.size() + 256);
    put<uint64_t>(mk("total_bytes wrong"), off_total, good.size() + 1);
    put<uint64_t>(mk("tensor offset near 2^64"), t0 + e_off, ~0ULL - 15);
    put<uint64_t>(mk("tensor offset+nbytes wraps"), t0 + e_off, 0ULL - 4096);
    put<uint64_t>(mk("tensor nbytes huge"), t0 + e_nb, ~0ULL);
    put<uint64_t>(mk("tensor offset before data_off"), t0 + e_off, 0);
    put<uint64_t>(mk("tensor offset unaligned"), t0 + e_off, hd.data_off + 4);
    put<int64_t>(mk("negative dimension"), t0 + e_shape, -1);
    put<int64_t>(mk("dimension 2^62 (product wraps)"), t0 + e_shape, 1LL << 62);
    put<uint32_t>(mk("rank 5"), t0 + e_ndim, 5);
    put<uint32_t>(mk("dtype 7"), t0 + e_dtype, 7);
    put<int32_t>(mk("d_model 0"), off_cfg + 4 * CFG_D_MODEL, 0);
    put<int32_t>(mk("d_model 2048"), off_cfg + 4 * CFG_D_MODEL, 2048);
    put<int32_t>(mk("heads negative"), off_cfg + 4 * CFG_HEADS, -4);
    put<int32_t>(mk("ffn 1<<30"), off_cfg + 4 * CFG_FFN, 1 << 30);
    put<int32_t>(mk("layers 0"), off_cfg + 4 * CFG_LAYERS, 0);
    if (hd.kind == WT_KIND_ENCODER) {
        auto& b = mk("fp16 engine with ffn_dim < d_model");
        put<uint32_t>(b, off_prec, WT_F16);
        put<int32_t>(b, off_cfg + 4 * CFG_FFN, hd.cfg[CFG_D_MODEL] / 2);
    } else {
        auto& b = mk("fp16 decoder with ffn_dim % 8 != 0");   // eight halves per 16-byte weight load
        put<uint32_t>(b, off_prec, WT_F16);
        put<int32_t>(b, off_cfg + 4 * CFG_FFN, hd.cfg[CFG_D_MODEL] + 4);
    }
    {
        auto& b = mk("truncated to the header");
        b.resize(sizeof(BlobHeader));
    }
    {
        auto& b = mk("truncated mid-table");
        b.resize(hd.table_off + sizeof(BlobTensor) / 2);
    }
    for (Case& c : must_reject) {
        // exact-size heap copy so that ASan sees any read past the end
        std::vector<unsigned char> exact(c.b.begin(), c.b.end());
        ParsedBlob q;
        rc = parse_blob(exact.data(), exact.size(), &q, err, sizeof err);
        CHECK(rc != WT_OK, "corruption accepted: %s", c.what);
    }

    // ---- (3) random mutations
    const uint64_t edge[] = {0, 1, 15, 16, 255, 256, 4095, 0x7fffffffULL, 0x80000000ULL, 0xffffffffULL, 0x100000000ULL,
                             0x7fffffffffffffffULL, 0x8000000000000000ULL, ~0ULL, ~0ULL - 15, ~0ULL - 151, (uint64_t)good.size(),
                             (uint64_t)good.size() - 1, (uint64_t)good.size() + 1, hd.data_off, hd.data_off - 1, hd.table_off};
    const size_t n_edge = sizeof edge / sizeof edge[0];
    const size_t head_bytes = (size_t)hd.data_off < good.size() ? (size_t)hd.data_off : good.size();
    long accepted = 0;
    for (long it = 0; it < iters; ++it) {
        std::vector<unsigned char> b = good;
        const int nmut = 1 + (int)(rng() % 4);
        for (int m = 0; m < nmut; ++m) {
            const int kind = (int)(rng() % 6);
            if (kind == 0) {  // flip a byte in header / table
                b[rng() % head_bytes] ^= (unsigned char)(1u << (rng() % 8));
            } else if (kind == 1) {  // overwrite a header u64 with an edge value
                const size_t offs[] = {off_table, off_data, off_total};
                put<uint64_t>(b, offs[rng() % 3], edge[rng() % n_edge]);
            } else if (kind == 2) {  // overwrite a table u64/i64 field of a random entry
                const size_t ent = t0 + (rng() % (hd.n_tensors ? hd.n_tensors : 1)) * sizeof(BlobTensor);
                const size_t fields[] = {e_off, e_nb, e_shape, e_shape + 8, e_shape + 16, e_shape + 24};
                put<uint64_t>(b, ent + fields[rng() % 6], edge[rng() % n_edge]);
            } else if (kind == 3) {  // config ints
                put<int32_t>(b, off_cfg + 4 * (rng() % 9), (int32_t)edge[rng() % n_edge]);
            } else if (kind == 4) {  // n_tensors / precision / kind / version
                put<uint32_t>(b, 8 + 4 * (rng() % 4), (uint32_t)edge[rng() % n_edge]);
            } else {  // truncate (and sometimes patch total_bytes to match, so the later checks are reached)
                const size_t nl = rng() % (b.size() + 1);
                b.resize(nl);
                if (rng() & 1) put<uint64_t>(b, off_total, (uint64_t)nl);
            }
        }
        std::vector<unsigned char> exact(b.begin(), b.end());
        ParsedBlob q;
        rc = parse_blob(exact.empty() ? (const void*)"" : (const void*)exact.data(), exact.size(), &q, err, sizeof err);
        if (rc == WT_OK) {
            ++accepted;
            check_accepted(q, exact);
        }
    }

    // ---- (4) shape inference with random descriptors
    const char* names[] = {"data", "length", "encoder_hidden_states", "self_past_key", "self_past_value", "cross_past_key",
                           "cross_past_value", "past_self_cache_mask", "past_cross_cache_mask", "bogus", ""};
    long shapes_ok = 0;
    for (long it = 0; it < iters; ++it) {
        wt_tensor_desc in[12], out[8];
        const int n_in = (int)(rng() % 12);
        for (int i = 0; i < n_in; ++i) {
            memset(&in[i], 0, sizeof in[i]);
            if (rng() % 16 == 0) memset(in[i].name, 'x', sizeof in[i].name);  // unterminated name
            else snprintf(in[i].name, sizeof in[i].name, "%s", i < 9 && rng() % 8 ? names[i] : names[rng() % 11]);
            in[i].dtype = (int)(rng() % 5) - (rng() % 16 == 0);
            in[i].ndim = rng() % 8 ? (int)(rng() % 5) : (int)(int32_t)edge[rng() % n_edge];
            for (int k = 0; k < WT_MAX_DIMS; ++k) {
                const int64_t typical[] = {1, pb.dims.L, pb.dims.H, pb.dims.S, pb.dims.S + 1, 64, pb.dims.d, pb.dims.C, 2 * pb.dims.S, pb.dims.T, pb.dims.T + 1, 3};
                in[i].shape[k] = rng() % 6 ? typical[rng() % 12] : (int64_t)edge[rng() % n_edge];
            }
        }
        int n_out = (int)(rng() % 9);
        ShapeState st;
        rc = infer_shapes(pb.dims, in, n_in, out, &n_out, &st, err, sizeof err);
        CHECK((rc == WT_OK) == st.ok, "ShapeState.ok disagrees with the return code");
        if (rc == WT_OK) {
            ++shapes_ok;
            CHECK(n_out >= 1 && n_out <= 5, "n_out");
            for (int i = 0; i < n_out; ++i)
                for (int k = 0; k < out[i].ndim; ++k) CHECK(out[i].shape[k] >= 1, "non-positive output dim");
        }
    }
    printf("fuzz_host_logic: %zu targeted corruptions rejected, %ld random blobs (%ld accepted), %ld descriptor sets (%ld accepted), %d failures\n",
           must_reject.size(), iters, accepted, iters, shapes_ok, g_fail);
    return g_fail ? 1 : 0;
}
