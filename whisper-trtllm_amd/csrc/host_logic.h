// Host-only logic of the C-ABI: serialized-engine ("weight pack") parsing, config validation and the Session-style
// shape inference.  No HIP types or calls: this file and host_logic.cpp also build with plain g++ under
// -fsanitize=address,undefined for the CPU mutation test (tests/native/fuzz_host_logic.cpp, tests/test_host_sanitizer.py).
// Reference call sites replaced: tensorrt_llm/runtime/session.py:54 (from_serialized_engine) and :116-146 (infer_shapes).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/whisper_trtllm_amd.h"

namespace wt {

constexpr int HEAD_DIM = 64;  // every Whisper size uses 64-wide heads (d_model / n_heads)

// ---- serialized engine layout, written by whisper-trtllm_amd/engine_pack.py --------------------------------
// [BlobHeader][BlobTensor * n_tensors][raw tensor bytes, each 256-byte aligned]
struct BlobHeader {
    char magic[8];  // "WTENGINE"
    uint32_t version, kind, precision, n_tensors;
    int32_t cfg[24];  // see CFG_* below
    uint64_t table_off, data_off, total_bytes;
};
struct BlobTensor {
    char name[96];
    uint32_t dtype, ndim;
    int64_t shape[4];
    uint64_t offset, nbytes;  // offset from blob start
};
static_assert(sizeof(BlobHeader) == 144, "blob header layout");
static_assert(sizeof(BlobTensor) == 152, "blob tensor layout");
enum { CFG_D_MODEL = 0, CFG_HEADS, CFG_LAYERS, CFG_FFN, CFG_MELS, CFG_SRC_POS, CFG_TGT_POS, CFG_VOCAB, CFG_TIED };
constexpr uint32_t BLOB_VERSION = 2;

struct EngineDims {
    int kind = 0, precision = 0;
    int d = 0, H = 0, L = 0, F = 0, C = 0, S = 0, T = 0, V = 0, tied = 0;
};

struct ParsedBlob {
    BlobHeader hd;
    EngineDims dims;
    std::vector<BlobTensor> tensors;  // validated: in-bounds, aligned, known dtype, shape x element size == nbytes
};

// Every check is overflow-safe (no sum of file-controlled 64-bit terms).  Returns WT_OK or a WT_E_* code with a message in err.
int parse_blob(const void* blob, size_t nbytes, ParsedBlob* out, char* err, size_t errlen);

// shapes remembered between infer_shapes and run (Session keeps them in the execution context, session.py:137-146)
struct ShapeState {
    bool ok = false;
    int c_B = 1, c_s = 0, c_ms = 0, c_mc = 0;
};
int infer_shapes(const EngineDims& e, const wt_tensor_desc* in, int n_in, wt_tensor_desc* out, int* n_out, ShapeState* st,
                 char* err, size_t errlen);

}  // namespace wt
