// Log-mel front-end on the GPU: waveform -> Whisper's 80 x 3000 log-mel features.
// Reference: HF WhisperFeatureExtractor._np_extract_fbank_features (feature_extraction_whisper.py:94-111) over
// audio_utils.spectrogram (audio_utils.py:267-452): zero-pad/trim to 30 s, reflect-pad 200, Hann-400 frames with hop 160,
// |rfft|^2, slaney mel filter bank (80 x 201), log10(max(1e-10, .)), drop the last frame, clamp to max-8, (x+4)/4.
// The DFT and the mel projection are dense contractions and run on the fp32 MFMA GEMM (gemm_f32_kernel); framing,
// power, log/max and the normalising transpose are small coalesced kernels.
#include "../../include/whisper_trtllm_amd.h"
#include "wt_common.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

using namespace wt;

int wt_set_error(int code, const char* fmt, ...);  // engine.hip

struct wt_logmel {
    int device = 0, n_fft = 0, hop = 0, n_mels = 0, n_frames = 0, n_bins = 0, n_samples = 0;
    int ndft = 0, npw = 0;   // padded widths: DFT output columns (re|pad|im|pad), power columns
    float *dft = nullptr, *window = nullptr, *filt = nullptr;
    int cap = 0;
    float *frames = nullptr, *spec = nullptr, *pw = nullptr, *mel = nullptr, *bmax = nullptr;
};

// frames[(b*T + t)][j] = padded[t*hop + j] * window[j]; padded = reflect-pad(n_fft/2) of the zero-padded/trimmed waveform
__global__ __launch_bounds__(256) void frame_window_kernel(const float* __restrict__ audio, int n_in, int n_samples,
                                                           const float* __restrict__ window, float* __restrict__ frames,
                                                           int T, int n_fft, int hop) {
    const int t = blockIdx.x, b = blockIdx.y;
    const float* x = audio + (size_t)b * n_in;
    float* out = frames + ((size_t)b * T + t) * n_fft;
    const int half = n_fft / 2;
    for (int j = threadIdx.x; j < n_fft; j += 256) {
        int i = t * hop + j - half;                 // index into the 30 s signal
        if (i < 0) i = -i;                          // np.pad(mode="reflect")
        if (i >= n_samples) i = 2 * (n_samples - 1) - i;
        const float v = i < n_in ? x[i] : 0.f;      // zero padding up to 30 s
        out[j] = v * window[j];
    }
}

// pw[m][f] = re^2 + im^2 for f < n_bins (re at column f, im at column im_off + f of spec); padding columns = 0
__global__ __launch_bounds__(256) void power_kernel(const float* __restrict__ spec, float* __restrict__ pw, size_t rows,
                                                    int ndft, int npw, int n_bins, int im_off) {
    const size_t total = rows * npw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t m = i / npw;
        const int f = (int)(i - m * npw);
        float v = 0.f;
        if (f < n_bins) {
            const float re = spec[m * ndft + f], im = spec[m * ndft + im_off + f];
            v = re * re + im * im;
        }
        pw[i] = v;
    }
}

// mel[m][c] -> log10(max(1e-10, mel)) in place + per-block maximum (one block = 64 frames of one utterance)
__global__ __launch_bounds__(256) void log_max_kernel(float* __restrict__ mel, float* __restrict__ bmax, int T, int n_mels) {
    __shared__ float red[4];
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    float mx = -INFINITY;
    const int n = min(64, T - t0) * n_mels;
    float* base = mel + ((size_t)b * T + t0) * n_mels;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = log10f(fmaxf(1e-10f, base[i]));
        base[i] = v;
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) bmax[(size_t)b * gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// out[b][c][t] = (max(logmel[b][t][c], utterance_max - 8) + 4) / 4
__global__ __launch_bounds__(256) void normalize_transpose_kernel(const float* __restrict__ mel, const float* __restrict__ bmax,
                                                                  int nblk, float* __restrict__ out, int T, int n_mels) {
    __shared__ float tile[64][129];
    __shared__ float s_max;
    const int b = blockIdx.y, t0 = blockIdx.x * 64;
    if (threadIdx.x < 64) {
        float mx = -INFINITY;
        for (int i = threadIdx.x; i < nblk; i += 64) mx = fmaxf(mx, bmax[(size_t)b * nblk + i]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (threadIdx.x == 0) s_max = mx;
    }
    const int nt = min(64, T - t0);
    const float* base = mel + ((size_t)b * T + t0) * n_mels;
    for (int i = threadIdx.x; i < nt * n_mels; i += 256) tile[i / n_mels][i % n_mels] = base[i];
    __syncthreads();
    const float floor_v = s_max - 8.0f;
    for (int i = threadIdx.x; i < n_mels * 64; i += 256) {
        const int c = i >> 6, tl = i & 63;
        if (tl < nt) out[((size_t)b * n_mels + c) * T + t0 + tl] = (fmaxf(tile[tl][c], floor_v) + 4.0f) * 0.25f;
    }
}

extern "C" int wt_logmel_create(int device, int n_fft, int hop, int n_mels, int n_frames, const float* dft, int ndft,
                                const float* window, const float* filters, int npw, wt_logmel** out) {
    if (!out || !dft || !window || !filters) return wt_set_error(WT_E_INVALID, "wt_logmel_create: null argument");
    *out = nullptr;
    const int n_bins = n_fft / 2 + 1;
    if (n_fft <= 0 || (n_fft & 3) || hop <= 0 || n_mels <= 0 || n_mels > 128 || n_frames <= 0 || ndft < 2 * n_bins || (ndft & 1) ||
        npw < n_bins || (npw & 3))
        return wt_set_error(WT_E_INVALID, "wt_logmel_create: bad geometry (n_fft %d hop %d n_mels %d frames %d ndft %d npw %d)", n_fft, hop,
                            n_mels, n_frames, ndft, npw);
    wt::DeviceGuard guard(device);
    if (guard.err != hipSuccess) return wt_set_error(WT_E_HIP, "hipSetDevice(%d) failed", device);
    wt_logmel* h = new wt_logmel();
    h->device = device; h->n_fft = n_fft; h->hop = hop; h->n_mels = n_mels; h->n_frames = n_frames; h->n_bins = n_bins;
    h->n_samples = n_frames * hop; h->ndft = ndft; h->npw = npw;
    const size_t nd = (size_t)ndft * n_fft, nf = (size_t)n_mels * npw;
    if (hipMalloc((void**)&h->dft, nd * 4) != hipSuccess || hipMalloc((void**)&h->window, (size_t)n_fft * 4) != hipSuccess ||
        hipMalloc((void**)&h->filt, nf * 4) != hipSuccess) {
        wt_logmel_destroy(h);  // frees whichever tables were already allocated
        return wt_set_error(WT_E_NOMEM, "wt_logmel_create: table allocation failed");
    }
    if (hipMemcpy(h->dft, dft, nd * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->window, window, (size_t)n_fft * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->filt, filters, nf * 4, hipMemcpyHostToDevice) != hipSuccess) {
        wt_logmel_destroy(h);
        return wt_set_error(WT_E_HIP, "wt_logmel_create: table upload failed");
    }
    *out = h;
    return WT_OK;
}

extern "C" void wt_logmel_destroy(wt_logmel* h) {
    if (!h) return;
    wt::DeviceGuard guard(h->device);
    for (float* p : {h->dft, h->window, h->filt, h->frames}) if (p) hipFree(p);
    delete h;
}

extern "C" int wt_logmel_forward(wt_logmel* h, const float* audio, int batch, int n_in, float* mel_out, void* stream) {
    if (!h || !audio || !mel_out || batch < 1 || n_in < 1) return wt_set_error(WT_E_INVALID, "wt_logmel_forward: bad arguments");
    wt::DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return wt_set_error(WT_E_HIP, "hipSetDevice failed");
    hipStream_t s = (hipStream_t)stream;
    const int T = h->n_frames;
    const size_t M = (size_t)batch * T;
    const int nblk = (T + 63) / 64;
    if (batch > h->cap) {
        if (h->frames) hipFree(h->frames);
        const size_t floats = M * h->n_fft + M * h->ndft + M * h->npw + M * h->n_mels + (size_t)batch * nblk + 64;
        if (hipMalloc((void**)&h->frames, floats * 4) != hipSuccess) {
            h->frames = nullptr; h->cap = 0;
            return wt_set_error(WT_E_NOMEM, "wt_logmel_forward: workspace allocation (%zu bytes) failed", floats * 4);
        }
        h->spec = h->frames + M * h->n_fft;
        h->pw = h->spec + M * h->ndft;
        h->mel = h->pw + M * h->npw;
        h->bmax = h->mel + M * h->n_mels;
        h->cap = batch;
    }
    hipLaunchKernelGGL(frame_window_kernel, dim3(T, batch), dim3(256), 0, s, audio, n_in, h->n_samples, h->window, h->frames, T,
                       h->n_fft, h->hop);
    GemmParams g;
    memset(&g, 0, sizeof g);
    g.A = h->frames; g.lda = h->n_fft; g.a_rows_per_batch = (int)M; g.W = h->dft; g.M = (int)M; g.N = h->ndft; g.K = h->n_fft;
    g.C = h->spec; g.ldc = h->ndft; g.c_rows_per_batch = (int)M;
    hipError_t e = launch_gemm_f32(g, s);
    if (e != hipSuccess) return wt_set_error(WT_E_HIP, "DFT GEMM launch failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(power_kernel, dim3(2048), dim3(256), 0, s, h->spec, h->pw, M, h->ndft, h->npw, h->n_bins, h->ndft / 2);
    memset(&g, 0, sizeof g);
    g.A = h->pw; g.lda = h->npw; g.a_rows_per_batch = (int)M; g.W = h->filt; g.M = (int)M; g.N = h->n_mels; g.K = h->npw;
    g.C = h->mel; g.ldc = h->n_mels; g.c_rows_per_batch = (int)M;
    e = launch_gemm_f32(g, s);
    if (e != hipSuccess) return wt_set_error(WT_E_HIP, "mel GEMM launch failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(log_max_kernel, dim3(nblk, batch), dim3(256), 0, s, h->mel, h->bmax, T, h->n_mels);
    hipLaunchKernelGGL(normalize_transpose_kernel, dim3(nblk, batch), dim3(256), 0, s, h->mel, h->bmax, nblk, mel_out, T, h->n_mels);
    e = hipGetLastError();
    if (e != hipSuccess) return wt_set_error(WT_E_HIP, "front-end kernel launch failed: %s", hipGetErrorString(e));
    return WT_OK;
}
