// Log-mel front-end on the GPU: waveform -> Whisper's 80 x 3000 log-mel features.
// Reference: HF WhisperFeatureExtractor._np_extract_fbank_features (feature_extraction_whisper.py:94-111) over
// audio_utils.spectrogram (audio_utils.py:267-452): zero-pad/trim to 30 s, reflect-pad 200, Hann-400 frames with hop 160,
// |rfft|^2, slaney mel filter bank (80 x 201), log10(max(1e-10, .)), drop the last frame, clamp to max-8, (x+4)/4.
// Framing + windowed DFT + power run in fp64 on the matrix cores (dft_power_f64_kernel), the mel projection on the fp32 MFMA GEMM
// (gemm_f32_kernel); log/max and the normalising transpose are small coalesced kernels.
#include "../../include/whisper_trtllm_amd.h"
#include "wt_common.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <type_traits>
#include <vector>

using namespace wt;

int wt_set_error(int code, const char* fmt, ...);  // engine.hip

struct wt_logmel {
    int device = 0, n_fft = 0, hop = 0, n_mels = 0, n_frames = 0, n_bins = 0, n_samples = 0;
    int npw = 0, nbt = 0;    // power columns (n_bins rounded up to 4), 16-bin tiles of the DFT
    double *frag = nullptr, *window = nullptr;   // fp64 DFT matrix in MFMA fragment order, fp64 Hann window
    float* filt = nullptr;
    int cap = 0;
    float *pw = nullptr, *mel = nullptr, *bmax = nullptr;
};

// ---- windowed DFT + power spectrum in fp64 on the matrix cores ----------------------------------------------------------------
// The reference frames the waveform in float64, takes np.fft.rfft in float64, stores the spectrum as complex64 and squares its
// magnitude in float64 (audio_utils.py:399-428).  Bins on the spectral-leakage floor are sums of 400 products that cancel to
// ~1e-7 of their terms, so an fp32 DFT misses them by up to 2e-3 of the feature range (round 2); v_mfma_f64_16x16x4_f64 makes the
// DFT exact to fp64 rounding at 8 GFLOP per 8 x 30 s -- ~0.2 ms of the chip's fp64 matrix rate.
//   grid (ceil(T / 64), batch); 4 waves: wave = (frame group of 32) x (half of the bin tiles).  The block's 63*hop + n_fft samples
//   sit in LDS as fp32, the A fragment a[frame][k] = (double)x[frame*hop + k] * window[k] is formed in registers; the DFT matrix
//   is pre-laid in fragment order (tile n = 2*bin_tile + {cos, -sin}, k-step, lane), so a B fragment is one coalesced 512-byte load
//   from L2 (1.3 MB in all) feeding two MFMAs.  cos and -sin tiles of the same bins land in the same lane and register, so
//   |X|^2 = fl32(re)^2 + fl32(im)^2 (the reference's complex64 rounding) needs no exchange.
// Fragment maps (cdna guide: f64 MFMA): A row = lane & 15, k = lane >> 4; B col = lane & 15, k = lane >> 4;
// C/D col = lane & 15, row = (lane >> 4) + 4 * reg.
typedef double d4v __attribute__((ext_vector_type(4)));
constexpr int DFT_FRAMES = 64;       // frames per workgroup
constexpr int DFT_MAX_BT = 7;        // bin tiles per wave half (n_fft <= 446)

__global__ __launch_bounds__(256) void dft_power_f64_kernel(const float* __restrict__ audio, int n_in, int n_samples,
                                                            const double* __restrict__ window, const double* __restrict__ frag,
                                                            float* __restrict__ pw, int T, int n_fft, int hop, int npw, int nbt) {
    extern __shared__ __attribute__((aligned(16))) double dft_smem[];
    double* s_w = dft_smem;                                        // [n_fft]
    float* s_x = reinterpret_cast<float*>(dft_smem + n_fft);       // [(DFT_FRAMES - 1) * hop + n_fft]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fg = wave & 1, half = wave >> 1;
    const int b = blockIdx.y, t0 = blockIdx.x * DFT_FRAMES;
    const int span = (DFT_FRAMES - 1) * hop + n_fft, ksteps = n_fft >> 2;
    const float* x = audio + (size_t)b * n_in;
    for (int i = tid; i < span; i += 256) {
        long long sidx = (long long)t0 * hop + i - n_fft / 2;     // index into the 30 s signal
        if (sidx < 0) sidx = -sidx;                               // np.pad(mode="reflect")
        if (sidx >= n_samples) sidx = 2ll * (n_samples - 1) - sidx;
        s_x[i] = (sidx >= 0 && sidx < n_in) ? x[sidx] : 0.f;      // zero padding up to 30 s (and frames past T in the last block)
    }
    for (int i = tid; i < n_fft; i += 256) s_w[i] = window[i];
    __syncthreads();
    const int bt_mid = (nbt + 1) >> 1;
    const int bt0 = half ? bt_mid : 0, nb = half ? nbt - bt_mid : bt_mid;   // this wave's bin tiles [bt0, bt0 + nb)
    d4v acc[2][2 * DFT_MAX_BT];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 2 * DFT_MAX_BT; ++j) acc[r][j] = d4v{0.0, 0.0, 0.0, 0.0};
    const int frow = fg * 32 + (lane & 15), kl = lane >> 4;
    const double* fp = frag + (size_t)(2 * bt0) * ksteps * 64 + lane;
    // B fragments of k-step kk + 1 are requested before the MFMAs of step kk (two static register sets): the L2 round trip hides
    // under 2 * 2 * nb MFMAs of 64 clocks each
    double bq[2][2 * DFT_MAX_BT];
    auto bload = [&](auto set_c, const int kk) {
        constexpr int set = decltype(set_c)::value;
#pragma unroll
        for (int j = 0; j < 2 * DFT_MAX_BT; ++j)
            if (j < 2 * nb) bq[set][j] = fp[((size_t)j * ksteps + kk) * 64];
    };
    auto kstep = [&](auto set_c, const int kk) {
        constexpr int set = decltype(set_c)::value;
        if (kk + 1 < ksteps) bload(std::integral_constant<int, set ^ 1>{}, kk + 1);
        const int k = kk * 4 + kl;
        const double wv = s_w[k];
        const double a0 = (double)s_x[frow * hop + k] * wv;
        const double a1 = (double)s_x[(frow + 16) * hop + k] * wv;
#pragma unroll
        for (int j = 0; j < 2 * DFT_MAX_BT; ++j)
            if (j < 2 * nb) {   // wave-uniform
                acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bq[set][j], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bq[set][j], acc[1][j], 0, 0, 0);
            }
    };
    bload(std::integral_constant<int, 0>{}, 0);
    for (int kk = 0; kk < ksteps; kk += 2) {
        kstep(std::integral_constant<int, 0>{}, kk);
        if (kk + 1 < ksteps) kstep(std::integral_constant<int, 1>{}, kk + 1);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < DFT_MAX_BT; ++q)
            if (q < nb) {
                const int bin = (bt0 + q) * 16 + (lane & 15);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float re = (float)acc[r][2 * q][i], im = (float)acc[r][2 * q + 1][i];   // complex64 storage of the reference
                    const double pwr = (double)re * (double)re + (double)im * (double)im;
                    const int t = t0 + fg * 32 + r * 16 + (lane >> 4) + 4 * i;
                    if (t < T && bin < npw) pw[((size_t)b * T + t) * npw + bin] = (float)pwr;
                }
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

extern "C" int wt_logmel_create(int device, int n_fft, int hop, int n_mels, int n_frames, const double* window,
                                const float* filters, int npw, wt_logmel** out) {
    if (!out || !window || !filters) return wt_set_error(WT_E_INVALID, "wt_logmel_create: null argument");
    *out = nullptr;
    const int n_bins = n_fft / 2 + 1, nbt = (n_bins + 15) / 16;
    if (n_fft <= 0 || (n_fft & 3) || hop <= 0 || n_mels <= 0 || n_mels > 128 || n_frames <= 0 || npw < n_bins || (npw & 3) ||
        npw > nbt * 16 || nbt > 2 * DFT_MAX_BT || (size_t)n_fft * 8 + ((size_t)(DFT_FRAMES - 1) * hop + n_fft) * 4 > 160 * 1024)
        return wt_set_error(WT_E_INVALID, "wt_logmel_create: bad geometry (n_fft %d hop %d n_mels %d frames %d npw %d)", n_fft, hop,
                            n_mels, n_frames, npw);
    wt::DeviceGuard guard(device);
    if (guard.err != hipSuccess) return wt_set_error(WT_E_HIP, "hipSetDevice(%d) failed", device);
    wt_logmel* h = new wt_logmel();
    h->device = device; h->n_fft = n_fft; h->hop = hop; h->n_mels = n_mels; h->n_frames = n_frames; h->n_bins = n_bins;
    h->n_samples = n_frames * hop; h->npw = npw; h->nbt = nbt;
    // real-input DFT matrix in fp64, in the fragment order of dft_power_f64_kernel: tile n = 2 * bin_tile + part (0: cos, 1: -sin),
    // k-step kk, lane l  ->  row (bin) = 16 * bin_tile + (l & 15), column (sample) = 4 * kk + (l >> 4); bins >= n_bins are zero rows
    const int ksteps = n_fft / 4;
    const size_t nfrag = (size_t)2 * nbt * ksteps * 64, nf = (size_t)n_mels * npw;
    std::vector<double> frag(nfrag, 0.0);
    const double two_pi = 6.283185307179586476925286766559;
    for (int n = 0; n < 2 * nbt; ++n)
        for (int kk = 0; kk < ksteps; ++kk)
            for (int l = 0; l < 64; ++l) {
                const int bin = 16 * (n >> 1) + (l & 15), k = 4 * kk + (l >> 4);
                if (bin >= n_bins) continue;
                const double ang = two_pi * (double)(((long long)bin * k) % n_fft) / (double)n_fft;
                frag[((size_t)n * ksteps + kk) * 64 + l] = (n & 1) ? -sin(ang) : cos(ang);
            }
    if (hipMalloc((void**)&h->frag, nfrag * 8) != hipSuccess || hipMalloc((void**)&h->window, (size_t)n_fft * 8) != hipSuccess ||
        hipMalloc((void**)&h->filt, nf * 4) != hipSuccess) {
        wt_logmel_destroy(h);  // frees whichever tables were already allocated
        return wt_set_error(WT_E_NOMEM, "wt_logmel_create: table allocation failed");
    }
    hipStream_t up = nullptr;   // private non-blocking stream: no legacy-stream work while other host threads may be capturing graphs
    bool ok = hipStreamCreateWithFlags(&up, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMemcpyAsync(h->frag, frag.data(), nfrag * 8, hipMemcpyHostToDevice, up) == hipSuccess;
    ok = ok && hipMemcpyAsync(h->window, window, (size_t)n_fft * 8, hipMemcpyHostToDevice, up) == hipSuccess;
    ok = ok && hipMemcpyAsync(h->filt, filters, nf * 4, hipMemcpyHostToDevice, up) == hipSuccess;
    ok = ok && hipStreamSynchronize(up) == hipSuccess;
    if (up) hipStreamDestroy(up);
    if (!ok) {
        wt_logmel_destroy(h);
        return wt_set_error(WT_E_HIP, "wt_logmel_create: table upload failed");
    }
    const int smem = n_fft * 8 + ((DFT_FRAMES - 1) * hop + n_fft) * 4;
    if (smem > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(dft_power_f64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
        wt_logmel_destroy(h);
        return wt_set_error(WT_E_HIP, "wt_logmel_create: cannot reserve %d bytes of LDS", smem);
    }
    *out = h;
    return WT_OK;
}

extern "C" void wt_logmel_destroy(wt_logmel* h) {
    if (!h) return;
    wt::DeviceGuard guard(h->device);
    for (void* p : {(void*)h->frag, (void*)h->window, (void*)h->filt, (void*)h->pw}) if (p) hipFree(p);
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
        if (h->pw) hipFree(h->pw);
        const size_t floats = M * h->npw + M * h->n_mels + (size_t)batch * nblk + 64;
        if (hipMalloc((void**)&h->pw, floats * 4) != hipSuccess) {
            h->pw = nullptr; h->cap = 0;
            return wt_set_error(WT_E_NOMEM, "wt_logmel_forward: workspace allocation (%zu bytes) failed", floats * 4);
        }
        h->mel = h->pw + M * h->npw;
        h->bmax = h->mel + M * h->n_mels;
        h->cap = batch;
    }
    const int smem = h->n_fft * 8 + ((DFT_FRAMES - 1) * h->hop + h->n_fft) * 4;
    hipLaunchKernelGGL(dft_power_f64_kernel, dim3((T + DFT_FRAMES - 1) / DFT_FRAMES, batch), dim3(256), smem, s, audio, n_in, h->n_samples,
                       h->window, h->frag, h->pw, T, h->n_fft, h->hop, h->npw, h->nbt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wt_set_error(WT_E_HIP, "DFT kernel launch failed: %s", hipGetErrorString(e));
    // mel projection: 201-term sums of non-negative products -- no cancellation, the fp32 MFMA GEMM is within 1e-6 relative
    GemmParams g;
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
