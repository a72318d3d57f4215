// Mel front-end (audio.py:115-210): zero ("constant") centre padding, frames of n_fft with hop,
// periodic Hann window, rfft magnitude, HTK mel filterbank matmul, log(max(., 1e-5)).
// One workgroup per frame: the 1024-point FFT runs in LDS (radix-2 DIT, fp32), the filterbank
// product and the log are fused behind it.  Output layout (frames, n_mels), last STFT frame dropped
// (audio.py:202) => frames = L / hop.
#include "../../include/f5tts_hip.h"
#include "common.hpp"

#define MEL_NFFT 1024
#define MEL_LOG2 10

__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ wave, long L, const float* __restrict__ window,
                                                  const float* __restrict__ fb, int hop, int n_mels, float* __restrict__ out) {
    __shared__ float re[MEL_NFFT], im[MEL_NFFT];
    __shared__ float twc[MEL_NFFT / 2], tws[MEL_NFFT / 2];
    const int tid = threadIdx.x;
    const long frame = blockIdx.x;
    const long start = frame * hop - MEL_NFFT / 2;
    for (int i = tid; i < MEL_NFFT; i += 256) {
        const long si = start + i;
        const float v = (si >= 0 && si < L) ? wave[si] * window[i] : 0.0f;
        const int r = (int)(__brev((unsigned)i) >> (32 - MEL_LOG2));
        re[r] = v;
        im[r] = 0.0f;
    }
    for (int i = tid; i < MEL_NFFT / 2; i += 256) {
        float sn, cs;
        sincosf(-6.283185307179586f * (float)i / (float)MEL_NFFT, &sn, &cs);
        twc[i] = cs;
        tws[i] = sn;
    }
    __syncthreads();
    for (int s = 1; s <= MEL_LOG2; ++s) {
        const int half = 1 << (s - 1);
        const int tstep = MEL_NFFT >> s;
        for (int j = tid; j < MEL_NFFT / 2; j += 256) {
            const int grp = j >> (s - 1), pos = j & (half - 1);
            const int i0 = (grp << s) + pos, i1 = i0 + half;
            const float wr = twc[pos * tstep], wi = tws[pos * tstep];
            const float xr = re[i1], xi = im[i1];
            const float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
            const float ur = re[i0], ui = im[i0];
            re[i0] = ur + tr;
            im[i0] = ui + ti;
            re[i1] = ur - tr;
            im[i1] = ui - ti;
        }
        __syncthreads();
    }
    // magnitude into re[0..512]
    for (int k = tid; k <= MEL_NFFT / 2; k += 256) {
        const float a = re[k], b = im[k];
        re[k] = sqrtf(a * a + b * b);
    }
    __syncthreads();
    const int nbin = MEL_NFFT / 2 + 1;
    for (int m = tid; m < n_mels; m += 256) {
        const float* f = fb + (size_t)m * nbin;
        float acc = 0.0f;
        for (int k = 0; k < nbin; ++k) acc += re[k] * f[k];
        out[frame * n_mels + m] = logf(fmaxf(acc, 1e-5f));
    }
}

extern "C" int f5_mel_spectrogram(const float* wave, int64_t L, const float* window, const float* filterbank, int n_fft, int hop,
                                  int n_mels, float* out, void* stream) {
    F5_REQUIRE(wave && window && filterbank && out, "mel: null pointer");
    F5_REQUIRE(n_fft == MEL_NFFT, "mel: only n_fft = 1024 is supported (got %d)", n_fft);
    F5_REQUIRE(hop > 0 && n_mels > 0, "mel: bad hop / n_mels");
    const long frames = L / hop;
    if (frames <= 0) return 0;
    hipLaunchKernelGGL(mel_kernel, dim3((unsigned)frames), dim3(256), 0, (hipStream_t)stream, wave, (long)L, window, filterbank,
                       hop, n_mels, out);
    F5_LAUNCH_CHECK();
    return 0;
}
