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
    wave += (long)blockIdx.y * L;                                   // batch element: waves are [B][L], outputs [B][frames][n_mels]
    out += (long)blockIdx.y * (long)gridDim.x * n_mels;
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

extern "C" int f5_mel_spectrogram_batch(const float* wave, int B, int64_t L, const float* window, const float* filterbank, int n_fft,
                                        int hop, int n_mels, float* out, void* stream) {
    F5_REQUIRE(wave && window && filterbank && out, "mel: null pointer");
    F5_REQUIRE(n_fft == MEL_NFFT, "mel: only n_fft = 1024 is supported (got %d)", n_fft);
    F5_REQUIRE(hop > 0 && n_mels > 0 && B >= 1 && B <= 65535, "mel: bad hop / n_mels / batch");
    const long frames = L / hop;
    if (frames <= 0) return 0;
    hipLaunchKernelGGL(mel_kernel, dim3((unsigned)frames, (unsigned)B), dim3(256), 0, (hipStream_t)stream, wave, (long)L, window,
                       filterbank, hop, n_mels, out);
    F5_LAUNCH_CHECK();
    return 0;
}
extern "C" int f5_mel_spectrogram(const float* wave, int64_t L, const float* window, const float* filterbank, int n_fft, int hop,
                                  int n_mels, float* out, void* stream) {
    return f5_mel_spectrogram_batch(wave, 1, L, window, filterbank, n_fft, hop, n_mels, out, stream);
}

// =================================================================================================
// Vocos ISTFT head (third-party vocos_mlx `Vocos.decode`, call site cfm.py:399-400; restated from the
// upstream gemelo-ai/vocos ISTFTHead): x[frame][0:513] = log-magnitude, x[frame][513:1026] = phase;
// S = min(exp(mag), 1e2) * (cos p + i sin p); wave = istft(S, n_fft=1024, hop=256, hann, center=True).
// Kernel 1: per frame, Hermitian-extend + inverse FFT in LDS, multiply by the window.
// Kernel 2: overlap-add of <= n_fft/hop frames per sample, divide by the window-square envelope, trim n_fft/2.
// =================================================================================================
__global__ __launch_bounds__(256) void istft_frames_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ window,
                                                           float* __restrict__ frames) {
    __shared__ float re[MEL_NFFT], im[MEL_NFFT];
    __shared__ float twc[MEL_NFFT / 2], tws[MEL_NFFT / 2];
    const int tid = threadIdx.x;
    const long f = blockIdx.x;
    const float* xf = x + f * (long)ldx;
    const int nbin = MEL_NFFT / 2 + 1;
    for (int k = tid; k < MEL_NFFT; k += 256) {
        const int kk = k < nbin ? k : MEL_NFFT - k;             // Hermitian extension X[N-k] = conj(X[k])
        const float mag = fminf(expf(xf[kk]), 100.0f);
        float sn, cs;
        sincosf(xf[nbin + kk], &sn, &cs);
        float vr = mag * cs, vi = mag * sn;
        if (k >= nbin) vi = -vi;
        if (k == 0 || k == MEL_NFFT / 2) vi = 0.0f;            // irfft ignores the imaginary part of DC / Nyquist
        const int r = (int)(__brev((unsigned)k) >> (32 - MEL_LOG2));
        re[r] = vr;
        im[r] = vi;
    }
    for (int i = tid; i < MEL_NFFT / 2; i += 256) {
        float sn, cs;
        sincosf(6.283185307179586f * (float)i / (float)MEL_NFFT, &sn, &cs);   // inverse transform: +i
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
    for (int i = tid; i < MEL_NFFT; i += 256) frames[f * MEL_NFFT + i] = re[i] * (1.0f / MEL_NFFT) * window[i];
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ window,
                                                        float* __restrict__ wave, int nframes, int hop, long out_len) {
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= out_len) return;
    frames += (long)blockIdx.y * nframes * MEL_NFFT;            // batch element
    wave += (long)blockIdx.y * out_len;
    const long pos = s + MEL_NFFT / 2;                          // position in the untrimmed signal
    int f1 = (int)(pos / hop);
    if (f1 > nframes - 1) f1 = nframes - 1;
    long f0l = (pos - MEL_NFFT) / hop + 1;
    if (pos - MEL_NFFT < 0) f0l = 0;
    const int f0 = (int)(f0l < 0 ? 0 : f0l);
    float acc = 0.0f, env = 0.0f;
    for (int f = f0; f <= f1; ++f) {
        const int off = (int)(pos - (long)f * hop);
        if (off >= 0 && off < MEL_NFFT) {
            acc += frames[(long)f * MEL_NFFT + off];
            const float w = window[off];
            env += w * w;
        }
    }
    wave[s] = acc / env;
}

// x [B * nframes][ldx] (rows of utterance b are b*nframes ...), frames_scratch [B * nframes][n_fft], wave [B][hop * (nframes - 1)]:
// two launches for the whole batch
int f5_launch_istft(const float* x, int ldx, const float* window, float* frames_scratch, float* wave, int B, int nframes, int hop,
                    hipStream_t s) {
    F5_REQUIRE(x && window && frames_scratch && wave, "istft: null pointer");
    F5_REQUIRE(B >= 1 && B <= 65535 && nframes >= 1 && hop > 0 && ldx >= MEL_NFFT + 2, "istft: bad sizes");
    hipLaunchKernelGGL(istft_frames_kernel, dim3((unsigned)(B * nframes)), dim3(256), 0, s, x, ldx, window, frames_scratch);
    const long out_len = (long)hop * (nframes - 1);
    if (out_len > 0)
        hipLaunchKernelGGL(istft_ola_kernel, dim3(f5_cdiv(out_len, 256), (unsigned)B), dim3(256), 0, s, frames_scratch, window, wave,
                           nframes, hop, out_len);
    F5_LAUNCH_CHECK();
    return 0;
}
extern "C" int f5_op_istft_batch(const float* x, int ldx, const float* window, float* frames_scratch, float* wave, int B, int nframes,
                                 int n_fft, int hop, void* stream) {
    F5_REQUIRE(n_fft == MEL_NFFT, "istft: only n_fft = 1024 is supported (got %d)", n_fft);
    return f5_launch_istft(x, ldx, window, frames_scratch, wave, B, nframes, hop, (hipStream_t)stream);
}
extern "C" int f5_op_istft(const float* x, int ldx, const float* window, float* frames_scratch, float* wave, int nframes,
                           int n_fft, int hop, void* stream) {
    return f5_op_istft_batch(x, ldx, window, frames_scratch, wave, 1, nframes, n_fft, hop, stream);
}
