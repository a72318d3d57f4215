// Initial noise of F5TTS.sample (cfm.py:369-375): per batch element `mx.random.seed(seed); mx.random.normal((mel, dur))`
// -- a CHANNEL-major draw -- zero padded to N frames and transposed to (N, mel).  MLX's generator is third-party arithmetic
// (mlx/random.cpp, not in /root/reference and not installable here); this is the published algorithm as restated in
// f5_tts_mlx_amd/rng.py, which stays the host-side definition the tests compare against:
//   key(seed) = (seed >> 32, seed & 0xffffffff);  key, sub = split(key)   (threefry2x32 over counters (0,2), (1,3));
//   bits(n)   = threefry2x32(sub, (i, i + half)), first outputs then second outputs, half = ceil(n / 2);
//   u = min(float(bits) / float(2^32 - 1), nextafter(1, 0)) * (1 - lo) + lo,  lo = nextafter(-1, 0)        (float32 steps);
//   z = sqrt(2) * erfinv(u)  evaluated in float64 and rounded to float32 once.
// Integer path bit-exact; the float path follows rng.py's rounding steps (no fused multiply-add).  One thread per counter pair.
#include "../../include/f5tts_hip.h"
#include "host_common.hpp"

__host__ __device__ inline void f5_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
    const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    const int rot[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
    uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
    for (int i = 0; i < 5; ++i) {
        for (int j = 0; j < 4; ++j) {
            const int r = rot[i & 1][j];
            x0 += x1;
            x1 = ((x1 << r) | (x1 >> (32 - r))) ^ x0;
        }
        x0 += ks[(i + 1) % 3];
        x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
    }
    o0 = x0;
    o1 = x1;
}

// erfinv in float64: Giles' single-precision polynomial as the starting point, two Halley steps on erf(x) - y = 0
__device__ inline double f5_erfinv(double y) {
    float w = -__logf((1.0f - (float)y) * (1.0f + (float)y));
    float p;
    if (w < 5.0f) {
        w -= 2.5f;
        p = 2.81022636e-08f;
        p = 3.43273939e-07f + p * w;
        p = -3.5233877e-06f + p * w;
        p = -4.39150654e-06f + p * w;
        p = 0.00021858087f + p * w;
        p = -0.00125372503f + p * w;
        p = -0.00417768164f + p * w;
        p = 0.246640727f + p * w;
        p = 1.50140941f + p * w;
    } else {
        w = sqrtf(w) - 3.0f;
        p = -0.000200214257f;
        p = 0.000100950558f + p * w;
        p = 0.00134934322f + p * w;
        p = -0.00367342844f + p * w;
        p = 0.00573950773f + p * w;
        p = -0.0076224613f + p * w;
        p = 0.00943887047f + p * w;
        p = 1.00167406f + p * w;
        p = 2.83297682f + p * w;
    }
    double x = (double)p * y;
    if (!(fabs(x) < 1e300)) x = y < 0 ? -6.0 : 6.0;              // |y| rounded to 1 inside the float polynomial
    for (int it = 0; it < 2; ++it) {
        const double e = erf(x) - y;
        const double d = 1.1283791670955126 * exp(-x * x);        // 2 / sqrt(pi) * exp(-x^2)
        x -= e / (d + x * e);                                       // Halley: f'' / f' = -2x
    }
    return x;
}

__device__ inline float f5_bits_to_normal(uint32_t bits) {
    float u = __fdiv_rn((float)bits, 4294967295.0f);                // float(2^32 - 1) == 2^32 in float32, as in numpy
    u = fminf(u, 0.99999994f);                                      // nextafter(1, 0)
    const float lo = -0.99999994f;                                  // nextafter(-1, 0)
    u = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, lo), u), lo);
    return (float)((double)1.41421354f * f5_erfinv((double)u));     // float32(sqrt(2)) promoted, one final rounding
}

// y0[b][t][c] for t < dur[b] from value index i = c * dur + t of the (mel, dur) draw; zeros for dur <= t < N
__global__ __launch_bounds__(256) void noise_normal_kernel(const uint32_t* __restrict__ subkeys, const int* __restrict__ durs, int N, int mel,
                                                           float* __restrict__ y0) {
    const int b = blockIdx.y;
    const int dur = durs[b];
    const long n = (long)mel * dur;
    const long half = (n + 1) / 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float* yb = y0 + (size_t)b * N * mel;
    if (i < half) {
        uint32_t o0, o1;
        f5_threefry2x32(subkeys[2 * b], subkeys[2 * b + 1], (uint32_t)i, (uint32_t)(i + half), o0, o1);
        {
            const int c = (int)(i / dur), t = (int)(i - (long)c * dur);
            yb[(size_t)t * mel + c] = f5_bits_to_normal(o0);
        }
        const long i2 = i + half;
        if (i2 < n) {
            const int c = (int)(i2 / dur), t = (int)(i2 - (long)c * dur);
            yb[(size_t)t * mel + c] = f5_bits_to_normal(o1);
        }
    }
    // zero padding: the same grid covers (N - dur) * mel pad elements of this batch element
    const long npad = (long)(N - dur) * mel;
    for (long q = i; q < npad; q += (long)gridDim.x * 256) yb[(size_t)dur * mel + q] = 0.0f;
}

// seeds: one 64-bit seed per batch element (the reference re-seeds with the SAME seed for every element, cfm.py:371-373: pass it
// B times); durations: host ints; y0: device [B][N][mel].  Kernel-only (host scalars travel as launch arguments through a small
// staging kernel), safe inside a stream capture.
extern "C" int f5_noise_normal(const uint64_t* seeds, int B, const int32_t* durations, int N, int mel, float* y0, void* scratch_words,
                               void* stream) {
    F5_REQUIRE(seeds && durations && y0 && scratch_words, "noise: null pointer");
    F5_REQUIRE(B >= 1 && B <= 256 && N >= 1 && mel >= 1, "noise: bad sizes (B <= 256)");
    uint32_t w[3 * 256];
    long maxn = 0;
    for (int b = 0; b < B; ++b) {
        F5_REQUIRE(durations[b] >= 1 && durations[b] <= N, "noise: duration[%d] = %d outside [1, N = %d]", b, durations[b], N);
        const uint32_t k0 = (uint32_t)(seeds[b] >> 32), k1 = (uint32_t)(seeds[b] & 0xFFFFFFFFu);
        // key, sub = split(key): bits(4) = threefry over counter pairs (0, 2), (1, 3), laid out [first outputs | second outputs]
        uint32_t a0, a1, b0, b1;
        f5_threefry2x32(k0, k1, 0u, 2u, a0, a1);
        f5_threefry2x32(k0, k1, 1u, 3u, b0, b1);
        (void)a0;
        (void)b0;
        w[2 * b] = a1;                     // bits = [a0, b0, a1, b1] -> sub key = (bits[2], bits[3])
        w[2 * b + 1] = b1;
        w[2 * B + b] = (uint32_t)durations[b];
        const long n = (long)mel * durations[b];
        if (n > maxn) maxn = n;
    }
    hipStream_t s = (hipStream_t)stream;
    uint32_t* dev = (uint32_t*)scratch_words;    // >= 3 * B words
    RC(f5_launch_stage_words(w, (size_t)3 * B, dev, s));
    const long half = (maxn + 1) / 2;
    const long padmax = (long)N * mel;
    long threads = half > 4096 ? half : 4096;
    if (threads > padmax) threads = padmax > half ? padmax : half;
    hipLaunchKernelGGL(noise_normal_kernel, dim3((unsigned)f5_cdiv(threads, 256), (unsigned)B), dim3(256), 0, s, dev, (const int*)(dev + 2 * B),
                       N, mel, y0);
    F5_LAUNCH_CHECK();
    return 0;
}
