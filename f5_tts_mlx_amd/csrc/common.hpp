// Shared device/host helpers for the gfx950 F5-TTS sampling engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define F5_WAVE 64

// ---- error handling (never throw across the C ABI) -------------------------------------------
void f5_set_error(const char* fmt, ...);
#define F5_HIP_CHECK(expr)                                                                        \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            f5_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)
#define F5_LAUNCH_CHECK() F5_HIP_CHECK(hipGetLastError())
#define F5_REQUIRE(cond, ...)                                                                     \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            f5_set_error(__VA_ARGS__);                                                            \
            return 2;                                                                             \
        }                                                                                         \
    } while (0)

// ---- bf16 helpers ----------------------------------------------------------------------------
// float -> bf16 round-to-nearest-even (same rounding as torch's .to(bfloat16)); hi/lo split for the
// 3-pass "bf16x3" precision mode: x ~= hi + lo with |x - hi - lo| <= 2^-17 |x|.
__host__ __device__ inline u16 f5_f2bf_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);  // NaN
    uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
    return (u16)(r >> 16);
}
__host__ __device__ inline float f5_bf_bits2f(u16 h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__device__ inline bf16_t f5_f2bf(float f) { return static_cast<bf16_t>(f); }
__device__ inline float f5_bf2f(bf16_t h) { return static_cast<float>(h); }
__device__ inline void f5_split(float f, bf16_t& hi, bf16_t& lo) {
    hi = static_cast<bf16_t>(f);
    lo = static_cast<bf16_t>(f - static_cast<float>(hi));
}

// pack two floats to bf16 pairs (element 0 in the low half); *_lo packs the rounding residuals
typedef float f5_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 f5_bf16x2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t f5_pack2(float a, float b) {
    // one v_cvt_pk_bf16_f32 (RNE); the scalar-cast form compiled to two converts + an SDWA or
    const f5_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f5_bf16x2));
}
__device__ inline uint32_t f5_pack2_lo(float a, float b) {
    const float ra = a - static_cast<float>(static_cast<bf16_t>(a));
    const float rb = b - static_cast<float>(static_cast<bf16_t>(b));
    return f5_pack2(ra, rb);
}

// ---- activations (MLX semantics, see oracle/f5_oracle.py) ------------------------------------
__device__ inline float f5_silu(float x) { return x / (1.0f + __expf(-x)); }
// tanh-approximated GELU (nn.GELU(approx="tanh"), dit.py:94,309):  0.5 x (1 + tanh(u)) == x / (1 + exp(-2u)),
// u = sqrt(2/pi) (x + 0.044715 x^3).  One v_exp_f32 + one v_rcp_f32 instead of a libm tanhf (~40 instructions);
// relative error ~2e-7 (rcp + exp2 ulp), exact limits (exp -> inf gives 0, exp -> 0 gives x).
__device__ inline float f5_gelu_tanh(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u = k0 * (x + k1 * x * x * x);
    const float t = __builtin_amdgcn_exp2f(-2.8853900817779268f * u);   // exp(-2u)
    return x * __builtin_amdgcn_rcpf(1.0f + t);
}
__device__ inline float f5_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
// Mish (nn.Mish, dit.py:35,37): x tanh(softplus(x)); with w = e^x: tanh(log(1+w)) = (w^2 + 2w) / (w^2 + 2w + 2)
__device__ inline float f5_mish(float x) {
    const float w = __builtin_amdgcn_exp2f(1.4426950408889634f * fminf(x, 30.0f));
    const float n = w * (w + 2.0f);
    return x * n * __builtin_amdgcn_rcpf(n + 2.0f);
}

// ---- wave helpers ----------------------------------------------------------------------------
__device__ inline float f5_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float f5_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int f5_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- MX (OCP microscaling) helpers: e4m3 elements, one E8M0 scale per 32 consecutive elements -------------------------
// scale exponent e = ceil(log2(amax / 448)) so that amax / 2^e lies in (224, 448]: no element saturates (the OCP
// reference rule floor(log2(amax)) - 8 clips elements in (448, 512) * 2^e instead).  Returned as the biased E8M0 byte.
__device__ __forceinline__ int f5_mx_scale_byte(float amax) {
    const float y = amax * 0x1.24924ap-9f;            // amax / 448 with the fp32-rounded reciprocal (oracle uses the same constant)
    const uint32_t b = __float_as_uint(y);
    int e = (int)((b >> 23) & 255u) + ((b & 0x7FFFFFu) ? 1 : 0);    // biased exponent of the next power of two >= y
    if ((b >> 23) == 0) e = 0;                        // zero / denormal block
    return e > 254 ? 254 : e;
}
__device__ __forceinline__ float f5_mx_inv_scale(int e8) {          // 2^(127 - e8): multiply elements by this before the cast
    return __uint_as_float((uint32_t)(254 - e8) << 23);              // e8 in [0, 254] -> exponent field 254..0 (2^-127 flushes: block is zero)
}
__device__ __forceinline__ uint32_t f5_pack4_fp8(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.0f), 448.0f);              // v_cvt_pk_fp8_f32 does not saturate (probe: 500 -> NaN)
    b = fminf(fmaxf(b, -448.0f), 448.0f);
    c = fminf(fmaxf(c, -448.0f), 448.0f);
    d = fminf(fmaxf(d, -448.0f), 448.0f);
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (uint32_t)v;
}
