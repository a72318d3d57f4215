// Shared device/host helpers for the gfx950 F5-TTS sampling engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define F5_WAVE 64

// GEMM epilogues (csrc/gemm.hip); "BF16" in a name means "the 16-bit operand type of the build" (bf16 or fp16, op16.hpp)
enum F5Epi : int {
    EPI_F32 = 0,         // out_f32 = acc + bias
    EPI_BF16 = 1,        // out_bf  = bf16(acc + bias)
    EPI_GELU_TANH = 2,   // out_bf  = bf16(gelu_tanh(acc + bias))                 (dit.py:94-99)
    EPI_GELU_ERF = 3,    // out_f32 = gelu_erf(acc + bias)                        (convnext_v2.py:50-51)
    EPI_RESID_GATE = 4,  // out_f32 += gate[col] * ((acc + bias) * keep[row])     (dit.py:172-173,319,323)
    EPI_QKV_ROPE = 5,    // q,k: rope(acc + bias) -> qk[row][col]; v -> vt[b,h][d][n] (dit.py:136-158)
    EPI_ADDROWS = 6,     // out_f32 = acc + addrows[row][col]; out_bf = bf16(same)  (dit.py:250 split GEMM)
    EPI_RESID_KEEP = 7,  // out_f32 = (resid[row][col] + acc + bias) * keep[row]  (convnext_v2.py:53-54, dit.py:225)
    EPI_GELU_ERF_BF16 = 8,  // out_bf = bf16(gelu_erf(acc + bias))              (Vocos ConvNeXt block)
};

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

// ---- host-side float <-> 16-bit operand bits (weight upload), round to nearest even -----------------------
// bf16: same rounding as torch's .to(bfloat16); fp16: same as torch's .to(float16) except that finite values beyond
// +-65504 saturate instead of becoming inf (the device producers saturate too, op16.hpp).
inline u16 f5_f2bf_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);  // NaN
    uint32_t r = u + 0x7fffu + ((u >> 16) & 1u);
    return (u16)(r >> 16);
}
inline float f5_bf_bits2f(u16 h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline u16 f5_f2h_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const u16 sign = (u16)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (u16)(sign | 0x7e00u);                    // NaN
    if (a >= 0x477ff000u) return (u16)(sign | 0x7bffu);                   // >= 65520 (would round to inf), inf: saturate
    if (a < 0x33000001u) return sign;                                     // <= 2^-25: rounds to zero
    const int e = (int)(a >> 23) - 127;                                   // unbiased exponent, in [-25, 15]
    uint32_t m = (a & 0x7fffffu) | 0x800000u;                             // 24-bit significand
    int shift = e >= -14 ? 13 : 13 + (-14 - e);                           // bits dropped (subnormals drop more)
    const uint32_t half = 1u << (shift - 1), mask = (1u << shift) - 1u;
    uint32_t q = m >> shift;
    const uint32_t rem = m & mask;
    if (rem > half || (rem == half && (q & 1u))) ++q;
    // q is the 11-bit significand (normal) or the subnormal mantissa; adding the biased exponent lets a carry roll over
    const uint32_t he = e >= -14 ? (uint32_t)(e + 15 - 1) << 10 : 0u;     // exponent field minus the implicit bit of q
    return (u16)(sign | (u16)(he + q));
}
inline float f5_h_bits2f(u16 h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 31u, m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) {
            u = sign;
        } else {                                                         // subnormal: m * 2^-24
            float f = (float)m * (1.0f / 16777216.0f);
            memcpy(&u, &f, 4);
            u |= sign;
        }
    } else if (e == 31) {
        u = sign | 0x7f800000u | (m << 13);
    } else {
        u = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// ---- activations (MLX semantics, see oracle/f5_oracle.py) ------------------------------------
__device__ inline float f5_silu(float x) { return x / (1.0f + __expf(-x)); }
// tanh-approximated GELU (nn.GELU(approx="tanh"), dit.py:94,309):  0.5 x (1 + tanh(u)) == x / (1 + exp(-2u)),
// u = sqrt(2/pi) (x + 0.044715 x^3).  One v_exp_f32 + one v_rcp_f32 instead of a libm tanhf (~40 instructions);
// relative error ~2e-7 (rcp + exp2 ulp), exact limits (exp -> inf gives 0, exp -> 0 gives x).
// Evaluated as x / (1 + exp2(x (C0 + C1 x^2))), C0 = -2 log2(e) sqrt(2/pi), C1 = 0.044715 C0: mul, fma, mul, exp2, add, rcp, mul.
// f5_gelu_tanh2 is the same arithmetic on TWO elements with packed-f32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32):
// bit-identical per element to the scalar form.  A wave issues one VALU instruction per ~5.4 cycles whatever its width
// (tools/probes/valu_rate.hip: v_fma_f32 and v_pk_fma_f32 both 5.4, v_exp_f32 / v_rcp_f32 9.5), and the GEMM epilogues are bound by
// exactly that issue rate (profiles/r05): two elements per instruction is half the arithmetic issue time.
typedef float f5_v2f __attribute__((ext_vector_type(2)));
#define F5_GELU_C0 (-2.3022081983f)
#define F5_GELU_C1 (-0.10294324f)
__device__ inline float f5_gelu_tanh(float x) {
    const float p = __builtin_fmaf(x * x, F5_GELU_C1, F5_GELU_C0);
    const float t = __builtin_amdgcn_exp2f(x * p);                      // exp(-2u)
    return x * __builtin_amdgcn_rcpf(1.0f + t);
}
__device__ __forceinline__ f5_v2f f5_gelu_tanh2(f5_v2f x) {
    const f5_v2f c1 = {F5_GELU_C1, F5_GELU_C1}, c0 = {F5_GELU_C0, F5_GELU_C0}, one = {1.0f, 1.0f};
    const f5_v2f a = x * __builtin_elementwise_fma(x * x, c1, c0);
    f5_v2f t = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    t = t + one;
    const f5_v2f r = {__builtin_amdgcn_rcpf(t[0]), __builtin_amdgcn_rcpf(t[1])};
    return x * r;
}
__device__ inline float f5_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
// Mish (nn.Mish, dit.py:35,37): x tanh(softplus(x)); with w = e^x: tanh(log(1+w)) = (w^2 + 2w) / (w^2 + 2w + 2)
__device__ inline float f5_mish(float x) {
    const float w = __builtin_amdgcn_exp2f(1.4426950408889634f * fminf(x, 30.0f));
    const float n = w * (w + 2.0f);
    return x * n * __builtin_amdgcn_rcpf(n + 2.0f);
}

// ---- wave helpers ----------------------------------------------------------------------------
// All-lanes butterfly reductions (offsets 32, 16, 8, 4, 2, 1; every lane ends with the result) WITHOUT the six LDS round trips of
// __shfl_xor (ds_bpermute_b32): lanes i / i + 32 exchange with v_permlane32_swap, rows of 16 lanes with v_permlane16_swap, and inside a
// row a rotation by 8 / 4 / 2 / 1 (DPP row_ror, fused into the add) delivers the butterfly partner because after the previous step the
// values are already symmetric under the larger offsets.  Same pairs added in the same order: bit-identical to the __shfl_xor loop
// (tools/probes: 262 144 wave sums compared on the GPU, 0 different).  One-row-per-wave kernels are latency chains at batch 1.
template <int CTRL>
__device__ __forceinline__ float f5_dpp_row(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float f5_xor32(float v, unsigned lane) {
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((lane & 32u) ? s[0] : s[1]);
}
__device__ __forceinline__ float f5_xor16(float v, unsigned lane) {
    const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((lane & 16u) ? s[0] : s[1]);
}
__device__ inline float f5_wave_sum(float v) {
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    v += f5_xor32(v, lane);
    v += f5_xor16(v, lane);
    v += f5_dpp_row<0x128>(v);   // row_ror:8
    v += f5_dpp_row<0x124>(v);   // row_ror:4
    v += f5_dpp_row<0x122>(v);   // row_ror:2
    v += f5_dpp_row<0x121>(v);   // row_ror:1
    return v;
}
__device__ inline float f5_wave_max(float v) {
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    v = fmaxf(v, f5_xor32(v, lane));
    v = fmaxf(v, f5_xor16(v, lane));
    v = fmaxf(v, f5_dpp_row<0x128>(v));
    v = fmaxf(v, f5_dpp_row<0x124>(v));
    v = fmaxf(v, f5_dpp_row<0x122>(v));
    v = fmaxf(v, f5_dpp_row<0x121>(v));
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
