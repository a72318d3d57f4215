// 16-bit MFMA operand type of this build of the kernels.  Every kernel source (gemm / attention / convpos / rowops) is compiled
// TWICE: with F5_F16=0 the operands are bf16 (namespace f5bf, precisions "bf16", "bf16x3", "mxfp8") and with F5_F16=1 they
// are IEEE fp16 (namespace f5hf, precision "f16": v_mfma_f32_32x32x16_f16 runs at the bf16 rate and keeps 11 significand
// bits instead of 8, which is what brings the one-pass mode inside the 1e-3 mel-L1 parity gate; fp16's range is enough for
// LN-normalised activations, RoPE'd q/k, softmax probabilities and weights, and every producer saturates at +-65504).
// The engine (compiled once) includes the op headers under both namespaces.  This header has no include guard on purpose.
#include "common.hpp"

#ifndef F5_F16
#define F5_F16 0
#endif
#undef F5_NS
#undef F5_MFMA32
#if F5_F16
#define F5_NS f5hf
#define F5_MFMA32 __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
#define F5_NS f5bf
#define F5_MFMA32 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif

#if F5_F16
#ifndef F5_OP16_HPP_F16
#define F5_OP16_HPP_F16
#define F5_OP16_HPP_BODY
#endif
#else
#ifndef F5_OP16_HPP_BF16
#define F5_OP16_HPP_BF16
#define F5_OP16_HPP_BODY
#endif
#endif
#ifdef F5_OP16_HPP_BODY
#undef F5_OP16_HPP_BODY
namespace F5_NS {
#if F5_F16
typedef _Float16 op16_t;
#else
typedef __bf16 op16_t;
#endif
typedef op16_t op16x8 __attribute__((ext_vector_type(8)));
typedef op16_t op16x4 __attribute__((ext_vector_type(4)));
typedef op16_t f5_op16x2 __attribute__((ext_vector_type(2)));
typedef float f5_f32x2 __attribute__((ext_vector_type(2)));

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// float -> operand type, round to nearest even.  fp16 saturates instead of producing inf (|x| > 65504 does not occur on
// this path with sane weights, but an inf would poison a whole attention row; the clamp is one v_med3_f32)
__device__ __forceinline__ float f5_sat(float f) {
#if F5_F16
    return __builtin_amdgcn_fmed3f(f, -65504.0f, 65504.0f);
#else
    return f;
#endif
}
__device__ inline op16_t f5_f2op(float f) { return static_cast<op16_t>(f5_sat(f)); }
__device__ inline float f5_op2f(op16_t h) { return static_cast<float>(h); }
// hi/lo split for the 3-pass precision mode: x ~= hi + lo
__device__ inline void f5_split(float f, op16_t& hi, op16_t& lo) {
    hi = static_cast<op16_t>(f5_sat(f));
    lo = static_cast<op16_t>(f - static_cast<float>(hi));
}
// pack two floats into an operand pair (element 0 in the low half): one v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 (RNE)
__device__ inline uint32_t f5_pack2(float a, float b) {
    const f5_f32x2 v = {f5_sat(a), f5_sat(b)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f5_op16x2));
}
// no clamp: for values known to be bounded (softmax probabilities, normalised attention output)
__device__ inline uint32_t f5_pack2_bounded(float a, float b) {
    const f5_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f5_op16x2));
}
// ---- range detector of the fp16 build (round 6).  Every producer of a 16-bit MFMA operand keeps the largest magnitude it packs
// (one v_max3_f32 per PAIR: the clamp above hides an overflow, this does not) and ORs F5_STATUS_SATURATED into the call's status
// word when a value beyond +-65 504 went through f5_sat -- once per wave and kernel, and only on a hit.  In the bf16 build the
// tracker is empty and every call compiles away (bf16 has fp32's range).
#define F5_STATUS_SATURATED 4
#if F5_F16
struct f5_sat_t {
    float m = 0.0f;
};
__device__ __forceinline__ void f5_sat_see(f5_sat_t& t, float a) { t.m = fmaxf(t.m, fabsf(a)); }
__device__ __forceinline__ void f5_sat_see2(f5_sat_t& t, float a, float b) { t.m = fmaxf(fmaxf(t.m, fabsf(a)), fabsf(b)); }
__device__ __forceinline__ void f5_sat_commit(const f5_sat_t& t, int* flag) {
    // every lane that saw one reports it (a rare path; callers may sit inside divergent code, so no "lane 0 speaks for the wave");
    // inf counts, NaN does not reach here: fmaxf drops it
    if (flag != nullptr && t.m > 65504.0f) atomicOr(flag, F5_STATUS_SATURATED);
}
// the same detector WITHOUT a vector register: the hit mask of the wave is ORed into a scalar pair (one v_max_f32 + one v_cmp per
// pair instead of one v_max3_f32).  For code that has no VGPR to spare -- the V tiles of the 256 x 256 QKV kernel sit at 256 registers
// and spilled with the one-register tracker (tests/test_isa.py).
struct f5_sat_s {
    unsigned long long hit = 0;
};
__device__ __forceinline__ void f5_sat_see2(f5_sat_s& t, float a, float b) { t.hit |= __builtin_amdgcn_ballot_w64(fmaxf(fabsf(a), fabsf(b)) > 65504.0f); }
__device__ __forceinline__ void f5_sat_commit(const f5_sat_s& t, int* flag) {
    if (flag != nullptr && t.hit != 0) atomicOr(flag, F5_STATUS_SATURATED);
}
#else
struct f5_sat_t {};
struct f5_sat_s {};
__device__ __forceinline__ void f5_sat_see2(f5_sat_s&, float, float) {}
__device__ __forceinline__ void f5_sat_commit(const f5_sat_s&, int*) {}
__device__ __forceinline__ void f5_sat_see(f5_sat_t&, float) {}
__device__ __forceinline__ void f5_sat_see2(f5_sat_t&, float, float) {}
__device__ __forceinline__ void f5_sat_commit(const f5_sat_t&, int*) {}
#endif
__device__ inline uint32_t f5_pack2(float a, float b, f5_sat_t& t) {
    f5_sat_see2(t, a, b);
    return f5_pack2(a, b);
}
__device__ inline uint32_t f5_pack2(float a, float b, f5_sat_s& t) {
    f5_sat_see2(t, a, b);
    return f5_pack2(a, b);
}
__device__ inline void f5_split(float f, op16_t& hi, op16_t& lo, f5_sat_t& t) {
    f5_sat_see(t, f);
    f5_split(f, hi, lo);
}
__device__ inline uint32_t f5_pack2_lo(float a, float b) {
    const float ra = a - static_cast<float>(static_cast<op16_t>(a));
    const float rb = b - static_cast<float>(static_cast<op16_t>(b));
    return f5_pack2_bounded(ra, rb);
}
#endif
}  // namespace F5_NS
#endif
