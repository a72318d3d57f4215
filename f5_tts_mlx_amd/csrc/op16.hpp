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
__device__ inline uint32_t f5_pack2_lo(float a, float b) {
    const float ra = a - static_cast<float>(static_cast<op16_t>(a));
    const float rb = b - static_cast<float>(static_cast<op16_t>(b));
    return f5_pack2_bounded(ra, rb);
}
#endif
}  // namespace F5_NS
#endif
