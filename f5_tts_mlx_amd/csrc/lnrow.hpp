// LayerNorm (no affine) + adaLN modulation of ONE row held by ONE wave: y = LN(x) * (1 + scale) + shift (dit.py:270,289,321).
// Shared by ln_modulate_kernel (rowops.hip) and by the LN tail fused behind the residual GEMMs (gemm.hip) so that both produce
// the same bits: the fused and the stand-alone path are interchangeable (graph == eager, fused == unfused are tested bitwise).
#include "op16.hpp"

#if F5_F16
#ifndef F5_LNROW_HPP_F16
#define F5_LNROW_HPP_F16
#define F5_LNROW_HPP_BODY
#endif
#else
#ifndef F5_LNROW_HPP_BF16
#define F5_LNROW_HPP_BF16
#define F5_LNROW_HPP_BODY
#endif
#endif
#ifdef F5_LNROW_HPP_BODY
#undef F5_LNROW_HPP_BODY
namespace F5_NS {

// v: the row, NV float4 per lane (lane l holds columns i*256 + 4*l .. + 3); dim = NV * 256
template <int NV>
__device__ __forceinline__ void f5_ln_modulate_row(const f32x4 (&v)[NV], const float* __restrict__ scale,
                                                   const float* __restrict__ shift, op16_t* __restrict__ out_hi,
                                                   op16_t* __restrict__ out_lo, size_t row, int lane, float eps, int* sat_flag = nullptr) {
    constexpr int DIM = NV * 256;
    // the modulation vectors do not depend on the row: all of them are requested before the reductions.  Loaded chunk by chunk
    // inside the output loop, each chunk cost a full memory round trip behind the previous chunk's store (the compiler's
    // `s_waitcnt vmcnt(0)` for the new loads also waits for that store)
    f32x4 scv[NV], shv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        scv[i] = *reinterpret_cast<const f32x4*>(scale + i * 256 + lane * 4);
        shv[i] = *reinterpret_cast<const f32x4*>(shift + i * 256 + lane * 4);
    }
    __builtin_amdgcn_sched_barrier(0);                 // (the scheduler otherwise sinks them behind the reductions again)
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = f5_wave_sum(sum) * (1.0f / DIM);
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            sq += d * d;
        }
    const float var = f5_wave_sum(sq) * (1.0f / DIM);
    const float rstd = rsqrtf(var + eps);
    f5_sat_t trk;                                      // fp16 build: the modulated row is an MFMA operand without a hard bound (op16.hpp)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        const f32x4 sc = scv[i], sh = shv[i];
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * (1.0f + sc[e]) + sh[e];
        *reinterpret_cast<u32x2*>(out_hi + row * DIM + c) = u32x2{f5_pack2(y[0], y[1], trk), f5_pack2(y[2], y[3], trk)};
        if (out_lo) *reinterpret_cast<u32x2*>(out_lo + row * DIM + c) = u32x2{f5_pack2_lo(y[0], y[1]), f5_pack2_lo(y[2], y[3])};
    }
    f5_sat_commit(trk, sat_flag);
}

}  // namespace F5_NS
#endif
