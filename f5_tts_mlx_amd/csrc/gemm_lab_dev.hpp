// Lab-only device code shared by gemm.hip and gemm_lab.hip (F5_LAB builds): the residual update by no-return L2 atomics.
#pragma once
#include "gemm.hpp"
#include "gemm_dev.hpp"
namespace F5_NS {
// x += gate * ((acc + bias) * keep) (EPI_RESID_GATE) with the ADD done by the L2's atomic units (global_atomic_add_f32 without
// return value), straight from the accumulator registers.  MEASURED SLOWER, kept as an experiment behind gemm flag 8
// (f5_debug_set_gemm_flags; tools/r2b_ab.py, profiles/r02/attention_nomax_and_resid_atomic_ab.txt).  Idea: the load / add / store forms
// (gemm_epilogue, staged_epilogue_resid) make every wave wait for its 8 B / element round trip to HBM at the end of its tile
// (490 MB per launch at batch 32, all 256 workgroups of a round enter the epilogue together, the matrix cores idle meanwhile);
// a no-return atomic is fire-and-forget, so the workgroup would retire and the CU's next tile start its main loop while the
// memory side applies the adds.  Every element receives exactly ONE add per launch (split-K partials are summed in LDS first),
// so the result is deterministic and equals the load / add / store form up to the product gate * v being rounded before the
// add.  Result on MI355X: out-proj 267 vs 193 us, FF2 360 vs 300 us at M = 59 968; 13.3 vs 12.6 and 18.5 vs 17.2 us at
// M = 1 874; sample() 1 320 vs 1 254 ms at batch 32 -- the L2 atomic units retire the 61 M adds of a launch at ~1.5 TB/s
// equivalent, slower than the 5.3 TB/s the plain read-modify-write reaches, and the queued atomics hold up the next tile's
// operand loads instead of hiding under its MFMAs.
// Lane layout of a 32x32 accumulator block: register r of lane (hi, lcol) is row 8*(r>>2) + 4*hi + (r&3), column lcol, so
// one instruction updates two 128-byte row segments.
template <int MBW, int NBW, bool GUARD>
__device__ __forceinline__ void atomic_epilogue_resid_impl(const F5GemmArgs& p, f32x16 (&acc)[MBW][NBW], int row0, int colbase,
                                                           int lane) {
    const int hi = lane >> 5, lcol = lane & 31;
    float bcol[NBW], gcol[NBW];
    bool colok[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int c = colbase + nb * 32 + lcol;
        colok[nb] = !GUARD || c < p.N;
        bcol[nb] = (p.bias != nullptr && colok[nb]) ? p.bias[c] : 0.0f;
        gcol[nb] = colok[nb] ? p.gate[c] : 0.0f;
    }
    const bool keep_words = p.rowkeep != nullptr && (reinterpret_cast<uintptr_t>(p.rowkeep) & 3) == 0;
    char* const xbase = reinterpret_cast<char*>(p.out_f32);
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int rowb = row0 + mb * 32 + rg * 8 + hi * 4;          // 4 consecutive rows, rowb % 4 == 0
            uint32_t kw = 0x01010101u;                                   // keep bytes of the 4 rows
            if (p.rowkeep != nullptr) {
                if (keep_words && (!GUARD || rowb + 3 < p.M)) {
                    kw = *reinterpret_cast<const uint32_t*>(p.rowkeep + rowb);
                } else {
                    kw = 0;
#pragma unroll
                    for (int ri = 0; ri < 4; ++ri)
                        if (!GUARD || rowb + ri < p.M) kw |= (uint32_t)p.rowkeep[rowb + ri] << (8 * ri);
                }
            }
            // 32-bit BYTE offset from the uniform base (host-checked: M * ldo * 4 < 4 GiB)
            uint32_t boff = ((uint32_t)rowb * (uint32_t)p.ldo + (uint32_t)(colbase + lcol)) * 4u;
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const float kp = ((kw >> (8 * ri)) & 0xffu) ? 1.0f : 0.0f;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
                    if (!GUARD || (rowb + ri < p.M && colok[nb]))
                        __hip_atomic_fetch_add(reinterpret_cast<float*>(xbase + boff + nb * 128),
                                               gcol[nb] * ((acc[mb][nb][rg * 4 + ri] + bcol[nb]) * kp), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                boff += (uint32_t)p.ldo * 4u;
            }
        }
}
template <int MBW, int NBW>
__device__ __forceinline__ void atomic_epilogue_resid(const F5GemmArgs& p, f32x16 (&acc)[MBW][NBW], int row0, int colbase,
                                                      int lane) {
    // interior wave tiles (all but the last row / column of tiles): no per-element guards, the 16 * MBW * NBW atomics of a
    // lane issue back to back
    if (row0 + 32 * MBW <= p.M && colbase + 32 * NBW <= p.N) atomic_epilogue_resid_impl<MBW, NBW, false>(p, acc, row0, colbase, lane);
    else atomic_epilogue_resid_impl<MBW, NBW, true>(p, acc, row0, colbase, lane);
}

}  // namespace F5_NS
