// Device-side pieces shared by the GEMM kernels of gfx950 (included once per operand build, inside namespace F5_NS):
// the K-tile width, the 16-byte XOR swizzle of a [rows][64] 16-bit LDS tile, global_load_lds, and the fused epilogues
// (plain per-element stores for small tiles, LDS-staged 16-byte row segments for 128-row wave tiles).
#pragma once
#include "gemm.hpp"

namespace F5_NS {

#define BK 64

// Measurement build only (F5_PROBE=1 bash build.sh -> libf5tts_hip_probe.so, tools/r5_epilogue_probe.py): ablation switches of the
// staged epilogues, read from F5GemmArgs::debug_flags.  In the product build they are compile-time false and leave no code.
#ifndef F5_PROBE
#define F5_PROBE 0
#endif
#define F5_PROBE_NOSTORE(p_) (F5_PROBE && ((p_).debug_flags & 0x10000))   // the epilogue runs, its global stores do not (q / k / FF1 / x)
#define F5_PROBE_NOVSTORE(p_) (F5_PROBE && ((p_).debug_flags & 0x8000))   // the same for the transposed V tiles of the QKV projection
#define F5_PROBE_NOMATH(p_) (F5_PROBE && ((p_).debug_flags & 0x20000))    // no GELU / rotation arithmetic (and no table loads)
#define F5_PROBE_NT(p_) (F5_PROBE && ((p_).debug_flags & 0x40000))        // residual stream read / written non-temporally
template <typename T>
__device__ __forceinline__ void f5_probe_sink(const T& v) {
    asm volatile("" ::"v"(v));
}

// ---- packed-f32 epilogue arithmetic (round 5).  The staged epilogues are bound by the VALU ISSUE rate of their two waves per SIMD (one
// instruction per ~5.4 cycles and wave whatever its width, tools/probes/valu_rate.hip), so their per-element arithmetic -- the LN-fold
// factors, GELU's polynomial, the rotation, the bias -- runs on element PAIRS with v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: the same
// IEEE operations per element, half the instructions.  A scalar that is broadcast to both halves goes through f5_bc2 (its own
// register: the compiler then broadcasts the LOW half, op_sel_hi = 0); the form whose low result reads a HIGH half (op_sel = 1) is the
// one that misbehaves next to MFMAs (build.sh, tests/test_isa.py) and must not appear: the ISA test scans every operand for it.
__device__ __forceinline__ f5_f32x2 f5_bc2(float s) {
    asm volatile("" : "+v"(s));      // volatile: a plain asm is hoisted above the K loop together with its operand -- a value requested
    return f5_f32x2{s, s};           // before the loop (FoldPre) then gets its vmcnt(0) INSIDE the loop (tests/test_isa.py caught it)
}
__device__ __forceinline__ f5_f32x2 f5_fma2(f5_f32x2 a, f5_f32x2 b, f5_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ int swz_off(int row, int chunk) {
    return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

// Epilogue.  All global LOADS an epilogue needs (bias, gate, residual / addrows values, rope cos/sin, row
// masks) are issued in batches BEFORE the dependent stores: a load placed between stores to a possibly
// aliasing pointer is serialised by the compiler (one HBM round trip per element, ~40 us per tile).
template <int EPI, int MB, int NB>
__device__ __forceinline__ void gemm_epilogue(const F5GemmArgs& p, f32x16 (&acc)[MB][NB], int m0, int n0, int wm, int wn,
                                               int lane) {
    const int hi = lane >> 5;
    const int lcol = lane & 31;
    f5_sat_t trk;                  // fp16 build: largest magnitude that goes through a 16-bit pack (op16.hpp), reported at the end
    int col[NB];
    bool colok[NB];
    float bcol[NB], gcol[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        col[nb] = n0 + wn * (32 * NB) + nb * 32 + lcol;
        colok[nb] = col[nb] < p.N;
        bcol[nb] = (EPI != EPI_ADDROWS && p.bias != nullptr && colok[nb]) ? p.bias[col[nb]] : 0.0f;
        gcol[nb] = (EPI == EPI_RESID_GATE && colok[nb]) ? p.gate[col[nb]] : 0.0f;
    }
    // waited for ONCE, here: a value whose first use sits inside the per-row conditionals gets the compiler's `s_waitcnt vmcnt(0)` in
    // every one of those blocks, where it also waits for all the stores issued so far (the stores of a lane then leave one memory
    // round trip apart; found in the conv-pos kernel, profiles/r03/convpos_epilogue_ab.txt)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+v"(bcol[nb]), "+v"(gcol[nb]));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int rowblk = m0 + wm * (32 * MB) + mb * 32 + hi * 4;   // row of (rg, ri) = rowblk + rg*8 + ri
        // ---- batched loads for this 32-row block --------------------------------------------------
        float pre[16][NB];
        uint8_t keep[16];
        if (EPI == EPI_RESID_GATE || EPI == EPI_ADDROWS || EPI == EPI_RESID_KEEP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rowblk + (r >> 2) * 8 + (r & 3);
                const bool rowok = row < p.M;
                keep[r] = 1;
                if (EPI != EPI_ADDROWS && p.rowkeep != nullptr && rowok) keep[r] = p.rowkeep[row];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float v = 0.0f;
                    if (rowok && colok[nb]) {
                        if (EPI == EPI_RESID_GATE) v = p.out_f32[(size_t)row * p.ldo + col[nb]];
                        if (EPI == EPI_ADDROWS) v = p.addrows[(size_t)row * p.ldadd + col[nb]];
                        if (EPI == EPI_RESID_KEEP) v = p.resid[(size_t)row * p.ldres + col[nb]];
                    }
                    pre[r][nb] = v;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+v"(pre[r][nb]));     // (the batch is waited for once, see above)
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int rowbase = rowblk + rg * 8;
            int nbase = 0, bbase = 0;
            float rc[4][NB], rs[4][NB];
            if (EPI == EPI_QKV_ROPE) {
                bbase = rowbase / p.seq_len;
                nbase = rowbase - bbase * p.seq_len;
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) {
                    int n = nbase + ri;
                    if (n >= p.seq_len) n -= p.seq_len;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const bool isqk = col[nb] < 2 * p.dmodel;
                        const int j = (col[nb] & 63) >> 1;
                        const float qs = (p.q_premul != 0.0f && col[nb] < p.dmodel) ? p.q_premul : 1.0f;
                        rc[ri][nb] = (isqk && rowbase + ri < p.M) ? p.rope_cos[n * 32 + j] * qs : 1.0f;
                        rs[ri][nb] = (isqk && rowbase + ri < p.M) ? p.rope_sin[n * 32 + j] * qs : 0.0f;
                    }
                }
            }
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const int row = rowbase + ri;
                const bool rowok = row < p.M;
                const int r = rg * 4 + ri;
                int n = nbase + ri, b = bbase;
                if (EPI == EPI_QKV_ROPE) {
                    if (n >= p.seq_len) {
                        n -= p.seq_len;
                        b += 1;
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int c = col[nb];
                    float v = acc[mb][nb][r] + bcol[nb];
                    if (EPI == EPI_QKV_ROPE) {
                        const float partner = __shfl_xor(v, 1, 64);
                        if (rowok && colok[nb]) {
                            if (c < 2 * p.dmodel) {
                                // (explicit product + fma: the same two roundings as the transposed tiles, whatever the compiler contracts)
                                const float o = __builtin_fmaf((c & 1) ? partner : -partner, rs[ri][nb], v * rc[ri][nb]);
                                op16_t h, l;
                                f5_split(o, h, l, trk);
                                p.out_bf[0][(size_t)row * p.ldob + c] = h;
                                if (p.out_bf[1]) p.out_bf[1][(size_t)row * p.ldob + c] = l;
                            } else {
                                const int c2 = c - 2 * p.dmodel;
                                const int head = c2 >> 6, d = c2 & 63;
                                const size_t off = ((size_t)(b * p.heads + head) * 64 + d) * p.npad + n;
                                op16_t h, l;
                                f5_split(v, h, l, trk);
                                p.vt[0][off] = h;
                                if (p.vt[1]) p.vt[1][off] = l;
                            }
                        }
                    } else if (rowok && colok[nb]) {
                        if (EPI == EPI_F32) {
                            p.out_f32[(size_t)row * p.ldo + c] = v;
                        } else if (EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_GELU_ERF_BF16) {
                            if (EPI == EPI_GELU_TANH) v = f5_gelu_tanh(v);
                            if (EPI == EPI_GELU_ERF_BF16) v = f5_gelu_erf(v);
                            op16_t h, l;
                            f5_split(v, h, l, trk);
                            p.out_bf[0][(size_t)row * p.ldob + c] = h;
                            if (p.out_bf[1]) p.out_bf[1][(size_t)row * p.ldob + c] = l;
                        } else if (EPI == EPI_GELU_ERF) {
                            p.out_f32[(size_t)row * p.ldo + c] = f5_gelu_erf(v);
                        } else if (EPI == EPI_RESID_GATE) {
                            if (keep[r] == 0) v = 0.0f;
                            const float xn = pre[r][nb] + gcol[nb] * v;
                            // fused LN tail: another XCD's workgroup reads these rows back inside this kernel -> agent-scope
                            // (sc1, write-through) store; the XCDs' L2s are not coherent with each other
                            if (p.ln_counter) __hip_atomic_store(&p.out_f32[(size_t)row * p.ldo + c], xn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            else p.out_f32[(size_t)row * p.ldo + c] = xn;
                        } else if (EPI == EPI_ADDROWS) {
                            v += pre[r][nb];
                            p.out_f32[(size_t)row * p.ldo + c] = v;
                            op16_t h, l;
                            f5_split(v, h, l, trk);
                            p.out_bf[0][(size_t)row * p.ldob + c] = h;
                            if (p.out_bf[1]) p.out_bf[1][(size_t)row * p.ldob + c] = l;
                        } else if (EPI == EPI_RESID_KEEP) {
                            v += pre[r][nb];
                            if (keep[r] == 0) v = 0.0f;
                            p.out_f32[(size_t)row * p.ldo + c] = v;
                        }
                    }
                }
            }
        }
    }
    f5_sat_commit(trk, p.sat_flag);
}

#define V2_HALF_ELEMS (128 * BK)

__device__ __forceinline__ void glds16(const op16_t* gptr, op16_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0);
}

// ---- LN-modulate folded into the consumer GEMM (F5GemmArgs::fold_*)
// What a FOLD epilogue needs from memory, requested BEFORE the K loop -- since round 5 ahead of the prologue's operand loads and pinned
// behind the first K tile's wait, see fold_prefetch_pin -- (10 registers through the loop): a one-workgroup-per-CU kernel
// has nothing to hide a load round trip at the head of its epilogue behind -- requested there it cost 1 us per tile (round 4:
// +8 us on the FF1 launch of batch 32, +4 on QKV).  Transposed tiles (lane = token): rr[mb] = row factors of the lane's token in each
// 32-row block, c[0] / c[1] = c1 / c2 of column col0 + lane.  Straight V tiles (lane = feature, rows in registers): rr[i] = row factors
// of row i * 64 + lane (spread to the wave through LDS later), c[0..3] = c1 / c2 of the lane's two columns.
struct FoldPre {
    f5_f32x2 rr[4];
    float c[4];
};
template <int MBW>
__device__ __forceinline__ void fold_prefetch_tr(const F5GemmArgs& p, FoldPre& f, int row0, int col0, int lane) {
    static_assert(MBW <= 4, "row blocks per wave");
    const f5_f32x2* rf = reinterpret_cast<const f5_f32x2*>(p.fold_rowf);
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
        int grow = row0 + mb * 32 + (lane & 31);
        if (grow > p.M - 1) grow = p.M - 1;
        f.rr[mb] = rf[grow];
    }
    f.c[0] = p.fold_c1[col0 + lane];
    f.c[1] = p.fold_c2[col0 + lane];
}
// The requests are issued BEFORE the prologue's operand loads and pinned right behind the wait for the first K tile (which they are
// older than): no register load is pending while the K loop runs.  Left pending across the loop (round 4), the compiler guards every
// register it believes such a load may still write -- a changed register allocation was enough to put a `s_waitcnt vmcnt(0)` in front
// of an address temporary INSIDE the loop: every K step drained the operand ring (tests/test_isa.py flags such a wait).  All fields are
// defined on every path (a field loaded on one path and undefined on the other escapes the pin: the phi is a different register).
__device__ __forceinline__ void fold_prefetch_pin(FoldPre& f) {
    asm volatile("" : "+v"(f.rr[0]), "+v"(f.rr[1]), "+v"(f.rr[2]), "+v"(f.rr[3]), "+v"(f.c[0]), "+v"(f.c[1]), "+v"(f.c[2]), "+v"(f.c[3]));
}
__device__ __forceinline__ void fold_prefetch_clear(FoldPre& f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f.rr[i] = f5_f32x2{0.0f, 0.0f};
        f.c[i] = 0.0f;
    }
}
template <int MBW>
__device__ __forceinline__ void fold_prefetch_v(const F5GemmArgs& p, FoldPre& f, int row0, int col0, int lane) {
    const f5_f32x2* rf = reinterpret_cast<const f5_f32x2*>(p.fold_rowf);
#pragma unroll
    for (int i = 0; i < (32 * MBW + 63) / 64; ++i) {
        int grow = row0 + i * 64 + lane;
        if (grow > p.M - 1) grow = p.M - 1;
        f.rr[i] = rf[grow];
    }
    const int lcol = lane & 31;
    f.c[0] = p.fold_c1[col0 + lcol];
    f.c[1] = p.fold_c1[col0 + 32 + lcol];
    f.c[2] = p.fold_c2[col0 + lcol];
    f.c[3] = p.fold_c2[col0 + 32 + lcol];
}

// ---- row factors from the producer's slice statistics, inside the consumer (F5GemmArgs::fold_stats, round 6).  Requested with the other
// fold operands BEFORE the prologue's operand loads (fold_stats_request_*), merged right behind the wait for the first K tile
// (fold_stats_finish_*: ~100 VALU instructions while the second K tile is in flight); only FoldPre::rr stays live through the K loop.
// Transposed tiles (lane = token): the two lane halves hold the same 32 tokens -- half 0 takes slices 0-7, half 1 slices 8-15, two
// v_permlane32_swap exchanges complete the row.  Straight V tiles (lane = row): 16 slices per lane and row.  Sums are taken as
// (slices 0-7 in order) + (slices 8-15 in order) everywhere, f5_fold_rows_kernel<16> included: one bit pattern per row whoever computes it.
// NT = 8 * (rows per lane of a transposed tile) = 16 * (rows per lane of a straight V tile): 16 for 64 x 64 wave tiles, 32 for 128 x 64
template <int NT>
struct FoldStatsPre {
    f5_f32x2 t[NT];           // transposed tiles: [mb * 8 + i] = slice (8 half + i) of row block mb; straight tiles: [r * 16 + k]
    float sh[NT / 8];
};
template <int NT>
__device__ __forceinline__ void fold_stats_clear(FoldStatsPre<NT>& q) {       // every field defined on every path (see fold_prefetch_pin)
#pragma unroll
    for (int i = 0; i < NT; ++i) q.t[i] = f5_f32x2{0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < NT / 8; ++i) q.sh[i] = 0.0f;
}
template <int NT>
__device__ __forceinline__ void fold_stats_pin(FoldStatsPre<NT>& q) {
#pragma unroll
    for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(q.t[i]));
#pragma unroll
    for (int i = 0; i < NT / 8; ++i) asm volatile("" : "+v"(q.sh[i]));
}
__device__ __forceinline__ f5_f32x2 fold_row_factor(float m2_lo, float m2_hi, float mean, float eps) {
    const float rstd = rsqrtf((m2_lo + m2_hi) * (1.0f / 1024.0f) + eps);
    return f5_f32x2{rstd, rstd * mean};
}
template <int MBW>
__device__ __forceinline__ void fold_stats_request_tr(const F5GemmArgs& p, FoldStatsPre<8 * MBW>& q, FoldPre& f, int row0, int col0, int lane) {
    const int half = lane >> 5;
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
        int grow = row0 + mb * 32 + (lane & 31);
        if (grow > p.M - 1) grow = p.M - 1;
        const f5_f32x2* sp = reinterpret_cast<const f5_f32x2*>(p.fold_stats) + (size_t)(half * 8) * p.fold_stats_ld + grow;
#pragma unroll
        for (int i = 0; i < 8; ++i) q.t[mb * 8 + i] = sp[(size_t)i * p.fold_stats_ld];
        q.sh[mb] = p.fold_shift != nullptr ? p.fold_shift[grow] : 0.0f;
    }
    f.c[0] = p.fold_c1[col0 + lane];
    f.c[1] = p.fold_c2[col0 + lane];
}
template <int MBW>
__device__ __forceinline__ void fold_stats_finish_tr(const F5GemmArgs& p, const FoldStatsPre<8 * MBW>& q, FoldPre& f, int row0, int lane, bool write_mean) {
    const int half = lane >> 5;
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
        float s = q.t[mb * 8][0];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += q.t[mb * 8 + i][0];
        const float so = f5_xor32(s, (unsigned)lane);
        const float mean = ((half ? so : s) + (half ? s : so)) * (1.0f / 1024.0f);
        float m2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float dm = q.t[mb * 8 + i][0] * (1.0f / 64.0f) - mean;
            m2 += q.t[mb * 8 + i][1] + 64.0f * dm * dm;
        }
        const float mo = f5_xor32(m2, (unsigned)lane);
        f.rr[mb] = fold_row_factor(half ? mo : m2, half ? m2 : mo, mean, p.fold_eps);
        const int grow = row0 + mb * 32 + (lane & 31);
        if (write_mean && half == 0 && grow < p.M) p.fold_mean_out[grow] = q.sh[mb] + mean;
    }
}
template <int MBW>
__device__ __forceinline__ void fold_stats_request_v(const F5GemmArgs& p, FoldStatsPre<8 * MBW>& q, FoldPre& f, int row0, int col0, int lane) {
    constexpr int NR = (32 * MBW + 63) / 64;
    static_assert(NR * 16 <= 8 * MBW, "one row per 64 rows of the wave tile");
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        int grow = row0 + i * 64 + lane;
        if (grow > p.M - 1) grow = p.M - 1;
        const f5_f32x2* sp = reinterpret_cast<const f5_f32x2*>(p.fold_stats) + grow;
#pragma unroll
        for (int k = 0; k < 16; ++k) q.t[i * 16 + k] = sp[(size_t)k * p.fold_stats_ld];
    }
    const int lcol = lane & 31;
    f.c[0] = p.fold_c1[col0 + lcol];
    f.c[1] = p.fold_c1[col0 + 32 + lcol];
    f.c[2] = p.fold_c2[col0 + lcol];
    f.c[3] = p.fold_c2[col0 + 32 + lcol];
}
template <int MBW>
__device__ __forceinline__ void fold_stats_finish_v(const F5GemmArgs& p, const FoldStatsPre<8 * MBW>& q, FoldPre& f) {
    constexpr int NR = (32 * MBW + 63) / 64;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        float s_lo = q.t[i * 16][0], s_hi = q.t[i * 16 + 8][0];
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            s_lo += q.t[i * 16 + k][0];
            s_hi += q.t[i * 16 + 8 + k][0];
        }
        const float mean = (s_lo + s_hi) * (1.0f / 1024.0f);
        float m_lo = 0.0f, m_hi = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float d0 = q.t[i * 16 + k][0] * (1.0f / 64.0f) - mean, d1 = q.t[i * 16 + 8 + k][0] * (1.0f / 64.0f) - mean;
            m_lo += q.t[i * 16 + k][1] + 64.0f * d0 * d0;
            m_hi += q.t[i * 16 + 8 + k][1] + 64.0f * d1 * d1;
        }
        f.rr[i] = fold_row_factor(m_lo, m_hi, mean, p.fold_eps);
    }
}

// ---- LDS-staged epilogues (used by the 256x256 and the 128x256 kernels).  A wave owns a (32*MBW) x (32*NBW) tile
// and a private LDS region; the MFMA C layout (lane = column, registers = rows) is turned into 16-byte global accesses
// in full row segments.  bf16 row-major outputs (FF1 / q / k / plain bf16): 32-row passes, [32][W+8] hi (+ lo).
// V (QKV columns >= 2*dmodel) is written TRANSPOSED, Vt[(b*H+h)*64+d][n]: staged [d][64 tokens (+8)] and stored along
// the token axis; those 16-byte stores may be only 2-byte aligned (legal on gfx950, tools/probes/unaligned.hip) and
// are split element-wise where a chunk crosses a batch-element boundary.
// FOLD (V tiles only): the LN-modulate fold of F5GemmArgs::fold_* -- value = rstd * acc - rstd * mean * c1[col] + c2[col], the row
// factors (finished by f5_launch_fold_rows, requested before the K loop: FoldPre) through the wave's LDS scratch `fl`.
template <int EPI, int MBW, int NBW, bool VONLY = false, bool FOLD = false>
__device__ __forceinline__ void staged_epilogue_bf16(const F5GemmArgs& p, f32x16 (&acc)[MBW][NBW], op16_t* reg, int row0,
                                                     int colbase, int lane, float* fl = nullptr, const FoldPre* pre = nullptr) {
    static_assert(!FOLD || VONLY, "the straight q / k / FF1 tiles have no folded form");
    constexpr int W = 32 * NBW;
    constexpr int LD = W + 8;
    constexpr int CPR = W / 8;             // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;          // rows per store instruction
    const int hi = lane >> 5, lcol = lane & 31;
    const bool two = p.out_bf[1] != nullptr;
    f5_sat_t trk;
    f5_sat_s trkv;                         // V tiles: the scalar form (no VGPR to spare there, op16.hpp)
    float bcol[NBW], c1col[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        if (FOLD) {
            static_assert(!FOLD || NBW == 2, "FoldPre holds two columns per lane");
            bcol[nb] = pre->c[2 + (nb & 1)];
            c1col[nb] = pre->c[nb & 1];
        } else {
            bcol[nb] = p.bias ? p.bias[colbase + nb * 32 + lcol] : 0.0f;
            c1col[nb] = 0.0f;
        }
    }
    if (FOLD) {                                          // the row factors requested before the K loop, spread to the wave through LDS,
#pragma unroll                                           // planar: fl[row] = rstd, fl[32 MBW + row] = rstd (mean - m): four consecutive rows
        for (int i = 0; i < (32 * MBW + 63) / 64; ++i)   // of either are one 16-byte read = two register PAIRS for the packed fold
            if (i * 64 + lane < 32 * MBW) {
                fl[i * 64 + lane] = pre->rr[i][0];
                fl[32 * MBW + i * 64 + lane] = pre->rr[i][1];
            }
        __builtin_amdgcn_wave_barrier();
    }
    const bool is_v = VONLY || ((EPI == EPI_QKV_ROPE) && (colbase >= 2 * p.dmodel));   // VONLY: the q / k tiles went elsewhere

    if (!is_v) {
        op16_t* rh = reg;
        op16_t* rl = reg + 32 * LD;
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) {
            const int rowblk = row0 + mb * 32;
            // RoPE factors of the whole 32-row block, issued as ONE batch of loads (in a one-round launch the epilogue is a
            // latency chain: four dependent batches cost four round trips).  The table index depends on the column only through
            // its position inside the head, i.e. on the parity of nb; rows >= M read a valid entry and are never stored.
            constexpr int PAR = NBW >= 2 ? 2 : 1;
            float rc[4][4][PAR], rs[4][4][PAR];
            if (EPI == EPI_QKV_ROPE) {
                // q columns carry the softmax scale * log2(e) (F5GemmArgs::q_premul) folded into their rotation factors
                const float qs = (p.q_premul != 0.0f && colbase < p.dmodel) ? p.q_premul : 1.0f;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int nbase = (rowblk + rg * 8 + hi * 4) % p.seq_len;
#pragma unroll
                    for (int ri = 0; ri < 4; ++ri) {
                        int n = nbase + ri;
                        if (n >= p.seq_len) n -= p.seq_len;
#pragma unroll
                        for (int q = 0; q < PAR; ++q) {
                            const int j = ((q * 32 + lcol) & 63) >> 1;
                            rc[rg][ri][q] = p.rope_cos[n * 32 + j] * qs;
                            rs[rg][ri][q] = p.rope_sin[n * 32 + j] * qs;
                        }
                    }
                }
            }
            // bias + GELU on register PAIRS (two consecutive rows of the lane's column), in place: packed-f32 arithmetic, see f5_bc2
            if (EPI != EPI_QKV_ROPE) {
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    const f5_f32x2 bv = f5_bc2(bcol[nb]);
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f5_f32x2 t = f5_f32x2{acc[mb][nb][r], acc[mb][nb][r + 1]} + bv;
                        if (EPI == EPI_GELU_TANH) t = f5_gelu_tanh2(t);
                        if (EPI == EPI_GELU_ERF_BF16) t = f5_f32x2{f5_gelu_erf(t[0]), f5_gelu_erf(t[1])};
                        acc[mb][nb][r] = t[0];
                        acc[mb][nb][r + 1] = t[1];
                    }
                }
            }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
#pragma unroll
                for (int ri = 0; ri < 4; ++ri) {
                    const int r = rg * 4 + ri;
                    const int lrow = ri + 8 * rg + 4 * hi;
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) {
                        float v = EPI != EPI_QKV_ROPE ? acc[mb][nb][r] : acc[mb][nb][r] + bcol[nb];
                        if (EPI == EPI_QKV_ROPE) {
                            const float partner = __shfl_xor(v, 1, 64);
                            const float c = rc[rg][ri][nb & (PAR - 1)], sn = rs[rg][ri][nb & (PAR - 1)];
                            v = __builtin_fmaf((lcol & 1) ? partner : -partner, sn, v * c);   // even column v c - partner s, odd v c + partner s
                        }
                        op16_t h, l;
                        f5_split(v, h, l, trk);
                        rh[lrow * LD + nb * 32 + lcol] = h;
                        if (two) rl[lrow * LD + nb * 32 + lcol] = l;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 32 / RPI; ++i) {
                const int lrow = i * RPI + lane / CPR, chunk = lane % CPR;
                const int grow = rowblk + lrow;
                if (grow < p.M) {
                    const size_t off = (size_t)grow * p.ldob + colbase + chunk * 8;
                    *reinterpret_cast<u32x4*>(p.out_bf[0] + off) = *reinterpret_cast<const u32x4*>(&rh[lrow * LD + chunk * 8]);
                    if (two)
                        *reinterpret_cast<u32x4*>(p.out_bf[1] + off) = *reinterpret_cast<const u32x4*>(&rl[lrow * LD + chunk * 8]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        // ---- V: transposed staging, [W d-rows][32*MP tokens (+8 pad)] per pass of MP 32-row blocks, hi then lo
        constexpr int MP = MBW >= 2 ? 2 : 1;
        static_assert(MBW % MP == 0, "row blocks per pass");
        constexpr int TLD = 32 * MP + 8;
        constexpr int CPD = 4 * MP;           // 16-byte chunks per d-row
        const int head0 = (colbase - 2 * p.dmodel) >> 6;
        const bool two_v = p.vt[1] != nullptr;
#pragma unroll
        for (int mq = 0; mq < MBW / MP; ++mq) {
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                if (part == 1 && !two_v) break;
#pragma unroll
                for (int mb = 0; mb < MP; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            float v[4];
                            const f32x16& ac = acc[mq * MP + mb][nb];
                            const f5_f32x2 a01 = {ac[rg * 4 + 0], ac[rg * 4 + 1]}, a23 = {ac[rg * 4 + 2], ac[rg * 4 + 3]};
                            f5_f32x2 v01, v23;
                            if (FOLD) {
                                // rows (mq*MP + mb)*32 + rg*8 + 4*hi + [0, 4): their rstd / rstd (mean - m) quads (planar scratch)
                                const int r4 = (mq * MP + mb) * 32 + rg * 8 + 4 * hi;
                                const f32x4 rs4 = *reinterpret_cast<const f32x4*>(fl + r4);
                                const f32x4 rm4 = *reinterpret_cast<const f32x4*>(fl + 32 * MBW + r4);
                                const f5_f32x2 dv = f5_bc2(bcol[nb]), ncv = f5_bc2(-c1col[nb]);
                                v01 = f5_fma2(f5_f32x2{rs4[0], rs4[1]}, a01, f5_fma2(f5_f32x2{rm4[0], rm4[1]}, ncv, dv));
                                v23 = f5_fma2(f5_f32x2{rs4[2], rs4[3]}, a23, f5_fma2(f5_f32x2{rm4[2], rm4[3]}, ncv, dv));
                            } else {
                                const f5_f32x2 bv = f5_bc2(bcol[nb]);
                                v01 = a01 + bv;
                                v23 = a23 + bv;
                            }
                            v[0] = v01[0];
                            v[1] = v01[1];
                            v[2] = v23[0];
                            v[3] = v23[1];
                            const u32x2 pk = part == 0 ? u32x2{f5_pack2(v[0], v[1], trkv), f5_pack2(v[2], v[3], trkv)}
                                                       : u32x2{f5_pack2_lo(v[0], v[1]), f5_pack2_lo(v[2], v[3])};
                            const int tok = mb * 32 + rg * 8 + 4 * hi;
                            *reinterpret_cast<u32x2*>(&reg[(nb * 32 + lcol) * TLD + tok]) = pk;
                        }
                __builtin_amdgcn_wave_barrier();
                u16* dstbase = reinterpret_cast<u16*>(p.vt[part]);
#pragma unroll
                for (int i = 0; i < W * CPD / 64; ++i) {
                    const int c = i * 64 + lane;
                    const int d = c / CPD, t0 = (c % CPD) * 8;
                    const int grow = row0 + mq * (32 * MP) + t0;
                    if (grow < p.M) {
                        const int b = grow / p.seq_len;
                        const int n = grow - b * p.seq_len;
                        const u32x4 val = *reinterpret_cast<const u32x4*>(&reg[d * TLD + t0]);
                        const int head = head0 + (d >> 6), dd = d & 63;
                        u16* dst = dstbase + ((size_t)(b * p.heads + head) * 64 + dd) * p.npad + n;
                        if (F5_PROBE_NOVSTORE(p)) {
                            f5_probe_sink(val);
                        } else if (n + 8 <= p.seq_len && grow + 8 <= p.M) {
                            *reinterpret_cast<u32x4*>(dst) = val;       // may be only 2-byte aligned: legal on gfx950
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const int g2 = grow + e;
                                if (g2 < p.M) {
                                    const int b2 = g2 / p.seq_len, n2 = g2 - b2 * p.seq_len;
                                    dstbase[((size_t)(b2 * p.heads + head) * 64 + dd) * p.npad + n2] = (u16)(val[e >> 1] >> (16 * (e & 1)));
                                }
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    f5_sat_commit(trk, p.sat_flag);
    f5_sat_commit(trkv, p.sat_flag);
}

// 16-bit row-major outputs (FF1 + GELU, plain 16-bit) from a TRANSPOSED accumulator tile, D = W A^T: lane = token, registers =
// 4 consecutive features per group (feature = 8*(r>>2) + 4*hi + (r&3)).  staged_epilogue_bf16 above turns the usual layout
// (lane = feature) into row segments with one 2-byte LDS write per element; here a lane's 4 features are ONE 8-byte LDS write
// into the same [32 tokens][W + 8] image, and the read-back / 16-byte global stores in full row segments are unchanged: 8x
// fewer LDS write instructions per tile, bias lane-uniform.  The wave picks the layout by the operand order of its MFMAs
// (f5_gemm256_kernel, V2_MM).
// FOLD: the LN-modulate fold of F5GemmArgs::fold_* (lane = token: its two row factors are one 8-byte LDS read per 32-token block).
template <int EPI, int MBW, int NBW, bool FOLD = false>
__device__ __forceinline__ void staged_epilogue_tr(const F5GemmArgs& p, f32x16 (&acc)[MBW][NBW], op16_t* reg, int row0, int colbase,
                                                   int lane, float* fl = nullptr, const FoldPre* pre = nullptr) {
    constexpr int W = 32 * NBW;
    constexpr int LD = W + 8;
    constexpr int CPR = W / 8;             // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;          // rows per store instruction
    const int hi = lane >> 5, lcol = lane & 31;
    const bool two = p.out_bf[1] != nullptr;
    op16_t* rh = reg;
    op16_t* rl = reg + 32 * LD;
    f5_sat_t trk;
    f32x4 b4[NBW][4];                      // bias of the lane's feature groups (the same for every 32-token block)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
            b4[nb][rg] = (!FOLD && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + colbase + nb * 32 + rg * 8 + hi * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    f5_f32x2 rrv[MBW];
    if (FOLD) {
        // everything the fold needs was requested before the K loop (FoldPre): the lane's row factors are registers, the wave's W
        // columns of c1 | c2 go through the LDS scratch (read back as 16-byte broadcasts, once)
        static_assert(!FOLD || W == 64, "one column of c1 | c2 per lane");
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) rrv[mb] = pre->rr[mb];
        fl[lane] = pre->c[0];
        fl[W + lane] = pre->c[1];
        __builtin_amdgcn_wave_barrier();
    }
    // the lane's c1 / c2 quads, read back ONCE (b4 = c2): LDS reads between the staging passes would serialise on lgkmcnt(0) with the
    // staging writes -- 32 LDS round trips per tile, measured +16 us on the FF1 launch of batch 32
    f32x4 c1r[FOLD ? NBW : 1][4];
    if (FOLD) {
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                c1r[nb][rg] = *reinterpret_cast<const f32x4*>(&fl[nb * 32 + rg * 8 + hi * 4]);
                b4[nb][rg] = *reinterpret_cast<const f32x4*>(&fl[W + nb * 32 + rg * 8 + hi * 4]);
            }
    }
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
        const int rowblk = row0 + mb * 32;
        f5_f32x2 rr = {1.0f, 0.0f};
        if (FOLD) rr = rrv[mb];
        const f5_f32x2 r0v = f5_bc2(rr[0]), nr1v = f5_bc2(-rr[1]);       // the lane's row factors, broadcast to both halves of a pair
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 c1q = FOLD ? c1r[nb][rg] : f32x4{0.f, 0.f, 0.f, 0.f}, c2q = b4[nb][rg];
                f5_f32x2 v[2];                                           // features (rg*4 + 0, 1) and (rg*4 + 2, 3) of this token
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f5_f32x2 a = {acc[mb][nb][rg * 4 + 2 * h], acc[mb][nb][rg * 4 + 2 * h + 1]};
                    const f5_f32x2 c2h = {c2q[2 * h], c2q[2 * h + 1]};
                    if (FOLD) {
                        const f5_f32x2 c1h = {c1q[2 * h], c1q[2 * h + 1]};
                        v[h] = f5_fma2(r0v, a, f5_fma2(nr1v, c1h, c2h));  // rstd acc + (c2 - rstd (mean - m) c1)
                    } else {
                        v[h] = a + c2h;
                    }
                    if (EPI == EPI_GELU_TANH && !F5_PROBE_NOMATH(p)) v[h] = f5_gelu_tanh2(v[h]);
                    if (EPI == EPI_GELU_ERF_BF16) v[h] = f5_f32x2{f5_gelu_erf(v[h][0]), f5_gelu_erf(v[h][1])};
                }
                const int so = lcol * LD + nb * 32 + rg * 8 + hi * 4;
                *reinterpret_cast<u32x2*>(&rh[so]) = u32x2{f5_pack2(v[0][0], v[0][1], trk), f5_pack2(v[1][0], v[1][1], trk)};
                if (two) *reinterpret_cast<u32x2*>(&rl[so]) = u32x2{f5_pack2_lo(v[0][0], v[0][1]), f5_pack2_lo(v[1][0], v[1][1])};
            }
        __builtin_amdgcn_wave_barrier();
        // read the staged rows back in ONE batch (inside the `grow < M` blocks every read was waited for on its own in front of its
        // store: four LDS round trips per 32-row block), then store
        u32x4 back[32 / RPI];
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) back[i] = *reinterpret_cast<const u32x4*>(&rh[(i * RPI + lane / CPR) * LD + (lane % CPR) * 8]);
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) asm volatile("" : "+v"(back[i]));
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) {
            const int lrow = i * RPI + lane / CPR, chunk = lane % CPR;
            const int grow = rowblk + lrow;
            if (F5_PROBE_NOSTORE(p)) {
                f5_probe_sink(back[i]);
            } else if (grow < p.M) {
                const size_t off = (size_t)grow * p.ldob + colbase + chunk * 8;
                *reinterpret_cast<u32x4*>(p.out_bf[0] + off) = back[i];
                if (two) *reinterpret_cast<u32x4*>(p.out_bf[1] + off) = *reinterpret_cast<const u32x4*>(&rl[lrow * LD + chunk * 8]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    f5_sat_commit(trk, p.sat_flag);
}

// The q / k column tiles of the QKV projection from a TRANSPOSED accumulator tile (256x256 kernel): as staged_epilogue_tr, with the
// rotation applied in-lane -- the pair (2i, 2i+1) sits in neighbouring registers, so no cross-lane exchange -- from PAIR-major
// tables ([dim_head/2][positions]: the 32 lanes of a half wave hold 32 consecutive tokens and read 128 contiguous bytes per
// factor; the straight tile reads a token-major table with 2 lines per load but needs 64 loads and 32 lane swaps per 32 x 64
// block).  q_premul is folded into the q tables by the host.  dit.py:136-158.
template <int MBW, int NBW, bool FOLD = false>
__device__ __forceinline__ void staged_epilogue_tr_rope(const F5GemmArgs& p, f32x16 (&acc)[MBW][NBW], op16_t* reg, int row0,
                                                        int colbase, int lane, float* fl = nullptr, const FoldPre* pre = nullptr) {
    constexpr int W = 32 * NBW;
    constexpr int LD = W + 8;
    constexpr int CPR = W / 8;
    constexpr int RPI = 64 / CPR;
    const int hi = lane >> 5, lcol = lane & 31;
    const bool two = p.out_bf[1] != nullptr;
    const bool isq = colbase < p.dmodel;
    // group-major factors (gemm.hpp rope_g4*): element (g, n) = 16 bytes.  The lane's part of the address -- token n, and hi picks the
    // odd group of the pair of groups a (nb, rg) step covers -- is a 32-bit byte offset; the (nb, rg) part is wave-uniform and goes
    // into a scalar base: global_load_dwordx4 v, v_off, s[base], no 64-bit VALU address arithmetic (round 5 counted 200 of the 860
    // executed VALU instructions of this epilogue as such)
    typedef __attribute__((address_space(1))) const char gcchar;
    typedef __attribute__((address_space(1))) const f32x4 gcf32x4;
    gcchar* const tb = (gcchar*)(isq ? p.rope_g4q : p.rope_g4k);
    const uint32_t gstep = (uint32_t)p.seq_len * 16u;             // bytes from group g to group g + 1
    op16_t* rh = reg;
    op16_t* rl = reg + 32 * LD;
    f5_sat_t trk;
    f5_f32x2 rrv[MBW];
    if (FOLD) {                                              // as staged_epilogue_tr: FoldPre -> registers / LDS scratch
        static_assert(!FOLD || W == 64, "one column of c1 | c2 per lane");
#pragma unroll
        for (int mb = 0; mb < MBW; ++mb) rrv[mb] = pre->rr[mb];
        fl[lane] = pre->c[0];
        fl[W + lane] = pre->c[1];
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
        const int rowblk = row0 + mb * 32;
        int row = rowblk + lcol;
        if (row > p.M - 1) row = p.M - 1;                    // rows past the end compute a valid rotation and are never stored
        const int n = row % p.seq_len;
        uint32_t loff = (uint32_t)(hi * p.seq_len + n) * 16u;
        f5_f32x2 rr = {1.0f, 0.0f};
        if (FOLD) rr = rrv[mb];
        const f5_f32x2 nr1v = f5_bc2(-rr[1]);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            // one batch of loads per 32-feature block: 16 rotation factors + 4 bias quads (the accumulators leave ~90 free VGPRs)
            float c0[4], c1[4], s0[4], s1[4];
            f32x4 b4[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int c = colbase + nb * 32 + rg * 8 + hi * 4;
                if (F5_PROBE_NOMATH(p)) {
                    c0[rg] = c1[rg] = 1.0f;
                    s0[rg] = s1[rg] = 0.0f;
                } else {
                    // features c .. c + 3 of a 64-wide head = rotation pairs 2g, 2g + 1 with g = ((c & 63) >> 2): its hi part is in loff
                    const int g0 = (((colbase & 63) + nb * 32 + rg * 8) & 63) >> 2;   // wave-uniform (colbase is a multiple of 32)
                    gcchar* gb = tb + (size_t)g0 * gstep;
                    asm volatile("" : "+v"(loff));                                   // (keeps the zero-extension next to the load: saddr form)
                    const f32x4 cs = *(gcf32x4*)(gb + loff);
                    c0[rg] = cs[0];
                    c1[rg] = cs[1];
                    s0[rg] = cs[2];
                    s1[rg] = cs[3];
                }
                if (FOLD) {
                    const f32x4 c1q = *reinterpret_cast<const f32x4*>(&fl[c - colbase]), c2q = *reinterpret_cast<const f32x4*>(&fl[W + c - colbase]);
                    const f5_f32x2 lo = f5_fma2(nr1v, f5_f32x2{c1q[0], c1q[1]}, f5_f32x2{c2q[0], c2q[1]});
                    const f5_f32x2 hi2 = f5_fma2(nr1v, f5_f32x2{c1q[2], c1q[3]}, f5_f32x2{c2q[2], c2q[3]});
                    b4[rg] = f32x4{lo[0], lo[1], hi2[0], hi2[1]};
                } else {
                    b4[rg] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            const f5_f32x2 rsv = f5_bc2(FOLD ? rr[0] : 1.0f);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                // fold / bias on pairs (packed), then the rotation: the pair (2i, 2i + 1) is one register pair -- its cos product is
                // one packed multiply, the two sin terms need the halves crossed and stay scalar fmas (a packed form would read a
                // high half into a low result: the select that must not sit next to MFMAs)
                f5_f32x2 a01 = {acc[mb][nb][rg * 4 + 0], acc[mb][nb][rg * 4 + 1]}, a23 = {acc[mb][nb][rg * 4 + 2], acc[mb][nb][rg * 4 + 3]};
                const f5_f32x2 b01 = {b4[rg][0], b4[rg][1]}, b23 = {b4[rg][2], b4[rg][3]};
                if (FOLD) {
                    a01 = f5_fma2(rsv, a01, b01);
                    a23 = f5_fma2(rsv, a23, b23);
                } else {
                    a01 = a01 + b01;
                    a23 = a23 + b23;
                }
                // the same two roundings as the straight tile: even column fma(-partner, s, v c), odd column fma(partner, s, v c)
                const f5_f32x2 t01 = a01 * f5_bc2(c0[rg]), t23 = a23 * f5_bc2(c1[rg]);
                const float o0 = __builtin_fmaf(-a01[1], s0[rg], t01[0]), o1 = __builtin_fmaf(a01[0], s0[rg], t01[1]);
                const float o2 = __builtin_fmaf(-a23[1], s1[rg], t23[0]), o3 = __builtin_fmaf(a23[0], s1[rg], t23[1]);
                const int so = lcol * LD + nb * 32 + rg * 8 + hi * 4;
                *reinterpret_cast<u32x2*>(&rh[so]) = u32x2{f5_pack2(o0, o1, trk), f5_pack2(o2, o3, trk)};
                if (two) *reinterpret_cast<u32x2*>(&rl[so]) = u32x2{f5_pack2_lo(o0, o1), f5_pack2_lo(o2, o3)};
            }
        }
        __builtin_amdgcn_wave_barrier();
        // read the staged rows back in ONE batch (inside the `grow < M` blocks every read was waited for on its own in front of its
        // store: four LDS round trips per 32-row block), then store
        u32x4 back[32 / RPI];
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) back[i] = *reinterpret_cast<const u32x4*>(&rh[(i * RPI + lane / CPR) * LD + (lane % CPR) * 8]);
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) asm volatile("" : "+v"(back[i]));
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) {
            const int lrow = i * RPI + lane / CPR, chunk = lane % CPR;
            const int grow = rowblk + lrow;
            if (F5_PROBE_NOSTORE(p)) {
                f5_probe_sink(back[i]);
            } else if (grow < p.M) {
                const size_t off = (size_t)grow * p.ldob + colbase + chunk * 8;
                *reinterpret_cast<u32x4*>(p.out_bf[0] + off) = back[i];
                if (two) *reinterpret_cast<u32x4*>(p.out_bf[1] + off) = *reinterpret_cast<const u32x4*>(&rl[lrow * LD + chunk * 8]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    f5_sat_commit(trk, p.sat_flag);
}

// x += gate * ((acc + bias) * keep)  (dit.py:319,323): fp32 tile staged [32 rows][W+4] so that the read-modify-write of
// the residual stream uses 16-byte accesses; the residual values are loaded before the LDS round trip.
// PRE (single-round launches, MBW = 1): the rows of x, the keep bytes and the row shifts were requested BEFORE the K loop
// (resid_staged_preload): in a one-round launch the epilogue's load round trip is on the critical chain (gemm.hip ResidPre)
template <int NBW>
struct ResidStagedPre {
    f32x4 xr[32 / (64 / (32 * NBW / 4))];
    uint32_t kraw[32 / (64 / (32 * NBW / 4))];
    float xsh[32 / (64 / (32 * NBW / 4))];
    float bcol[NBW];          // bias of the lane's columns (accumulator layout)
    f32x4 g4, sc4;            // gate and 1 + scale of the lane's 16-byte chunk (row-segment layout)
};
template <int NBW>
__device__ __forceinline__ void resid_staged_preload(const F5GemmArgs& p, ResidStagedPre<NBW>& q, int row0, int colbase, int lane) {
    constexpr int CPR = 32 * NBW / 4, RPI = 64 / CPR, NI = 32 / RPI;
    const int chunk = lane % CPR;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int grow = row0 + i * RPI + lane / CPR;
        const bool ok = grow < p.M;
        q.xr[i] = ok ? *reinterpret_cast<const f32x4*>(p.out_f32 + (size_t)grow * p.ldo + colbase + chunk * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        q.kraw[i] = (ok && p.rowkeep != nullptr) ? (uint32_t)p.rowkeep[grow] : 1u;
        q.xsh[i] = (ok && p.x16_out != nullptr && p.x16_shift != nullptr) ? p.x16_shift[grow] : 0.0f;
    }
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) q.bcol[nb] = p.bias ? p.bias[colbase + nb * 32 + (lane & 31)] : 0.0f;
    q.g4 = *reinterpret_cast<const f32x4*>(p.gate + colbase + chunk * 4);
    q.sc4 = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
    if (p.x16_out != nullptr) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.x16_scale + colbase + chunk * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) q.sc4[e] += t[e];
    }
}
template <int MBW, int NBW, bool PRE = false>
__device__ __forceinline__ void staged_epilogue_resid(const F5GemmArgs& p, f32x16 (&acc)[MBW][NBW], float* reg, int row0,
                                                      int colbase, int lane, const ResidStagedPre<NBW>* pre = nullptr) {
    static_assert(!PRE || MBW == 1, "the preloaded form covers one 32-row block");
    constexpr int W = 32 * NBW;
    constexpr int LD = W + 4;
    constexpr int CPR = W / 4;             // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;          // rows per instruction
    constexpr int NI = 32 / RPI;           // instructions per 32-row pass
    const int hi = lane >> 5, lcol = lane & 31;
    float bcol[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) bcol[nb] = PRE ? pre->bcol[nb] : (p.bias ? p.bias[colbase + nb * 32 + lcol] : 0.0f);
    const int chunk = lane % CPR;
    f32x4 g4 = PRE ? pre->g4 : *reinterpret_cast<const f32x4*>(p.gate + colbase + chunk * 4);
    asm volatile("" : "+v"(g4));
    f32x4 sc4 = {1.0f, 1.0f, 1.0f, 1.0f};               // LN fold: 1 + the scale of the LN that follows (F5GemmArgs::x16_scale)
    if (PRE) {
        sc4 = pre->sc4;
    } else if (p.x16_out != nullptr) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.x16_scale + colbase + chunk * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) sc4[e] += t[e];
    }
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
        const int rowblk = row0 + mb * 32;
        f32x4 xr[NI];
        float xsh[NI];                     // LN fold: the row shift m of the folded operand (F5GemmArgs::x16_shift)
        uint32_t kraw[NI];                 // the keep byte as loaded: converting it inside this loop put a `s_waitcnt vmcnt(0)` behind
                                           // every byte load, i.e. behind every x-row load -- the rows of a lane were fetched one
                                           // memory round trip after the other
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (PRE) {
                xr[i] = pre->xr[i];
                kraw[i] = pre->kraw[i];
                xsh[i] = pre->xsh[i];
                continue;
            }
            const int grow = rowblk + i * RPI + lane / CPR;
            const bool ok = grow < p.M;
            if (F5_PROBE_NT(p))
                xr[i] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.out_f32 + (size_t)grow * p.ldo + colbase + chunk * 4))
                           : f32x4{0.f, 0.f, 0.f, 0.f};
            else
            xr[i] = ok ? *reinterpret_cast<const f32x4*>(p.out_f32 + (size_t)grow * p.ldo + colbase + chunk * 4)
                       : f32x4{0.f, 0.f, 0.f, 0.f};
            kraw[i] = (ok && p.rowkeep != nullptr) ? (uint32_t)p.rowkeep[grow] : 1u;
            xsh[i] = (CPR == 16 && ok && p.x16_out != nullptr && p.x16_shift != nullptr) ? p.x16_shift[grow] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) reg[lrow * LD + nb * 32 + lcol] = acc[mb][nb][r] + bcol[nb];
        }
        // the x rows (requested above, in flight during the staging writes) are waited for ONCE, here.  Their first use used to be
        // inside the `grow < M` blocks below: the compiler's `s_waitcnt vmcnt(0)` for them sat in every block and also waited for
        // the store of the previous block -- the eight 16-byte stores of a lane left one memory round trip apart
#pragma unroll
        for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(xr[i]), "+v"(kraw[i]), "+v"(xsh[i]));
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int lrow = i * RPI + lane / CPR;
            const int grow = rowblk + lrow;
            const f32x4 v = *reinterpret_cast<const f32x4*>(&reg[lrow * LD + chunk * 4]);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = xr[i][e] + g4[e] * (kraw[i] != 0u ? v[e] : 0.0f);   // a select, like gemm_epilogue: a non-finite
                                                                                                  // accumulator of a masked row must not reach x
            if (F5_PROBE_NOSTORE(p)) f5_probe_sink(o);
            else if (F5_PROBE_NT(p)) {
                if (grow < p.M) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(p.out_f32 + (size_t)grow * p.ldo + colbase + chunk * 4));
            } else
            if (grow < p.M) *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)grow * p.ldo + colbase + chunk * 4) = o;
            if (CPR == 16 && p.x16_out != nullptr) {          // (64-column wave tiles only: the launcher refuses the others)
                // LN fold (F5GemmArgs): (x - m)(1 + s) in the operand type is the next GEMM's A operand, m = the row's mean at the
                // previous LayerNorm.  Slice statistics of d = x - m over this wave's 64 columns: the sum by a 16-lane DPP rotation
                // all-reduce (the 16 lanes of a DPP row hold one row segment; every lane ends up with the total), then the sum of
                // squares about the SLICE mean the same way -- f5_fold_rows_kernel merges the slices without cancellation
                const bool rok = grow < p.M;
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = rok ? o[e] - xsh[i] : 0.0f;
                float ps = (d[0] + d[1]) + (d[2] + d[3]);
                ps += f5_dpp_row<0x128>(ps);
                ps += f5_dpp_row<0x124>(ps);
                ps += f5_dpp_row<0x122>(ps);
                ps += f5_dpp_row<0x121>(ps);
                const float ms = ps * (1.0f / 64.0f);
                const float e0 = d[0] - ms, e1 = d[1] - ms, e2 = d[2] - ms, e3 = d[3] - ms;
                float pq = rok ? (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3) : 0.0f;
                pq += f5_dpp_row<0x128>(pq);
                pq += f5_dpp_row<0x124>(pq);
                pq += f5_dpp_row<0x122>(pq);
                pq += f5_dpp_row<0x121>(pq);
#if F5_F16
                if (p.x16_overflow != nullptr) {             // fail loudly: f5_pack2 saturates, the engine reports the flag (f5_sample_status)
                    const bool ov = rok && !(fmaxf(fmaxf(fabsf(d[0] * sc4[0]), fabsf(d[1] * sc4[1])),
                                                   fmaxf(fabsf(d[2] * sc4[2]), fabsf(d[3] * sc4[3]))) <= 65504.0f);
                    if (__any(ov) && lane == 0) atomicOr(p.x16_overflow, 1);
                }
#endif
                if (rok) {
                    *reinterpret_cast<u32x2*>(p.x16_out + (size_t)grow * p.ldx16 + colbase + chunk * 4) =
                        u32x2{f5_pack2(d[0] * sc4[0], d[1] * sc4[1]), f5_pack2(d[2] * sc4[2], d[3] * sc4[3])};
                    if (chunk == 0)
                        *reinterpret_cast<f5_f32x2*>(p.stats_out + ((size_t)(colbase >> 6) * p.stats_ld + grow) * 2) = f5_f32x2{ps, pq};
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// a wave-uniform pointer the compiler can see is uniform (SGPR pair): lets global_load_lds use the "SGPR base + 32-bit VGPR
// offset" addressing form, i.e. no per-load 64-bit VALU add
__device__ __forceinline__ const char* v2_uniform_ptr(const void* ptr) {
    const uint64_t u = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

}  // namespace F5_NS
