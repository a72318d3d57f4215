// 256 x 256 x 64 MFMA GEMM for gfx950, the large-shape kernel of the DiT block (batch >~ 8):  C[M,N] = A[M,K] W[N,K]^T with the
// fused epilogues of gemm_dev.hpp (QKV + bias + RoPE + head split dit.py:136-158, out-proj / FF2 + gated fp32 residual update
// dit.py:167-173,319,323, FF1 + GELU-tanh dit.py:94-99).
//
// Geometry: 512 threads = 8 waves (2 x 4), wave tile 128 x 64 = 4 x 2 accumulators of v_mfma_f32_32x32x16 (128 registers);
// operands go HBM -> LDS with global_load_lds (lane-linear image, the 16-byte XOR swizzle sits on the SOURCE address and on the
// read); LDS = [A0, A1, B0, B1][2 K tiles][128 x 64] = 128 KB, one workgroup per CU.
//
// Schedule ("role split"): the two waves of a SIMD are wave w (rows 0-127, group 0) and wave w + 4 (rows 128-255, group 1).  A K
// tile is 4 phases; a phase is a LOAD segment (fragment ds_reads + global_load_lds issue + the counted vmcnt) and a MATRIX segment
// (8 MFMAs = 256 cycles of the SIMD's matrix pipe), each closed by a workgroup barrier.  Group 1 runs ONE barrier behind group 0,
// so in every interval between two barriers one wave of each SIMD streams MFMAs (s_setprio 1) while its partner reads LDS and
// issues loads: the matrix pipe alternates between the two waves instead of idling through a lock-step read burst (the previous
// kernel: all 8 waves read, then all 8 multiply; 1 150-1 200 TF main loop = 55 % pipe time).  CDNA4 guide, "256^2 8-phase template".
//
//   interval        2q              2q+1            2q+2
//   group 0     LOAD(q)          MATRIX(q)       LOAD(q+1)
//   group 1     MATRIX(q-1)      LOAD(q)         MATRIX(q)
//
// Phases of K tile t in ring buffer par = t & 1 (a wave reads ONE A half, wm, and ONE B half, wn >> 1):
//   1: read A rows 0-63 (8 ds_read_b128), B cols 0-31 (4)    quadrant (0,0)    issue A0(t+1) -> buffer 1-par
//   2: read B cols 32-63 (4)                                  quadrant (0,1)    issue A1(t+1) -> buffer 1-par
//   3: read A rows 64-127 (8)                                 quadrant (1,1)
//   4: -                                                      quadrant (1,0)    issue B0(t+2), B1(t+2) -> buffer par; vmcnt(4)
// Hazards (a staged buffer is ordered for a reader only by every issuing wave's counted vmcnt followed by a barrier the reader
// has passed; the groups are one interval apart):
//   RAW: tile t+1 is complete when both groups have passed the vmcnt(4) of phase (t,4) (intervals 8t+6 / 8t+7: everything but
//        the two B halves just issued has landed); its first reader is group 0's LOAD(t+1,1) in interval 8t+8.
//   WAR: a slot last read in phase q (reads retired by the lgkmcnt wait at the head of MATRIX(q): intervals 2q+1 / 2q+2) is
//        re-staged in phase >= q+2 (intervals >= 2q+4 / 2q+5).  A[1-par] was last read in phase (t-1,3) -> issued in (t,1), (t,2);
//        B[par] was last read in phase (t,2) -> issued in (t,4).
// The accumulation order of every accumulator is the same as in the lock-step kernel: results are bit-identical to it.
//
// Row tail: when the last row tile holds <= 128 valid rows it is numbered LAST (workgroup ids are dealt to XCDs round-robin,
// so the cheap tiles spread over the XCDs and fill the tail of the launch) and its dead waves / row halves skip their reads and
// MFMAs, the dead A half is not staged: QKV at M = 59 968 is 2 808 whole tiles (10.97 rounds of 256) + 12 quarter tiles instead
// of 2 820 tiles = 11.02 rounds that cost 12.
#include "gemm.hpp"
#include "gemm_dev.hpp"

namespace F5_NS {

#if F5_PROBE
// measurement build: a timeline of every workgroup (100 MHz wall clock): [0] start, [1] prologue staged, [2] K loop done (wave 0),
// [3] epilogue done wave 0, [4] epilogue done wave 4, [5] HW_ID, [6] XCC_ID, [7] K loop done (wave 4); read by f5_probe_read_ts
#define F5_PROBE_MAXWG 4096
__device__ unsigned long long f5_probe_ts[F5_PROBE_MAXWG * 8];
#define F5_PROBE_TS(idx_, wave_)                                                                              \
    if (lane == 0 && wave == (wave_) && bid < F5_PROBE_MAXWG) f5_probe_ts[bid * 8 + (idx_)] = wall_clock64();
#else
#define F5_PROBE_TS(idx_, wave_)
#endif

#define G256_BARRIER()                         \
    {                                          \
        __builtin_amdgcn_sched_barrier(0);     \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_s_barrier();          \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
    }

// QT (EPI_QKV_ROPE with pair-major rotation tables): q / k column tiles accumulated transposed (staged_epilogue_tr_rope), V tiles
// straight
// FOLD: the LN-fold consumer (F5GemmArgs::fold_*) as its own instantiation: its requests are issued and pinned on every path of the
// kernel (gemm_dev.hpp fold_prefetch_pin), the plain kernels carry none of it
// FOLD: 0 = plain, 1 = row factors from memory (fold_rowf), 2 = merged here from the producer's slice statistics (fold_stats)
template <int EPI, bool QT, int FOLD>
__global__ __launch_bounds__(512) void f5_gemm256_kernel(F5GemmArgs p, int tiles_n, int nfull, int tiles_mf) {
    __shared__ __attribute__((aligned(16))) op16_t smem[2 * 4 * V2_HALF_ELEMS];   // [A0,A1,B0,B1][ring buffer][128*64]

    const int bid = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int kt = p.K / BK;
    const int T = kt * p.nseg;
#if F5_PROBE
    F5_PROBE_TS(0, 0);
    if (lane == 0 && wave == 0 && bid < F5_PROBE_MAXWG) {
        f5_probe_ts[bid * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
        f5_probe_ts[bid * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) & 0xF;   // HW_REG_XCC_ID (| tn << 8 | tm << 20 below)
    }
    {   // measurement build: phase-shift part of the first round of workgroups (do the CUs' synchronised epilogue bursts cost time?)
        const int sg = (p.debug_flags >> 20) & 0xff;       // delay step, units of 0.5 us (100 MHz wall clock)
        const int mode = (p.debug_flags >> 28) & 3;        // who waits: 0 = every other CU slot of an XCD, 1 = every other XCD, 2 = four phases
        if (sg != 0 && bid < 256) {
            const int ph = mode == 0 ? ((bid >> 3) & 1) : mode == 1 ? (bid & 1) : (((bid >> 3) & 1) + 2 * (bid & 1));
            if (ph != 0) {
                const unsigned long long t0 = wall_clock64(), dt = (unsigned long long)ph * sg * 50ull;
                while (wall_clock64() - t0 < dt) __builtin_amdgcn_s_sleep(20);
            }
        }
    }
#endif

    // ---- workgroup id -> tile.  Whole tiles: each XCD (private L2; workgroup b runs on XCD b % 8) walks a contiguous chunk of
    // the tile list; the list is n-fastest, or BAND-major (bands of nband column tiles walked row by row) so that a chunk's W
    // panels stay in the XCD's 4 MB L2 while the A panels stream through once.  Tail tiles (<= 128 valid rows) come last.
    int tm, tn;
    if (bid < nfull) {
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = nfull >> 3, r = nfull & 7;
        const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        if (p.nband > 0) {
            const int per_band = tiles_mf * p.nband;
            const int band = tile / per_band, r_ = tile - band * per_band;
            tm = r_ / p.nband;
            tn = band * p.nband + (r_ - tm * p.nband);
        } else {
            tm = tile / tiles_n;
            tn = tile - tm * tiles_n;
        }
    } else {
        tm = tiles_mf;
        tn = bid - nfull;
    }
    const int m0 = tm * 256, n0 = tn * 256;
#if F5_PROBE
    if (lane == 0 && wave == 0 && bid < F5_PROBE_MAXWG) f5_probe_ts[bid * 8 + 6] |= ((unsigned long long)tn << 8) | ((unsigned long long)tm << 20);
#endif
    const int rows_wave = p.M - m0 - wm * 128;        // valid rows of this wave's 128-row half (wave-uniform)
    const bool act_lo = rows_wave > 0;                // phases 1, 2: rows 0-63 of the half
    const bool act_hi = rows_wave > 64;               // phases 3, 4: rows 64-127
    const bool stage_a1 = p.M - m0 > 128;             // workgroup-uniform: the second A half holds valid rows

    const int frow = lane & 31;
    const int fk = lane >> 5;
    // fragment read pointers, one per 16-wide K sub-step; ring buffer, row block and quadrant are ds_read immediates
    const op16_t* pa[4];
    const op16_t* pb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        pa[ks] = smem + (wm * 2) * V2_HALF_ELEMS + swz_off(frow, ks * 2 + fk);
        pb[ks] = smem + ((2 + (wn >> 1)) * 2) * V2_HALF_ELEMS + swz_off((wn & 1) * 64 + frow, ks * 2 + fk);
    }

    // ---- staging addresses: 2 chunks per thread per half tile.  Linear chunk q_ = j*512 + tid of the [128][8] half-tile image:
    // row = q_ >> 3, slot = q_ & 7, source chunk = slot ^ ((row >> 1) & 7)
    uint32_t srcA[2][2], srcB[2][2];   // [half][j] BYTE offsets (without k0), added to a uniform pointer (saddr form)
    int ldsoff[2];                     // [j] element offset of this WAVE's 1 KB destination inside a half tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q_ = j * 512 + tid;
        const int row = q_ >> 3, slot = q_ & 7;
        const int chunk = slot ^ ((row >> 1) & 7);
        ldsoff[j] = (j * 512 + wave * 64) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int gr = m0 + h * 128 + row;
            if (gr > p.M - 1) gr = p.M - 1;
            if (p.a_row_mod > 0) gr = gr % p.a_row_mod;
            srcA[h][j] = ((uint32_t)gr * (uint32_t)p.lda + chunk * 8) * 2u;
            srcB[h][j] = ((uint32_t)(n0 + h * 128 + row) * (uint32_t)p.ldw + chunk * 8) * 2u;
        }
    }
#define G256_ISSUE_A(par_, h_, Ap_, k0_)                                                            \
    {                                                                                               \
        op16_t* dst_ = smem + ((h_) * 2 + (par_)) * V2_HALF_ELEMS;                                  \
        const char* src_ = reinterpret_cast<const char*>(Ap_);                                      \
        const uint32_t kb_ = (uint32_t)(k0_) * 2u;                                                  \
        glds16(reinterpret_cast<const op16_t*>(src_ + (srcA[(h_)][0] + kb_)), dst_ + ldsoff[0]);    \
        glds16(reinterpret_cast<const op16_t*>(src_ + (srcA[(h_)][1] + kb_)), dst_ + ldsoff[1]);    \
    }
#define G256_ISSUE_B(par_, h_, Wp_, k0_)                                                            \
    {                                                                                               \
        op16_t* dst_ = smem + ((2 + (h_)) * 2 + (par_)) * V2_HALF_ELEMS;                            \
        const char* src_ = reinterpret_cast<const char*>(Wp_);                                      \
        const uint32_t kb_ = (uint32_t)(k0_) * 2u;                                                  \
        glds16(reinterpret_cast<const op16_t*>(src_ + (srcB[(h_)][0] + kb_)), dst_ + ldsoff[0]);    \
        glds16(reinterpret_cast<const op16_t*>(src_ + (srcB[(h_)][1] + kb_)), dst_ + ldsoff[1]);    \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // LN fold: what the epilogue needs from memory is requested here, ahead of the operand loads (gemm_dev.hpp fold_prefetch_pin)
    static_assert(!FOLD || EPI == EPI_QKV_ROPE || EPI == EPI_GELU_TANH, "fold consumers");
    constexpr bool fold = FOLD != 0;
    const int row0 = m0 + wm * 128, col0 = n0 + wn * 64;
    // 16-bit row-major outputs: the tile is accumulated TRANSPOSED (operands swapped in every MFMA) for staged_epilogue_tr.  Both
    // loop copies end in their own epilogue: no join with 128 live accumulator registers.
    constexpr bool TR_EPI = (EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_GELU_ERF_BF16);
    const bool tr_path = (TR_EPI && (p.debug_flags & 16384) == 0) || (QT && n0 < 2 * p.dmodel);       // workgroup-uniform
    FoldPre fpre;
    fold_prefetch_clear(fpre);
    FoldStatsPre<FOLD == 2 ? 32 : 8> spre;
    fold_stats_clear(spre);
    if (FOLD == 1) {
        if (tr_path) fold_prefetch_tr<4>(p, fpre, row0, col0, lane);
        else fold_prefetch_v<4>(p, fpre, row0, col0, lane);
    }
    if constexpr (FOLD == 2) {
        if (tr_path) fold_stats_request_tr<4>(p, spre, fpre, row0, col0, lane);
        else fold_stats_request_v<4>(p, spre, fpre, row0, col0, lane);
    }

    // ---- prologue: K tile 0 (4 halves) + the B halves of K tile 1; running state = (segment, K offset) of tile tt+1 (A halves)
    // and tile tt+2 (B halves); segment 0 = A.hi W.hi, 1 = A.lo W.hi, 2 = A.hi W.lo (bf16x3)
    int a_seg = 0, a_k0 = 0, b_seg = 0, b_k0 = 0;
    {
        G256_ISSUE_A(0, 0, p.A[0], 0);
        if (stage_a1) G256_ISSUE_A(0, 1, p.A[0], 0);
        G256_ISSUE_B(0, 0, p.W[0], 0);
        G256_ISSUE_B(0, 1, p.W[0], 0);
        a_k0 = BK;
        if (a_k0 == p.K) {
            a_k0 = 0;
            ++a_seg;
        }
        b_seg = a_seg;
        b_k0 = a_k0;
    }
    if (fold) {                        // K tile 0 and the (older) fold requests waited for in full and pinned BEFORE tile 1's B halves
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // are issued: those have most of a K tile to land either way
        if constexpr (FOLD == 2) {                           // the slice statistics have landed: merge them into this lane's row factors
            fold_stats_pin(spre);
            if (tr_path) fold_stats_finish_tr<4>(p, spre, fpre, row0, lane, p.fold_mean_out != nullptr && n0 == 0 && wn == 0);
            else fold_stats_finish_v<4>(p, spre, fpre);
        }
        fold_prefetch_pin(fpre);
    }
    if (1 < T) {
        const op16_t* Wp1 = b_seg == 2 ? p.W[1] : p.W[0];
        G256_ISSUE_B(1, 0, Wp1, b_k0);
        G256_ISSUE_B(1, 1, Wp1, b_k0);
        b_k0 += BK;
        if (b_k0 == p.K) {
            b_k0 = 0;
            ++b_seg;
        }
        if (!fold) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    G256_BARRIER();
    F5_PROBE_TS(1, 0);
    if (wm == 1) G256_BARRIER();       // group 1 starts one interval late (paired with group 0's first LOAD barrier)

    op16x8 af[2][4], bfr[2][4];
    const op16_t* Apn = a_seg == 1 ? p.A[1] : p.A[0];   // operand (bf16x3 segment) pointers of the tiles being staged
    const op16_t* Wpn = b_seg == 2 ? p.W[1] : p.W[0];
#define G256_FRAG_A(PAR, ks, rowoff) (*reinterpret_cast<const op16x8*>(pa[ks] + (PAR) * V2_HALF_ELEMS + (rowoff) * BK))
#define G256_FRAG_B(PAR, ks, rowoff) (*reinterpret_cast<const op16x8*>(pb[ks] + (PAR) * V2_HALF_ELEMS + (rowoff) * BK))
#define G256_MM(TR_, A_, B_, C_) ((TR_) ? F5_MFMA32(B_, A_, C_, 0, 0, 0) : F5_MFMA32(A_, B_, C_, 0, 0, 0))
#define G256_MATRIX(TR_, ON_, R0_, NQ_)                                                                                     \
    {                                                                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                                      \
        if (ON_) {                                                                                                          \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                \
                _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                            \
                    acc[(R0_) + mb][NQ_] = G256_MM(TR_, af[mb][ks], bfr[NQ_][ks], acc[(R0_) + mb][NQ_]);                    \
        }                                                                                                                   \
        __builtin_amdgcn_s_setprio(0);                                                                                      \
        G256_BARRIER();                                                                                                     \
    }
#define G256_KSTEP(PAR, tt, TR_)                                                                                            \
    {                                                                                                                       \
        /* phase 1 */                                                                                                       \
        if (act_lo) {                                                                                                       \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                              \
                _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) af[mb][ks] = G256_FRAG_A(PAR, ks, mb * 32);                \
                bfr[0][ks] = G256_FRAG_B(PAR, ks, 0);                                                                       \
            }                                                                                                               \
        }                                                                                                                   \
        if ((tt) + 1 < T) G256_ISSUE_A(1 - PAR, 0, Apn, a_k0);                                                              \
        G256_BARRIER();                                                                                                     \
        G256_MATRIX(TR_, act_lo, 0, 0);                                                                                     \
        /* phase 2 */                                                                                                       \
        if (act_lo) {                                                                                                       \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) bfr[1][ks] = G256_FRAG_B(PAR, ks, 32);                         \
        }                                                                                                                   \
        if ((tt) + 1 < T && stage_a1) G256_ISSUE_A(1 - PAR, 1, Apn, a_k0);                                                  \
        G256_BARRIER();                                                                                                     \
        G256_MATRIX(TR_, act_lo, 0, 1);                                                                                     \
        /* phase 3 */                                                                                                       \
        if (act_hi) {                                                                                                       \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                \
                _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) af[mb][ks] = G256_FRAG_A(PAR, ks, 64 + mb * 32);           \
        }                                                                                                                   \
        G256_BARRIER();                                                                                                     \
        G256_MATRIX(TR_, act_hi, 2, 1);                                                                                     \
        /* phase 4: both B halves of tile t+2 into this tile's buffer (last read two phases ago), then the tile-t+1 wait */  \
        if ((tt) + 2 < T) {                                                                                                 \
            G256_ISSUE_B(PAR, 0, Wpn, b_k0);                                                                                \
            G256_ISSUE_B(PAR, 1, Wpn, b_k0);                                                                                \
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                                \
        } else {                                                                                                            \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
        }                                                                                                                   \
        G256_BARRIER();                                                                                                     \
        G256_MATRIX(TR_, act_hi, 2, 0);                                                                                     \
        a_k0 += BK;                                                                                                         \
        if (a_k0 == p.K) {                                                                                                  \
            a_k0 = 0;                                                                                                       \
            ++a_seg;                                                                                                        \
            Apn = a_seg == 1 ? p.A[1] : p.A[0];                                                                             \
        }                                                                                                                   \
        b_k0 += BK;                                                                                                         \
        if (b_k0 == p.K) {                                                                                                  \
            b_k0 = 0;                                                                                                       \
            ++b_seg;                                                                                                        \
            Wpn = b_seg == 2 ? p.W[1] : p.W[0];                                                                             \
        }                                                                                                                   \
    }

    op16_t* stage = smem + wave * 8192;                 // this wave's private 16 KB of epilogue staging
    float* fl = reinterpret_cast<float*>(stage + 6144); // its last 4 KB: row factors of a folded LN-modulate (the staged tiles use <= 9 KB)
    if (tr_path) {
        for (int tt = 0; tt < T; tt += 2) {
            G256_KSTEP(0, tt, true);
            if (tt + 1 < T) G256_KSTEP(1, tt + 1, true);
        }
        F5_PROBE_TS(2, 0);
        F5_PROBE_TS(7, 4);
        if (wm == 0) G256_BARRIER();                    // group 0 waits for group 1's last MATRIX segment: the ring is dead
        if ((p.debug_flags & 1) || !act_lo) return;     // flag 1 = timing experiment: main loop only
        if (fold) {                                     // LN-modulate folded into this GEMM (F5GemmArgs::fold_*; workgroup-uniform)
            if (QT) staged_epilogue_tr_rope<4, 2, true>(p, acc, stage, row0, col0, lane, fl, &fpre);
            else staged_epilogue_tr<EPI, 4, 2, true>(p, acc, stage, row0, col0, lane, fl, &fpre);
            F5_PROBE_TS(3, 0);
            F5_PROBE_TS(4, 4);
            return;
        }
        if (QT) staged_epilogue_tr_rope<4, 2>(p, acc, stage, row0, col0, lane);
        else staged_epilogue_tr<EPI, 4, 2>(p, acc, stage, row0, col0, lane);
        F5_PROBE_TS(3, 0);
        F5_PROBE_TS(4, 4);
        return;
    }
    for (int tt = 0; tt < T; tt += 2) {
        G256_KSTEP(0, tt, false);
        if (tt + 1 < T) G256_KSTEP(1, tt + 1, false);
    }
    F5_PROBE_TS(2, 0);
    F5_PROBE_TS(7, 4);
    if (wm == 0) G256_BARRIER();
    if (p.debug_flags & 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    if (!act_lo) return;
    if (QT) {                                                               // (the q / k tiles finished above)
        if (fold) staged_epilogue_bf16<EPI, 4, 2, true, true>(p, acc, stage, row0, col0, lane, fl, &fpre);
        else staged_epilogue_bf16<EPI, 4, 2, true>(p, acc, stage, row0, col0, lane);
    } else if (EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_GELU_ERF_BF16 || EPI == EPI_QKV_ROPE) {
        staged_epilogue_bf16<EPI, 4, 2>(p, acc, stage, row0, col0, lane);
    } else if (EPI == EPI_RESID_GATE) {
        staged_epilogue_resid<4, 2>(p, acc, reinterpret_cast<float*>(stage), row0, col0, lane);
    } else {
        gemm_epilogue<EPI, 4, 2>(p, acc, m0, n0, wm, wn, lane);
    }
    F5_PROBE_TS(3, 0);
    F5_PROBE_TS(4, 4);
#undef G256_KSTEP
#undef G256_MATRIX
#undef G256_MM
#undef G256_FRAG_A
#undef G256_FRAG_B
#undef G256_ISSUE_A
#undef G256_ISSUE_B
}

int f5_gemm_nband = 4;   // column-tile band width of the tile numbering (0 = n fastest).  4 = 2 MB of W per band at K = 1024:
                         // sample() at batch 32 1 244 vs 1 260-1 275 ms (profiles/r02/gemm_nband_ab.txt)

template <int EPI>
static int launch256(const F5GemmArgs& a, hipStream_t stream) {
    const int tiles_n = a.N / 256;
    const int tail_rows = a.M % 256;
    const bool cheap_tail = tail_rows != 0 && tail_rows <= 128;     // numbered last, runs a fraction of a tile
    const int tiles_mf = cheap_tail ? a.M / 256 : f5_cdiv(a.M, 256);
    const int nfull = tiles_mf * tiles_n;
    const int ntiles = nfull + (cheap_tail ? tiles_n : 0);
    F5GemmArgs ab = a;
    ab.nband = (f5_gemm_nband > 0 && tiles_n > f5_gemm_nband && tiles_n % f5_gemm_nband == 0) ? f5_gemm_nband : 0;
    // staged_epilogue_tr reads the bias as 16-byte quads: an unaligned bias vector takes the straight-order path
    if (a.bias != nullptr && (reinterpret_cast<uintptr_t>(a.bias) & 15) != 0) ab.debug_flags |= 16384;
    constexpr bool CAN_FOLD = EPI == EPI_QKV_ROPE || EPI == EPI_GELU_TANH;
    // (f5_launch_gemm has checked the fold's preconditions: transposed q / k tiles, K = 1024 for the statistics form)
    const int fold = !CAN_FOLD ? 0 : (ab.fold_stats != nullptr ? 2 : (ab.fold_rowf != nullptr ? 1 : 0));
    constexpr bool QT = EPI == EPI_QKV_ROPE;
    if (EPI == EPI_QKV_ROPE && ab.rope_g4k != nullptr) {
        if (fold == 2) hipLaunchKernelGGL((f5_gemm256_kernel<EPI, QT, CAN_FOLD ? 2 : 0>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, nfull, tiles_mf);
        else if (fold == 1) hipLaunchKernelGGL((f5_gemm256_kernel<EPI, QT, CAN_FOLD ? 1 : 0>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, nfull, tiles_mf);
        else hipLaunchKernelGGL((f5_gemm256_kernel<EPI, QT, 0>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, nfull, tiles_mf);
    } else {
        if (fold == 2) hipLaunchKernelGGL((f5_gemm256_kernel<EPI, false, CAN_FOLD ? 2 : 0>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, nfull, tiles_mf);
        else if (fold == 1) hipLaunchKernelGGL((f5_gemm256_kernel<EPI, false, CAN_FOLD ? 1 : 0>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, nfull, tiles_mf);
        else hipLaunchKernelGGL((f5_gemm256_kernel<EPI, false, 0>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, nfull, tiles_mf);
    }
    F5_LAUNCH_CHECK();
    return 0;
}

// Preconditions (checked by f5_launch_gemm): N % 256 == 0, M >= 256, K % 64 == 0, operands below 2 GiB (32-bit byte offsets)
int f5_launch_gemm256(const F5GemmArgs& a, int epi, hipStream_t stream) {
    F5_REQUIRE(a.N % 256 == 0 && a.M >= 256 && a.K % BK == 0, "gemm256: needs N %% 256 == 0, M >= 256, K %% 64 == 0");
    F5_REQUIRE((size_t)(a.a_row_mod > 0 ? a.a_row_mod : a.M) * a.lda < (1ull << 31) && (size_t)(a.N + 256) * a.ldw < (1ull << 31),
               "gemm256: operands must stay below 4 GiB (32-bit byte offsets)");
    switch (epi) {
        case EPI_F32: return launch256<EPI_F32>(a, stream);
        case EPI_BF16: return launch256<EPI_BF16>(a, stream);
        case EPI_GELU_TANH: return launch256<EPI_GELU_TANH>(a, stream);
        case EPI_GELU_ERF: return launch256<EPI_GELU_ERF>(a, stream);
        case EPI_RESID_GATE: return launch256<EPI_RESID_GATE>(a, stream);
        case EPI_QKV_ROPE: return launch256<EPI_QKV_ROPE>(a, stream);
        case EPI_ADDROWS: return launch256<EPI_ADDROWS>(a, stream);
        case EPI_RESID_KEEP: return launch256<EPI_RESID_KEEP>(a, stream);
        case EPI_GELU_ERF_BF16: return launch256<EPI_GELU_ERF_BF16>(a, stream);
        default: f5_set_error("gemm256: unknown epilogue %d", epi); return 2;
    }
}
}  // namespace F5_NS

#if F5_PROBE && F5_F16
extern "C" int f5_probe_read_ts(unsigned long long* host, int words, int clear) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(f5hf::f5_probe_ts), (size_t)words * 8) != hipSuccess) return 1;
    if (clear) {
        void* d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(f5hf::f5_probe_ts)) != hipSuccess || hipMemset(d, 0, sizeof(unsigned long long) * F5_PROBE_MAXWG * 8) != hipSuccess)
            return 2;
    }
    return 0;
}
#endif
