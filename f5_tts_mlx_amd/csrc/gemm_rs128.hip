// 128 x 256 x 64 MFMA GEMM for gfx950 with the role-split schedule of gemm256.hip, for SMALL row counts (batch 1: M = 2 x 937
// rows): one round of 8-wave workgroups in which every workgroup is a latency chain of K / 64 steps.
//
// Why: at batch 1 a DiT-block GEMM is ONE round of workgroups; its time is a fixed cost plus K / 64 times the duration of one K
// step of one workgroup.  The lock-step ring kernels (gemm.hip) spend ~1.4 us per step on a 128 x 256 tile -- three times the
// MFMA time of the step -- because the fragment reads wait for the landed tile, the MFMAs wait for the reads, and all eight waves
// do the same thing at the same time (tools/ring_ablate.py: loads, reads and MFMAs overlap pairwise, never all three).  The chip
// is far from its power limit here (MFMA busy ~25 %), so unlike at batch 32 the schedule IS the lever.
//
// Geometry: 512 threads = 8 waves = 2 groups (rows 0-63 / 64-127) x 4 column waves, wave tile 64 x 64 = 2 x 2 accumulators of
// v_mfma_f32_32x32x16 (64 registers).  LDS = 3 ring slots x [A0 (64 x 64), A1, B0 (128 x 64), B1] = 3 x 48 KB = 144 KB.
// Group 1 runs one barrier behind group 0: in every interval one wave of each SIMD multiplies (8 MFMAs, s_setprio 1) while
// its partner reads fragments and issues global_load_lds.
//
// Phases of K step t (slot t % 3; a wave reads ONE A half -- its group's -- and ONE B half, wn >> 1):
//   1: read A (2 x 4 ds_read_b128), B columns 0-31 (4)     accumulators (., 0)     issue A0, A1 of step t+2 -> slot (t+2) % 3
//   2: read B columns 32-63 (4)                            accumulators (., 1)     issue B0, B1 of step t+2 -> slot (t+2) % 3; vmcnt(6)
// Hazards (staggered groups, see gemm256.hip): a slot last read in phase q may be re-staged in phase >= q + 2.  Slot (t+2) % 3
// held step t-1: its A halves were last read in phase (t-1, 1), its B halves in (t-1, 2) -> A re-staged in (t, 1) [+2], B in
// (t, 2) [+2].  RAW: step t+1 (A issued in (t-1, 1), B in (t-1, 2)) is complete when both groups have passed the vmcnt(6) of phase
// (t, 2) -- everything but the 6 loads of step t+2 just issued has landed -- one barrier before its first reader (t+1, 1).
// Accumulation order per accumulator = K ascending in 16-wide chunks, as in every other GEMM kernel here: bit-identical results.
#include "gemm.hpp"
#include "gemm_dev.hpp"

namespace F5_NS {

#define RS_BARRIER()                           \
    {                                          \
        __builtin_amdgcn_sched_barrier(0);     \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_s_barrier();          \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
    }

// FOLD: the LN-fold consumer (F5GemmArgs::fold_*) as its own instantiation -- its requests are issued and pinned on EVERY path of the
// kernel (a request under a run-time condition leaves the compiler's wait bookkeeping "pending" on the other path, and it then guards the
// registers inside the K loop: gemm_dev.hpp fold_prefetch_pin), and the plain kernels carry none of it
// FOLD: 0 = plain, 1 = row factors from memory (fold_rowf), 2 = merged here from the producer's slice statistics (fold_stats)
template <int EPI, bool QT, int FOLD>
__global__ __launch_bounds__(512) void f5_gemm_rs128_kernel(F5GemmArgs p, int tiles_n, int ntiles, int tiles_m) {
    constexpr int AH = 64 * BK;                 // elements of an A half (64 rows)
    constexpr int BH = 128 * BK;                // elements of a B half (128 rows)
    constexpr int SLOT = 2 * AH + 2 * BH;       // 24576 elements = 48 KB
    __shared__ __attribute__((aligned(16))) op16_t smem[3 * SLOT];

    // everything the prologue needs from the argument block is requested in ONE scalar-load clause: left to itself the compiler loads
    // the fields where they are first used -- three dependent s_load / s_waitcnt rounds before the first tile request of a kernel
    // that is a single latency chain
    asm volatile("" ::"s"(p.seq_len), "s"(p.A[0]), "s"(p.W[0]), "s"(p.lda), "s"(p.ldw), "s"(p.K), "s"(p.nseg), "s"(p.M), "s"(p.a_row_mod),
                 "s"(p.debug_flags), "s"(tiles_n), "s"(ntiles), "s"(tiles_m));
    const int bid = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int kt = p.K / BK;
    const int T = kt * p.nseg;

    // Workgroups are dealt to the 8 XCDs round-robin by block id, each XCD has a private L2, and what an L2 does not hold comes
    // over the fabric (the weights of 22 blocks never stay on chip).  Two numberings:
    //  * XCD GRID 2 x 4 (tiles_m even, tiles_n a multiple of 4): XCD x owns row-tile half x & 1 and column-tile quarter x >> 1, so
    //    an L2 fetches HALF of A and a QUARTER of W.  Batch-1 QKV (16 x 12 tiles): 15 + 12.6 MB of operand fetches;
    //  * XCD-chunked list, M fastest (the fallback, and debug flag 4096): an XCD's consecutive tiles share W panels but every L2
    //    fetches ALL of A: 30 + 8.4 MB for the same launch (PMC: 52.6 MB per launch including the 11.5 MB of output).
    const int xcd = bid & 7, idx = bid >> 3;
    int tn, tm;
    if ((tiles_m & 1) == 0 && (tiles_n & 3) == 0 && (p.debug_flags & 4096) == 0) {
        const int hm = tiles_m >> 1, qn = tiles_n >> 2;
        const int tl = idx / hm;
        tm = (xcd & 1) * hm + (idx - tl * hm);
        tn = (xcd >> 1) * qn + tl;
    } else {
        const int q = ntiles >> 3, r = ntiles & 7;
        const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tn = tile / tiles_m;
        tm = tile - tn * tiles_m;
    }
    const int n0 = tn * 256;
    // Row tiles.  Plain: 128 consecutive rows of the [M][K] operand.  EPI_QKV_ROPE (rows = [batch element][position]): tiles are
    // laid out PER BATCH ELEMENT (tiles_per_elem = ceil(seq_len / 128) each) and never straddle two of them, so a tile's first
    // position is a multiple of 128 and every 8-token piece of V^T[b, h, d][n] it writes is a 16-byte ALIGNED store -- with
    // consecutive-row tiles and seq_len = 937 the pieces of every element but the first start at odd positions, and 16-byte
    // stores at 2-byte alignment cost 6 us of a 28 us launch (profiles/r03/qkv_b1_alignment.txt).  m_end = one past the last row
    // this tile may touch.
    int m0, m_end;
    if (EPI == EPI_QKV_ROPE) {
        const int tpe = (p.seq_len + 127) >> 7;
        const int b = tm / tpe, i = tm - b * tpe;
        m0 = b * p.seq_len + i * 128;
        m_end = (b + 1) * p.seq_len;
    } else {
        m0 = tm * 128;
        m_end = p.M;
    }
    p.M = m_end;                                          // staging clamp and epilogue row bounds of THIS tile
    const bool active = m_end - m0 - wm * 64 > 0;         // this wave's 64 rows hold valid rows (wave-uniform)

    const int frow = lane & 31;
    const int fk = lane >> 5;
    const op16_t* pa[4];
    const op16_t* pb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        pa[ks] = smem + wm * AH + swz_off(frow, ks * 2 + fk);
        pb[ks] = smem + 2 * AH + (wn >> 1) * BH + swz_off((wn & 1) * 64 + frow, ks * 2 + fk);
    }
    // staging: an A half is 512 16-byte chunks = 1 per thread, a B half 1024 = 2 per thread.  Linear chunk c of a half: row = c >> 3,
    // slot = c & 7, source chunk = slot ^ ((row >> 1) & 7)
    uint32_t srcA[2], srcB[2][2];
    int dstA, dstB[2];
    {
        const int row = tid >> 3, slot = tid & 7;
        const int chunk = slot ^ ((row >> 1) & 7);
        dstA = wave * 64 * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int gr = m0 + h * 64 + row;
            if (gr > p.M - 1) gr = p.M - 1;
            if (p.a_row_mod > 0) gr = gr % p.a_row_mod;
            srcA[h] = ((uint32_t)gr * (uint32_t)p.lda + chunk * 8) * 2u;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = j * 512 + tid;
        const int row = c >> 3, slot = c & 7;
        const int chunk = slot ^ ((row >> 1) & 7);
        dstB[j] = (j * 512 + wave * 64) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) srcB[h][j] = ((uint32_t)(n0 + h * 128 + row) * (uint32_t)p.ldw + chunk * 8) * 2u;
    }
    // running (segment, K byte offset) of the next A pair / B pair to stage (bf16x3: segment 0 = A.hi W.hi, 1 = A.lo W.hi, 2 = A.hi W.lo)
    int a_seg = 0, b_seg = 0;
    uint32_t a_kb = 0, b_kb = 0;
    const uint32_t k_bytes = (uint32_t)p.K * 2u;
#define RS_ISSUE_A(off_)                                                                                      \
    {                                                                                                         \
        const char* src_ = reinterpret_cast<const char*>(a_seg == 1 ? p.A[1] : p.A[0]);                       \
        glds16(reinterpret_cast<const op16_t*>(src_ + (srcA[0] + a_kb)), smem + (off_) + dstA);               \
        glds16(reinterpret_cast<const op16_t*>(src_ + (srcA[1] + a_kb)), smem + (off_) + AH + dstA);          \
        a_kb += BK * 2u;                                                                                      \
        if (a_kb == k_bytes) {                                                                                \
            a_kb = 0;                                                                                         \
            ++a_seg;                                                                                          \
        }                                                                                                     \
    }
#define RS_ISSUE_B(off_)                                                                                      \
    {                                                                                                         \
        const char* src_ = reinterpret_cast<const char*>(b_seg == 2 ? p.W[1] : p.W[0]);                       \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                         \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                     \
                glds16(reinterpret_cast<const op16_t*>(src_ + (srcB[h][j] + b_kb)), smem + (off_) + 2 * AH + h * BH + dstB[j]); \
        b_kb += BK * 2u;                                                                                      \
        if (b_kb == k_bytes) {                                                                                \
            b_kb = 0;                                                                                         \
            ++b_seg;                                                                                          \
        }                                                                                                     \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // LN fold: what the epilogue needs from memory is requested here, ahead of the operand loads (gemm_dev.hpp fold_prefetch_pin)
    static_assert(!FOLD || EPI == EPI_QKV_ROPE || EPI == EPI_GELU_TANH, "fold consumers");
    constexpr bool fold = FOLD != 0;
    const int row0 = m0 + wm * 64, col0 = n0 + wn * 64;
    constexpr bool TR_EPI = (EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_GELU_ERF_BF16);
    const bool tr_path = (TR_EPI && (p.debug_flags & 16384) == 0) || (QT && n0 < 2 * p.dmodel);       // workgroup-uniform
    FoldPre fpre;
    fold_prefetch_clear(fpre);
    FoldStatsPre<FOLD == 2 ? 16 : 8> spre;
    fold_stats_clear(spre);
    if (FOLD == 1) {                                    // (waves past the last row read clamped rows and never use them)
        if (tr_path) fold_prefetch_tr<2>(p, fpre, row0, col0, lane);
        else fold_prefetch_v<2>(p, fpre, row0, col0, lane);
    }
    if constexpr (FOLD == 2) {
        if (tr_path) fold_stats_request_tr<2>(p, spre, fpre, row0, col0, lane);
        else fold_stats_request_v<2>(p, spre, fpre, row0, col0, lane);
    }

    // ---- prologue: steps 0 and 1 (slots 0, 1); step 0 must have landed.  With the fold, step 0 (and the requests above, which are
    // older) is waited for in full and pinned before step 1 is issued: step 1 has a whole K step to land either way
    RS_ISSUE_A(0);
    RS_ISSUE_B(0);
    if (FOLD == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fold_prefetch_pin(fpre);
    }
    if (1 < T) {
        RS_ISSUE_A(SLOT);
        RS_ISSUE_B(SLOT);
        if (FOLD != 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (FOLD == 2) {
        // the statistics form serves single-round launches, where the time of the kernel is one workgroup's chain: step 1 is already in
        // flight (the wait above left its 6 loads pending; the statistics and step 0 are older and have landed), the merge below runs under it
        fold_stats_pin(spre);
        if (tr_path) fold_stats_finish_tr<2>(p, spre, fpre, row0, lane, p.fold_mean_out != nullptr && n0 == 0 && wn == 0);
        else fold_stats_finish_v<2>(p, spre, fpre);
        fold_prefetch_pin(fpre);
    }
    RS_BARRIER();
    if (wm == 1) RS_BARRIER();          // group 1 starts one interval late

    int rd = 0, wr = 2 * SLOT;          // element offsets of the slot of step t and of the slot re-staged during step t (step t+2)
    op16x8 af[2][4], bfr[2][4];
#define RS_MM(TR_, A_, B_, C_) ((TR_) ? F5_MFMA32(B_, A_, C_, 0, 0, 0) : F5_MFMA32(A_, B_, C_, 0, 0, 0))
#define RS_MATRIX(TR_, NQ_)                                                                                                 \
    {                                                                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                                      \
        if (active) {                                                                                                       \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                \
                _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                            \
                    acc[mb][NQ_] = RS_MM(TR_, af[mb][ks], bfr[NQ_][ks], acc[mb][NQ_]);                                      \
        }                                                                                                                   \
        __builtin_amdgcn_s_setprio(0);                                                                                      \
        RS_BARRIER();                                                                                                       \
    }
#define RS_STEP(tt, TR_)                                                                                                    \
    {                                                                                                                       \
        /* phase 1 */                                                                                                       \
        if (active) {                                                                                                       \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                              \
                _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                            \
                    af[mb][ks] = *reinterpret_cast<const op16x8*>(pa[ks] + rd + mb * 32 * BK);                              \
                bfr[0][ks] = *reinterpret_cast<const op16x8*>(pb[ks] + rd);                                                 \
            }                                                                                                               \
        }                                                                                                                   \
        if ((tt) + 2 < T) RS_ISSUE_A(wr);                                                                                   \
        RS_BARRIER();                                                                                                       \
        RS_MATRIX(TR_, 0);                                                                                                  \
        /* phase 2 */                                                                                                       \
        if (active) {                                                                                                       \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) bfr[1][ks] = *reinterpret_cast<const op16x8*>(pb[ks] + rd + 32 * BK); \
        }                                                                                                                   \
        if ((tt) + 2 < T) {                                                                                                 \
            RS_ISSUE_B(wr);                                                                                                 \
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   /* step t+1 landed; the 6 loads of step t+2 stay in flight */ \
        } else {                                                                                                            \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                \
        }                                                                                                                   \
        RS_BARRIER();                                                                                                       \
        RS_MATRIX(TR_, 1);                                                                                                  \
        wr = rd;                                                                                                            \
        rd += SLOT;                                                                                                         \
        if (rd == 3 * SLOT) rd = 0;                                                                                         \
    }

    op16_t* stage = smem + wave * 4608;                 // 9 KB of private epilogue staging per wave (32 x (64 + 8) hi + lo)
    float* fl = reinterpret_cast<float*>(smem + 8 * 4608) + wave * 128;     // behind the eight stages: row factors of a folded LN-modulate
    if (tr_path) {
        for (int tt = 0; tt < T; ++tt) RS_STEP(tt, true);
        if (wm == 0) RS_BARRIER();                      // group 0 waits for group 1's last MATRIX segment: the ring is dead
        if ((p.debug_flags & 1) || !active) return;
        // (requesting the rotation factors before the K loop -- 64 more live registers -- was measured neutral: 25.3 vs 25.6 us)
        if (fold) {                                     // LN-modulate folded into this GEMM (F5GemmArgs::fold_*; workgroup-uniform)
            if (QT) staged_epilogue_tr_rope<2, 2, true>(p, acc, stage, row0, col0, lane, fl, &fpre);
            else staged_epilogue_tr<EPI, 2, 2, true>(p, acc, stage, row0, col0, lane, fl, &fpre);
            return;
        }
        if (QT) staged_epilogue_tr_rope<2, 2>(p, acc, stage, row0, col0, lane);
        else staged_epilogue_tr<EPI, 2, 2>(p, acc, stage, row0, col0, lane);
        return;
    }
    for (int tt = 0; tt < T; ++tt) RS_STEP(tt, false);
    if (wm == 0) RS_BARRIER();
    if (p.debug_flags & 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    if (!active) return;
    if (QT) {
        if (fold) staged_epilogue_bf16<EPI, 2, 2, true, true>(p, acc, stage, row0, col0, lane, fl, &fpre);
        else staged_epilogue_bf16<EPI, 2, 2, true>(p, acc, stage, row0, col0, lane);
    } else if (EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_GELU_ERF_BF16 || EPI == EPI_QKV_ROPE) {
        staged_epilogue_bf16<EPI, 2, 2>(p, acc, stage, row0, col0, lane);
    } else if (EPI == EPI_RESID_GATE) {
        staged_epilogue_resid<2, 2>(p, acc, reinterpret_cast<float*>(stage), row0, col0, lane);
    } else {
        gemm_epilogue<EPI, 2, 2>(p, acc, m0, n0, wm, wn, lane);
    }
#undef RS_STEP
#undef RS_MATRIX
#undef RS_MM
#undef RS_ISSUE_A
#undef RS_ISSUE_B
}

template <int EPI>
static int launch_rs128(const F5GemmArgs& a, hipStream_t stream) {
    // QKV: row tiles per batch element (see the kernel); everything else: consecutive rows
    const int tiles_m = (EPI == EPI_QKV_ROPE) ? (a.M / a.seq_len) * f5_cdiv(a.seq_len, 128) : f5_cdiv(a.M, 128);
    const int tiles_n = a.N / 256;
    const int ntiles = tiles_m * tiles_n;
    F5GemmArgs ab = a;
    if (a.bias != nullptr && (reinterpret_cast<uintptr_t>(a.bias) & 15) != 0) ab.debug_flags |= 16384;
    constexpr bool CAN_FOLD = EPI == EPI_QKV_ROPE || EPI == EPI_GELU_TANH;
    // (f5_launch_gemm has checked the fold's preconditions: transposed q / k tiles, K = 1024 for the statistics form)
    const int fold = !CAN_FOLD ? 0 : (ab.fold_stats != nullptr ? 2 : (ab.fold_rowf != nullptr ? 1 : 0));
    constexpr bool QT = EPI == EPI_QKV_ROPE;
    if (EPI == EPI_QKV_ROPE && ab.rope_g4k != nullptr) {
        if (fold == 2) hipLaunchKernelGGL((f5_gemm_rs128_kernel<EPI, QT, CAN_FOLD ? 2 : 0>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, ntiles, tiles_m);
        else if (fold == 1) hipLaunchKernelGGL((f5_gemm_rs128_kernel<EPI, QT, CAN_FOLD ? 1 : 0>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, ntiles, tiles_m);
        else hipLaunchKernelGGL((f5_gemm_rs128_kernel<EPI, QT, 0>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, ntiles, tiles_m);
    } else {
        if (fold == 2) hipLaunchKernelGGL((f5_gemm_rs128_kernel<EPI, false, CAN_FOLD ? 2 : 0>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, ntiles, tiles_m);
        else if (fold == 1) hipLaunchKernelGGL((f5_gemm_rs128_kernel<EPI, false, CAN_FOLD ? 1 : 0>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, ntiles, tiles_m);
        else hipLaunchKernelGGL((f5_gemm_rs128_kernel<EPI, false, 0>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, ntiles, tiles_m);
    }
    F5_LAUNCH_CHECK();
    return 0;
}

// Preconditions (checked by f5_launch_gemm): N % 256 == 0, K % 64 == 0, operands below 2 GiB (32-bit byte offsets)
int f5_launch_gemm_rs128(const F5GemmArgs& a, int epi, hipStream_t stream) {
    F5_REQUIRE(a.N % 256 == 0 && a.M >= 1 && a.K % BK == 0, "gemm_rs128: needs N %% 256 == 0, K %% 64 == 0");
    F5_REQUIRE(epi != EPI_QKV_ROPE || (a.seq_len > 0 && a.M % a.seq_len == 0), "gemm_rs128(qkv): M must be a multiple of seq_len");
    switch (epi) {
        case EPI_F32: return launch_rs128<EPI_F32>(a, stream);
        case EPI_BF16: return launch_rs128<EPI_BF16>(a, stream);
        case EPI_GELU_TANH: return launch_rs128<EPI_GELU_TANH>(a, stream);
        case EPI_RESID_GATE: return launch_rs128<EPI_RESID_GATE>(a, stream);
        case EPI_QKV_ROPE: return launch_rs128<EPI_QKV_ROPE>(a, stream);
        default: f5_set_error("gemm_rs128: unsupported epilogue %d", epi); return 2;
    }
}
}  // namespace F5_NS
