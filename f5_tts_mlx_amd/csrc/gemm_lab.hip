// Lab build only (-DF5_LAB=1): GEMM kernels that were measured and superseded or rejected, kept for A/B runs --
//   * the lock-step 256x256 kernel of rounds 1-2 (all eight waves read, then all eight multiply), with its stream-K / hybrid
//     schedules, x-tile prefetch and atomic-residual experiments (f5_debug_set_gemm_big_kernel 4, _streamk, gemm flags 8 / 512+)
//   * the 128x256 two-workgroups-per-CU kernel of round 1 (f5_debug_set_gemm_big_kernel 3)
// The product library does not contain them (csrc/build.sh: F5_LAB=1 bash build.sh).
#include "gemm.hpp"
#include "gemm_dev.hpp"
#include "gemm_lab_dev.hpp"

namespace F5_NS {
extern int f5_gemm_debug_flags;

// =================================================================================================
// v2: 256x256x64 block tile, 512 threads = 8 waves (2 x 4), wave tile 128x64 = 4x2 accumulators of
// v_mfma_f32_32x32x16_bf16 (128 acc registers).  Operands go HBM -> LDS directly with
// global_load_lds (16 B per lane, no VGPR staging); the LDS image of each 128-row half tile is
// lane-linear, so the XOR swizzle is applied to the per-lane SOURCE address and again on the read
// (CDNA4 guide rule 21).  LDS = 2 K-tiles x 4 half tiles (A0,A1,B0,B1) x 16 KB = 128 KB, one
// workgroup per CU.  A K-tile is consumed in 4 phases (one 64x32 C quadrant x K=64 = 8 MFMAs each);
// every phase also issues ONE half tile (2 global_load_lds per lane) of a future K-tile into the
// slot whose last reader finished a phase earlier:
//     tile t, phase 1: A0(t+1)   phase 2: A1(t+1)   phase 3: B0(t+2)   phase 4: B1(t+2)
// (B halves are last read in phase 2, A halves in phase 3).  Waits are COUNTED: at the end of a
// K-tile `s_waitcnt vmcnt(4)` retires everything except the two B halves issued for tile t+2, which
// stay in flight across the barrier.  Barriers: end of phases 2, 3 (WAR on the slots about to be
// overwritten) and 4 (RAW for the next tile).
// =================================================================================================

// SK = stream-K scheduling: the grid is one persistent workgroup per CU and workgroup `rid` owns the contiguous range
// [rid*W/P, (rid+1)*W/P) of the W = ntiles*T K-steps (tile-major).  A range is: the HEAD of a tile that the next range
// finishes (done FIRST: partial sums -> sk_part[rid], flag), the TAIL of a tile begun by the previous range (waits for
// that partial, adds it in fixed order head + tail, runs the epilogue), and whole tiles.  Because the first spans differ
// in length from CU to CU, the epilogues (bursts of HBM writes: x += ... is 8 B per output) of different CUs no longer
// coincide and run under other CUs' main loops; the last round is also perfectly balanced.  Results are deterministic
// (fixed summation order); they differ from the data-parallel schedule only in fp32 summation order of split tiles.
// QT (EPI_QKV_ROPE only): q / k column tiles accumulated transposed (staged_epilogue_tr_rope), V tiles straight; its own
// instantiation so that the straight q / k epilogue does not sit in the same 256-register budget
template <int EPI, bool SK, bool QT = false>
__global__ __launch_bounds__(512) void f5_gemm256_kernel(F5GemmArgs p, int tiles_n, int ntiles, float* sk_part, int* sk_flag,
                                                         int* sk_err, int sk_hybrid) {
    __shared__ __attribute__((aligned(16))) op16_t smem[2 * 4 * V2_HALF_ELEMS];   // [A0,A1,B0,B1][ring buffer][128*64]

    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int kt = p.K / BK;
    const int T = kt * p.nseg;

    // ---- this workgroup's K-step range -> spans.  Positions index the tiles of this XCD's chunk (the same contiguous
    // chunk of tile ids the one-tile-per-workgroup launch gives an XCD) in COLUMN-major order of the ragged matrix
    // [round][CU]: position idx*R + k is tile chunk + k*cpx + idx, so the CUs of an XCD sit on neighbouring tiles at any
    // time (shared A / W panels stay in the 4 MB L2) exactly like successive rounds of the plain launch.
    int rid = 0, chunk0 = 0, chunk_sk = 0, cpx = 1, Rr = 0, rem = 0, ndp = 0;
    long w0, w1;
    {
        const int q = ntiles >> 3, r = ntiles & 7;
        chunk0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        if (SK) {
            cpx = gridDim.x >> 3;                      // CUs (workgroups) per XCD; ntiles >= gridDim.x (host-checked)
            int nx = q + (xcd < r ? 1 : 0);            // tiles in this XCD's chunk
            // hybrid: whole rounds run one tile per workgroup in lockstep (CUs of an XCD stream the same K slices of shared
            // panels through the L2 together); only the last 1..2 rounds' worth of tiles is split stream-K for balance
            ndp = sk_hybrid ? (nx / cpx - 1) : 0;
            if (ndp < 0) ndp = 0;
            nx -= ndp * cpx;
            chunk_sk = chunk0 + ndp * cpx;
            Rr = nx / cpx;
            rem = nx - Rr * cpx;
            rid = xcd * cpx + idx;
            const long Wx = (long)nx * T;
            w0 = (long)idx * Wx / cpx;
            w1 = (long)(idx + 1) * Wx / cpx;
        } else {
            w0 = (long)idx * T;
            w1 = w0 + T;
        }
    }
    const int first_pos = (int)(w0 / T), t_first = (int)(w0 - (long)first_pos * T);   // tail span [t_first, T) when t_first != 0
    const int last_pos = (int)(w1 / T), t_last = (int)(w1 - (long)last_pos * T);      // head span [0, t_last) when t_last != 0
    const int nh = (SK && t_last != 0) ? 1 : 0, nt = (SK && t_first != 0) ? 1 : 0;
    const int full_begin = first_pos + nt;
    const int nspan = ndp + nh + nt + (last_pos - full_begin);

    const int frow = lane & 31;
    const int fk = lane >> 5;
    // fragment read pointers, one per 16-wide K sub-step: everything else (ring buffer, row block, quadrant) is a compile-time
    // offset that lands in the ds_read offset field, so the main loop spends no VALU instruction on LDS addressing (on this
    // chip nothing else issues on a SIMD while an MFMA is in flight, tools/probes/coissue.hip: every non-MFMA instruction of
    // the loop is paid in full)
    // (LDS layout [A0,A1,B0,B1][ring buffer][128 x 64]: the ring-buffer offset, 16 KB, is an immediate as well)
    const op16_t* pa[4];
    const op16_t* pb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        pa[ks] = smem + (wm * 2) * V2_HALF_ELEMS + swz_off(frow, ks * 2 + fk);
        pb[ks] = smem + ((2 + (wn >> 1)) * 2) * V2_HALF_ELEMS + swz_off((wn & 1) * 64 + frow, ks * 2 + fk);
    }
    for (int sp = 0; sp < nspan; ++sp) {
    int kind, pos, t0, t1;                            // kind: 0 whole tile, 1 head (publish partial), 2 tail (consume partial)
    const int ss = sp - ndp;
    if (ss < 0) {
        kind = 0; pos = 0; t0 = 0; t1 = T;            // lockstep round sp: tile chunk0 + sp*cpx + idx
    } else if (ss < nh) {
        kind = 1; pos = last_pos; t0 = 0; t1 = t_last;
    } else if (ss < nh + nt) {
        kind = 2; pos = first_pos; t0 = t_first; t1 = T;
    } else {
        kind = 0; pos = full_begin + (ss - nh - nt); t0 = 0; t1 = T;
    }
    int tile = chunk0 + pos;
    if (SK && ss < 0) {
        tile = chunk0 + sp * cpx + idx;
    } else if (SK) {                                         // column-major position -> (column c, round k) of the ragged [round][CU] matrix
        int c, k;
        if (pos < rem * (Rr + 1)) {
            c = pos / (Rr + 1);
            k = pos - c * (Rr + 1);
        } else {
            const int p2 = pos - rem * (Rr + 1);
            c = p2 / Rr;
            k = p2 - c * Rr;
            c += rem;
        }
        tile = chunk_sk + k * cpx + c;
    }
    if (sp > 0) __syncthreads();                      // the previous span's epilogue staging is done with the LDS
    int ln = lane;                                    // opaque per span: keeps the epilogue / partial-tile address math from
    if (SK) asm volatile("" : "+v"(ln));              // being hoisted out of the span loop (hundreds of spilled VGPRs)
    // tile -> (tm, tn).  n fastest, or BAND-major when the launcher set p.nband: the column tiles are cut into bands of nband,
    // a band is walked row by row.  An XCD's contiguous chunk of tiles then stays inside one band: its W panels
    // (nband x 512 KB at K = 1024) stay resident in the XCD's 4 MB L2 while the A panels stream through once, instead of all
    // tiles_n W panels being re-fetched for every round of 32 tiles (QKV at M = 59 968: FETCH_SIZE 1.50 GB per launch, 3x the
    // operand bytes, with n-fastest numbering).
    int tm, tn;
    if (p.nband > 0) {
        const int per_band = (ntiles / tiles_n) * p.nband;          // tiles_m * nband
        const int band = tile / per_band, r_ = tile - band * per_band;
        tm = r_ / p.nband;
        tn = band * p.nband + (r_ - tm * p.nband);
    } else {
        tm = tile / tiles_n;
        tn = tile - tm * tiles_n;
    }
    const int m0 = tm * 256, n0 = tn * 256;

    // ---- staging addresses: 2 chunks per thread per half tile -----------------------------------
    // linear chunk q_ = j*512 + tid of the [128][8] half-tile image; row = q_>>3, slot = q_&7,
    // source chunk = slot ^ ((row>>1)&7)
    uint32_t srcA[2][2], srcB[2][2];   // [half][j] BYTE offsets (without k0): 32-bit, added to a uniform pointer (saddr form)
    int ldsoff[2];                   // [j] element offset of this WAVE's 1 KB destination inside a half tile
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

    // issue one half tile (A half h / B half h) into ring buffer `par` (K tiles alternate buffers, the first tile of a span uses
    // buffer 0, so the parity is a compile-time constant in the 2x unrolled loop); the operand
    // (bf16x3 segment) pointer and the K offset of the tile are running values, not recomputed (tt / kt is ~20 SALU instructions
    // and sat in front of every one of the four issue points of a K step)
#define V2_ISSUE_A(par_, h_, Ap_, k0_)                                                              \
    {                                                                                               \
        op16_t* dst_ = smem + ((h_) * 2 + (par_)) * V2_HALF_ELEMS;                                  \
        const char* src_ = reinterpret_cast<const char*>(Ap_);                                      \
        const uint32_t kb_ = (uint32_t)(k0_) * 2u;                                                  \
        glds16(reinterpret_cast<const op16_t*>(src_ + (srcA[(h_)][0] + kb_)), dst_ + ldsoff[0]);    \
        glds16(reinterpret_cast<const op16_t*>(src_ + (srcA[(h_)][1] + kb_)), dst_ + ldsoff[1]);    \
    }
#define V2_ISSUE_B(par_, h_, Wp_, k0_)                                                              \
    {                                                                                               \
        op16_t* dst_ = smem + ((2 + (h_)) * 2 + (par_)) * V2_HALF_ELEMS;                            \
        const char* src_ = reinterpret_cast<const char*>(Wp_);                                      \
        const uint32_t kb_ = (uint32_t)(k0_) * 2u;                                                  \
        glds16(reinterpret_cast<const op16_t*>(src_ + (srcB[(h_)][0] + kb_)), dst_ + ldsoff[0]);    \
        glds16(reinterpret_cast<const op16_t*>(src_ + (srcB[(h_)][1] + kb_)), dst_ + ldsoff[1]);    \
    }
    // (segment, K offset) of K-tile tt: segment 0 = A.hi W.hi, 1 = A.lo W.hi, 2 = A.hi W.lo
#define V2_SEGK(tt_, seg_, k0_)              \
    const int seg_ = (tt_) / kt;             \
    const int k0_ = ((tt_) - seg_ * kt) * BK;
#define V2_BARRIER()                                   \
    {                                                  \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_s_barrier();                  \
        asm volatile("" ::: "memory");                 \
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // ---- experiment, MEASURED SLOWER, off by default (gemm flag 512/1024/2048 = prefetch 1/4, 1/2 or all of the tile):
    // residual-update launches touch the 128-byte lines of their x tile BEFORE the main loop (one dword per line, value
    // unused), hoping that the read half of the epilogue's read-modify-write is then served by the L2 / Infinity Cache and the
    // HBM reads happen while the matrix cores work.  Out-proj at M = 59 968: 186 us without, 188 / 193 / 205 us with 1/4,
    // 1/2, all lines; sample() at batch 32 1 296-1 301 vs 1 313 ms (profiles/r02/resid_preload_prefetch_ab.txt): a round's
    // x tiles (8 MB per XCD) do not survive the operand stream in the 4 MB L2, and the early reads delay the first operand tiles.
    // The loads are older than every operand load: the counted vmcnt waits of the main loop cover them.
    float xpf[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (EPI == EPI_RESID_GATE && (p.debug_flags & (512 | 1024 | 2048)) && kind != 1) {
        const int npf = (p.debug_flags & 2048) ? 4 : ((p.debug_flags & 1024) ? 2 : 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < npf) {
                const int li = j * 512 + tid;                     // line of the 256 x 256 fp32 tile: 8 lines per row
                int row = m0 + (li >> 3);
                if (row > p.M - 1) row = p.M - 1;
                const float* ptr = p.out_f32 + (size_t)row * p.ldo + n0 + (li & 7) * 32;
                asm volatile("global_load_dword %0, %1, off" : "=v"(xpf[j]) : "v"(ptr) : "memory");
            }
    }

    // ---- prologue: tile 0 (4 halves) + B halves of tile 1 -----------------------------------------
    int a_seg, a_k0, b_seg, b_k0;                     // running state: tile tt+1 (A halves) and tile tt+2 (B halves)
    {
        V2_SEGK(t0, s0_, k00_);
        const op16_t* Ap0 = s0_ == 1 ? p.A[1] : p.A[0];
        const op16_t* Wp0 = s0_ == 2 ? p.W[1] : p.W[0];
        V2_ISSUE_A(0, 0, Ap0, k00_);
        V2_ISSUE_A(0, 1, Ap0, k00_);
        V2_ISSUE_B(0, 0, Wp0, k00_);
        V2_ISSUE_B(0, 1, Wp0, k00_);
        a_seg = s0_;
        a_k0 = k00_ + BK;
        if (a_k0 == p.K) {
            a_k0 = 0;
            ++a_seg;
        }
        b_seg = a_seg;
        b_k0 = a_k0;
    }
    if (t0 + 1 < t1) {
        const op16_t* Wp1 = b_seg == 2 ? p.W[1] : p.W[0];
        V2_ISSUE_B(1, 0, Wp1, b_k0);
        V2_ISSUE_B(1, 1, Wp1, b_k0);
        b_k0 += BK;
        if (b_k0 == p.K) {
            b_k0 = 0;
            ++b_seg;
        }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    V2_BARRIER();

    // one K tile out of ring buffer PAR (compile-time): 4 phases of 8 MFMAs, each issuing one half tile of a later K tile
    op16x8 af[2][4], bfr[2][4];
#define V2_FRAG_A(PAR, ks, rowoff) (*reinterpret_cast<const op16x8*>(pa[ks] + (PAR) * V2_HALF_ELEMS + (rowoff) * BK))
#define V2_FRAG_B(PAR, ks, rowoff) (*reinterpret_cast<const op16x8*>(pb[ks] + (PAR) * V2_HALF_ELEMS + (rowoff) * BK))
#define V2_MM(TR_, A_, B_, C_) ((TR_) ? F5_MFMA32(B_, A_, C_, 0, 0, 0) : F5_MFMA32(A_, B_, C_, 0, 0, 0))
#define V2_KSTEP(PAR, tt, TR_)                                                                                             \
    {                                                                                                                   \
        /* phase 1: A(mq=0), B(nq=0); quadrant (0,0); issue A0(t+1) */                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                              \
            _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) af[mb][ks] = V2_FRAG_A(PAR, ks, mb * 32);                  \
            bfr[0][ks] = V2_FRAG_B(PAR, ks, 0);                                                                         \
        }                                                                                                               \
        if ((tt) + 1 < t1) V2_ISSUE_A(1 - PAR, 0, Apn, a_k0);                                            \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                \
            _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                            \
                acc[mb][0] = V2_MM(TR_, af[mb][ks], bfr[0][ks], acc[mb][0]);                                            \
        /* phase 2: B(nq=1); quadrant (0,1); issue A1(t+1) */                                                           \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) bfr[1][ks] = V2_FRAG_B(PAR, ks, 32);                           \
        if ((tt) + 1 < t1) V2_ISSUE_A(1 - PAR, 1, Apn, a_k0);                                            \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                \
            _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                            \
                acc[mb][1] = V2_MM(TR_, af[mb][ks], bfr[1][ks], acc[mb][1]);                                            \
        V2_BARRIER(); /* every wave has finished reading the B halves of this tile */                                   \
        /* phase 3: A(mq=1); quadrant (1,1); issue B0(t+2) */                                                           \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                \
            _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) af[mb][ks] = V2_FRAG_A(PAR, ks, 64 + mb * 32);             \
        if ((tt) + 2 < t1) V2_ISSUE_B(PAR, 0, Wpn, b_k0);                                                \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                \
            _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                            \
                acc[2 + mb][1] = V2_MM(TR_, af[mb][ks], bfr[1][ks], acc[2 + mb][1]);                                    \
        V2_BARRIER(); /* every wave has finished reading the A halves of this tile */                                   \
        /* phase 4: quadrant (1,0) from registers; issue B1(t+2) */                                                     \
        if ((tt) + 2 < t1) V2_ISSUE_B(PAR, 1, Wpn, b_k0);                                                \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                \
            _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                            \
                acc[2 + mb][0] = V2_MM(TR_, af[mb][ks], bfr[0][ks], acc[2 + mb][0]);                                    \
        /* next tile's operands: everything but the two B halves just issued for tile t+2 must have landed */           \
        if ((tt) + 2 < t1) {                                                                             \
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                            \
        } else {                                                                                                        \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                            \
        }                                                                                                               \
        V2_BARRIER();                                                                                                   \
        a_k0 += BK;                                                                                                     \
        if (a_k0 == p.K) {                                                                                              \
            a_k0 = 0;                                                                                                   \
            ++a_seg;                                                                                                    \
            Apn = a_seg == 1 ? p.A[1] : p.A[0];                                                                         \
        }                                                                                                               \
        b_k0 += BK;                                                                                                     \
        if (b_k0 == p.K) {                                                                                              \
            b_k0 = 0;                                                                                                   \
            ++b_seg;                                                                                                    \
            Wpn = b_seg == 2 ? p.W[1] : p.W[0];                                                                         \
        }                                                                                                               \
    }
    const op16_t* Apn = a_seg == 1 ? p.A[1] : p.A[0];   // operand (bf16x3 segment) pointers of the tiles being staged
    const op16_t* Wpn = b_seg == 2 ? p.W[1] : p.W[0];
    // 16-bit row-major outputs: the tile is accumulated TRANSPOSED (operands swapped in every MFMA) for staged_epilogue_tr; the
    // straight order stays selectable for A/B (gemm flag 16384).  Both loop copies end in their own epilogue: no join with 128
    // live accumulator registers.
    constexpr bool TR_EPI = !SK && (EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_GELU_ERF_BF16);
    if ((TR_EPI && (p.debug_flags & 16384) == 0) || (QT && n0 < 2 * p.dmodel)) {       // workgroup-uniform
        for (int tt = t0; tt < t1; tt += 2) {
            V2_KSTEP(0, tt, true);
            if (tt + 1 < t1) V2_KSTEP(1, tt + 1, true);
        }
        if ((p.debug_flags & 1) == 0) {
            if (QT) staged_epilogue_tr_rope<4, 2>(p, acc, smem + wave * 8192, m0 + wm * 128, n0 + wn * 64, ln);
            else staged_epilogue_tr<EPI, 4, 2>(p, acc, smem + wave * 8192, m0 + wm * 128, n0 + wn * 64, ln);
        }
        continue;
    }
    for (int tt = t0; tt < t1; tt += 2) {
        V2_KSTEP(0, tt, false);
        if (tt + 1 < t1) V2_KSTEP(1, tt + 1, false);
    }
#undef V2_KSTEP
#undef V2_MM
#undef V2_FRAG_A
#undef V2_FRAG_B

    // Partial tiles cross XCDs, whose L2s are not coherent.  No agent-scope fences here: a release fence writes back and an
    // acquire fence invalidates the WHOLE L2 of the XCD (measured: the operand panels of all 32 CUs get refetched and the
    // kernel runs 1.65x slower).  Instead the payload and the flag use relaxed agent-scope atomics, i.e. plain sc1
    // (write-through / L2-bypassing) stores and loads, ordered by s_waitcnt vmcnt(0) + the workgroup barrier.
    if (SK && kind == 1) {
        float* dst = sk_part + (size_t)rid * 65536 + (size_t)wave * 8192 + ln;   // [wave][acc block][reg][lane] fp32
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    __hip_atomic_store(dst + ((i * 2 + j) * 16 + e) * 64, acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&sk_flag[rid], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
    }
    if (SK && kind == 2) {
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(&sk_flag[rid - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1 << 20)) {             // never hang the GPU: flag the error, results will be wrong
                    atomicExch(sk_err, 1);
                    break;
                }
            }
            __hip_atomic_store(&sk_flag[rid - 1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
        }
        __syncthreads();
        const float* src = sk_part + (size_t)(rid - 1) * 65536 + (size_t)wave * 8192 + ln;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float part[16];                        // 16 loads in flight at a time (all 128 at once would spill)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    part[e] = __hip_atomic_load(src + ((i * 2 + j) * 16 + e) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = part[e] + acc[i][j][e];
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    if (EPI == EPI_RESID_GATE) asm volatile("" ::"v"(xpf[0]), "v"(xpf[1]), "v"(xpf[2]), "v"(xpf[3]));   // prefetch registers live until here
    if (p.debug_flags & 1) {   // timing experiment: main loop only
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
        continue;
    }
    if (QT) {                                                               // (the q / k tiles finished above)
        staged_epilogue_bf16<EPI, 4, 2, true>(p, acc, smem + wave * 8192, m0 + wm * 128, n0 + wn * 64, ln);
    } else if (EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_GELU_ERF_BF16 || EPI == EPI_QKV_ROPE) {
        staged_epilogue_bf16<EPI, 4, 2>(p, acc, smem + wave * 8192, m0 + wm * 128, n0 + wn * 64, ln);
    } else if (EPI == EPI_RESID_GATE) {
        if (p.debug_flags & 8) atomic_epilogue_resid<4, 2>(p, acc, m0 + wm * 128, n0 + wn * 64, ln);     // experiment, see atomic_epilogue_resid
        else staged_epilogue_resid<4, 2>(p, acc, reinterpret_cast<float*>(smem + wave * 8192), m0 + wm * 128, n0 + wn * 64, ln);
    } else {
        gemm_epilogue<EPI, 4, 2>(p, acc, m0, n0, wm, wn, ln);
    }
    }   // spans
}
// stream-K scratch (process-wide, one device): partial tiles [P][256*256] fp32, flags, error word.  Allocated outside of
// any stream capture by f5_gemm_streamk_init(), which the debug hook calls when the schedule is switched on.
static float* g_sk_part = nullptr;
static int* g_sk_flag = nullptr;
static int g_sk_P = 0;
int f5_gemm_streamk = 0;          // large shapes: 0 = one tile per workgroup, 1 = stream-K over all K-steps, 2 = hybrid (lockstep
                                  // rounds + stream-K tail)
int f5_gemm_streamk_init() {
    if (g_sk_part) return 0;
    int dev = 0, cus = 0;
    F5_HIP_CHECK(hipGetDevice(&dev));
    F5_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus -= cus % 8;
    F5_REQUIRE(cus >= 8, "stream-K: unexpected CU count %d", cus);
    float* part = nullptr;
    int* flag = nullptr;
    F5_HIP_CHECK(hipMalloc(&part, (size_t)cus * 65536 * sizeof(float)));
    F5_HIP_CHECK(hipMalloc(&flag, (size_t)(cus + 64) * sizeof(int)));
    F5_HIP_CHECK(hipMemset(flag, 0, (size_t)(cus + 64) * sizeof(int)));
    g_sk_part = part;
    g_sk_flag = flag;
    g_sk_P = cus;
    return 0;
}
int f5_gemm_streamk_error() {      // 1 if a consumer ever timed out waiting for a partial tile (results invalid)
    if (!g_sk_flag) return 0;
    int v = 0;
    if (hipMemcpy(&v, g_sk_flag + g_sk_P, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}
extern int f5_gemm_nband;   // gemm256.hip
template <int EPI>
static int launch_v2(const F5GemmArgs& a, hipStream_t stream) {
    F5_REQUIRE((size_t)(a.a_row_mod > 0 ? a.a_row_mod : a.M) * a.lda < (1ull << 31) && (size_t)(a.N + 256) * a.ldw < (1ull << 31),
               "gemm: operands of the 256x256 kernel must stay below 4 GiB (32-bit byte offsets)");
    const int tiles_m = f5_cdiv(a.M, 256), tiles_n = a.N / 256;
    const int ntiles = tiles_m * tiles_n;
    F5GemmArgs ab = a;
    ab.nband = (f5_gemm_nband > 0 && tiles_n > f5_gemm_nband && tiles_n % f5_gemm_nband == 0 && !f5_gemm_streamk) ? f5_gemm_nband : 0;
    // staged_epilogue_tr reads the bias as 16-byte quads: an unaligned bias vector takes the straight-order path
    if (a.bias != nullptr && (reinterpret_cast<uintptr_t>(a.bias) & 15) != 0) ab.debug_flags |= 16384;
    if (f5_gemm_streamk && g_sk_part && ntiles >= g_sk_P) {
        hipLaunchKernelGGL((f5_gemm256_kernel<EPI, true>), dim3(g_sk_P), dim3(512), 0, stream, a, tiles_n, ntiles, g_sk_part,
                           g_sk_flag, g_sk_flag + g_sk_P, f5_gemm_streamk == 2 ? 1 : 0);
    } else if (EPI == EPI_QKV_ROPE && ab.rope_g4k != nullptr) {
        hipLaunchKernelGGL((f5_gemm256_kernel<EPI, false, EPI == EPI_QKV_ROPE>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, ntiles,
                           (float*)nullptr, (int*)nullptr, (int*)nullptr, 0);
    } else {
        hipLaunchKernelGGL((f5_gemm256_kernel<EPI, false>), dim3(ntiles), dim3(512), 0, stream, ab, tiles_n, ntiles,
                           (float*)nullptr, (int*)nullptr, (int*)nullptr, 0);
    }
    F5_LAUNCH_CHECK();
    return 0;
}


// =================================================================================================
// v3: 128x256x32 block tile, 256 threads = 4 waves (2 x 2), wave tile 64x128 (2x4 accumulators), global_load_lds
// ring of 3 K-tiles of 24 KB => 72 KB of LDS and <= 256 registers, i.e. TWO workgroups per CU.  Rationale (measured
// with the skip-epilogue ablation, tools/gemm_ablate.py): at K = 1024 the epilogue is 26-42 % of a 256x256 tile's time
// and is bound by the CU's store path / HBM (x += ... moves 8 B per output), during which the matrix pipes idle.  With
// two resident workgroups that are half a tile out of phase (the second wave of workgroups starts with a one-off
// sleep), one workgroup's epilogue runs under the other's main loop.  64-byte LDS rows: swizzle chunk ^= (row>>2)&3.
// =================================================================================================
#define V3_BK 32
__device__ __forceinline__ int swz32(int row, int chunk) { return row * V3_BK + ((chunk ^ ((row >> 2) & 3)) << 3); }

// PRIO (issue priority between the two co-resident workgroups of a CU): 0 = s_setprio 1 around the MFMA clusters (round 1:
// measured no overlap of one workgroup's epilogue with the other's main loop), 1 = no priority changes, 2 = the EPILOGUE runs at
// priority 3 and the main loop at 0, so the epilogue's VALU / LDS / store instructions issue in the gaps of the partner's MFMAs
template <int EPI, int PRIO>
__global__ __launch_bounds__(256, 2) void f5_gemm_v3_kernel(F5GemmArgs p, int tiles_n, int ntiles, int stagger_cycles) {
    constexpr int BMt = 128, BNt = 256, NST = 3;
    constexpr int NA = 2, NW = 4, G = NA + NW;
    constexpr int STAGE = (BMt + BNt) * V3_BK;          // 12288 elements = 24 KB
    __shared__ __attribute__((aligned(16))) op16_t smem[NST * STAGE];

    const int bid = blockIdx.x;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // tile numbering: n fastest (neighbouring tiles share the A panel) or, when tiles_n < 0, m fastest with
    // tiles_m = -tiles_n (neighbouring tiles share the W panel: better when the whole A operand fits in an XCD's L2)
    int tm, tn;
    if (tiles_n > 0) {
        tm = tile / tiles_n;
        tn = tile - tm * tiles_n;
    } else {
        tn = tile / (-tiles_n);
        tm = tile - tn * (-tiles_n);
    }
    const int m0 = tm * BMt, n0 = tn * BNt;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // de-phase the two co-resident workgroups: blocks 256..511 (second slot of every CU in dispatch order) start late
    if (bid >= 256 && bid < 512) {
        for (int c = 0; c < stagger_cycles; c += 64 * 100) __builtin_amdgcn_s_sleep(100);
    }

    size_t a_src[NA], w_src[NW];
    int a_dst[NA], w_dst[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q_ = i * 256 + tid;
        const int row = q_ >> 2, chunk = (q_ & 3) ^ ((row >> 2) & 3);
        int gr = m0 + row;
        if (gr > p.M - 1) gr = p.M - 1;
        if (p.a_row_mod > 0) gr = gr % p.a_row_mod;
        a_src[i] = (size_t)gr * p.lda + chunk * 8;
        a_dst[i] = (i * 256 + wave * 64) * 8;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int q_ = i * 256 + tid;
        const int row = q_ >> 2, chunk = (q_ & 3) ^ ((row >> 2) & 3);
        w_src[i] = (size_t)(n0 + row) * p.ldw + chunk * 8;
        w_dst[i] = BMt * V3_BK + (i * 256 + wave * 64) * 8;
    }
    const int kt = p.K / V3_BK;
    const int T = kt * p.nseg;
#define V3_ISSUE(tt_)                                                                                        \
    {                                                                                                        \
        const int seg_ = (tt_) / kt;                                                                         \
        const int k0_ = ((tt_) - seg_ * kt) * V3_BK;                                                         \
        op16_t* st_ = smem + ((tt_) % NST) * STAGE;                                                          \
        const op16_t* Ap_ = (seg_ == 1) ? p.A[1] : p.A[0];                                                   \
        const op16_t* Wp_ = (seg_ == 2) ? p.W[1] : p.W[0];                                                   \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) glds16(Ap_ + a_src[i] + k0_, st_ + a_dst[i]);         \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) glds16(Wp_ + w_src[i] + k0_, st_ + w_dst[i]);         \
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    V3_ISSUE(0);
    if (T > 1) V3_ISSUE(1);

    const int frow = lane & 31;
    const int fk = lane >> 5;
    for (int tt = 0; tt < T; ++tt) {
        if (tt + 1 < T) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");     // tile tt landed, tile tt+1 may be in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tt + 2 < T) V3_ISSUE(tt + 2);                                 // slot of tile tt-1: every wave is past it

        const op16_t* sA = smem + (tt % NST) * STAGE;
        const op16_t* sB = sA + BMt * V3_BK;
        if (PRIO == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            op16x8 af[2], bfr[4];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) af[mb] = *reinterpret_cast<const op16x8*>(&sA[swz32(wm * 64 + mb * 32 + frow, ks * 2 + fk)]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) bfr[nb] = *reinterpret_cast<const op16x8*>(&sB[swz32(wn * 128 + nb * 32 + frow, ks * 2 + fk)]);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc[mb][nb] = F5_MFMA32(af[mb], bfr[nb], acc[mb][nb], 0, 0, 0);
        }
        if (PRIO == 0) __builtin_amdgcn_s_setprio(0);
    }
    if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
    if (p.debug_flags & 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    // every wave must be done reading the ring before it is reused as epilogue staging space
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    op16_t* reg = smem + wave * 9216;     // 18 KB per wave
    if (EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_GELU_ERF_BF16 || EPI == EPI_QKV_ROPE) {
        staged_epilogue_bf16<EPI, 2, 4>(p, acc, reg, m0 + wm * 64, n0 + wn * 128, lane);
    } else if (EPI == EPI_RESID_GATE) {
        staged_epilogue_resid<2, 4>(p, acc, reinterpret_cast<float*>(reg), m0 + wm * 64, n0 + wn * 128, lane);
    } else {
        gemm_epilogue<EPI, 2, 4>(p, acc, m0, n0, wm, wn, lane);
    }
}

int f5_gemm_v3_prio = 0;       // 128x256 kernel: 0 = priority to the MFMA clusters, 1 = none, 2 = priority to the epilogue
int f5_gemm_v3_stagger = -1;   // cycles of initial delay for workgroups 256..511 (-1: auto = half a tile's main loop)
template <int EPI>
static int launch_v3(const F5GemmArgs& a, hipStream_t stream) {
    const int tiles_m = f5_cdiv(a.M, 128), tiles_n = a.N / 256;
    const int ntiles = tiles_m * tiles_n;
    int stagger = f5_gemm_v3_stagger;
    if (stagger < 0) stagger = (a.K / V3_BK) * a.nseg * 16 * 32;   // ~ half of (K tiles x 16 MFMAs x 32 cycles x 2 workgroups)
    if (f5_gemm_v3_prio == 1) hipLaunchKernelGGL((f5_gemm_v3_kernel<EPI, 1>), dim3(ntiles), dim3(256), 0, stream, a, tiles_n, ntiles, stagger);
    else if (f5_gemm_v3_prio == 2) hipLaunchKernelGGL((f5_gemm_v3_kernel<EPI, 2>), dim3(ntiles), dim3(256), 0, stream, a, tiles_n, ntiles, stagger);
    else hipLaunchKernelGGL((f5_gemm_v3_kernel<EPI, 0>), dim3(ntiles), dim3(256), 0, stream, a, tiles_n, ntiles, stagger);
    F5_LAUNCH_CHECK();
    return 0;
}


int f5_launch_gemm_lab_v2(const F5GemmArgs& a, int epi, hipStream_t stream) {
    switch (epi) {
        case EPI_F32: return launch_v2<EPI_F32>(a, stream);
        case EPI_BF16: return launch_v2<EPI_BF16>(a, stream);
        case EPI_GELU_TANH: return launch_v2<EPI_GELU_TANH>(a, stream);
        case EPI_GELU_ERF: return launch_v2<EPI_GELU_ERF>(a, stream);
        case EPI_RESID_GATE: return launch_v2<EPI_RESID_GATE>(a, stream);
        case EPI_QKV_ROPE: return launch_v2<EPI_QKV_ROPE>(a, stream);
        case EPI_ADDROWS: return launch_v2<EPI_ADDROWS>(a, stream);
        case EPI_RESID_KEEP: return launch_v2<EPI_RESID_KEEP>(a, stream);
        case EPI_GELU_ERF_BF16: return launch_v2<EPI_GELU_ERF_BF16>(a, stream);
        default: f5_set_error("gemm lab v2: unknown epilogue %d", epi); return 2;
    }
}
int f5_launch_gemm_lab_v3(const F5GemmArgs& a, int epi, hipStream_t stream) {
    switch (epi) {
        case EPI_F32: return launch_v3<EPI_F32>(a, stream);
        case EPI_BF16: return launch_v3<EPI_BF16>(a, stream);
        case EPI_GELU_TANH: return launch_v3<EPI_GELU_TANH>(a, stream);
        case EPI_GELU_ERF: return launch_v3<EPI_GELU_ERF>(a, stream);
        case EPI_RESID_GATE: return launch_v3<EPI_RESID_GATE>(a, stream);
        case EPI_QKV_ROPE: return launch_v3<EPI_QKV_ROPE>(a, stream);
        case EPI_ADDROWS: return launch_v3<EPI_ADDROWS>(a, stream);
        case EPI_RESID_KEEP: return launch_v3<EPI_RESID_KEEP>(a, stream);
        case EPI_GELU_ERF_BF16: return launch_v3<EPI_GELU_ERF_BF16>(a, stream);
        default: f5_set_error("gemm lab v3: unknown epilogue %d", epi); return 2;
    }
}
}  // namespace F5_NS
