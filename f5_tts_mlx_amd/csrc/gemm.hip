// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T, fp32 accumulate, fused epilogues.
//
// Structure (v1): 128x128x64 block tile, 256 threads = 4 waves in a 2x2 grid, each wave owns a
// 64x64 sub-tile as 2x2 v_mfma_f32_32x32x16_bf16 accumulators (64 acc VGPRs).  A and W tiles are
// staged global -> VGPR -> LDS (double buffered, one barrier per K tile); the LDS image of a
// [128][64] bf16 tile is XOR-swizzled at 16-byte granularity, chunk' = chunk ^ ((row >> 1) & 7),
// which makes the ds_read_b128 fragment reads conflict free (MI355X guide, LDS section).
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous range of tiles.
//
// The "bf16x3" precision mode (nseg = 3) runs the K loop three times over (A_hi,W_hi), (A_lo,W_hi),
// (A_hi,W_lo): products are then exact to ~2^-17 relative, i.e. fp32-class results from bf16 MFMA
// at 3x the matrix work.
#include "gemm.hpp"
#include "gemm_dev.hpp"
#include "lnrow.hpp"

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

// ---- EPI_RESID_GATE of the small-tile ring kernel with everything the update needs ALREADY IN REGISTERS ---------------------
// At M = 2*937 the launch is one round of workgroups and its run time is one workgroup's dependency chain; the plain epilogue
// appends "load x, bias, gate, keep -> wait a memory round trip -> add -> store" to it.  The values do not depend on the
// product, so the waves that will run the epilogue request them before the first operand tile (ResidPre) and the round trip
// overlaps the whole K loop; the epilogue is then arithmetic + stores.  Same arithmetic as gemm_epilogue: x + gate * (v * keep).
// Measured at M = 1874 (tools/r2c_ab.py, gemm flag 256 = loads in the epilogue): out-proj 11.6 vs 12.6 us, FF2 16.6 vs 17.5 us,
// sample() at batch 1 77.5 vs 80.5 ms (profiles/r02/resid_preload_prefetch_ab.txt).
template <int MB, int NB>
struct ResidPre {
    float x[MB][16][NB];
    float bias[NB], gate[NB];
    uint32_t keep[MB][4];       // keep bytes of rows (rg*8 + hi*4 + 0..3) of row block mb
};
template <int MB, int NB>
__device__ __forceinline__ void resid_preload(const F5GemmArgs& p, ResidPre<MB, NB>& q, int row0, int colbase, int lane) {
    const int hi = lane >> 5, lcol = lane & 31;
    const bool keep_words = p.rowkeep != nullptr && (reinterpret_cast<uintptr_t>(p.rowkeep) & 3) == 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int c = colbase + nb * 32 + lcol;
        const bool ok = c < p.N;
        q.bias[nb] = (p.bias != nullptr && ok) ? p.bias[c] : 0.0f;
        q.gate[nb] = ok ? p.gate[c] : 0.0f;
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int rowb = row0 + mb * 32 + rg * 8 + hi * 4;
            uint32_t kw = 0x01010101u;
            if (p.rowkeep != nullptr) {
                if (keep_words && rowb + 3 < p.M) {
                    kw = *reinterpret_cast<const uint32_t*>(p.rowkeep + rowb);
                } else {
                    kw = 0;
#pragma unroll
                    for (int ri = 0; ri < 4; ++ri)
                        if (rowb + ri < p.M) kw |= (uint32_t)p.rowkeep[rowb + ri] << (8 * ri);
                }
            }
            q.keep[mb][rg] = kw;
#pragma unroll
            for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int row = rowb + ri, c = colbase + nb * 32 + lcol;
                    q.x[mb][rg * 4 + ri][nb] = (row < p.M && c < p.N) ? p.out_f32[(size_t)row * p.ldo + c] : 0.0f;
                }
        }
}
template <int MB, int NB>
__device__ __forceinline__ void resid_epilogue_preloaded(const F5GemmArgs& p, f32x16 (&acc)[MB][NB], const ResidPre<MB, NB>& q, int row0,
                                                         int colbase, int lane) {
    const int hi = lane >> 5, lcol = lane & 31;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const int row = row0 + mb * 32 + rg * 8 + hi * 4 + ri;
                const bool kp = ((q.keep[mb][rg] >> (8 * ri)) & 0xffu) != 0;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int c = colbase + nb * 32 + lcol;
                    float v = acc[mb][nb][rg * 4 + ri] + q.bias[nb];
                    if (!kp) v = 0.0f;
                    if (row < p.M && c < p.N) p.out_f32[(size_t)row * p.ldo + c] = q.x[mb][rg * 4 + ri][nb] + q.gate[nb] * v;
                }
            }
}

// ---- LN-modulate fused behind the residual update (EPI_RESID_GATE of the small-tile kernels, batch-1-sized problems) ------
// At M = 2*937 every launch is one round of workgroups and costs ~2 us of launch / drain on top of its work, and the
// stand-alone LN-modulate kernels are 2 of the 7 launches of a DiT block (5.2 us each).  Instead, every workgroup of the
// residual GEMM publishes its part of x with agent-scope stores, drains them (vmcnt(0)), and bumps the counter of its row
// block; the workgroup that arrives LAST (no one waits, so no deadlock and no ordering assumption) re-reads the rows with
// agent-scope (sc1, L2-bypassing) loads -- the other column tiles were written from other XCDs -- and runs the same
// per-row code as ln_modulate_kernel (lnrow.hpp: identical bits).  It re-arms the counter for the next launch.
__device__ __forceinline__ f32x4 f5_ld_agent_f32x4(const float* ptr) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
template <int NV>
__device__ __forceinline__ void resid_ln_rows(const F5GemmArgs& p, int row0, int row1, int wave, int nwaves, int lane) {
    // RB rows per wave are in flight at a time: the agent-scope loads come from the Infinity Cache / HBM (~1-2 us), a row at a
    // time the tail of a 64-row block took ~10 us (measured: 88 vs 75 ms per sample); RB = 4 keeps 16 * NV VGPRs live
    constexpr int RB = 4;
    for (int rbase = row0 + wave; rbase < row1; rbase += RB * nwaves) {
        f32x4 v[RB][NV];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int row = rbase + j * nwaves < row1 ? rbase + j * nwaves : row1 - 1;     // clamped rows are loaded, not used
            const float* xr = p.out_f32 + (size_t)row * p.ldo;
#pragma unroll
            for (int i = 0; i < NV; ++i) v[j][i] = f5_ld_agent_f32x4(xr + i * 256 + lane * 4);
        }
        // the loads above are invisible to the compiler's own waitcnt bookkeeping: drain them by hand; every value passes
        // through an (empty) asm statement behind the wait so that nothing reads it earlier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < RB; ++j)
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(v[j][i])::"memory");
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int row = rbase + j * nwaves;
            if (row < row1)
                f5_ln_modulate_row<NV>(v[j], p.ln_scale, p.ln_shift, p.ln_out[0], p.ln_out[1], (size_t)row, lane, p.ln_eps);
        }
    }
}
// called by EVERY wave of the workgroup (also the K-split groups that took no part in the epilogue) after the epilogue
__device__ __forceinline__ void resid_ln_tail(const F5GemmArgs& p, int tile_m, int bm_rows, int ntiles_n) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's x stores have reached the coherence point
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(p.ln_counter + tile_m, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == ntiles_n - 1;
        if (last) __hip_atomic_store(p.ln_counter + tile_m, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int row0 = tile_m * bm_rows;
    const int row1 = row0 + bm_rows < p.M ? row0 + bm_rows : p.M;
    switch (p.N >> 8) {
        case 1: resid_ln_rows<1>(p, row0, row1, wave, nwaves, lane); break;
        case 2: resid_ln_rows<2>(p, row0, row1, wave, nwaves, lane); break;
        case 3: resid_ln_rows<3>(p, row0, row1, wave, nwaves, lane); break;
        default: resid_ln_rows<4>(p, row0, row1, wave, nwaves, lane); break;
    }
}

// block tile = (64*MB) x (64*NB), 4 waves in a 2x2 grid, wave tile = (32*MB) x (32*NB)
// Small-tile kernels: bf16-output epilogues go through the LDS-staged 16-byte-store path when the whole block tile lies
// inside N (the direct path issues 2-byte stores, and for QKV strided 2-byte V^T stores).  Needs one barrier (the K ring
// is dead afterwards) and, for the QKV head split, wave tiles that are whole heads (32*NB % 64 == 0).
template <int EPI, int NB>
__device__ __forceinline__ constexpr bool small_tile_staged() {
    return (NB & (NB - 1)) == 0 &&
           ((EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_GELU_ERF_BF16) || (EPI == EPI_QKV_ROPE && NB % 2 == 0));
}
template <int NB>
__device__ __forceinline__ constexpr int small_tile_stage_elems() {   // bf16 elements of LDS per wave
    return 2 * 32 * (32 * NB + 8) > (32 * NB) * 40 ? 2 * 32 * (32 * NB + 8) : (32 * NB) * 40;
}

template <int EPI, int MB, int NB>
__global__ __launch_bounds__(256) void f5_gemm_kernel(F5GemmArgs p, int tiles_n, int ntiles) {
    constexpr int BMt = 64 * MB, BNt = 64 * NB;
    constexpr int NA = MB * 2, NW = NB * 2;      // 16-byte chunks staged per thread for A / W
    __shared__ __attribute__((aligned(16))) op16_t smem[2][(BMt + BNt) * BK];  // [buffer][A tile | W tile]

    // XCD-aware, bijective remap: workgroup b runs on XCD b % 8; give each XCD a contiguous range
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
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // 32-bit BYTE offsets added to a uniform operand pointer (SGPR base + VGPR offset loads)
    uint32_t a_off[NA], w_off[NW];
    int sa_off[NA], sw_off[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int qd = tid + 256 * i;
        const int srow = qd >> 3, schunk = qd & 7;
        int gr = m0 + srow;
        if (gr > p.M - 1) gr = p.M - 1;
        if (p.a_row_mod > 0) gr = gr % p.a_row_mod;
        a_off[i] = ((uint32_t)gr * (uint32_t)p.lda + schunk * 8) * 2u;
        sa_off[i] = swz_off(srow, schunk);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int qd = tid + 256 * i;
        const int srow = qd >> 3, schunk = qd & 7;
        w_off[i] = ((uint32_t)(n0 + srow) * (uint32_t)p.ldw + schunk * 8) * 2u;
        sw_off[i] = BMt * BK + swz_off(srow, schunk);
    }

    const int T = (p.K / BK) * p.nseg;

    // tiles are loaded in order: running (segment, K offset) of the next tile to load
    int ld_seg = 0, ld_k0 = 0;
    u32x4 ra[NA], rb[NW];
#define LOAD_TILE()                                                                                           \
    {                                                                                                         \
        const char* Ap_ = reinterpret_cast<const char*>((ld_seg == 1) ? p.A[1] : p.A[0]);                     \
        const char* Wp_ = reinterpret_cast<const char*>((ld_seg == 2) ? p.W[1] : p.W[0]);                     \
        const uint32_t kb_ = (uint32_t)ld_k0 * 2u;                                                            \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const u32x4*>(Ap_ + (a_off[i] + kb_)); \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) rb[i] = *reinterpret_cast<const u32x4*>(Wp_ + (w_off[i] + kb_)); \
        ld_k0 += BK;                                                                                          \
        if (ld_k0 == p.K) {                                                                                   \
            ld_k0 = 0;                                                                                        \
            ++ld_seg;                                                                                         \
        }                                                                                                     \
    }
#define STORE_TILE(buf_)                                                                                      \
    {                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) *reinterpret_cast<u32x4*>(&smem[buf_][sa_off[i]]) = ra[i]; \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) *reinterpret_cast<u32x4*>(&smem[buf_][sw_off[i]]) = rb[i]; \
    }

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    LOAD_TILE();
    STORE_TILE(0);
    __syncthreads();

    const int frow = lane & 31;
    const int fk = lane >> 5;
    // fragment read pointers per K sub-step; buffer, row block and A / W part are ds_read immediates (K loop unrolled 2x)
    const op16_t* pa[4];
    const op16_t* pb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        pa[ks] = &smem[0][0] + swz_off(wm * (32 * MB) + frow, ks * 2 + fk);
        pb[ks] = &smem[0][0] + BMt * BK + swz_off(wn * (32 * NB) + frow, ks * 2 + fk);
    }
    static_assert(2 * (BMt + BNt) * BK * 2 <= 65536, "both buffers must be addressable by ds_read immediates");
#define REG_STEP(CUR, tt_)                                                                                    \
    {                                                                                                         \
        if ((tt_) + 1 < T) LOAD_TILE();                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                    \
            op16x8 af[MB], bfr[NB];                                                                           \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                 \
                af[mb] = *reinterpret_cast<const op16x8*>(pa[ks] + (CUR) * (BMt + BNt) * BK + mb * 32 * BK);  \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                 \
                bfr[nb] = *reinterpret_cast<const op16x8*>(pb[ks] + (CUR) * (BMt + BNt) * BK + nb * 32 * BK); \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                 \
                _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                             \
                    acc[mb][nb] = F5_MFMA32(af[mb], bfr[nb], acc[mb][nb], 0, 0, 0); \
        }                                                                                                     \
        if ((tt_) + 1 < T) STORE_TILE(1 - (CUR));                                                             \
        __syncthreads();                                                                                      \
    }
    for (int tt = 0; tt < T; tt += 2) {
        REG_STEP(0, tt);
        if (tt + 1 < T) REG_STEP(1, tt + 1);
    }
#undef REG_STEP
#undef LOAD_TILE
#undef STORE_TILE

    if (small_tile_staged<EPI, NB>() && n0 + BNt <= p.N && (p.debug_flags & 2) == 0) {
        static_assert(!small_tile_staged<EPI, NB>() || 4 * small_tile_stage_elems<NB>() <= 2 * (BMt + BNt) * BK, "staging fits");
        __syncthreads();
        staged_epilogue_bf16<EPI, MB, NB>(p, acc, &smem[0][0] + wave * small_tile_stage_elems<NB>(), m0 + wm * (32 * MB),
                                          n0 + wn * (32 * NB), lane);
        return;
    }
    if (EPI == EPI_RESID_GATE && p.ln_counter == nullptr && (p.debug_flags & 8) != 0) {
        atomic_epilogue_resid<MB, NB>(p, acc, m0 + wm * (32 * MB), n0 + wn * (32 * NB), lane);
        return;
    }
    gemm_epilogue<EPI, MB, NB>(p, acc, m0, n0, wm, wn, lane);
    if (EPI == EPI_RESID_GATE && p.ln_counter) resid_ln_tail(p, tm, BMt, (p.N + BNt - 1) / BNt);
}

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
    } else if (EPI == EPI_QKV_ROPE && ab.rope_cos_tk != nullptr) {
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
// v1r: the small-tile kernel with global_load_lds staging into a ring of NST K-tiles (NST-1 tiles in flight,
// counted vmcnt, one barrier per K tile).  Used for small-batch shapes (M ~ 2 * 937 rows) where the chip is only
// filled by 64x64 / 64x128 tiles and the old one-tile-deep register prefetch left every iteration latency bound.
// =================================================================================================
int f5_gemm_order = 0;   // 0 auto, 1 force n-fastest, 2 force m-fastest (ring kernel tile numbering)
static bool gemm_mfast(const F5GemmArgs& a) {
    if (f5_gemm_order == 1) return false;
    if (f5_gemm_order == 2) return true;
    const size_t a_bytes = (size_t)(a.a_row_mod > 0 ? a.a_row_mod : a.M) * a.K * 2 * (a.nseg == 3 ? 2 : 1);
    return a_bytes <= (size_t)4 << 20;     // whole A operand fits in one XCD's 4 MB L2
}

// ABL (timing experiments only, results are garbage; tools/ring_ablate.py): 1 = no operand loads after the prologue, 2 = no MFMAs,
// 4 = no LDS fragment reads, 8 = no workgroup barrier -- what a K step of a lone workgroup is made of
template <int EPI, int MB, int NB, int NST, int WM = 2, int WN = 2, int KS = 1, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN * KS) void f5_gemm_ring_kernel(F5GemmArgs p, int tiles_n, int ntiles) {
    // WM x WN waves per K group, wave tile 32*MB x 32*NB; KS groups split the K tiles round-robin (group g owns tiles
    // g, g+KS, ...; its own ring) and are summed in group order through LDS at the end: small-M problems are one round
    // of workgroups whose run time is a single workgroup's chain of K steps, which KS cuts by KS.
    constexpr int NTG = 64 * WM * WN;                   // threads per K group
    constexpr int BMt = 32 * MB * WM, BNt = 32 * NB * WN;
    constexpr int NA = BMt * 8 / NTG, NW = BNt * 8 / NTG;
    static_assert(NA * NTG == BMt * 8 && NW * NTG == BNt * 8, "tile rows must divide over the threads");
    constexpr int G = NA + NW;                          // global_load_lds per thread per K tile
    constexpr int STAGE = (BMt + BNt) * BK;             // elements
    constexpr int RING = NST * STAGE;
    static_assert(KS * RING * 2 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) op16_t smem_all[KS * RING];

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
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave_all / (WM * WN), wave = wave_all % (WM * WN);
    const int tg = tid - grp * NTG;
    const int wm = wave / WN, wn = wave % WN;
    op16_t* smem = smem_all + grp * RING;

    // staging: 32-bit BYTE offsets added to a uniform operand pointer (SGPR base + VGPR offset form of global_load_lds)
    uint32_t a_src[NA], w_src[NW];
    int a_dst[NA], w_dst[NW];                           // wave-uniform LDS element offsets inside a stage
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q_ = i * NTG + tg;
        const int row = q_ >> 3, chunk = (q_ & 7) ^ ((row >> 1) & 7);
        int gr = m0 + row;
        if (gr > p.M - 1) gr = p.M - 1;
        if (p.a_row_mod > 0) gr = gr % p.a_row_mod;
        a_src[i] = ((uint32_t)gr * (uint32_t)p.lda + chunk * 8) * 2u;
        a_dst[i] = (i * NTG + wave * 64) * 8;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int q_ = i * NTG + tg;
        const int row = q_ >> 3, chunk = (q_ & 7) ^ ((row >> 1) & 7);
        w_src[i] = ((uint32_t)(n0 + row) * (uint32_t)p.ldw + chunk * 8) * 2u;
        w_dst[i] = BMt * BK + (i * NTG + wave * 64) * 8;
    }
    const int kt = p.K / BK;
    const int T = kt * p.nseg;
    const int Tg = T > grp ? (T - grp + KS - 1) / KS : 0;    // K tiles of this group
    const int nit = (T + KS - 1) / KS;                       // iterations (= tiles of group 0)
    // the group's K tiles are staged in order: running (segment, K offset) of the next tile to issue (global tile jj*KS + grp)
    int is_seg = 0, is_k0 = grp * BK;
    while (is_k0 >= p.K) {
        is_k0 -= p.K;
        ++is_seg;
    }
#define RING_ISSUE(ST_)                                                                                      \
    {                                                                                                        \
        op16_t* st_ = smem + (ST_) * STAGE;                                                                  \
        const char* Ap_ = reinterpret_cast<const char*>((is_seg == 1) ? p.A[1] : p.A[0]);                    \
        const char* Wp_ = reinterpret_cast<const char*>((is_seg == 2) ? p.W[1] : p.W[0]);                    \
        const uint32_t kb_ = (uint32_t)is_k0 * 2u;                                                           \
        _Pragma("unroll") for (int i = 0; i < NA; ++i)                                                       \
            glds16(reinterpret_cast<const op16_t*>(Ap_ + (a_src[i] + kb_)), st_ + a_dst[i]);                 \
        _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                       \
            glds16(reinterpret_cast<const op16_t*>(Wp_ + (w_src[i] + kb_)), st_ + w_dst[i]);                 \
        is_k0 += KS * BK;                                                                                    \
        while (is_k0 >= p.K) {                   /* at most once unless K < KS * 64 */                       \
            is_k0 -= p.K;                                                                                    \
            ++is_seg;                                                                                        \
        }                                                                                                    \
    }

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // residual-update launches: the epilogue waves request x / bias / gate / keep now (see ResidPre); these loads are older than
    // every operand load, so the counted vmcnt waits below cover them (in-order return) and stay valid
    constexpr bool PRE_RESID = EPI == EPI_RESID_GATE && MB * NB <= 3;   // 64 more live VGPRs would spill the 2x2 wave tile
    ResidPre<PRE_RESID ? MB : 1, PRE_RESID ? NB : 1> rpre;
    const bool use_pre = PRE_RESID && p.ln_counter == nullptr && (p.debug_flags & (8 | 256)) == 0;
    if (PRE_RESID && use_pre && grp == 0) resid_preload<PRE_RESID ? MB : 1, PRE_RESID ? NB : 1>(p, rpre, m0 + wm * (32 * MB), n0 + wn * (32 * NB), lane);

#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
        if (st < Tg) RING_ISSUE(st);

    const int frow = lane & 31;
    const int fk = lane >> 5;
    // fragment read pointers per K sub-step (and per 64 KB window of the ring): stage, row block and A / W part are ds_read
    // immediates, the K loop is unrolled NST times so the stage is a constant (every non-MFMA instruction of the loop is paid
    // in full on this chip: tools/probes/coissue.hip)
    constexpr int SPS = 65536 / (STAGE * 2) < NST ? 65536 / (STAGE * 2) : NST;   // stages per 64 KB window (16-bit ds_read offsets)
    static_assert(SPS >= 1, "a stage must fit the ds_read offset field");
    constexpr int NSET = (NST + SPS - 1) / SPS;                      // windows = pointer sets
    const op16_t* pa[NSET][4];
    const op16_t* pb[NSET][4];
#pragma unroll
    for (int w2 = 0; w2 < NSET; ++w2)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            pa[w2][ks] = smem + w2 * SPS * STAGE + swz_off(wm * (32 * MB) + frow, ks * 2 + fk);
            pb[w2][ks] = smem + w2 * SPS * STAGE + BMt * BK + swz_off(wn * (32 * NB) + frow, ks * 2 + fk);
        }
    op16x8 abl_frag;
#pragma unroll
    for (int e = 0; e < 8; ++e) abl_frag[e] = (op16_t)0;
    if (ABL & 4) asm volatile("" : "+v"(abl_frag));
#define RING_STEP(ST_, jj_, TR_)                                                                                       \
    {                                                                                                               \
        /* tile jj must have landed; up to NST-2 younger tiles may stay in flight (conservative vmcnt(0) at the tail) */ \
        if ((jj_) + NST - 2 < Tg) {                                                                                 \
            if (NST == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");                              \
            else if (NST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");                             \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
        } else {                                                                                                    \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
        }                                                                                                           \
        asm volatile("" ::: "memory");                                                                              \
        if (!(ABL & 8)) __builtin_amdgcn_s_barrier();                                                               \
        asm volatile("" ::: "memory");                                                                              \
        if (!(ABL & 1) && (jj_) + NST - 1 < Tg) RING_ISSUE(((ST_) + NST - 1) % NST); /* refills the slot consumed one iteration ago */ \
        if (!(KS > 1 && (jj_) >= Tg)) {              /* wave-uniform: a group without a tile left still meets the barrier */ \
            constexpr int W2 = (ST_) / SPS, SL = (ST_) % SPS;                                                        \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                      \
                op16x8 af[MB], bfr[NB];                                                                             \
                _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                   \
                    af[mb] = (ABL & 4) ? abl_frag : *reinterpret_cast<const op16x8*>(pa[W2][ks] + SL * STAGE + mb * 32 * BK); \
                _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                   \
                    bfr[nb] = (ABL & 4) ? abl_frag : *reinterpret_cast<const op16x8*>(pb[W2][ks] + SL * STAGE + nb * 32 * BK); \
                _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                                   \
                    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {                                             \
                        if (ABL & 2) asm volatile("" ::"v"(af[mb]), "v"(bfr[nb]));                                  \
                        else if (TR_) acc[mb][nb] = F5_MFMA32(bfr[nb], af[mb], acc[mb][nb], 0, 0, 0);               \
                        else acc[mb][nb] = F5_MFMA32(af[mb], bfr[nb], acc[mb][nb], 0, 0, 0);                        \
                    }                                                                                               \
            }                                                                                                       \
        }                                                                                                           \
    }
    // EPI_QKV_ROPE with pair-major rotation tables on the 8-wave instances: the q / k wave tiles are accumulated transposed and
    // leave through staged_epilogue_tr_rope (wave-uniform choice; the block tile lies inside one of the q | k | v ranges, so
    // every wave of the workgroup takes the same side and meets the same barriers).  The 4-wave instances keep the straight
    // tiles: an earlier LDS-free variant of this path returned stale values on a few lanes there, nondeterministically
    // (profiles/r02/qkv_direct_epilogue_rejected.txt), and the cause was not found.
    constexpr bool QKV_TR_OK = EPI == EPI_QKV_ROPE && (32 * NB) % 64 == 0 && (NB & (NB - 1)) == 0 && KS == 1 && WM * WN == 8;
    if (QKV_TR_OK && p.rope_cos_tk != nullptr && n0 + wn * (32 * NB) < 2 * p.dmodel) {
        for (int jj = 0; jj < nit; jj += NST) {
            RING_STEP(0, jj, true);
            if (NST > 1 && jj + 1 < nit) RING_STEP(1 % NST, jj + 1, true);
            if (NST > 2 && jj + 2 < nit) RING_STEP(2 % NST, jj + 2, true);
            if (NST > 3 && jj + 3 < nit) RING_STEP(3 % NST, jj + 3, true);
        }
        __syncthreads();                                  // the ring is dead: its LDS becomes the staging area
        staged_epilogue_tr_rope<MB, NB>(p, acc, smem_all + wave * small_tile_stage_elems<NB>(), m0 + wm * (32 * MB), n0 + wn * (32 * NB),
                                        lane);
        return;
    }
    for (int jj = 0; jj < nit; jj += NST) {
        RING_STEP(0, jj, false);
        if (NST > 1 && jj + 1 < nit) RING_STEP(1 % NST, jj + 1, false);
        if (NST > 2 && jj + 2 < nit) RING_STEP(2 % NST, jj + 2, false);
        if (NST > 3 && jj + 3 < nit) RING_STEP(3 % NST, jj + 3, false);
    }
#undef RING_STEP
#undef RING_ISSUE
    constexpr int STG = small_tile_staged<EPI, NB>() ? WM * WN * small_tile_stage_elems<NB>() : 0;   // staging area (bf16 elements)
    if (KS > 1) {
        // partial sums of groups 1.. -> LDS (behind the epilogue staging area), added by group 0 in group order
        constexpr int RED = MB * NB * 16 * 64;          // floats per wave
        static_assert(KS == 1 || STG * 2 + (KS - 1) * WM * WN * RED * 4 <= KS * RING * 2, "reduction area fits in the rings");
        float* red = reinterpret_cast<float*>(smem_all + STG);
        __syncthreads();
        if (grp > 0) {
            float* dst = red + (size_t)((grp - 1) * (WM * WN) + wave) * RED + lane;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) dst[((mb * NB + nb) * 16 + e) * 64] = acc[mb][nb][e];
        }
        __syncthreads();
        if (grp > 0) {
            if (EPI == EPI_RESID_GATE && p.ln_counter) resid_ln_tail(p, tm, BMt, (p.N + BNt - 1) / BNt);
            return;
        }
#pragma unroll
        for (int g = 1; g < KS; ++g) {
            const float* src = red + (size_t)((g - 1) * (WM * WN) + wave) * RED + lane;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mb][nb][e] += src[((mb * NB + nb) * 16 + e) * 64];
        }
    }
    if (small_tile_staged<EPI, NB>() && n0 + BNt <= p.N && (p.debug_flags & 2) == 0) {
        static_assert(STG <= RING, "staging fits");
        if (KS == 1) __syncthreads();
        staged_epilogue_bf16<EPI, MB, NB>(p, acc, smem_all + wave * small_tile_stage_elems<NB>(), m0 + wm * (32 * MB),
                                          n0 + wn * (32 * NB), lane);
        return;
    }
    if (EPI == EPI_RESID_GATE && p.ln_counter == nullptr && (p.debug_flags & 8) != 0) {
        atomic_epilogue_resid<MB, NB>(p, acc, m0 + wm * (32 * MB), n0 + wn * (32 * NB), lane);
        return;
    }
    if constexpr (PRE_RESID) {
        if (use_pre) {
            resid_epilogue_preloaded<MB, NB>(p, acc, rpre, m0 + wm * (32 * MB), n0 + wn * (32 * NB), lane);
            return;
        }
    }
    gemm_epilogue<EPI, MB, NB>(p, acc, m0, n0, wm, wn, lane);
    if (EPI == EPI_RESID_GATE && p.ln_counter) resid_ln_tail(p, tm, BMt, (p.N + BNt - 1) / BNt);
}

template <int EPI, int MB, int NB>
static int launch_ring(const F5GemmArgs& a, hipStream_t stream) {
    const int tiles_m = f5_cdiv(a.M, 64 * MB), tiles_n = f5_cdiv(a.N, 64 * NB);
    const int ntiles = tiles_m * tiles_n;
    const int order = gemm_mfast(a) ? -tiles_m : tiles_n;
    // ring depth: keep TWO workgroups per CU (<= 80 KB of LDS each): 64x64 tiles take 4 stages (64 KB), 64x128 take 3 (72 KB)
    constexpr int NST = (MB + NB) * 8 * 4 <= 80 ? 4 : 3;
    hipLaunchKernelGGL((f5_gemm_ring_kernel<EPI, MB, NB, NST>), dim3(ntiles), dim3(256), 0, stream, a, order, ntiles);
    F5_LAUNCH_CHECK();
    return 0;
}

// 8-wave ring kernel for ONE round of workgroups on a small-M problem (M = 2*937 rows at batch 1): 4 x 2 waves, wave tile
// 32 x 32*NB, block tile 128 x 64*NB (NB = 3: 128x192 -> 15 x 16 = 240 workgroups for the QKV projection, NB = 2: 128x128 ->
// 240 for FF1).  One workgroup per CU with 8 waves and 3-4 K tiles in flight moves a third less L2->LDS traffic than the
// 64x128 tiles (the bound there: tools/b1_decompose.py) without leaving CUs idle.
// in-workgroup split-K ring kernels (tile overrides 10 / 11): 64x128 tile, 2 x (2x2 waves, wave tile 32x64), 3 stages per group
// (out-proj / FF2 at batch 1: 240 workgroups, K chain halved); 128x128 tile, 2 x (2x2 waves, wave tile 64x64), 2 stages
template <int EPI, int MB>
static int launch_ring_ks2(const F5GemmArgs& a, hipStream_t stream) {
    constexpr int BMt = 64 * MB, BNt = 128;
    const int tiles_m = f5_cdiv(a.M, BMt), tiles_n = f5_cdiv(a.N, BNt);
    const int ntiles = tiles_m * tiles_n;
    const int order = gemm_mfast(a) ? -tiles_m : tiles_n;
    constexpr int NST = MB == 1 ? 3 : 2;
    hipLaunchKernelGGL((f5_gemm_ring_kernel<EPI, MB, 2, NST, 2, 2, 2>), dim3(ntiles), dim3(512), 0, stream, a, order, ntiles);
    F5_LAUNCH_CHECK();
    return 0;
}

template <int EPI, int NB>
static int launch_ring8(const F5GemmArgs& a, hipStream_t stream) {
    constexpr int BMt = 128, BNt = 64 * NB;
    const int tiles_m = f5_cdiv(a.M, BMt), tiles_n = a.N / BNt;
    const int ntiles = tiles_m * tiles_n;
    const int order = gemm_mfast(a) ? -tiles_m : tiles_n;
    constexpr int NST = (BMt + BNt) * 64 * 2 * 4 <= 128 * 1024 ? 4 : 3;
    hipLaunchKernelGGL((f5_gemm_ring_kernel<EPI, 1, NB, NST, 4, 2>), dim3(ntiles), dim3(512), 0, stream, a, order, ntiles);
    F5_LAUNCH_CHECK();
    return 0;
}

// 128x256 ring tiles for ONE round on the QKV projection at batch 1 (M = 1874, N = 3072: 15 x 12 = 180 workgroups), two
// wave layouts (tile overrides 12 / 13): 8 waves of 64x64 and 8 waves of 32x128, i.e. 1.0 / 1.25 KB of LDS fragment reads
// per MFMA against 1.5 for the 64x128 tile of 4 waves of 32x64, and half the L2->LDS bytes per flop.  Measured in-graph
// (tools/qkv_tiles_bench.py): 27.0-28.1 / 24.7-24.8 us against 25.5-27.2 us for the register-staged 64x128 default, and
// 22.3-23.1 us at M = 937 where only 96 workgroups exist: a lone workgroup takes ~1.4 us per K tile whatever the fill, three
// times its MFMA time -- neither L2 bytes nor occupancy is what bounds this shape.  Kept as overrides, not selected.
template <int EPI, int MB, int NB, int WM, int WN, int KS = 1, int ABL = 0>
static int launch_ring_wide(const F5GemmArgs& a, hipStream_t stream) {
    constexpr int BMt = 32 * MB * WM, BNt = 32 * NB * WN;
    F5_REQUIRE(a.N % BNt == 0, "gemm: this tile needs N %% %d == 0", BNt);
    const int tiles_m = f5_cdiv(a.M, BMt), tiles_n = a.N / BNt;
    const int ntiles = tiles_m * tiles_n;
    const int order = gemm_mfast(a) ? -tiles_m : tiles_n;
    hipLaunchKernelGGL((f5_gemm_ring_kernel<EPI, MB, NB, 3, WM, WN, KS, ABL>), dim3(ntiles), dim3(64 * WM * WN * KS), 0, stream, a, order,
                       ntiles);
    F5_LAUNCH_CHECK();
    return 0;
}
// timing-only ablations of the ring main loop (debug flags bits 4-7), bf16 epilogue, tile 13 (128x256, 8 waves) and tile 10
// (64x128, two K groups of 4 waves)
template <int ABL>
static int launch_ring_ablate(const F5GemmArgs& a, int sel, hipStream_t stream) {
    return sel == 13 ? launch_ring_wide<EPI_BF16, 1, 4, 4, 2, 1, ABL>(a, stream) : launch_ring_wide<EPI_BF16, 1, 2, 2, 2, 2, ABL>(a, stream);
}
static int launch_ring_ablate(const F5GemmArgs& a, int sel, int abl, hipStream_t stream) {
    switch (abl) {
        case 1: return launch_ring_ablate<1>(a, sel, stream);
        case 2: return launch_ring_ablate<2>(a, sel, stream);
        case 4: return launch_ring_ablate<4>(a, sel, stream);
        case 8: return launch_ring_ablate<8>(a, sel, stream);
        case 6: return launch_ring_ablate<6>(a, sel, stream);     // loads + barrier only
        case 7: return launch_ring_ablate<7>(a, sel, stream);     // barrier + loop skeleton
        case 9: return launch_ring_ablate<9>(a, sel, stream);     // LDS reads + MFMAs, no loads, no barrier
        case 15: return launch_ring_ablate<15>(a, sel, stream);   // loop skeleton
        default: f5_set_error("gemm: ablation %d is not instantiated", abl); return 2;
    }
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

template <int EPI, int MB, int NB>
static int launch_cfg(const F5GemmArgs& a, hipStream_t stream) {
    const int tiles_m = f5_cdiv(a.M, 64 * MB), tiles_n = f5_cdiv(a.N, 64 * NB);
    const int ntiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((f5_gemm_kernel<EPI, MB, NB>), dim3(ntiles), dim3(256), 0, stream, a, tiles_n, ntiles);
    F5_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// MX-fp8 GEMM (BASELINE configs[4]): e4m3 operands with one E8M0 scale per 32 K elements, fp32 accumulate, on
// v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 MFMA rate, gfx950 only).  Same skeleton as the 256x256 bf16 kernel: the
// LDS image of a half tile is again 128 rows x 128 BYTES (= 128 K elements now), staged with global_load_lds, XOR
// swizzled on the source address, 4 phases per K tile with the same issue / counted-wait schedule.  Operand layout of the
// instruction (tools/probes/mxfp8.hip, mxscale.hip): lane l holds row l&31; its bytes 0-15 belong to MX block 0 and bytes
// 16-31 to MX block 1 of the 64-wide K step (k = 32*(j/16) + 16*(l>>5) + j%16), i.e. two ds_read_b128 of non-adjacent 16-byte
// chunks; the scale of block b comes from lane (l&31) + 32b, byte `opsel` of its scale VGPR.  The scales of a K tile (4 bytes
// per row: one dword) ride along as ONE extra 4-byte global_load_lds per thread (threads 0-255: A rows, 256-511: W rows).
// =================================================================================================
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void glds4(const uint8_t* gptr, uint8_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 4, 0, 0);
}
__device__ __forceinline__ void glds16b(const uint8_t* gptr, uint8_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0);
}
__device__ __forceinline__ int f8_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }   // bytes
// operand bytes of lane (row, h = lane>>5) for the 64-wide K sub-tile ks: bytes 0-15 = k [16h, 16h+16) of MX block 2ks,
// bytes 16-31 = k [32+16h, 48+16h) of MX block 2ks+1 (tools/probes/mxscale.hip: the hardware's block b is bytes
// [16b, 16b+16) of BOTH lane halves, scaled by the E8M0 of lane row + 32b) -> 16-byte chunks 4ks+h and 4ks+2+h of the row
__device__ __forceinline__ i32x8 f8_frag(const uint8_t* half, int row, int ks, int h) {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(half + f8_off(row, 4 * ks + h));
    const u32x4 hi = *reinterpret_cast<const u32x4*>(half + f8_off(row, 4 * ks + 2 + h));
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}

// fp8 + E8M0 output of GELU(acc + bias): the wave's 32x64 block goes through LDS as fp32 rows, then every lane owns 8
// consecutive columns (4 lanes = one 32-column MX block: amax by two lane shuffles), scales to (224, 448], packs 8 bytes.
__device__ __forceinline__ void staged_epilogue_gelu_f8(const F5GemmArgs& p, f32x16 (&acc)[4][2], float* reg, int row0,
                                                        int colbase, int lane) {
    constexpr int LD = 64 + 4;
    const int hi = lane >> 5, lcol = lane & 31;
    float bcol[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) bcol[nb] = p.bias ? p.bias[colbase + nb * 32 + lcol] : 0.0f;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int rowblk = row0 + mb * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = (r & 3) + 8 * (r >> 2) + 4 * hi;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) reg[lrow * LD + nb * 32 + lcol] = f5_gelu_tanh(acc[mb][nb][r] + bcol[nb]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {                     // 8 rows per pass: 8 lanes x 8 columns per row
            const int lrow = i * 8 + (lane >> 3), c0 = (lane & 7) * 8;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(&reg[lrow * LD + c0]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(&reg[lrow * LD + c0 + 4]);
            float am = fmaxf(fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))),
                             fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3]))));
            am = fmaxf(am, __shfl_xor(am, 1, 64));
            am = fmaxf(am, __shfl_xor(am, 2, 64));
            const int e8 = f5_mx_scale_byte(am);
            const float inv = f5_mx_inv_scale(e8);
            const int grow = rowblk + lrow;
            if (grow < p.M) {
                const u32x2 pk = {f5_pack4_fp8(v0[0] * inv, v0[1] * inv, v0[2] * inv, v0[3] * inv),
                                  f5_pack4_fp8(v1[0] * inv, v1[1] * inv, v1[2] * inv, v1[3] * inv)};
                *reinterpret_cast<u32x2*>(p.out8 + (size_t)grow * p.ldo8 + colbase + c0) = pk;
                if ((lane & 3) == 0) p.out8s[(size_t)grow * (p.N >> 5) + ((colbase + c0) >> 5)] = (uint8_t)e8;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int EPI>
__global__ __launch_bounds__(512) void f5_gemm256f8_kernel(F5GemmArgs p, int tiles_n, int ntiles) {
    constexpr int HALF = 128 * 128;                     // bytes per half tile
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * 4 * HALF + 2 * 2048];   // [dbuf][A0,A1,B0,B1] + [dbuf][512 scale dwords]
    uint8_t* sscale = smem + 2 * 4 * HALF;

    const int bid = blockIdx.x;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    uint32_t srcA[2][2], srcB[2][2];   // [half][j] byte offsets (without k0); 32-bit: uniform base + VGPR offset addressing
    int ldsoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q_ = j * 512 + tid;
        const int row = q_ >> 3, slot = q_ & 7;
        const int chunk = slot ^ ((row >> 1) & 7);
        ldsoff[j] = (j * 512 + wave * 64) * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int gr = m0 + h * 128 + row;
            if (gr > p.M - 1) gr = p.M - 1;
            srcA[h][j] = (uint32_t)gr * (uint32_t)p.lda8 + chunk * 16;
            srcB[h][j] = (uint32_t)(n0 + h * 128 + row) * (uint32_t)p.ldw8 + chunk * 16;
        }
    }
    // scale source: threads 0..255 -> A row m0 + tid, 256..511 -> W row n0 + tid - 256; one dword (4 K blocks) per K tile
    const int ksc = p.K >> 5;                            // scale bytes per row
    const uint8_t* ssrc;
    {
        int gr = m0 + tid;
        if (gr > p.M - 1) gr = p.M - 1;
        ssrc = tid < 256 ? p.As + (size_t)gr * ksc : p.Ws + (size_t)(n0 + tid - 256) * ksc;
    }
    const int T = p.K >> 7;                              // K tiles of 128

#define F8_ISSUE_A(tt_, h_)                                                                         \
    {                                                                                               \
        uint8_t* dst_ = smem + ((h_) * 2 + ((tt_) & 1)) * HALF;                                     \
        glds16b(p.A8 + (srcA[(h_)][0] + (uint32_t)(tt_) * 128u), dst_ + ldsoff[0]);                 \
        glds16b(p.A8 + (srcA[(h_)][1] + (uint32_t)(tt_) * 128u), dst_ + ldsoff[1]);                 \
    }
#define F8_ISSUE_B(tt_, h_)                                                                         \
    {                                                                                               \
        uint8_t* dst_ = smem + ((2 + (h_)) * 2 + ((tt_) & 1)) * HALF;                               \
        glds16b(p.W8 + (srcB[(h_)][0] + (uint32_t)(tt_) * 128u), dst_ + ldsoff[0]);                 \
        glds16b(p.W8 + (srcB[(h_)][1] + (uint32_t)(tt_) * 128u), dst_ + ldsoff[1]);                 \
    }
#define F8_ISSUE_S(tt_) glds4(ssrc + (tt_) * 4, sscale + ((tt_) & 1) * 2048 + wave * 256)
#define F8_BARRIER()                                   \
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

    // ---- prologue: tile 0 (4 halves + scales) + B halves of tile 1
    F8_ISSUE_A(0, 0);
    F8_ISSUE_A(0, 1);
    F8_ISSUE_B(0, 0);
    F8_ISSUE_B(0, 1);
    F8_ISSUE_S(0);
    if (T > 1) {
        F8_ISSUE_B(1, 0);
        F8_ISSUE_B(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    F8_BARRIER();

    const int frow = lane & 31;
    const int fk = lane >> 5;
    // fragment / scale read pointers per (ring buffer, K sub-tile, chunk): row-block and quadrant offsets are immediates, so the
    // loop issues no VALU instruction for LDS addressing (nothing co-issues with an MFMA on this chip: tools/probes/coissue.hip)
    const int brow0 = (wn & 1) * 64;
    // LDS layout [A0,A1,B0,B1][ring buffer][128 x 128 B]: the ring-buffer offset (16 KB) also fits the ds_read offset field
    const uint8_t* qa[2][2];
    const uint8_t* qb[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
            qa[ks][w2] = smem + (wm * 2) * HALF + f8_off(frow, 4 * ks + 2 * w2 + fk);
            qb[ks][w2] = smem + ((2 + (wn >> 1)) * 2) * HALF + f8_off(brow0 + frow, 4 * ks + 2 * w2 + fk);
        }
    const uint32_t* sca = reinterpret_cast<const uint32_t*>(sscale) + wm * 128 + frow;
    const uint32_t* scb = reinterpret_cast<const uint32_t*>(sscale) + 256 + (wn >> 1) * 128 + brow0 + frow;
    i32x8 af[2][2], bfr[2][2];
    int sa[2], sb[2];       // scale dword of the row, shifted so that byte 0 / byte 2 = this lane's K block of sub-tile 0 / 1
#define F8_FRAG(Q, PAR, ks, rowoff)                                                                            \
    ([&]() {                                                                                                   \
        const u32x4 lo_ = *reinterpret_cast<const u32x4*>(Q[ks][0] + (PAR) * HALF + (rowoff) * 128);            \
        const u32x4 hi_ = *reinterpret_cast<const u32x4*>(Q[ks][1] + (PAR) * HALF + (rowoff) * 128);            \
        return i32x8{(int)lo_[0], (int)lo_[1], (int)lo_[2], (int)lo_[3], (int)hi_[0], (int)hi_[1], (int)hi_[2], (int)hi_[3]}; \
    }())
#define F8_MFMA2(ACC, AF, BF, SA, SB)                                                                          \
    ACC = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(AF[0], BF[0], ACC, 0, 0, 0, SA, 0, SB);              \
    ACC = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(AF[1], BF[1], ACC, 0, 0, 2, SA, 2, SB);
#define F8_KSTEP(PAR, tt)                                                                                      \
    {                                                                                                          \
        /* phase 1: A(mq=0), B(nq=0); quadrant (0,0); issue A0(t+1) + scales(t+1) */                           \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) {                                                     \
            sa[mb] = (int)(sca[(PAR) * 512 + mb * 32] >> (8 * fk));                                                     \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) af[mb][ks] = F8_FRAG(qa, PAR, ks, mb * 32);       \
        }                                                                                                      \
        sb[0] = (int)(scb[(PAR) * 512] >> (8 * fk));                                                                \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) bfr[0][ks] = F8_FRAG(qb, PAR, ks, 0);                 \
        if ((tt) + 1 < T) {                                                                                    \
            F8_ISSUE_A((tt) + 1, 0);                                                                           \
            F8_ISSUE_S((tt) + 1);                                                                              \
        }                                                                                                      \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) { F8_MFMA2(acc[mb][0], af[mb], bfr[0], sa[mb], sb[0]) } \
        /* phase 2: B(nq=1); quadrant (0,1); issue A1(t+1) */                                                  \
        sb[1] = (int)(scb[(PAR) * 512 + 32] >> (8 * fk));                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) bfr[1][ks] = F8_FRAG(qb, PAR, ks, 32);                \
        if ((tt) + 1 < T) F8_ISSUE_A((tt) + 1, 1);                                                             \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) { F8_MFMA2(acc[mb][1], af[mb], bfr[1], sa[mb], sb[1]) } \
        F8_BARRIER(); /* every wave has finished reading the B halves of this tile */                          \
        /* phase 3: A(mq=1); quadrant (1,1); issue B0(t+2) */                                                  \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) {                                                     \
            sa[mb] = (int)(sca[(PAR) * 512 + 64 + mb * 32] >> (8 * fk));                                                \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) af[mb][ks] = F8_FRAG(qa, PAR, ks, 64 + mb * 32);  \
        }                                                                                                      \
        if ((tt) + 2 < T) F8_ISSUE_B((tt) + 2, 0);                                                             \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) { F8_MFMA2(acc[2 + mb][1], af[mb], bfr[1], sa[mb], sb[1]) } \
        F8_BARRIER(); /* every wave has finished reading the A halves (and the scales) of this tile */         \
        /* phase 4: quadrant (1,0) from registers; issue B1(t+2) */                                            \
        if ((tt) + 2 < T) F8_ISSUE_B((tt) + 2, 1);                                                             \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) { F8_MFMA2(acc[2 + mb][0], af[mb], bfr[0], sa[mb], sb[0]) } \
        /* next tile's operands and scales: everything but the two B halves issued for tile t+2 must have landed */ \
        if ((tt) + 2 < T) {                                                                                    \
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                   \
        } else {                                                                                               \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
        }                                                                                                      \
        F8_BARRIER();                                                                                          \
    }
    for (int tt = 0; tt < T; tt += 2) {
        F8_KSTEP(0, tt);
        if (tt + 1 < T) F8_KSTEP(1, tt + 1);
    }
#undef F8_KSTEP
#undef F8_MFMA2
#undef F8_FRAG
#undef F8_ISSUE_A
#undef F8_ISSUE_B
#undef F8_ISSUE_S
#undef F8_BARRIER

    op16_t* stage = reinterpret_cast<op16_t*>(smem) + wave * 8192;
    if (EPI == EPI_BF16 || EPI == EPI_QKV_ROPE) {
        staged_epilogue_bf16<EPI, 4, 2>(p, acc, stage, m0 + wm * 128, n0 + wn * 64, lane);
    } else if (EPI == EPI_GELU_TANH) {
        staged_epilogue_gelu_f8(p, acc, reinterpret_cast<float*>(stage), m0 + wm * 128, n0 + wn * 64, lane);
    } else if (EPI == EPI_RESID_GATE) {
        if (p.debug_flags & 8) atomic_epilogue_resid<4, 2>(p, acc, m0 + wm * 128, n0 + wn * 64, lane);
        else staged_epilogue_resid<4, 2>(p, acc, reinterpret_cast<float*>(stage), m0 + wm * 128, n0 + wn * 64, lane);
    } else {
        gemm_epilogue<EPI, 4, 2>(p, acc, m0, n0, wm, wn, lane);
    }
}

template <int EPI>
static int launch_f8(const F5GemmArgs& a, hipStream_t stream) {
    const int tiles_m = f5_cdiv(a.M, 256), tiles_n = a.N / 256;
    const int ntiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((f5_gemm256f8_kernel<EPI>), dim3(ntiles), dim3(512), 0, stream, a, tiles_n, ntiles);
    F5_LAUNCH_CHECK();
    return 0;
}
extern int f5_gemm_debug_flags;
int f5_launch_gemm_f8(const F5GemmArgs& a_in, int epi, hipStream_t stream) {
    F5GemmArgs a = a_in;
    a.debug_flags = f5_gemm_debug_flags;
    F5_REQUIRE(epi != EPI_RESID_GATE || (size_t)a.M * a.ldo * 4 < (1ull << 32), "gemm_f8(resid): the residual stream must stay below 4 GiB");
    F5_REQUIRE(a.M > 0 && a.N > 0 && a.N % 256 == 0 && a.K > 0 && a.K % 128 == 0,
               "gemm_f8: bad shape M=%d N=%d K=%d (N %% 256 == 0, K %% 128 == 0)", a.M, a.N, a.K);
    F5_REQUIRE(a.A8 && a.W8 && a.As && a.Ws, "gemm_f8: null operand");
    F5_REQUIRE(a.lda8 % 16 == 0 && a.ldw8 % 16 == 0 && a.lda8 >= a.K && a.ldw8 >= a.K, "gemm_f8: leading dims must be multiples of 16 and >= K");
    F5_REQUIRE((size_t)a.M * a.lda8 < (1ull << 32) && (size_t)(a.N + 256) * a.ldw8 < (1ull << 32), "gemm_f8: operands must be < 4 GiB");
    switch (epi) {
        case EPI_F32: return launch_f8<EPI_F32>(a, stream);
        case EPI_BF16: return launch_f8<EPI_BF16>(a, stream);
        case EPI_GELU_TANH:
            F5_REQUIRE(a.out8 && a.out8s && a.ldo8 >= a.N, "gemm_f8(gelu): fp8 output buffers missing");
            return launch_f8<EPI_GELU_TANH>(a, stream);
        case EPI_RESID_GATE: return launch_f8<EPI_RESID_GATE>(a, stream);
        case EPI_QKV_ROPE:
            F5_REQUIRE(a.dmodel % 128 == 0 && a.N == 3 * a.dmodel, "gemm_f8(qkv): N must be 3*dmodel, dmodel %% 128 == 0");
            return launch_f8<EPI_QKV_ROPE>(a, stream);
        default: f5_set_error("gemm_f8: unsupported epilogue %d", epi); return 2;
    }
}

// rows of fp32 -> e4m3 + E8M0 (one wave per row pass of 256 columns: lane = 4 consecutive columns, 8 lanes = one block)
__global__ __launch_bounds__(256) void quantize_mx_kernel(const float* __restrict__ x, int ldx, uint8_t* __restrict__ q, int ldq,
                                                          uint8_t* __restrict__ sc, int rows, int cols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    for (int c0 = lane * 4; c0 < cols; c0 += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + c0);
        float am = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        am = fmaxf(am, __shfl_xor(am, 1, 64));
        am = fmaxf(am, __shfl_xor(am, 2, 64));
        am = fmaxf(am, __shfl_xor(am, 4, 64));
        const int e8 = f5_mx_scale_byte(am);
        const float inv = f5_mx_inv_scale(e8);
        *reinterpret_cast<uint32_t*>(q + (size_t)row * ldq + c0) = f5_pack4_fp8(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
        if ((lane & 7) == 0) sc[(size_t)row * (cols >> 5) + (c0 >> 5)] = (uint8_t)e8;
    }
}
int f5_launch_quantize_mx(const float* x, int ldx, uint8_t* q, int ldq, uint8_t* sc, int rows, int cols, hipStream_t stream) {
    F5_REQUIRE(rows > 0 && cols > 0 && cols % 32 == 0 && ldx % 4 == 0 && ldq % 4 == 0, "quantize_mx: cols must be a multiple of 32");
    hipLaunchKernelGGL(quantize_mx_kernel, dim3(f5_cdiv(rows, 4)), dim3(256), 0, stream, x, ldx, q, ldq, sc, rows, cols);
    F5_LAUNCH_CHECK();
    return 0;
}

// tile shape: the largest of 128x128 / 64x128 / 64x64 that still gives the 256 CUs >= 1.5 workgroups each
// (small-batch shapes such as M = 1874 are otherwise a fraction of one wave of tiles)
int f5_gemm_tile_override = 0;  // 0 auto, 1 = 128x128, 2 = 64x128, 3 = 64x64, 4 = 256x256 v2, 5 = 64x128 ring, 6 = 64x64 ring,
                                // 7 = 128x256 v3, 8 = 128x192 8-wave ring, 9 = 128x128 8-wave ring, 10 / 11 = 64x128 / 128x128 split-K ring
int f5_gemm_debug_flags = 0;
int f5_gemm_big_kernel = 2;       // auto mode, large shapes: 2 = 256x256 role-split schedule (gemm256.hip), 3 = 128x256 (2 WG/CU), 4 = 256x256 lock-step
int f5_gemm_qkv_small_tile = 0;   // small-M QKV projection with pair-major tables: 0 = auto tiles, 12 / 13 = 8-wave 128x256 ring, transposed q / k
int f5_gemm_ring_default = 1;   // auto mode: small tiles use the global_load_lds ring kernel
// the large-shape kernels (256x256, 128x256) have no fused LN tail (at those sizes LN-modulate is HBM-bound, not launch-bound)
static bool gemm_uses_big_kernel(const F5GemmArgs& a) {
    const long t256 = (long)f5_cdiv(a.M, 256) * (a.N / 256);
    const bool v2ok = (a.N % 256 == 0) && (a.M >= 256);
    const int sel = f5_gemm_tile_override;
    return sel == 7 || sel == 4 || (sel == 0 && v2ok && t256 >= 512);
}
bool f5_gemm_resid_ln_fusable(const F5GemmArgs& a) {
    return !gemm_uses_big_kernel(a) && a.N % 256 == 0 && a.N >= 256 && a.N <= 1024 && a.ldo == a.N && a.M <= 64 * 65536;
}

template <int EPI>
static int launch_epi(const F5GemmArgs& a, hipStream_t stream) {
    if (a.ln_counter) {
        F5_REQUIRE(EPI == EPI_RESID_GATE && f5_gemm_resid_ln_fusable(a) && a.ln_scale && a.ln_shift && a.ln_out[0],
                   "gemm: the fused LN tail needs EPI_RESID_GATE on a small-tile shape (f5_gemm_resid_ln_fusable) and ln_* set");
    }
    const long t128 = (long)f5_cdiv(a.M, 128) * f5_cdiv(a.N, 128);
    const long t64x128 = (long)f5_cdiv(a.M, 64) * f5_cdiv(a.N, 128);
    int sel = f5_gemm_tile_override;
    const long t256 = (long)f5_cdiv(a.M, 256) * (a.N / 256);
    const bool v2ok = (a.N % 256 == 0) && (a.M >= 256);
    if (sel == 7 || (sel == 0 && v2ok && t256 >= 512 && f5_gemm_big_kernel == 3)) {
        F5_REQUIRE(v2ok && a.K % V3_BK == 0, "gemm: the 128x256 kernel needs N %% 256 == 0 and M >= 256");
        return launch_v3<EPI>(a, stream);
    }
    if (sel == 4 || (sel == 0 && v2ok && t256 >= 512)) {
        F5_REQUIRE(v2ok, "gemm: the 256x256 kernel needs N %% 256 == 0 and M >= 256");
        if (f5_gemm_big_kernel == 4 || f5_gemm_streamk) return launch_v2<EPI>(a, stream);      // lock-step predecessor (A/B)
        if constexpr (EPI == EPI_F32 || EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_RESID_GATE || EPI == EPI_QKV_ROPE) {
            if (f5_gemm_big_kernel == 5) return f5_launch_gemm128(a, EPI, stream);              // 128x256, two workgroups per CU
        }
        return f5_launch_gemm256(a, EPI, stream);
    }
    if constexpr (EPI == EPI_BF16) {
        const int abl = (a.debug_flags >> 4) & 15;
        if (abl && (sel == 13 || sel == 10) && a.N % 256 == 0) return launch_ring_ablate(a, sel, abl, stream);
    }
    if constexpr (EPI == EPI_QKV_ROPE) {
        // batch-1-sized QKV projection with pair-major tables: one round of 8-wave 128 x 256 tiles with transposed q / k wave tiles
        // (f5_gemm_qkv_small_tile = 13 / 12) instead of 64 x 128 register-staged tiles (0)
        if (sel == 0 && (f5_gemm_qkv_small_tile == 12 || f5_gemm_qkv_small_tile == 13) && a.rope_cos_tk != nullptr && a.N % 256 == 0) {
            const long t128x256 = (long)f5_cdiv(a.M, 128) * (a.N / 256);
            if (t128x256 <= 256) sel = f5_gemm_qkv_small_tile;
        }
    }
    if (sel == 12 || sel == 13) {
        if constexpr (EPI == EPI_QKV_ROPE || EPI == EPI_BF16 || EPI == EPI_GELU_TANH) {
            if (a.N % 256 == 0)
                return sel == 12 ? launch_ring_wide<EPI, 2, 2, 2, 4>(a, stream) : launch_ring_wide<EPI, 1, 4, 4, 2>(a, stream);
        }
        sel = 0;
    }
    if (sel == 10) return launch_ring_ks2<EPI, 1>(a, stream);
    if (sel == 11) return launch_ring_ks2<EPI, 2>(a, stream);
    if (sel == 8 || sel == 9) {
        const int bn = sel == 8 ? 192 : 128;
        if (a.N % bn == 0) return sel == 8 ? launch_ring8<EPI, 3>(a, stream) : launch_ring8<EPI, 2>(a, stream);
        sel = 0;
    }
    if (sel == 0 && f5_gemm_ring_default) {
        // one round of 8-wave workgroups (measured at M = 937 / 1874, tools/ring8_bench.py): 128x128 tiles when they fill
        // 70-100 % of the CUs (FF1 at batch 1: 15.0 vs 17.5 us), else 64x128 tiles with the K tiles split over two wave
        // groups (out-proj 12.4 vs 13.4 us, FF2 18.2 vs 20-21 us)
        if (a.N % 128 == 0 && t128 >= 176 && t128 <= 256) return launch_ring8<EPI, 2>(a, stream);
        if (t64x128 >= 176 && t64x128 <= 256) return launch_ring_ks2<EPI, 1>(a, stream);
    }
    if (sel == 0 || sel == 4 || sel == 7) sel = t128 >= 384 ? 1 : (t64x128 >= 384 ? 2 : 3);
    if (EPI == EPI_QKV_ROPE && sel == 3) sel = 2;
    if (EPI == EPI_QKV_ROPE && sel == 6) sel = 5;  // the V^T / head mapping wants >= one whole head per tile column
    if (sel == 5) return launch_ring<EPI, 1, 2>(a, stream);
    if (sel == 6) return launch_ring<EPI, 1, 1>(a, stream);
    if (sel == 1) return launch_cfg<EPI, 2, 2>(a, stream);
    if (f5_gemm_ring_default) {
        // the ring kernels hold 2 workgroups per CU (512 slots); the register-staged 64x128 kernel needs only 48 KB of
        // LDS (3 per CU, 768 slots): prefer it when that turns two rounds of tiles into one (QKV at M = 2*937: 720 tiles)
        if (sel == 2 && t64x128 > 512 && t64x128 <= 768) return launch_cfg<EPI, 1, 2>(a, stream);
        return sel == 2 ? launch_ring<EPI, 1, 2>(a, stream) : launch_ring<EPI, 1, 1>(a, stream);
    }
    if (sel == 2) return launch_cfg<EPI, 1, 2>(a, stream);
    return launch_cfg<EPI, 1, 1>(a, stream);
}

int f5_launch_gemm(const F5GemmArgs& a_in, int epi, hipStream_t stream) {
    F5GemmArgs a = a_in;
    a.debug_flags = f5_gemm_debug_flags;
    F5_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % BK == 0, "gemm: bad shape M=%d N=%d K=%d (K must be a multiple of %d)",
               a.M, a.N, a.K, BK);
    F5_REQUIRE(a.nseg == 1 || a.nseg == 3, "gemm: nseg must be 1 or 3");
    F5_REQUIRE(a.A[0] && a.W[0], "gemm: null operand");
    F5_REQUIRE(a.nseg == 1 || (a.A[1] && a.W[1]), "gemm: bf16x3 needs lo operands");
    F5_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm: leading dims must be multiples of 8");
    F5_REQUIRE((size_t)(a.a_row_mod > 0 ? a.a_row_mod : a.M) * a.lda < (1ull << 31) && (size_t)(a.N + 256) * a.ldw < (1ull << 31),
               "gemm: operands must stay below 4 GiB (the kernels use 32-bit byte offsets)");
    F5_REQUIRE(epi != EPI_RESID_GATE || (size_t)a.M * a.ldo * 4 < (1ull << 32), "gemm(resid): the residual stream must stay below 4 GiB");
    switch (epi) {
        case EPI_F32: return launch_epi<EPI_F32>(a, stream);
        case EPI_BF16: return launch_epi<EPI_BF16>(a, stream);
        case EPI_GELU_TANH: return launch_epi<EPI_GELU_TANH>(a, stream);
        case EPI_GELU_ERF: return launch_epi<EPI_GELU_ERF>(a, stream);
        case EPI_RESID_GATE: return launch_epi<EPI_RESID_GATE>(a, stream);
        case EPI_QKV_ROPE:
            F5_REQUIRE(a.dmodel % 128 == 0 && a.N == 3 * a.dmodel, "gemm(qkv): N must be 3*dmodel, dmodel %% 128 == 0");
            F5_REQUIRE(a.rope_cos && a.rope_sin, "gemm(qkv): token-major rotation tables missing");
            if (a.rope_cos_tk || a.rope_cos_tq) {
                // pair-major tables = transposed q / k tiles on the 256x256 kernel: every 256-column tile must lie inside one of the
                // q | k | v ranges, the bias is read as 16-byte quads; gemm flag 16384 = A/B against the straight tiles
                F5_REQUIRE(a.rope_cos_tq && a.rope_sin_tq && a.rope_cos_tk && a.rope_sin_tk && a.rope_ldt >= a.seq_len,
                           "gemm(qkv): pair-major rotation tables incomplete");
                const bool ok = a.dmodel % 256 == 0 && (a.bias == nullptr || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
                                (f5_gemm_debug_flags & 16384) == 0;
                if (!ok) a.rope_cos_tq = a.rope_sin_tq = a.rope_cos_tk = a.rope_sin_tk = nullptr;
            }
            return launch_epi<EPI_QKV_ROPE>(a, stream);
        case EPI_ADDROWS: return launch_epi<EPI_ADDROWS>(a, stream);
        case EPI_RESID_KEEP: return launch_epi<EPI_RESID_KEEP>(a, stream);
        case EPI_GELU_ERF_BF16: return launch_epi<EPI_GELU_ERF_BF16>(a, stream);
        default: f5_set_error("gemm: unknown epilogue %d", epi); return 2;
    }
}
}  // namespace F5_NS
