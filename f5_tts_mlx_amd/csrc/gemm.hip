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
#include "rowops.hpp"     // f5_sat_flag_host
#include "lnrow.hpp"
#ifndef F5_LAB
#define F5_LAB 0
#endif
#if F5_LAB
#include "gemm_lab_dev.hpp"
#endif

namespace F5_NS {



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
#if F5_LAB
    if (EPI == EPI_RESID_GATE && p.ln_counter == nullptr && (p.debug_flags & 8) != 0) {    // experiment: residual update by L2 atomics
        atomic_epilogue_resid<MB, NB>(p, acc, m0 + wm * (32 * MB), n0 + wn * (32 * NB), lane);
        return;
    }
#endif
    gemm_epilogue<EPI, MB, NB>(p, acc, m0, n0, wm, wn, lane);
    if (EPI == EPI_RESID_GATE && p.ln_counter) resid_ln_tail(p, tm, BMt, (p.N + BNt - 1) / BNt);
}

// =================================================================================================
// v1r: the small-tile kernel with global_load_lds staging into a ring of NST K-tiles (NST-1 tiles in flight,
// counted vmcnt, one barrier per K tile).  Used for small-batch shapes (M ~ 2 * 937 rows) where the chip is only
// filled by 64x64 / 64x128 tiles and the old one-tile-deep register prefetch left every iteration latency bound.
// =================================================================================================
int f5_gemm_order = 0;   // 0 auto, 1 force n-fastest, 2 force m-fastest, 3 force band-major where it applies (ring kernel tile numbering)
static bool gemm_mfast(const F5GemmArgs& a) {
    if (f5_gemm_order == 1) return false;
    if (f5_gemm_order == 2) return true;
    const size_t a_bytes = (size_t)(a.a_row_mod > 0 ? a.a_row_mod : a.M) * a.K * 2 * (a.nseg == 3 ? 2 : 1);
    return a_bytes <= (size_t)4 << 20;     // whole A operand fits in one XCD's 4 MB L2
}
// the ring kernels' tile-order argument: tiles_n (n fastest), -tiles_m (m fastest) or tiles_n | band << 20 (band-major, see the kernel).
// Band-major is the automatic choice for launches of at most two rounds whose column tiles split into bands of 4: every L2 then fetches
// a compact block (fabric bytes per launch at batch 1, PMC: profiles/pmc_traffic.json; op-level A/B of the four orders:
// profiles/r04/ring_order_ab.json: out-proj 11.8 m-fastest / 10.7 n-fastest / 10.7-10.8 band-major us, FF1 14.7 / 14.8 / 14.4, FF2 17.5 /
// 17.0 / 16.9).
static int ring_order(const F5GemmArgs& a, int tiles_m, int tiles_n) {
    const bool band_ok = tiles_n % 4 == 0 && tiles_n >= 8 && (long)tiles_m * tiles_n <= 512 && tiles_m >= 8;
    if ((f5_gemm_order == 3 || f5_gemm_order == 0) && band_ok) return tiles_n | (4 << 20);
    if (f5_gemm_order == 3) return tiles_n;
    return gemm_mfast(a) ? -tiles_m : tiles_n;
}

// ABL (timing experiments only, results are garbage; tools/ring_ablate.py): 1 = no operand loads after the prologue, 2 = no MFMAs,
// 4 = no LDS fragment reads, 8 = no workgroup barrier -- what a K step of a lone workgroup is made of
// FOLD (round 6, batch-1-sized launches of the LN fold, gemm.hpp): 1 = EPI_RESID_GATE writes the folded operand and its slice statistics
// (staged_epilogue_resid with its loads requested before the K loop), 2 = EPI_GELU_TANH is a fold consumer that merges the producer's statistics itself (fold_stats), on
// transposed wave tiles like the large kernels
template <int EPI, int MB, int NB, int NST, int WM = 2, int WN = 2, int KS = 1, int ABL = 0, int FOLD = 0>
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

    // the argument-block fields of the prologue in ONE scalar-load clause (left alone the compiler loads each field where it is
    // first used: three dependent s_load / s_waitcnt rounds before the first tile request of a one-round launch)
    asm volatile("" ::"s"(p.A[0]), "s"(p.W[0]), "s"(p.lda), "s"(p.ldw), "s"(p.K), "s"(p.nseg), "s"(p.M), "s"(p.N), "s"(p.a_row_mod),
                 "s"(p.debug_flags), "s"(tiles_n), "s"(ntiles));
    if (EPI == EPI_RESID_GATE)                          // (the residual operands are requested before the K loop)
        asm volatile("" ::"s"(p.bias), "s"(p.out_f32), "s"(p.ldo), "s"(p.gate), "s"(p.rowkeep), "s"(p.ln_counter));
    const int bid = blockIdx.x;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // tile numbering: n fastest (neighbouring tiles share the A panel); when tiles_n < 0, m fastest with tiles_m = -tiles_n
    // (neighbouring tiles share the W panel: better when the whole A operand fits in an XCD's L2); with a band width in bits 20+
    // (ring_order), BAND-major: bands of `band` column tiles, n fastest inside a band -- an XCD's contiguous chunk of the list is then a
    // compact block of tiles (batch-1 out-projection, 30 x 8 tiles: 7.5 rows x 4 columns per XCD = a quarter of A and half of W per L2
    // instead of all of A and an eighth of W)
    int tm, tn;
    if (tiles_n >= (1 << 20)) {
        const int band = tiles_n >> 20, tnn = tiles_n & ((1 << 20) - 1);
        const int per_band = band * (ntiles / tnn);
        const int b = tile / per_band, rem = tile - b * per_band;
        tm = rem / band;
        tn = b * band + (rem - tm * band);
    } else if (tiles_n > 0) {
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
    if (PRE_RESID && use_pre && grp == 0 && FOLD != 1)
        resid_preload<PRE_RESID ? MB : 1, PRE_RESID ? NB : 1>(p, rpre, m0 + wm * (32 * MB), n0 + wn * (32 * NB), lane);
    static_assert(FOLD == 0 || (FOLD == 1 && PRE_RESID && NB == 2) || (FOLD == 2 && EPI == EPI_GELU_TANH && NB == 2 && KS == 1 && WM * WN == 8),
                  "fold producer = the preloaded residual epilogue on 64-column wave tiles; fold consumer = the 8-wave GELU launch");
    // fold producer: the LDS-staged residual epilogue of the large kernels (16-byte accesses in full row segments, slice statistics by
    // 16-lane DPP reductions), with its rows of x / keep bytes / shifts / column vectors requested here, before the K loop (a straight-tile
    // variant -- lane = column, a 32-lane butterfly per row -- measured +3.7 us per launch against +1.6 for this one, profiles/r06)
    ResidStagedPre<FOLD == 1 ? NB : 1> spre_p;
    if constexpr (FOLD == 1) {
        if (grp == 0) resid_staged_preload<NB>(p, spre_p, m0 + wm * (32 * MB), n0 + wn * (32 * NB), lane);
    }
    // fold consumer: the slice statistics of this lane's rows and its c1 | c2 columns -- older than every operand load, so the first
    // counted wait of the K loop covers them too
    FoldPre fpre_c;
    fold_prefetch_clear(fpre_c);
    FoldStatsPre<8> spre_c;
    fold_stats_clear(spre_c);
    if constexpr (FOLD == 2) fold_stats_request_tr<1>(p, spre_c, fpre_c, m0 + wm * 32, n0 + wn * 64, lane);

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
    // tiles (an LDS-free variant of this path returned wrong values on lanes 48-63 there in round 2; round 3 traced that to
    // SLP-packed f32 instructions next to other workgroups' MFMAs -- profiles/r03/pk_f32_next_to_mfma_hazard.txt -- which is why
    // this file is built with -fno-slp-vectorize; the 4-wave kernels gain nothing from transposed tiles at the sizes they serve).
    if constexpr (FOLD == 2) {
        // the statistics and c1 | c2 have landed with the first K tile (they are older): merge them into this lane's row factor now,
        // while the other tiles are in flight
        if (NST == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fold_stats_pin(spre_c);
        fold_stats_finish_tr<1>(p, spre_c, fpre_c, m0 + wm * 32, lane, p.fold_mean_out != nullptr && n0 == 0 && wn == 0);
        fold_prefetch_pin(fpre_c);
        for (int jj = 0; jj < nit; jj += NST) {
            RING_STEP(0, jj, true);
            if (NST > 1 && jj + 1 < nit) RING_STEP(1 % NST, jj + 1, true);
            if (NST > 2 && jj + 2 < nit) RING_STEP(2 % NST, jj + 2, true);
            if (NST > 3 && jj + 3 < nit) RING_STEP(3 % NST, jj + 3, true);
        }
        __syncthreads();                                  // the ring is dead: its LDS becomes the staging area
        static_assert(FOLD != 2 || WM * WN * (small_tile_stage_elems<NB>() * 2 + 512) <= RING * 2, "staging + c1 | c2 scratch fit in the ring");
        float* fl = reinterpret_cast<float*>(smem_all + WM * WN * small_tile_stage_elems<NB>()) + wave * 128;
        staged_epilogue_tr<EPI, MB, NB, true>(p, acc, smem_all + wave * small_tile_stage_elems<NB>(), m0 + wm * (32 * MB), n0 + wn * (32 * NB), lane,
                                              fl, &fpre_c);
        return;
    }
    constexpr bool QKV_TR_OK = EPI == EPI_QKV_ROPE && (32 * NB) % 64 == 0 && (NB & (NB - 1)) == 0 && KS == 1 && WM * WN == 8;
    if (QKV_TR_OK && p.rope_g4k != nullptr && n0 + wn * (32 * NB) < 2 * p.dmodel) {
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
#if F5_LAB
    if (EPI == EPI_RESID_GATE && p.ln_counter == nullptr && (p.debug_flags & 8) != 0) {    // experiment: residual update by L2 atomics
        atomic_epilogue_resid<MB, NB>(p, acc, m0 + wm * (32 * MB), n0 + wn * (32 * NB), lane);
        return;
    }
#endif
    if constexpr (FOLD == 1) {
        // staging area behind the K-split reduction area (other waves of the group may still be reading theirs)
        constexpr int RED_BYTES = (KS - 1) * WM * WN * MB * NB * 16 * 64 * 4;
        static_assert(RED_BYTES + WM * WN * 32 * (32 * NB + 4) * 4 <= KS * RING * 2, "staging behind the reduction area");
        float* stg = reinterpret_cast<float*>(reinterpret_cast<char*>(smem_all) + RED_BYTES) + wave * (32 * (32 * NB + 4));
        staged_epilogue_resid<MB, NB, true>(p, acc, stg, m0 + wm * (32 * MB), n0 + wn * (32 * NB), lane, &spre_p);
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
    const int order = ring_order(a, tiles_m, tiles_n);
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
    const int order = ring_order(a, tiles_m, tiles_n);
    constexpr int NST = MB == 1 ? 3 : 2;
    if constexpr (EPI == EPI_RESID_GATE && MB == 1) {
        if (a.x16_out != nullptr) {                       // LN-fold producer (f5_gemm_fold_small has checked the preconditions)
            hipLaunchKernelGGL((f5_gemm_ring_kernel<EPI, MB, 2, NST, 2, 2, 2, 0, 1>), dim3(ntiles), dim3(512), 0, stream, a, order, ntiles);
            F5_LAUNCH_CHECK();
            return 0;
        }
    }
    hipLaunchKernelGGL((f5_gemm_ring_kernel<EPI, MB, 2, NST, 2, 2, 2>), dim3(ntiles), dim3(512), 0, stream, a, order, ntiles);
    F5_LAUNCH_CHECK();
    return 0;
}

template <int EPI, int NB>
static int launch_ring8(const F5GemmArgs& a, hipStream_t stream) {
    constexpr int BMt = 128, BNt = 64 * NB;
    const int tiles_m = f5_cdiv(a.M, BMt), tiles_n = a.N / BNt;
    const int ntiles = tiles_m * tiles_n;
    const int order = ring_order(a, tiles_m, tiles_n);
    constexpr int NST = (BMt + BNt) * 64 * 2 * 4 <= 128 * 1024 ? 4 : 3;
    if constexpr (EPI == EPI_GELU_TANH && NB == 2) {
        if (a.fold_stats != nullptr) {                    // LN-fold consumer, statistics form (f5_gemm_fold_small has checked the preconditions)
            hipLaunchKernelGGL((f5_gemm_ring_kernel<EPI, 1, NB, NST, 4, 2, 1, 0, 2>), dim3(ntiles), dim3(512), 0, stream, a, order, ntiles);
            F5_LAUNCH_CHECK();
            return 0;
        }
    }
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
    const int order = ring_order(a, tiles_m, tiles_n);
    hipLaunchKernelGGL((f5_gemm_ring_kernel<EPI, MB, NB, 3, WM, WN, KS, ABL>), dim3(ntiles), dim3(64 * WM * WN * KS), 0, stream, a, order,
                       ntiles);
    F5_LAUNCH_CHECK();
    return 0;
}
#if F5_LAB
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
#endif  // F5_LAB

template <int EPI, int MB, int NB>
static int launch_cfg(const F5GemmArgs& a, hipStream_t stream) {
    const int tiles_m = f5_cdiv(a.M, 64 * MB), tiles_n = f5_cdiv(a.N, 64 * NB);
    const int ntiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((f5_gemm_kernel<EPI, MB, NB>), dim3(ntiles), dim3(256), 0, stream, a, tiles_n, ntiles);
    F5_LAUNCH_CHECK();
    return 0;
}

// tile shape: the largest of 128x128 / 64x128 / 64x64 that still gives the 256 CUs >= 1.5 workgroups each
// (small-batch shapes such as M = 1874 are otherwise a fraction of one wave of tiles)
int f5_gemm_tile_override = 0;  // 0 auto, 1 = 128x128, 2 = 64x128, 3 = 64x64, 4 = 256x256 v2, 5 = 64x128 ring, 6 = 64x64 ring,
                                // 7 = 128x256 v3, 8 = 128x192 8-wave ring, 9 = 128x128 8-wave ring, 10 / 11 = 64x128 / 128x128 split-K ring
int f5_gemm_debug_flags = 0;
#if F5_LAB
int f5_gemm_big_kernel = 2;       // large shapes: 2 = 256x256 role-split schedule (gemm256.hip, the product kernel), 3 = 128x256 v3 (2 WG/CU),
                                  // 4 = 256x256 lock-step (rounds 1-2), 5 = 128x256 with in-wave fragment prefetch (gemm128.hip)
extern int f5_gemm_streamk;
int f5_launch_gemm_lab_v2(const F5GemmArgs& a, int epi, hipStream_t stream);   // gemm_lab.hip
int f5_launch_gemm_lab_v3(const F5GemmArgs& a, int epi, hipStream_t stream);
#endif
int f5_gemm_qkv_small_tile = 0;   // small-M QKV projection with pair-major tables: 0 = auto tiles, 12 / 13 = 8-wave 128x256 ring, transposed q / k
int f5_gemm_ring_default = 1;   // auto mode: small tiles use the global_load_lds ring kernel
// the large-shape kernels (256x256, 128x256) have no fused LN tail (at those sizes LN-modulate is HBM-bound, not launch-bound)
static bool gemm_uses_big_kernel(const F5GemmArgs& a) {
    const long t256 = (long)f5_cdiv(a.M, 256) * (a.N / 256);
    const bool v2ok = (a.N % 256 == 0) && (a.M >= 256);
    const int sel = f5_gemm_tile_override;
    return sel == 7 || sel == 4 || (sel == 0 && v2ok && t256 >= 512);     // (7 = the lab build's 128x256 kernel)
}
bool f5_gemm_resid_ln_fusable(const F5GemmArgs& a) {
    return !gemm_uses_big_kernel(a) && a.N % 256 == 0 && a.N >= 256 && a.N <= 1024 && a.ldo == a.N && a.M <= 64 * 65536;
}

// mirrors launch_epi's two staged routes (a drift makes f5_launch_gemm fail loudly, never compute something else)
bool f5_gemm_runs_staged(const F5GemmArgs& a, int epi) {
    if (!(epi == EPI_RESID_GATE || epi == EPI_QKV_ROPE || epi == EPI_GELU_TANH) || a.ln_counter != nullptr || a.N % 256 != 0) return false;
    const int sel = f5_gemm_tile_override;
#if F5_LAB
    if (sel == 7 || f5_gemm_big_kernel != 2 || f5_gemm_streamk) return false;
#endif
    const long t256 = (long)f5_cdiv(a.M, 256) * (a.N / 256);
    const long t128 = (long)f5_cdiv(a.M, 128) * f5_cdiv(a.N, 128);
    if (sel == 4 || (sel == 0 && a.M >= 256 && t256 >= 512)) return a.M >= 256;
    const bool qkv_rows_ok = epi != EPI_QKV_ROPE || (a.seq_len > 0 && a.M % a.seq_len == 0);
    return qkv_rows_ok && (sel == 14 || (sel == 0 && t128 >= 384));
}

// The batch-1-sized route of the LN fold (round 6): true when f5_launch_gemm sends this launch to the single-round kernel that implements
// the fold for its role -- EPI_RESID_GATE: the 64 x 128 split-K ring kernel with the preloaded residual epilogue (producer: x16_out /
// stats_out), EPI_GELU_TANH: the 8-wave 128 x 128 ring kernel (consumer, statistics form only: fold_stats), EPI_QKV_ROPE: one round of
// role-split 128 x 256 tiles (consumer; `qkv_tr` = the group-major rotation tables will be set).  Same rules as launch_epi below.
bool f5_gemm_fold_small(const F5GemmArgs& a, int epi, bool qkv_tr) {
    if (f5_gemm_tile_override != 0 || !f5_gemm_ring_default || a.ln_counter != nullptr || a.nseg != 1 || a.N % 256 != 0 || a.M < 1) return false;
    const long t256 = (long)f5_cdiv(a.M, 256) * (a.N / 256);
    const long t128 = (long)f5_cdiv(a.M, 128) * f5_cdiv(a.N, 128);
    const long t64x128 = (long)f5_cdiv(a.M, 64) * f5_cdiv(a.N, 128);
    if ((a.M >= 256 && t256 >= 512) || t128 >= 384) return false;                 // the staged multi-round kernels take these
    if (epi == EPI_QKV_ROPE) {
        if (!qkv_tr || a.seq_len <= 0 || a.M % a.seq_len != 0 || f5_gemm_qkv_small_tile != 0) return false;
        const long t = (long)(a.M / a.seq_len) * f5_cdiv(a.seq_len, 128) * (a.N / 256);
        return t >= 176 && t <= 256;
    }
    if (epi == EPI_GELU_TANH) return t128 >= 176 && t128 <= 256;
    if (epi == EPI_RESID_GATE) return !(t128 >= 176 && t128 <= 256) && t64x128 >= 176 && t64x128 <= 256 && (a.debug_flags & (8 | 256)) == 0;
    return false;
}

// ---- constants of the LN fold (gemm.hpp): one wave holds FC_ROWS weight rows in registers as fp32 and streams every modulation
// vector past them (the vectors are L2-resident: nvec x 2 x K floats)
constexpr int FC_ROWS = 4, FC_MAXC = 8;
int f5_fold_consts_mfma = 1;     // 0 = the VALU kernel (A/B)
__global__ __launch_bounds__(256) void f5_fold_consts_kernel(const op16_t* __restrict__ w, int ldw, const float* __restrict__ bias,
                                                             const float* __restrict__ scale, const float* __restrict__ shift, size_t vec_stride,
                                                             int nvec, float* __restrict__ c1, float* __restrict__ c2, size_t out_stride, int N,
                                                             int K, F5FoldBatch bt) {
    // blockIdx.y = one of bt.count equally shaped problems (the same projection of every DiT block) at uniform strides
    w += (size_t)blockIdx.y * bt.w_stride;
    if (bias) bias += (size_t)blockIdx.y * bt.bias_stride;
    scale += (size_t)blockIdx.y * bt.mod_stride;
    shift += (size_t)blockIdx.y * bt.mod_stride;
    c1 += (size_t)blockIdx.y * bt.out_blk_stride;
    c2 += (size_t)blockIdx.y * bt.out_blk_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * FC_ROWS;
    if (n0 >= N) return;
    const int nchunk = K >> 8;                       // 256 columns per chunk, 4 per lane
    float wv[FC_ROWS][FC_MAXC][4];
#pragma unroll
    for (int r = 0; r < FC_ROWS; ++r) {
        const int n = n0 + r < N ? n0 + r : N - 1;
#pragma unroll
        for (int c = 0; c < FC_MAXC; ++c) {
            if (c < nchunk) {
                const op16x4 t = *reinterpret_cast<const op16x4*>(w + (size_t)n * ldw + c * 256 + lane * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) wv[r][c][e] = f5_op2f(t[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) wv[r][c][e] = 0.0f;
            }
        }
    }
    for (int v = 0; v < nvec; ++v) {
        const float* sv = scale + (size_t)v * vec_stride;
        const float* bv = shift + (size_t)v * vec_stride;
        float a1[FC_ROWS], a2[FC_ROWS];
#pragma unroll
        for (int r = 0; r < FC_ROWS; ++r) a1[r] = a2[r] = 0.0f;
#pragma unroll
        for (int c = 0; c < FC_MAXC; ++c) {
            if (c < nchunk) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sv + c * 256 + lane * 4);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bv + c * 256 + lane * 4);
#pragma unroll
                for (int r = 0; r < FC_ROWS; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a1[r] += wv[r][c][e] * (1.0f + s4[e]);
                        a2[r] += wv[r][c][e] * b4[e];
                    }
            }
        }
#pragma unroll
        for (int r = 0; r < FC_ROWS; ++r) {
            a1[r] = f5_wave_sum(a1[r]);
            a2[r] = f5_wave_sum(a2[r]);
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < FC_ROWS; ++r)
                if (n0 + r < N) {
                    c1[(size_t)v * out_stride + n0 + r] = a1[r];
                    c2[(size_t)v * out_stride + n0 + r] = a2[r] + (bias ? bias[n0 + r] : 0.0f);
                }
        }
    }
}

// The same constants on the matrix cores (round 6).  The VALU kernel above spends 48 cross-lane operations per modulation vector and four
// weight rows: 455 us per projection for the 31 evaluations x 22 blocks of a 32-point solve -- nothing at batch 32, 1.3 % of a batch-1
// sample().  Here a wave owns 32 output columns and all (<= 128) modulation vectors: C[v][n] = sum_k S[v][k] W[n][k] as
// v_mfma_f32_32x32x16 with W straight from memory (its rows ARE the operand: 8 consecutive k per lane, 16-byte loads) and S split on
// the fly into an operand-typed (hi, lo) pair, S ~ hi + lo (fp16: 22 significand bits, fp32-class; bf16: 16), two MFMAs per product.
// Every weight row is read once per launch (10.5 MB per block), the vectors 32x less often than before.
template <int NVB>
__global__ __launch_bounds__(256) void f5_fold_consts_mfma_kernel(const op16_t* __restrict__ w, int ldw, const float* __restrict__ bias,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  size_t vec_stride, int nvec, float* __restrict__ c1, float* __restrict__ c2,
                                                                  size_t out_stride, int N, int K, F5FoldBatch bt) {
    w += (size_t)blockIdx.y * bt.w_stride;
    if (bias) bias += (size_t)blockIdx.y * bt.bias_stride;
    scale += (size_t)blockIdx.y * bt.mod_stride;
    shift += (size_t)blockIdx.y * bt.mod_stride;
    c1 += (size_t)blockIdx.y * bt.out_blk_stride;
    c2 += (size_t)blockIdx.y * bt.out_blk_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * 32;
    if (n0 >= N) return;
    const int lr = lane & 31, kh = (lane >> 5) * 8;
    const int nrow = n0 + lr < N ? n0 + lr : N - 1;
    const op16_t* wp = w + (size_t)nrow * ldw + kh;
    const float* sp[NVB];
    const float* bp[NVB];
#pragma unroll
    for (int vb = 0; vb < NVB; ++vb) {
        const int v = vb * 32 + lr < nvec ? vb * 32 + lr : nvec - 1;
        sp[vb] = scale + (size_t)v * vec_stride + kh;
        bp[vb] = shift + (size_t)v * vec_stride + kh;
    }
    f32x16 a1[NVB], a2[NVB];
#pragma unroll
    for (int vb = 0; vb < NVB; ++vb)
#pragma unroll
        for (int e = 0; e < 16; ++e) a1[vb][e] = a2[vb][e] = 0.0f;
    auto split = [](const f32x4& lo4, const f32x4& hi4, float add, op16x8& h, op16x8& l) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (e < 4 ? lo4[e] : hi4[e - 4]) + add;
            const op16_t hh = f5_f2op(f);
            h[e] = hh;
            l[e] = static_cast<op16_t>(f - static_cast<float>(hh));
        }
    };
    for (int k0 = 0; k0 < K; k0 += 16) {
        const op16x8 wf = *reinterpret_cast<const op16x8*>(wp + k0);
#pragma unroll
        for (int vb = 0; vb < NVB; ++vb) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp[vb] + k0), s1 = *reinterpret_cast<const f32x4*>(sp[vb] + k0 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp[vb] + k0), b1 = *reinterpret_cast<const f32x4*>(bp[vb] + k0 + 4);
            op16x8 sh, sl, bh, bl;
            split(s0, s1, 1.0f, sh, sl);
            split(b0, b1, 0.0f, bh, bl);
            a1[vb] = F5_MFMA32(sh, wf, a1[vb], 0, 0, 0);
            a1[vb] = F5_MFMA32(sl, wf, a1[vb], 0, 0, 0);
            a2[vb] = F5_MFMA32(bh, wf, a2[vb], 0, 0, 0);
            a2[vb] = F5_MFMA32(bl, wf, a2[vb], 0, 0, 0);
        }
    }
    // C layout: lane = column n0 + lr, register r = vector vb * 32 + 8 (r >> 2) + 4 (lane >> 5) + (r & 3)
    const int n = n0 + lr;
    if (n < N) {
        const float bn = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int vb = 0; vb < NVB; ++vb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int v = vb * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (v < nvec) {
                    c1[(size_t)v * out_stride + n] = a1[vb][r];
                    c2[(size_t)v * out_stride + n] = a2[vb][r] + bn;
                }
            }
    }
}

// Slice statistics -> row factors.  A slice carries (sum d, sum (d - its own mean)^2) of 64 values d = x - m; the row's statistics are
// the group form of Chan's merge -- mean = sum of the slice sums / n, M2 = sum of the slice M2 + 64 sum (slice mean - mean)^2: every
// term is a sum of squares, nothing of the size of mean^2 is ever subtracted (the round-4 kernel computed E[x^2] - E[x]^2 from one-pass
// sums).  NS = 16 (dim 1024): all slice loads of a row are independent and issued together.
template <int NS>
__global__ __launch_bounds__(256) void f5_fold_rows_kernel(const float* __restrict__ stats, int ld, int nslice, int M, float eps,
                                                           float* __restrict__ rowf, float* __restrict__ row_shift) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const f5_f32x2* sp = reinterpret_cast<const f5_f32x2*>(stats) + m;
    const float shift0 = row_shift != nullptr ? row_shift[m] : 0.0f;
    float mean, m2 = 0.0f;
    const float inv_n = 1.0f / (64.0f * (float)nslice);
    if (NS > 0) {
        f5_f32x2 t[NS > 0 ? NS : 1];
#pragma unroll
        for (int i = 0; i < NS; ++i) t[i] = sp[(size_t)i * ld];
        // (slices 0 .. NS/2 - 1 in order) + (slices NS/2 .. in order): the order the consumers use when they merge the statistics
        // themselves (gemm_dev.hpp fold_stats_finish_*: two lane halves / two in-lane chains) -- one bit pattern per row either way
        float s_lo = t[0][0], s_hi = t[NS / 2][0];
#pragma unroll
        for (int i = 1; i < NS / 2; ++i) {
            s_lo += t[i][0];
            s_hi += t[NS / 2 + i][0];
        }
        mean = (s_lo + s_hi) * inv_n;
        float m_lo = 0.0f, m_hi = 0.0f;
#pragma unroll
        for (int i = 0; i < NS / 2; ++i) {
            const float d0 = t[i][0] * (1.0f / 64.0f) - mean, d1 = t[NS / 2 + i][0] * (1.0f / 64.0f) - mean;
            m_lo += t[i][1] + 64.0f * d0 * d0;
            m_hi += t[NS / 2 + i][1] + 64.0f * d1 * d1;
        }
        m2 = m_lo + m_hi;
    } else {
        float s = 0.0f;
        for (int i = 0; i < nslice; ++i) s += sp[(size_t)i * ld][0];
        mean = s * inv_n;
        for (int i = 0; i < nslice; ++i) {
            const f5_f32x2 t = sp[(size_t)i * ld];
            const float dm = t[0] * (1.0f / 64.0f) - mean;
            m2 += t[1] + 64.0f * dm * dm;
        }
    }
    const float rstd = rsqrtf(m2 * inv_n + eps);
    reinterpret_cast<f5_f32x2*>(rowf)[m] = f5_f32x2{rstd, rstd * mean};
    if (row_shift != nullptr) row_shift[m] = shift0 + mean;
}
int f5_launch_fold_rows(const float* stats, int ld, int nslice, int M, float eps, float* rowf, float* row_shift, hipStream_t stream) {
    F5_REQUIRE(stats && rowf && nslice >= 1 && M >= 1 && ld >= M, "fold_rows: bad arguments");
    if (nslice == 16) hipLaunchKernelGGL(f5_fold_rows_kernel<16>, dim3(f5_cdiv(M, 256)), dim3(256), 0, stream, stats, ld, nslice, M, eps, rowf, row_shift);
    else hipLaunchKernelGGL(f5_fold_rows_kernel<0>, dim3(f5_cdiv(M, 256)), dim3(256), 0, stream, stats, ld, nslice, M, eps, rowf, row_shift);
    F5_LAUNCH_CHECK();
    return 0;
}

int f5_launch_fold_consts(const op16_t* w, int ldw, const float* bias, const float* scale, const float* shift, size_t vec_stride, int nvec,
                          float* c1, float* c2, size_t out_stride, int N, int K, hipStream_t stream, const F5FoldBatch* batch) {
    F5FoldBatch bt = {1, 0, 0, 0, 0};
    if (batch) bt = *batch;
    F5_REQUIRE(bt.count >= 1 && bt.count <= 65535 && bt.mod_stride % 4 == 0 && bt.w_stride % 4 == 0, "fold_consts: bad batch");
    F5_REQUIRE(w && scale && shift && c1 && c2 && N > 0 && nvec > 0, "fold_consts: null argument");
    F5_REQUIRE(K % 256 == 0 && K <= 256 * FC_MAXC && ldw % 4 == 0 && vec_stride % 4 == 0, "fold_consts: K must be a multiple of 256, <= %d",
               256 * FC_MAXC);
    F5_REQUIRE(((reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 7) == 0,
               "fold_consts: unaligned operand");
    if (K % 16 == 0 && ldw % 8 == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 && nvec <= 128 && f5_fold_consts_mfma) {
        const dim3 grid(f5_cdiv(N, 4 * 32), bt.count);
#define FC_LAUNCH(NVB_) hipLaunchKernelGGL(f5_fold_consts_mfma_kernel<NVB_>, grid, dim3(256), 0, stream, w, ldw, bias, scale, shift, vec_stride, nvec, c1, c2, out_stride, N, K, bt)
        switch ((nvec + 31) / 32) {
            case 1: FC_LAUNCH(1); break;
            case 2: FC_LAUNCH(2); break;
            case 3: FC_LAUNCH(3); break;
            default: FC_LAUNCH(4); break;
        }
#undef FC_LAUNCH
        F5_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(f5_fold_consts_kernel, dim3(f5_cdiv(N, 4 * FC_ROWS), bt.count), dim3(256), 0, stream, w, ldw, bias, scale, shift,
                       vec_stride, nvec, c1, c2, out_stride, N, K, bt);
    F5_LAUNCH_CHECK();
    return 0;
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
#if F5_LAB
    if (sel == 7 || (sel == 0 && v2ok && t256 >= 512 && f5_gemm_big_kernel == 3)) {
        F5_REQUIRE(v2ok && a.K % 32 == 0, "gemm: the 128x256 kernel needs N %% 256 == 0 and M >= 256");
        return f5_launch_gemm_lab_v3(a, EPI, stream);
    }
#endif
    if (sel == 4 || (sel == 0 && v2ok && t256 >= 512)) {
        F5_REQUIRE(v2ok, "gemm: the 256x256 kernel needs N %% 256 == 0 and M >= 256");
#if F5_LAB
        F5_REQUIRE((a.x16_out == nullptr && a.fold_rowf == nullptr) || (f5_gemm_big_kernel == 2 && !f5_gemm_streamk),
                   "gemm: the LN fold needs the product 256x256 kernel");
        if (f5_gemm_big_kernel == 4 || f5_gemm_streamk) return f5_launch_gemm_lab_v2(a, EPI, stream);      // lock-step predecessor (A/B)
        if constexpr (EPI == EPI_F32 || EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_RESID_GATE || EPI == EPI_QKV_ROPE) {
            if (f5_gemm_big_kernel == 5) return f5_launch_gemm128(a, EPI, stream);              // 128x256, two workgroups per CU
        }
#endif
        return f5_launch_gemm256(a, EPI, stream);
    }
#if F5_LAB
    if constexpr (EPI == EPI_BF16) {
        const int abl = (a.debug_flags >> 4) & 15;
        if (abl && (sel == 13 || sel == 10) && a.N % 256 == 0) return launch_ring_ablate(a, sel, abl, stream);
    }
#endif
    if constexpr (EPI == EPI_F32 || EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_RESID_GATE || EPI == EPI_QKV_ROPE) {
        // role-split 128 x 256 tiles (gemm_rs128.hip): forced by tile 14, or by the QKV-only knob at batch-1-sized shapes
        long t128x256 = (long)f5_cdiv(a.M, 128) * (a.N / 256);
        if (EPI == EPI_QKV_ROPE && a.seq_len > 0) t128x256 = (long)(a.M / a.seq_len) * f5_cdiv(a.seq_len, 128) * (a.N / 256);   // per-element row tiles
        // QKV at batch-1-sized shapes: one round of role-split 128 x 256 tiles when they fill >= 70 % of the CUs (M = 2 x 937: 192 tiles,
        // 22.0 vs 27.3 us for the 64 x 128 register-staged tiles, sample() 78.3 -> 73.7-76.4 ms; smaller grids stay with the small
        // tiles: M = 3 x 431 22.4 vs 18.4 us).  f5_gemm_qkv_small_tile: 0 = this rule, 14 = whenever one round, 12 / 13 = lock-step ring.
        // (the role-split QKV epilogue deals row tiles per batch element: it needs whole sequences, other shapes keep the small tiles)
        const bool qkv_rows_ok = EPI != EPI_QKV_ROPE || (a.seq_len > 0 && a.M % a.seq_len == 0);
        const bool qkv14 = EPI == EPI_QKV_ROPE && sel == 0 && qkv_rows_ok && a.rope_g4k != nullptr && t128x256 <= 256 &&
                           (f5_gemm_qkv_small_tile == 14 || (f5_gemm_qkv_small_tile == 0 && t128x256 >= 176));
        // MID sizes (batch 2 ... 16: more than one round of small tiles, too few 256 x 256 tiles to fill the chip twice): the role-split
        // 128 x 256 kernel in several rounds instead of the register-staged 128 x 128 kernel of round 1, which is where the `t128 >= 384`
        // fallback below used to send them.  sample() with it forced on every block GEMM (tile 14): batch 2 121.6 -> 110.1 ms, batch 3
        // 167.7 -> 154.5, batch 4 203.7 -> 166.5; equal to the 256 x 256 kernel at batch 8 (328.5 vs 327.4) and 16 (642.5 vs 650.6), whose
        // N = 1024 GEMMs (t256 < 512) fell to the small kernels as well (profiles/r03/mid_batch_dispatch.txt)
        const bool mid = sel == 0 && t128 >= 384 && a.N % 256 == 0 && a.ln_counter == nullptr && qkv_rows_ok;
        if ((sel == 14 || qkv14 || mid) && a.N % 256 == 0 && a.ln_counter == nullptr) return f5_launch_gemm_rs128(a, EPI, stream);
        if (sel == 14) sel = 0;
    }
    {
        // the small single-round kernels implement the fold for exactly the launches f5_gemm_fold_small names (statistics form)
        const bool small = f5_gemm_fold_small(a, EPI, true);
        F5_REQUIRE((a.x16_out == nullptr || (small && EPI == EPI_RESID_GATE)) && a.fold_rowf == nullptr &&
                       (a.fold_stats == nullptr || (small && EPI == EPI_GELU_TANH)),
                   "gemm: the LN fold (x16_out / fold_rowf / fold_stats) needs a launch on the 256x256 or the role-split 128x256 kernel "
                   "(f5_gemm_runs_staged), or one of the batch-1-sized launches of f5_gemm_fold_small in the statistics form");
    }
    if constexpr (EPI == EPI_QKV_ROPE) {
        // batch-1-sized QKV projection with pair-major tables: one round of 8-wave 128 x 256 tiles with transposed q / k wave tiles
        // (f5_gemm_qkv_small_tile = 13 / 12) instead of 64 x 128 register-staged tiles (0)
        if (sel == 0 && (f5_gemm_qkv_small_tile == 12 || f5_gemm_qkv_small_tile == 13) && a.rope_g4k != nullptr && a.N % 256 == 0) {
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
    a.debug_flags |= f5_gemm_debug_flags;      // process-wide flags on top of the caller's (an engine's own option)
    if (a.sat_flag == nullptr) a.sat_flag = f5_sat_flag_host;      // fp16 range detector of the 16-bit epilogues (op16.hpp), or null
    F5_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % BK == 0, "gemm: bad shape M=%d N=%d K=%d (K must be a multiple of %d)",
               a.M, a.N, a.K, BK);
    F5_REQUIRE(a.nseg == 1 || a.nseg == 3, "gemm: nseg must be 1 or 3");
    F5_REQUIRE(a.A[0] && a.W[0], "gemm: null operand");
    F5_REQUIRE(a.nseg == 1 || (a.A[1] && a.W[1]), "gemm: bf16x3 needs lo operands");
    F5_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm: leading dims must be multiples of 8");
    F5_REQUIRE((size_t)(a.a_row_mod > 0 ? a.a_row_mod : a.M) * a.lda < (1ull << 31) && (size_t)(a.N + 256) * a.ldw < (1ull << 31),
               "gemm: operands must stay below 4 GiB (the kernels use 32-bit byte offsets)");
    F5_REQUIRE(epi != EPI_RESID_GATE || (size_t)a.M * a.ldo * 4 < (1ull << 32), "gemm(resid): the residual stream must stay below 4 GiB");
    if (a.x16_out != nullptr || a.stats_out != nullptr) {
        F5_REQUIRE(epi == EPI_RESID_GATE && a.x16_out && a.stats_out && a.x16_scale && a.N % 64 == 0 && a.ldx16 % 4 == 0 && a.stats_ld >= a.M &&
                       (reinterpret_cast<uintptr_t>(a.x16_scale) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.x16_out) & 7) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.stats_out) & 7) == 0,
                   "gemm: LN-fold producer needs EPI_RESID_GATE, x16_out + stats_out + x16_scale (16-byte aligned), N %% 64 == 0");
    }
    if (a.fold_rowf != nullptr || a.fold_stats != nullptr) {
        F5_REQUIRE((epi == EPI_QKV_ROPE || epi == EPI_GELU_TANH) && a.nseg == 1 && a.out_bf[1] == nullptr && a.fold_c1 && a.fold_c2 &&
                       ((reinterpret_cast<uintptr_t>(a.fold_c1) | reinterpret_cast<uintptr_t>(a.fold_c2)) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.fold_rowf) & 7) == 0 && (a.debug_flags & 16384) == 0,
                   "gemm: LN-fold consumer needs EPI_QKV_ROPE / EPI_GELU_TANH, one-pass operands, aligned fold_c1 / fold_c2 / fold_rowf");
        F5_REQUIRE(a.fold_stats == nullptr || (a.K == 1024 && a.fold_stats_ld >= a.M && (reinterpret_cast<uintptr_t>(a.fold_stats) & 7) == 0 &&
                                               a.fold_mean_out != a.fold_shift),
                   "gemm: the statistics form of the LN fold needs K = 1024 (16 slices), fold_stats_ld >= M and fold_mean_out != fold_shift");
        F5_REQUIRE(epi != EPI_QKV_ROPE || (a.rope_g4k && a.dmodel % 256 == 0), "gemm(qkv): the LN fold needs the transposed q / k tiles");
        a.bias = nullptr;                             // inside fold_c2
    }
    switch (epi) {
        case EPI_F32: return launch_epi<EPI_F32>(a, stream);
        case EPI_BF16: return launch_epi<EPI_BF16>(a, stream);
        case EPI_GELU_TANH: return launch_epi<EPI_GELU_TANH>(a, stream);
        case EPI_GELU_ERF: return launch_epi<EPI_GELU_ERF>(a, stream);
        case EPI_RESID_GATE: return launch_epi<EPI_RESID_GATE>(a, stream);
        case EPI_QKV_ROPE:
            F5_REQUIRE(a.dmodel % 128 == 0 && a.N == 3 * a.dmodel, "gemm(qkv): N must be 3*dmodel, dmodel %% 128 == 0");
            F5_REQUIRE(a.rope_cos && a.rope_sin, "gemm(qkv): token-major rotation tables missing");
            if (a.rope_g4k || a.rope_g4q) {
                // group-major tables = transposed q / k tiles on the staged kernels: every 256-column tile must lie inside one of the
                // q | k | v ranges, the bias is read as 16-byte quads; gemm flag 16384 = A/B against the straight tiles
                F5_REQUIRE(a.rope_g4q && a.rope_g4k && ((reinterpret_cast<uintptr_t>(a.rope_g4q) | reinterpret_cast<uintptr_t>(a.rope_g4k)) & 15) == 0 &&
                               (size_t)a.seq_len * 16 * 16 < (1ull << 31),
                           "gemm(qkv): group-major rotation tables incomplete / not 16-byte aligned");
                const bool ok = a.dmodel % 256 == 0 && (a.bias == nullptr || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) &&
                                (f5_gemm_debug_flags & 16384) == 0;
                if (!ok) a.rope_g4q = a.rope_g4k = nullptr;
            }
            return launch_epi<EPI_QKV_ROPE>(a, stream);
        case EPI_ADDROWS: return launch_epi<EPI_ADDROWS>(a, stream);
        case EPI_RESID_KEEP: return launch_epi<EPI_RESID_KEEP>(a, stream);
        case EPI_GELU_ERF_BF16: return launch_epi<EPI_GELU_ERF_BF16>(a, stream);
        default: f5_set_error("gemm: unknown epilogue %d", epi); return 2;
    }
}
}  // namespace F5_NS
