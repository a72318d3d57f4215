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

#define BK 64

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
                        rc[ri][nb] = (isqk && rowbase + ri < p.M) ? p.rope_cos[n * 32 + j] : 1.0f;
                        rs[ri][nb] = (isqk && rowbase + ri < p.M) ? p.rope_sin[n * 32 + j] : 0.0f;
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
                                const float o = (c & 1) ? (v * rc[ri][nb] + partner * rs[ri][nb])
                                                        : (v * rc[ri][nb] - partner * rs[ri][nb]);
                                bf16_t h, l;
                                f5_split(o, h, l);
                                p.out_bf[0][(size_t)row * p.ldob + c] = h;
                                if (p.out_bf[1]) p.out_bf[1][(size_t)row * p.ldob + c] = l;
                            } else {
                                const int c2 = c - 2 * p.dmodel;
                                const int head = c2 >> 6, d = c2 & 63;
                                const size_t off = ((size_t)(b * p.heads + head) * 64 + d) * p.npad + n;
                                bf16_t h, l;
                                f5_split(v, h, l);
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
                            bf16_t h, l;
                            f5_split(v, h, l);
                            p.out_bf[0][(size_t)row * p.ldob + c] = h;
                            if (p.out_bf[1]) p.out_bf[1][(size_t)row * p.ldob + c] = l;
                        } else if (EPI == EPI_GELU_ERF) {
                            p.out_f32[(size_t)row * p.ldo + c] = f5_gelu_erf(v);
                        } else if (EPI == EPI_RESID_GATE) {
                            if (keep[r] == 0) v = 0.0f;
                            p.out_f32[(size_t)row * p.ldo + c] = pre[r][nb] + gcol[nb] * v;
                        } else if (EPI == EPI_ADDROWS) {
                            v += pre[r][nb];
                            p.out_f32[(size_t)row * p.ldo + c] = v;
                            bf16_t h, l;
                            f5_split(v, h, l);
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
}

// block tile = (64*MB) x (64*NB), 4 waves in a 2x2 grid, wave tile = (32*MB) x (32*NB)
template <int EPI, int MB, int NB>
__global__ __launch_bounds__(256) void f5_gemm_kernel(F5GemmArgs p, int tiles_n, int ntiles) {
    constexpr int BMt = 64 * MB, BNt = 64 * NB;
    constexpr int NA = MB * 2, NW = NB * 2;      // 16-byte chunks staged per thread for A / W
    __shared__ __attribute__((aligned(16))) bf16_t smem[2][(BMt + BNt) * BK];  // [buffer][A tile | W tile]

    // XCD-aware, bijective remap: workgroup b runs on XCD b % 8; give each XCD a contiguous range
    const int bid = blockIdx.x;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BMt, n0 = tn * BNt;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    size_t a_off[NA], w_off[NW];
    int sa_off[NA], sw_off[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int qd = tid + 256 * i;
        const int srow = qd >> 3, schunk = qd & 7;
        int gr = m0 + srow;
        if (gr > p.M - 1) gr = p.M - 1;
        if (p.a_row_mod > 0) gr = gr % p.a_row_mod;
        a_off[i] = (size_t)gr * p.lda + schunk * 8;
        sa_off[i] = swz_off(srow, schunk);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int qd = tid + 256 * i;
        const int srow = qd >> 3, schunk = qd & 7;
        w_off[i] = (size_t)(n0 + srow) * p.ldw + schunk * 8;
        sw_off[i] = BMt * BK + swz_off(srow, schunk);
    }

    const int kt = p.K / BK;
    const int T = kt * p.nseg;

    u32x4 ra[NA], rb[NW];
#define LOAD_TILE(tt_)                                                                                        \
    {                                                                                                         \
        const int seg_ = (tt_) / kt;                                                                          \
        const int k0_ = ((tt_) - seg_ * kt) * BK;                                                             \
        const bf16_t* Ap_ = (seg_ == 1) ? p.A[1] : p.A[0];                                                    \
        const bf16_t* Wp_ = (seg_ == 2) ? p.W[1] : p.W[0];                                                    \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const u32x4*>(Ap_ + a_off[i] + k0_); \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) rb[i] = *reinterpret_cast<const u32x4*>(Wp_ + w_off[i] + k0_); \
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

    LOAD_TILE(0);
    STORE_TILE(0);
    __syncthreads();

    const int frow = lane & 31;
    const int fk = lane >> 5;
    for (int tt = 0; tt < T; ++tt) {
        const int cur = tt & 1;
        if (tt + 1 < T) LOAD_TILE(tt + 1);
        const bf16_t* sA = smem[cur];
        const bf16_t* sB = smem[cur] + BMt * BK;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[MB], bfr[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                af[mb] = *reinterpret_cast<const bf16x8*>(&sA[swz_off(wm * (32 * MB) + mb * 32 + frow, ks * 2 + fk)]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                bfr[nb] = *reinterpret_cast<const bf16x8*>(&sB[swz_off(wn * (32 * NB) + nb * 32 + frow, ks * 2 + fk)]);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], bfr[nb], acc[mb][nb], 0, 0, 0);
        }
        if (tt + 1 < T) STORE_TILE(cur ^ 1);
        __syncthreads();
    }

    gemm_epilogue<EPI, MB, NB>(p, acc, m0, n0, wm, wn, lane);
}

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
int f5_gemm_tile_override = 0;  // 0 auto, 1 = 128x128, 2 = 64x128, 3 = 64x64 (microbenchmarks)
template <int EPI>
static int launch_epi(const F5GemmArgs& a, hipStream_t stream) {
    const long t128 = (long)f5_cdiv(a.M, 128) * f5_cdiv(a.N, 128);
    const long t64x128 = (long)f5_cdiv(a.M, 64) * f5_cdiv(a.N, 128);
    int sel = f5_gemm_tile_override;
    if (sel == 0) sel = t128 >= 384 ? 1 : (t64x128 >= 384 ? 2 : 3);
    if (EPI == EPI_QKV_ROPE && sel == 3) sel = 2;  // the V^T / head mapping wants >= one whole head per tile column
    if (sel == 1) return launch_cfg<EPI, 2, 2>(a, stream);
    if (sel == 2) return launch_cfg<EPI, 1, 2>(a, stream);
    return launch_cfg<EPI, 1, 1>(a, stream);
}

int f5_launch_gemm(const F5GemmArgs& a, int epi, hipStream_t stream) {
    F5_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % BK == 0, "gemm: bad shape M=%d N=%d K=%d (K must be a multiple of %d)",
               a.M, a.N, a.K, BK);
    F5_REQUIRE(a.nseg == 1 || a.nseg == 3, "gemm: nseg must be 1 or 3");
    F5_REQUIRE(a.A[0] && a.W[0], "gemm: null operand");
    F5_REQUIRE(a.nseg == 1 || (a.A[1] && a.W[1]), "gemm: bf16x3 needs lo operands");
    F5_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm: leading dims must be multiples of 8");
    switch (epi) {
        case EPI_F32: return launch_epi<EPI_F32>(a, stream);
        case EPI_BF16: return launch_epi<EPI_BF16>(a, stream);
        case EPI_GELU_TANH: return launch_epi<EPI_GELU_TANH>(a, stream);
        case EPI_GELU_ERF: return launch_epi<EPI_GELU_ERF>(a, stream);
        case EPI_RESID_GATE: return launch_epi<EPI_RESID_GATE>(a, stream);
        case EPI_QKV_ROPE:
            F5_REQUIRE(a.dmodel % 128 == 0 && a.N == 3 * a.dmodel, "gemm(qkv): N must be 3*dmodel, dmodel %% 128 == 0");
            return launch_epi<EPI_QKV_ROPE>(a, stream);
        case EPI_ADDROWS: return launch_epi<EPI_ADDROWS>(a, stream);
        case EPI_RESID_KEEP: return launch_epi<EPI_RESID_KEEP>(a, stream);
        case EPI_GELU_ERF_BF16: return launch_epi<EPI_GELU_ERF_BF16>(a, stream);
        default: f5_set_error("gemm: unknown epilogue %d", epi); return 2;
    }
}
