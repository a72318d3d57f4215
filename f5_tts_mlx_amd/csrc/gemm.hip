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

#define BM 128
#define BN 128
#define BK 64

__device__ __forceinline__ int swz_off(int row, int chunk) {
    return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const F5GemmArgs& p, f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn,
                                               int lane) {
    const int hi = lane >> 5;
    const int lcol = lane & 31;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int rowbase = m0 + wm * 64 + mb * 32 + rg * 8 + hi * 4;
            int nbase = 0, bbase = 0;
            if (EPI == EPI_QKV_ROPE) {
                bbase = rowbase / p.seq_len;
                nbase = rowbase - bbase * p.seq_len;
            }
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const int row = rowbase + ri;
                const bool rowok = row < p.M;
                int n = nbase + ri, b = bbase;
                if (EPI == EPI_QKV_ROPE) {
                    if (n >= p.seq_len) {
                        n -= p.seq_len;
                        b += 1;
                    }
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int col = n0 + wn * 64 + nb * 32 + lcol;
                    const bool colok = col < p.N;
                    float v = acc[mb][nb][rg * 4 + ri];
                    if (EPI != EPI_ADDROWS) {
                        if (p.bias != nullptr && colok) v += p.bias[col];
                    }
                    if (EPI == EPI_QKV_ROPE) {
                        const float partner = __shfl_xor(v, 1, 64);
                        if (rowok && colok) {
                            if (col < 2 * p.dmodel) {
                                const int j = (col & 63) >> 1;
                                const float c = p.rope_cos[n * 32 + j], s = p.rope_sin[n * 32 + j];
                                const float o = (col & 1) ? (v * c + partner * s) : (v * c - partner * s);
                                bf16_t h, l;
                                f5_split(o, h, l);
                                p.out_bf[0][(size_t)row * p.ldob + col] = h;
                                if (p.out_bf[1]) p.out_bf[1][(size_t)row * p.ldob + col] = l;
                            } else {
                                const int c2 = col - 2 * p.dmodel;
                                const int head = c2 >> 6, d = c2 & 63;
                                const size_t off = ((size_t)(b * p.heads + head) * 64 + d) * p.npad + n;
                                bf16_t h, l;
                                f5_split(v, h, l);
                                p.vt[0][off] = h;
                                if (p.vt[1]) p.vt[1][off] = l;
                            }
                        }
                    } else if (rowok && colok) {
                        if (EPI == EPI_F32) {
                            p.out_f32[(size_t)row * p.ldo + col] = v;
                        } else if (EPI == EPI_BF16 || EPI == EPI_GELU_TANH) {
                            if (EPI == EPI_GELU_TANH) v = f5_gelu_tanh(v);
                            bf16_t h, l;
                            f5_split(v, h, l);
                            p.out_bf[0][(size_t)row * p.ldob + col] = h;
                            if (p.out_bf[1]) p.out_bf[1][(size_t)row * p.ldob + col] = l;
                        } else if (EPI == EPI_GELU_ERF) {
                            p.out_f32[(size_t)row * p.ldo + col] = f5_gelu_erf(v);
                        } else if (EPI == EPI_RESID_GATE) {
                            if (p.rowkeep != nullptr && p.rowkeep[row] == 0) v = 0.0f;
                            float* o = p.out_f32 + (size_t)row * p.ldo + col;
                            *o = *o + p.gate[col] * v;
                        } else if (EPI == EPI_ADDROWS) {
                            v += p.addrows[(size_t)row * p.ldadd + col];
                            p.out_f32[(size_t)row * p.ldo + col] = v;
                            bf16_t h, l;
                            f5_split(v, h, l);
                            p.out_bf[0][(size_t)row * p.ldob + col] = h;
                            if (p.out_bf[1]) p.out_bf[1][(size_t)row * p.ldob + col] = l;
                        } else if (EPI == EPI_RESID_KEEP) {
                            v += p.resid[(size_t)row * p.ldres + col];
                            if (p.rowkeep != nullptr && p.rowkeep[row] == 0) v = 0.0f;
                            p.out_f32[(size_t)row * p.ldo + col] = v;
                        }
                    }
                }
            }
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(256) void f5_gemm_kernel(F5GemmArgs p, int tiles_n, int ntiles) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2][2][BM * BK];  // [buffer][A|W][tile]

    // XCD-aware, bijective remap: workgroup b runs on XCD b % 8; give each XCD a contiguous range
    const int bid = blockIdx.x;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // staging assignment: 4 chunks (16 B) of A and of W per thread
    size_t a_off[4], w_off[4];
    int s_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int qd = tid + 256 * i;
        const int srow = qd >> 3, schunk = qd & 7;
        int gr = m0 + srow;
        if (gr > p.M - 1) gr = p.M - 1;
        if (p.a_row_mod > 0) gr = gr % p.a_row_mod;
        a_off[i] = (size_t)gr * p.lda + schunk * 8;
        w_off[i] = (size_t)(n0 + srow) * p.ldw + schunk * 8;
        s_off[i] = swz_off(srow, schunk);
    }

    const int kt = p.K / BK;
    const int T = kt * p.nseg;

    u32x4 ra[4], rb[4];
#define LOAD_TILE(tt_)                                                                     \
    {                                                                                      \
        const int seg_ = (tt_) / kt;                                                       \
        const int k0_ = ((tt_) - seg_ * kt) * BK;                                          \
        const bf16_t* Ap_ = (seg_ == 1) ? p.A[1] : p.A[0];                                 \
        const bf16_t* Wp_ = (seg_ == 2) ? p.W[1] : p.W[0];                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                    \
            ra[i] = *reinterpret_cast<const u32x4*>(Ap_ + a_off[i] + k0_);                 \
            rb[i] = *reinterpret_cast<const u32x4*>(Wp_ + w_off[i] + k0_);                 \
        }                                                                                  \
    }
#define STORE_TILE(buf_)                                                                   \
    {                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                    \
            *reinterpret_cast<u32x4*>(&smem[buf_][0][s_off[i]]) = ra[i];                   \
            *reinterpret_cast<u32x4*>(&smem[buf_][1][s_off[i]]) = rb[i];                   \
        }                                                                                  \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
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
        const bf16_t* sA = smem[cur][0];
        const bf16_t* sB = smem[cur][1];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                af[mb] = *reinterpret_cast<const bf16x8*>(&sA[swz_off(wm * 64 + mb * 32 + frow, ks * 2 + fk)]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                bfr[nb] = *reinterpret_cast<const bf16x8*>(&sB[swz_off(wn * 64 + nb * 32 + frow, ks * 2 + fk)]);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], bfr[nb], acc[mb][nb], 0, 0, 0);
        }
        if (tt + 1 < T) STORE_TILE(cur ^ 1);
        __syncthreads();
    }

    gemm_epilogue<EPI>(p, acc, m0, n0, wm, wn, lane);
}

template <int EPI>
static int launch_epi(const F5GemmArgs& a, hipStream_t stream) {
    const int tiles_m = f5_cdiv(a.M, BM), tiles_n = f5_cdiv(a.N, BN);
    const int ntiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((f5_gemm_kernel<EPI>), dim3(ntiles), dim3(256), 0, stream, a, tiles_n, ntiles);
    F5_LAUNCH_CHECK();
    return 0;
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
            F5_REQUIRE(a.dmodel % BN == 0 && a.N == 3 * a.dmodel, "gemm(qkv): N must be 3*dmodel, dmodel %% 128 == 0");
            return launch_epi<EPI_QKV_ROPE>(a, stream);
        case EPI_ADDROWS: return launch_epi<EPI_ADDROWS>(a, stream);
        case EPI_RESID_KEEP: return launch_epi<EPI_RESID_KEEP>(a, stream);
        default: f5_set_error("gemm: unknown epilogue %d", epi); return 2;
    }
}
