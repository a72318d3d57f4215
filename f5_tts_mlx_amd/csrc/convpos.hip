// ConvPositionEmbedding conv (dit.py:29-50): grouped Conv1d(C, C, k=31, groups=C/64, pad=15) + Mish on
// channels-last (b, n, c) input, as an implicit GEMM per group on v_mfma_f32_32x32x16_bf16:
//   out[n][co] = sum_{tap, ci} x[n + tap - 15][ci] * w[co][tap][ci]      (M = tokens, N = 64, K = taps*64)
// The A operand is a sliding window over ONE LDS-resident halo tile of x ((128 + taps - 1) rows x 64 ci),
// so x is read from HBM once per block; the per-tap weight slab [64 co][64 ci] is double-buffered.
// Sequence edges are zero padded per batch element (Conv1d padding); no mask (dit.py:251 passes none).
#include "convpos.hpp"
#include "rowops.hpp"     // f5_sat_flag_host

namespace F5_NS {

#define XLD 72  // LDS row stride (elements) for both tiles: 144 B rows => conflict-free ds_read_b128
#define CP_ROWS 128
#define CP_MAXTAPS 31

// TPS = taps per pipeline step: the weight slabs of TPS taps are fetched, stored and synchronised together (fewer, fatter
// steps).  Measured no better than TPS = 1 at batch 1 and worse at batch 32, where the 18 KB weight ring of TPS = 1 lets 3-4
// workgroups share a CU (tools/convpos_bench.py); what did help at batch 1 is the block numbering below (25.5 -> 19.8 us).
template <bool HP, int TPS>
__global__ __launch_bounds__(256) void f5_convpos_kernel(F5ConvPosArgs p) {
    constexpr int NP = HP ? 2 : 1;
    constexpr int HALO = CP_ROWS + CP_MAXTAPS - 1;
    __shared__ __attribute__((aligned(16))) op16_t sX[NP][HALO * XLD];
    __shared__ __attribute__((aligned(16))) op16_t sW[2][TPS][NP][64 * XLD];

    asm volatile("" ::"s"(p.in[0]), "s"(p.W[0]), "s"(p.bias), "s"(p.B), "s"(p.seq_len), "s"(p.groups), "s"(p.taps), "s"(p.ld), "s"(p.mode),
                 "s"(p.out_bf[0]), "s"(p.out_f32), "s"(p.ldo));              // the argument block in one scalar-load clause
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, lr = lane & 31;
    // workgroup -> (token tile, group, batch element).  Every workgroup of a group streams that group's 31 weight slabs
    // (254 KB); with the plain (x = token tile, y = group, z = batch) numbering the token tiles of a group sit on different
    // XCDs (XCD = linear id & 7) and every XCD's L2 ends up streaming ALL groups' weights.  The 1-D grid used when the group
    // count is a multiple of 8 deals groups to XCDs instead: XCD x hosts groups x, x + 8, ... with all their token tiles and
    // batch elements, so a group's weights are fetched into one L2 and shared by its workgroups.
    int n0, g, b;
    if (gridDim.y == 1 && gridDim.z == 1) {
        const int nx = (p.seq_len + CP_ROWS - 1) / CP_ROWS;
        const int per_group = nx * p.B;
        const int s_ = blockIdx.x >> 3;
        const int gl = s_ / per_group, rem = s_ - gl * per_group;
        g = gl * 8 + (blockIdx.x & 7);
        b = rem / nx;
        n0 = (rem - b * nx) * CP_ROWS;
    } else {
        n0 = blockIdx.x * CP_ROWS;
        g = blockIdx.y;
        b = blockIdx.z;
    }
    const int taps = p.taps;
    const int pad = taps >> 1;
    const int halo = CP_ROWS + taps - 1;
    const size_t rowbase = (size_t)b * p.seq_len;
    const int kdim = taps * 64;

    // ---- stage the x halo tile (zero outside [0, seq_len)) --------------------------------------
    for (int qd = tid; qd < halo * 8; qd += 256) {
        const int r = qd >> 3, c = qd & 7;
        const int n = n0 - pad + r;
        const bool ok = (n >= 0) && (n < p.seq_len);
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4*>(p.in[pp] + (rowbase + n) * p.ld + g * 64 + c * 8);
            *reinterpret_cast<u32x4*>(&sX[pp][r * XLD + c * 8]) = v;
        }
    }

    // ---- weight slab staging: 2 chunks per thread per part per tap ------------------------------
    u32x4 rw[TPS][NP][2];
#define CP_LOADW(t0_)                                                                                   \
    {                                                                                                   \
        _Pragma("unroll") for (int tt_ = 0; tt_ < TPS; ++tt_) {                                         \
            const int tap_ = (t0_) + tt_ < taps ? (t0_) + tt_ : taps - 1;   /* clamped slabs are loaded, never used */ \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                             \
                const int qd_ = tid + 256 * i;                                                          \
                const int r_ = qd_ >> 3, c_ = qd_ & 7;                                                  \
                _Pragma("unroll") for (int pp = 0; pp < NP; ++pp) rw[tt_][pp][i] =                      \
                    *reinterpret_cast<const u32x4*>(p.W[pp] + (size_t)(g * 64 + r_) * kdim + tap_ * 64 + c_ * 8); \
            }                                                                                           \
        }                                                                                               \
    }
#define CP_STOREW(buf_)                                                                                 \
    {                                                                                                   \
        _Pragma("unroll") for (int tt_ = 0; tt_ < TPS; ++tt_)                                           \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                             \
                const int qd_ = tid + 256 * i;                                                          \
                const int r_ = qd_ >> 3, c_ = qd_ & 7;                                                  \
                _Pragma("unroll") for (int pp = 0; pp < NP; ++pp)                                       \
                    *reinterpret_cast<u32x4*>(&sW[buf_][tt_][pp][r_ * XLD + c_ * 8]) = rw[tt_][pp][i];  \
            }                                                                                           \
    }

    f32x16 acc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        acc[0][e] = 0.0f;
        acc[1][e] = 0.0f;
    }

    CP_LOADW(0);
    CP_STOREW(0);
    __syncthreads();

    // mode 1 accumulates into x.  `out[off] += v` element by element in the epilogue made the compiler keep every
    // read-modify-write in program order (it cannot rule out that one store aliases the next load): 32 dependent memory round
    // trips per lane at the end of the kernel.  The old values are requested HERE, before the tap loop (clamped rows, so the loads
    // are straight-line), and are long there when the epilogue adds and stores.
    float old[2][16];
    if (p.mode != 0) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int n = n0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (n > p.seq_len - 1) n = p.seq_len - 1;
                old[nb][r] = p.out_f32[(rowbase + n) * p.ldo + g * 64 + nb * 32 + lr];
            }
    }

    for (int t0 = 0, step = 0; t0 < taps; t0 += TPS, ++step) {
        const int cur = step & 1;
        if (t0 + TPS < taps) CP_LOADW(t0 + TPS);
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
            const int t = t0 + tt;
            if (t < taps) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int aoff = (wave * 32 + lr + t) * XLD + ks * 16 + hi * 8;
                    const op16x8 a = *reinterpret_cast<const op16x8*>(&sX[0][aoff]);
                    op16x8 al = a;
                    if (HP) al = *reinterpret_cast<const op16x8*>(&sX[NP - 1][aoff]);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const int boff = (nb * 32 + lr) * XLD + ks * 16 + hi * 8;
                        const op16x8 w = *reinterpret_cast<const op16x8*>(&sW[cur][tt][0][boff]);
                        acc[nb] = F5_MFMA32(a, w, acc[nb], 0, 0, 0);
                        if (HP) {
                            const op16x8 wl = *reinterpret_cast<const op16x8*>(&sW[cur][tt][NP - 1][boff]);
                            acc[nb] = F5_MFMA32(al, w, acc[nb], 0, 0, 0);
                            acc[nb] = F5_MFMA32(a, wl, acc[nb], 0, 0, 0);
                        }
                    }
                }
            }
        }
        if (t0 + TPS < taps) CP_STOREW(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: + bias, Mish, store ----------------------------------------------------------
    // the two bias values are requested together and waited for ONCE, outside the per-row conditionals: with the load inside the
    // nb loop the compiler put its `s_waitcnt vmcnt(0)` for it into every conditional row block, where it also waits for all the
    // stores issued so far -- the stores of a lane went out one memory round trip apart
    float bias2[2] = {p.bias[g * 64 + lr], p.bias[g * 64 + 32 + lr]};
    asm volatile("" : "+v"(bias2[0]), "+v"(bias2[1]));
    f5_sat_t trk;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int co = g * 64 + nb * 32 + lr;
        const float bias = bias2[nb];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (n < p.seq_len) {
                const float v = f5_mish(acc[nb][r] + bias);
                const size_t off = (rowbase + n) * p.ldo + co;
                if (p.mode == 0) {
                    op16_t h, l;
                    f5_split(v, h, l, trk);
                    p.out_bf[0][off] = h;
                    if (p.out_bf[1]) p.out_bf[1][off] = l;
                } else {
                    p.out_f32[off] = old[nb][r] + v;
                }
            }
        }
    }
    f5_sat_commit(trk, p.sat_flag);
}

int f5_convpos_tps = 0;   // taps per pipeline step: 0 auto, 1 / 2 / 4 forced (f5_debug_set_convpos_tps)
int f5_convpos_xcd_map = 1;   // 0: plain 3-D block numbering (A/B)
int f5_launch_convpos(const F5ConvPosArgs& a, hipStream_t stream) {
    F5_REQUIRE(a.B > 0 && a.seq_len > 0 && a.groups > 0, "convpos: bad shape");
    F5_REQUIRE(a.C == a.groups * 64, "convpos: channels per group must be 64 (C=%d groups=%d)", a.C, a.groups);
    F5_REQUIRE(a.taps >= 1 && a.taps <= CP_MAXTAPS && (a.taps & 1), "convpos: taps must be odd and <= %d", CP_MAXTAPS);
    F5_REQUIRE(a.ld % 8 == 0, "convpos: ld must be a multiple of 8");
    dim3 grid(f5_cdiv(a.seq_len, CP_ROWS), a.groups, a.B);
    const long nwg = (long)grid.x * grid.y * grid.z;
    if (a.groups % 8 == 0 && f5_convpos_xcd_map) grid = dim3((unsigned)nwg, 1, 1);      // group-per-XCD numbering (see the kernel)
    // taps per pipeline step: measured (tools/convpos_bench.py) 1 is best at every size once the groups are dealt to XCDs
    // (batch 1: 19.8 / 21.1 / 21.3 us for 1 / 2 / 4; batch 32: 304 / 355 / 629 us); 2 and 4 stay selectable for experiments
    int tps = f5_convpos_tps;
    if (tps == 0) tps = 1;
    F5ConvPosArgs ab = a;
    if (ab.sat_flag == nullptr) ab.sat_flag = f5_sat_flag_host;
    if (a.nseg == 3) {
        F5_REQUIRE(a.in[1] && a.W[1], "convpos: bf16x3 needs lo operands");
        if (tps > 1) hipLaunchKernelGGL((f5_convpos_kernel<true, 2>), grid, dim3(256), 0, stream, ab);
        else hipLaunchKernelGGL((f5_convpos_kernel<true, 1>), grid, dim3(256), 0, stream, ab);
    } else if (tps >= 4) {
        hipLaunchKernelGGL((f5_convpos_kernel<false, 4>), grid, dim3(256), 0, stream, ab);
    } else if (tps >= 2) {
        hipLaunchKernelGGL((f5_convpos_kernel<false, 2>), grid, dim3(256), 0, stream, ab);
    } else {
        hipLaunchKernelGGL((f5_convpos_kernel<false, 1>), grid, dim3(256), 0, stream, ab);
    }
    F5_LAUNCH_CHECK();
    return 0;
}
}  // namespace F5_NS
