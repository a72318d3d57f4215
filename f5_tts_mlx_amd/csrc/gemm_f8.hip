// MX-fp8 GEMM for gfx950 (precision mode "mxfp8", BASELINE configs[4]): 256 x 256 x 128 tiles on v_mfma_scale_f32_32x32x64_f8f6f4,
// OCP e4m3 operands with one E8M0 scale per 32 consecutive K elements, the epilogues of the DiT block, and the row quantiser.
#include "gemm.hpp"
#include "gemm_dev.hpp"

namespace F5_NS {

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
        staged_epilogue_resid<4, 2>(p, acc, reinterpret_cast<float*>(stage), m0 + wm * 128, n0 + wn * 64, lane);
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
    a.debug_flags |= f5_gemm_debug_flags;      // process-wide flags on top of the caller's (an engine's own option)
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

}  // namespace F5_NS
