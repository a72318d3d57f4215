// 128 x 256 x 32 MFMA GEMM for gfx950, TWO workgroups per CU: the overlap experiment of round 3.
//
// Why: in the 256 x 256 kernel (one workgroup per CU) a tile's epilogue runs with the matrix cores idle, and every CU enters it at
// the same time: out-proj spends 80 of 185 us in a read-modify-write of the residual stream at the HBM roofline (6.1 TB/s) while
// HBM idles through the main loops.  Two independent workgroups per CU can be in different phases - one streams its epilogue
// while the other multiplies - but only if ONE workgroup alone keeps the matrix pipe busy; the 128 x 256 kernel of round 1
// (one barrier + exposed fragment reads per 16 MFMAs, 475 TF per workgroup alone or shared) could not, so its epilogues stayed
// additive.  Here every wave prefetches the fragments of K step s+1 into a second register set while its 16 MFMAs of step s run
// (a lone wave per SIMD then needs no partner to cover its LDS latency), 3-stage global_load_lds ring, counted vmcnt, one barrier
// per step.
//
// Geometry: 256 threads = 4 waves (2 x 2), wave tile 64 x 128 = 2 x 4 accumulators of v_mfma_f32_32x32x16 (128 registers) + two
// fragment sets of 12 x 4 registers; LDS = 3 stages x (128 + 256) rows x 32 k x 2 B = 72 KB (reused as epilogue staging).
// Per step s:  vmcnt (stage s+1 landed) -> lgkmcnt(0) -> barrier -> stage s+3 into the slot of stage s (every wave has read it) ->
// fragment reads of stage s+1 -> 16 MFMAs on the registers of stage s.
#include "gemm.hpp"
#include "gemm_dev.hpp"

namespace F5_NS {

#define G128_BK 32
__device__ __forceinline__ int g128_swz(int row, int chunk) { return row * G128_BK + ((chunk ^ ((row >> 2) & 3)) << 3); }

int f5_gemm128_pad_lds = 0;     // experiment: extra dynamic LDS per workgroup (bytes); > 8 KB leaves ONE workgroup per CU

template <int EPI, bool QT>
__global__ __launch_bounds__(256, 2) void f5_gemm128_kernel(F5GemmArgs p, int tiles_n, int ntiles, int tiles_m) {
    constexpr int BMt = 128, BNt = 256, NST = 3;
    constexpr int NA = 2, NW = 4, G = NA + NW;
    constexpr int STAGE = (BMt + BNt) * G128_BK;          // 12288 elements = 24 KB
    __shared__ __attribute__((aligned(16))) op16_t smem[NST * STAGE];

    const int bid = blockIdx.x;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    int tm, tn;
    if (p.nband > 0) {                                   // band-major: bands of nband column tiles walked row by row
        const int per_band = tiles_m * p.nband;
        const int band = tile / per_band, r_ = tile - band * per_band;
        tm = r_ / p.nband;
        tn = band * p.nband + (r_ - tm * p.nband);
    } else {
        tm = tile / tiles_n;
        tn = tile - tm * tiles_n;
    }
    const int m0 = tm * BMt, n0 = tn * BNt;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    uint32_t a_src[NA], w_src[NW];                       // byte offsets (without k0)
    int a_dst[NA], w_dst[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q_ = i * 256 + tid;
        const int row = q_ >> 2, chunk = (q_ & 3) ^ ((row >> 2) & 3);
        int gr = m0 + row;
        if (gr > p.M - 1) gr = p.M - 1;
        if (p.a_row_mod > 0) gr = gr % p.a_row_mod;
        a_src[i] = ((uint32_t)gr * (uint32_t)p.lda + chunk * 8) * 2u;
        a_dst[i] = (i * 256 + wave * 64) * 8;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int q_ = i * 256 + tid;
        const int row = q_ >> 2, chunk = (q_ & 3) ^ ((row >> 2) & 3);
        w_src[i] = ((uint32_t)(n0 + row) * (uint32_t)p.ldw + chunk * 8) * 2u;
        w_dst[i] = BMt * G128_BK + (i * 256 + wave * 64) * 8;
    }
    const int kt = p.K / G128_BK;
    const int T = kt * p.nseg;
    // (segment, K byte offset) of the NEXT stage to issue are running values (a division per issue sat on the critical path)
    int i_seg = 0;
    uint32_t i_kb = 0;
    const uint32_t k_bytes = (uint32_t)p.K * 2u;
#define G128_ISSUE(slot_) /* element offset of the stage slot */                                                                                   \
    {                                                                                                        \
        op16_t* st_ = smem + (slot_);                                                                        \
        const char* Ap_ = reinterpret_cast<const char*>((i_seg == 1) ? p.A[1] : p.A[0]);                     \
        const char* Wp_ = reinterpret_cast<const char*>((i_seg == 2) ? p.W[1] : p.W[0]);                     \
        _Pragma("unroll") for (int i = 0; i < NA; ++i)                                                       \
            glds16(reinterpret_cast<const op16_t*>(Ap_ + (a_src[i] + i_kb)), st_ + a_dst[i]);                \
        _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                       \
            glds16(reinterpret_cast<const op16_t*>(Wp_ + (w_src[i] + i_kb)), st_ + w_dst[i]);                \
        i_kb += G128_BK * 2u;                                                                                \
        if (i_kb == k_bytes) {                                                                               \
            i_kb = 0;                                                                                        \
            ++i_seg;                                                                                         \
        }                                                                                                    \
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int frow = lane & 31;
    const int fk = lane >> 5;
    // fragment read pointers inside a stage (the stage offset is a running scalar)
    const op16_t* pa[2];
    const op16_t* pb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        pa[ks] = smem + g128_swz(wm * 64 + frow, ks * 2 + fk);
        pb[ks] = smem + BMt * G128_BK + g128_swz(wn * 128 + frow, ks * 2 + fk);
    }
    int rd = STAGE, wr = 0;          // element offsets of the stage read in this step ((s+1) % 3) and of the slot re-staged (s % 3)
#define G128_READ(SET_, OFF_)                                                                                                \
    {                                                                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                   \
            _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                                 \
                af##SET_[ks][mb] = *reinterpret_cast<const op16x8*>(pa[ks] + (OFF_) + mb * 32 * G128_BK);                    \
            _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)                                                                 \
                bf##SET_[ks][nb] = *reinterpret_cast<const op16x8*>(pb[ks] + (OFF_) + nb * 32 * G128_BK);                    \
        }                                                                                                                    \
    }
#define G128_MM(TR_, A_, B_, C_) ((TR_) ? F5_MFMA32(B_, A_, C_, 0, 0, 0) : F5_MFMA32(A_, B_, C_, 0, 0, 0))
    // One K step.  The registers of stage s are in fragment set CUR; KIND 0 (steady state): stage s+1 is read into set NXT,
    // stage s+3 is staged into the slot of stage s, one stage stays in flight (vmcnt 6); KIND 1 / 2: the last steps that still read
    // (nothing left to stage; vmcnt 6 / 0); KIND 3: the last step.  In the steady state the 12 fragment reads and the 6 staging
    // loads are interleaved with the 16 MFMAs (sched_group_barrier pins the order inside the basic block): a lone wave per SIMD
    // issues its loads in the shadow of its own matrix instructions.
#define G128_STEP(CUR, NXT, KIND, TR_, s_)                                                                                   \
    {                                                                                                                        \
        if (KIND == 0 || (s_) + 2 < T) {                                                                                     \
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   /* stage s+1 landed, stage s+2 may be in flight */          \
        } else {                                                                                                             \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                 \
        }                                                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     /* this wave's reads of stage s have returned */             \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        __builtin_amdgcn_s_barrier();                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        if (KIND == 0 || (s_) + 1 < T) G128_READ(NXT, rd);                                                                   \
        if (KIND == 0 || (s_) + 3 < T) G128_ISSUE(wr);                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                     \
            _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                                 \
                _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)                                                             \
                    acc[mb][nb] = G128_MM(TR_, af##CUR[ks][mb], bf##CUR[ks][nb], acc[mb][nb]);                               \
        if (KIND == 0) {                                                                                                     \
            _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                                                              \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                           \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                           \
            }                                                                                                                \
            _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                                               \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                           \
                __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);                                                           \
            }                                                                                                                \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                               \
        }                                                                                                                    \
        wr = rd;                                                                                                             \
        rd += STAGE;                                                                                                         \
        if (rd == NST * STAGE) rd = 0;                                                                                       \
    }

    op16x8 af0[2][2], bf0[2][4], af1[2][2], bf1[2][4];
    G128_ISSUE(0);
    if (1 < T) G128_ISSUE(STAGE);
    if (2 < T) G128_ISSUE(2 * STAGE);
    if (2 < T) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else if (1 < T) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    G128_READ(0, 0);

    op16_t* stage = smem + wave * 9216;                   // 18 KB of private epilogue staging per wave
    const int row0 = m0 + wm * 64, col0 = n0 + wn * 128;
    constexpr bool TR_EPI = (EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_GELU_ERF_BF16);
    // fragment sets alternate; the last (up to) four steps are peeled: they stage nothing / read nothing
#define G128_LOOP(TR_)                                                                                                       \
    {                                                                                                                        \
        int s = 0;                                                                                                           \
        for (; s + 4 < T; s += 2) {        /* steady state: both steps read a stage and stage one */                       \
            G128_STEP(0, 1, 0, TR_, s);                                                                                      \
            G128_STEP(1, 0, 0, TR_, s + 1);                                                                                  \
        }                                                                                                                    \
        for (; s < T; s += 2) {            /* the last (up to) four steps: conditions at run time */                         \
            G128_STEP(0, 1, 1, TR_, s);                                                                                      \
            if (s + 1 < T) G128_STEP(1, 0, 1, TR_, s + 1);                                                                   \
        }                                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        __builtin_amdgcn_s_barrier();      /* every wave is done with the ring: it becomes staging space */                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
    }

    if ((TR_EPI && (p.debug_flags & 16384) == 0) || (QT && n0 < 2 * p.dmodel)) {       // workgroup-uniform
        G128_LOOP(true);
        if (p.debug_flags & 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
            return;
        }
        if (QT) staged_epilogue_tr_rope<2, 4>(p, acc, stage, row0, col0, lane);
        else staged_epilogue_tr<EPI, 2, 4>(p, acc, stage, row0, col0, lane);
        return;
    }
    G128_LOOP(false);
    if (p.debug_flags & 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    if (QT) {
        staged_epilogue_bf16<EPI, 2, 4, true>(p, acc, stage, row0, col0, lane);
    } else if (EPI == EPI_BF16 || EPI == EPI_GELU_TANH || EPI == EPI_GELU_ERF_BF16 || EPI == EPI_QKV_ROPE) {
        staged_epilogue_bf16<EPI, 2, 4>(p, acc, stage, row0, col0, lane);
    } else if (EPI == EPI_RESID_GATE) {
        staged_epilogue_resid<2, 4>(p, acc, reinterpret_cast<float*>(stage), row0, col0, lane);
    } else {
        gemm_epilogue<EPI, 2, 4>(p, acc, m0, n0, wm, wn, lane);
    }
#undef G128_LOOP
#undef G128_STEP
#undef G128_MM
#undef G128_READ
#undef G128_ISSUE
}

extern int f5_gemm_nband;
template <int EPI>
static int launch128(const F5GemmArgs& a, hipStream_t stream) {
    const int tiles_m = f5_cdiv(a.M, 128), tiles_n = a.N / 256;
    const int ntiles = tiles_m * tiles_n;
    F5GemmArgs ab = a;
    ab.nband = (f5_gemm_nband > 0 && tiles_n > f5_gemm_nband && tiles_n % f5_gemm_nband == 0) ? f5_gemm_nband : 0;
    if (a.bias != nullptr && (reinterpret_cast<uintptr_t>(a.bias) & 15) != 0) ab.debug_flags |= 16384;
    const size_t dyn = (size_t)f5_gemm128_pad_lds;
    if (EPI == EPI_QKV_ROPE && ab.rope_g4k != nullptr) {
        hipLaunchKernelGGL((f5_gemm128_kernel<EPI, EPI == EPI_QKV_ROPE>), dim3(ntiles), dim3(256), dyn, stream, ab, tiles_n, ntiles, tiles_m);
    } else {
        hipLaunchKernelGGL((f5_gemm128_kernel<EPI, false>), dim3(ntiles), dim3(256), dyn, stream, ab, tiles_n, ntiles, tiles_m);
    }
    F5_LAUNCH_CHECK();
    return 0;
}

int f5_launch_gemm128(const F5GemmArgs& a, int epi, hipStream_t stream) {
    F5_REQUIRE(a.N % 256 == 0 && a.M >= 128 && a.K % G128_BK == 0, "gemm128: needs N %% 256 == 0, M >= 128, K %% 32 == 0");
    F5_REQUIRE((size_t)(a.a_row_mod > 0 ? a.a_row_mod : a.M) * a.lda < (1ull << 31) && (size_t)(a.N + 256) * a.ldw < (1ull << 31),
               "gemm128: operands must stay below 4 GiB (32-bit byte offsets)");
    switch (epi) {
        case EPI_F32: return launch128<EPI_F32>(a, stream);
        case EPI_BF16: return launch128<EPI_BF16>(a, stream);
        case EPI_GELU_TANH: return launch128<EPI_GELU_TANH>(a, stream);
        case EPI_RESID_GATE: return launch128<EPI_RESID_GATE>(a, stream);
        case EPI_QKV_ROPE: return launch128<EPI_QKV_ROPE>(a, stream);
        default: f5_set_error("gemm128: unsupported epilogue %d", epi); return 2;
    }
}
}  // namespace F5_NS
