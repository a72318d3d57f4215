// probe (round 6): is the 256 x 256 main loop held by the LDS port, and does the vendor's wave tile lift it?
// The shipped kernel (8 waves, wave tile 128 x 64) reads 0.75 operand fragments per MFMA: 192 KB of ds_read_b128 + 64 KB of LDS-DMA
// writes per 64-wide K tile and CU = 2 048 cycles of the 128 B/clk LDS port for 2 048 cycles of MFMA per SIMD -- co-limited, and the
// loop sits at ~60 % pipe time (1.2 PF against 1 656 TF for registers only: profiles/r06/mfma_energy_probe.jsonl).  This probe:
//   ONE wave per SIMD (256 threads, 512 registers), wave tile 128 x 128 (0.5 fragments per MFMA: 128 KB of reads per 64 of K), macro tile
//   256 x 256, K tiles of 32 in a ring of FOUR 32 KB slots (128 KB; 64-byte rows, 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 3)),
//   operands by global_load_lds three K tiles ahead, fragments double-buffered per 16-wide K step, ONE workgroup barrier per K tile
//   (= per 32 MFMAs and wave), in front of the tile's second K step.
//   mode 0  main loop only            mode 2  C = A W^T stored as fp32 (checked against fp64 on the host)
//   mode 1  x += gate * (A W^T) in fp32 + a 16-bit copy (the memory traffic of the shipped residual epilogue, serial, straight from
//           the accumulators)
// build: hipcc --offload-arch=gfx950 -O3 wave128_gemm.hip -o bin/wave128_gemm      run: wave128_gemm <N> <K> [iters]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
constexpr int SLOT = 32768, B_OFF = 16384;
#ifndef ABL
#define ABL 0      // timing ablations of mode 0 (results wrong): 1 no vmcnt wait, 2 no barrier, 3 no LDS-DMA in the loop, 4 no fragment reads in the loop
#endif

struct Args {
    const _Float16* A;
    const _Float16* W;
    float* X;
    _Float16* X16;
    int panels, N, K;
    float gate;
};
typedef __attribute__((address_space(1))) char gchar;
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) _Float16 ghalf;
__device__ __forceinline__ void glds16(const gchar* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <typename T>
__device__ __forceinline__ T* uptr(T* ptr) {
    const uint64_t u = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}
#define vhere(v_) ({ asm volatile("" : "+v"(v_)); v_; })
#define FENCE() { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); }
#define OPAQUE(p_) { p_ = uptr(p_); asm volatile("" : "+s"(p_)); }

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_wave128(Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, fk = lane >> 5, hi = lane >> 5, lcol = lane & 31;
    const int K = p.K, KT = K >> 5;
    // workgroup -> tile: each XCD (workgroup b runs on XCD b % 8) walks a contiguous, n-fastest chunk of the tile list
    const int tiles_n = p.N >> 8, ntiles = p.panels * tiles_n;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, q = ntiles >> 3, r = ntiles & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;

    uint32_t fa[2], fb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const uint32_t ch = (uint32_t)((ks * 2 + fk) ^ ((frow >> 1) & 3)) * 16u;
        fa[ks] = (uint32_t)(wm * 128 + frow) * 64u + ch;
        fb[ks] = (uint32_t)B_OFF + (uint32_t)(wn * 128 + frow) * 64u + ch;
    }
    const int prow = tid >> 2;                                    // row of this lane inside a 64-row piece
    const uint32_t pch = (uint32_t)((tid & 3) ^ ((prow >> 1) & 3));
    uint32_t voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) voff[i] = ((uint32_t)(i * 64 + prow) * (uint32_t)K + pch * 8u) * 2u;
    const uint32_t wdst = (uint32_t)wave * 1024u;
    const gchar* Ab = (const gchar*)p.A + (size_t)m0 * K * 2;
    const gchar* Wb = (const gchar*)p.W + (size_t)n0 * K * 2;
#define PIECE(i_, a_, w_, slot_)                                                                              \
    {                                                                                                         \
        if ((i_) < 4) glds16((a_) + vhere(voff[(i_)]), smem + (slot_) + (i_) * 4096 + wdst);                   \
        else glds16((w_) + vhere(voff[(i_) - 4]), smem + (slot_) + B_OFF + ((i_) - 4) * 4096 + wdst);          \
    }
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    h8 FA[2][4], FB[2][4];
#define FRAGS(dst_, slot_, ks_)                                                                                                     \
    {                                                                                                                               \
        _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) FA[dst_][mb] = *reinterpret_cast<const h8*>(smem + (slot_) + fa[ks_] + mb * 2048); \
        _Pragma("unroll") for (int nb = 0; nb < 4; ++nb) FB[dst_][nb] = *reinterpret_cast<const h8*>(smem + (slot_) + fb[ks_] + nb * 2048); \
    }
#define MMS(cur_)                                                                                            \
    {                                                                                                        \
        _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)                                                     \
            _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) acc[mb][nb] = MFMA(FA[cur_][mb], FB[cur_][nb], acc[mb][nb], 0, 0, 0); \
    }
    // interleave request per K step: 16 MFMAs, 8 LDS reads, 4 LDS-DMA pieces
#define SCHED_STEP()                                                                    \
    {                                                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                          \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                          \
        }                                                                               \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                          \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
        }                                                                               \
    }
    // prologue: K tiles 0, 1, 2
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const gchar* a_ = Ab + t * 64;
        const gchar* w_ = Wb + t * 64;
        OPAQUE(a_);
        OPAQUE(w_);
#pragma unroll
        for (int i = 0; i < 8; ++i) PIECE(i, a_, w_, t * SLOT);
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FENCE();
    FRAGS(0, 0, 0);
    // K tile g in slot cur_, g + 1 in nxt_, g + 3 goes to fre_ (the slot of g - 1: every wave's reads of it returned before the barrier
    // of the previous tile).  Boundary: this wave's reads of slot cur_ have returned (lgkmcnt 0), its pieces of K tile g + 1 have
    // landed (12 younger operations may be outstanding: 8 of g + 2, 4 of g + 3), barrier, then the first fragments of g + 1.
// one 16-wide K step, interleaved BY HAND (a single wave has one issue slot: 16 MFMAs, 8 fragment reads and 4 DMA pieces must alternate,
// and the compiler clusters them when left alone): 8 groups of (2 MFMAs on FA / FB[cur_], 1 fragment read of the NEXT step into
// [1 - cur_], 1 DMA piece in the first four groups), each closed by a scheduling fence
#define STEP(cur_, slotn_, ksn_, pf_, a3_, w3_, fre_)                                                                    \
    {                                                                                                                    \
        _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                                               \
            {                                                                                                            \
                const int t0_ = 2 * q_, t1_ = 2 * q_ + 1;                                                                \
                acc[t0_ & 3][t0_ >> 2] = MFMA(FA[cur_][t0_ & 3], FB[cur_][t0_ >> 2], acc[t0_ & 3][t0_ >> 2], 0, 0, 0);   \
                acc[t1_ & 3][t1_ >> 2] = MFMA(FA[cur_][t1_ & 3], FB[cur_][t1_ >> 2], acc[t1_ & 3][t1_ >> 2], 0, 0, 0);   \
            }                                                                                                            \
            if (ABL != 4) {                                                                                              \
                if (q_ < 4) FA[1 - (cur_)][q_] = *reinterpret_cast<const h8*>(smem + (slotn_) + fa[ksn_] + q_ * 2048);   \
                else FB[1 - (cur_)][q_ - 4] = *reinterpret_cast<const h8*>(smem + (slotn_) + fb[ksn_] + (q_ - 4) * 2048); \
            }                                                                                                            \
            if (ABL != 3 && q_ < 4) PIECE((pf_) + q_, a3_, w3_, (fre_) * SLOT);                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
        }                                                                                                                \
    }
#define KTILE(g_, cur_, nxt_, fre_)                                                                          \
    {                                                                                                        \
        const int g3_ = (g_) + 3 < KT ? (g_) + 3 : KT - 1;   /* past the end: re-load the last tile into a slot nobody reads */ \
        const gchar* a3_ = Ab + g3_ * 64;                                                                    \
        const gchar* w3_ = Wb + g3_ * 64;                                                                    \
        OPAQUE(a3_);                                                                                         \
        OPAQUE(w3_);                                                                                         \
        __builtin_amdgcn_s_waitcnt(0xC07F);      /* lgkmcnt(0): fragments (g, 0).  The BUILTIN, not asm: the compiler's own wait        \
                                                    insertion sees it; an asm wait it does not, and with LDS-DMA pending its own wait    \
                                                    before the first MFMA is a full lgkmcnt(0) behind a fragment read it hoisted there */ \
        FENCE();                                                                                             \
        STEP(0, (cur_) * SLOT, 1, 0, a3_, w3_, fre_);                                                        \
        __builtin_amdgcn_s_waitcnt(0xC07F);      /* fragments (g, 1): every read of slot cur_ has returned */ \
        if (ABL != 1 && ABL != 3) __builtin_amdgcn_s_waitcnt(0x0F7C);      /* vmcnt(12) */                    \
        if (ABL != 2) __builtin_amdgcn_s_barrier();                                                          \
        FENCE();                                                                                             \
        STEP(1, (nxt_) * SLOT, 0, 4, a3_, w3_, fre_);                                                        \
    }
    for (int g = 0; g < KT; g += 4) {
        KTILE(g, 0, 1, 3);
        KTILE(g + 1, 1, 2, 0);
        KTILE(g + 2, 2, 3, 1);
        KTILE(g + 3, 3, 0, 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"a"(acc[i][j]));
        return;
    }
    // serial epilogue straight from the accumulators: lane holds column nb * 32 + lcol, rows mb * 32 + 8 (e >> 2) + 4 hi + (e & 3)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int eq = 0; eq < 4; ++eq) {
            float xv[4][4];
            if (MODE == 1) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const size_t o = (size_t)(m0 + wm * 128 + mb * 32 + eq * 8 + hi * 4 + i) * p.N + n0 + wn * 128 + nb * 32 + lcol;
                        xv[nb][i] = p.X[o];
                    }
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const size_t o = (size_t)(m0 + wm * 128 + mb * 32 + eq * 8 + hi * 4 + i) * p.N + n0 + wn * 128 + nb * 32 + lcol;
                    if (MODE == 1) {
                        const float v = __builtin_fmaf(p.gate, acc[mb][nb][eq * 4 + i], xv[nb][i]);
                        p.X[o] = v;
                        p.X16[o] = (_Float16)v;
                    } else {
                        p.X[o] = acc[mb][nb][eq * 4 + i];
                    }
                }
        }
}

// ---- host --------------------------------------------------------------------------------------------------------------------
__global__ void fill_normal(_Float16* d, size_t n, float scale, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t a = (uint32_t)i * 2654435761u + seed, b = (uint32_t)(i >> 32) ^ (seed * 40503u);
        a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; a += b;
        uint32_t c = a * 0x9e3779b9u + 0x7f4a7c15u;
        c ^= c >> 16; c *= 0x7feb352du; c ^= c >> 15; c *= 0x846ca68bu; c ^= c >> 16;
        const float u1 = ((a >> 8) + 1) * (1.0f / 16777217.0f), u2 = (c >> 8) * (1.0f / 16777216.0f);
        d[i] = (_Float16)(scale * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2));
    }
}
#define CK(x) { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } }

template <int MODE>
static int run(const Args& a, int grid, int iters, float* ms_out) {
    const int lds = 4 * SLOT;
    CK(hipFuncSetAttribute((const void*)k_wave128<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_wave128<MODE>, dim3(grid), dim3(256), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_wave128<MODE>, dim3(grid), dim3(256), lds, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / iters;
    return 0;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1024;
    const int K = argc > 2 ? atoi(argv[2]) : 1024;
    const int iters = argc > 3 ? atoi(argv[3]) : 20;
    const int panels = 235, M = panels * 256;
    if (N % 256 != 0 || K % 128 != 0) { fprintf(stderr, "N %% 256, K %% 128\n"); return 1; }
    _Float16 *A, *W, *X16;
    float* X;
    CK(hipMalloc(&A, (size_t)M * K * 2));
    CK(hipMalloc(&W, (size_t)N * K * 2));
    CK(hipMalloc(&X, (size_t)M * N * 4));
    CK(hipMalloc(&X16, (size_t)M * N * 2));
    CK(hipMemset(X, 0, (size_t)M * N * 4));
    fill_normal<<<2048, 256>>>(A, (size_t)M * K, 1.0f, 1u);
    fill_normal<<<2048, 256>>>(W, (size_t)N * K, 0.03125f, 2u);
    CK(hipDeviceSynchronize());
    Args a{A, W, X, X16, panels, N, K, 0.5f};
    const int grid = panels * (N / 256);
    const double flops = 2.0 * M * (double)N * K;
    // correctness: mode 2 (C stored) sampled against fp64
    if (ABL == 0) {
        const int lds = 4 * SLOT;
        CK(hipFuncSetAttribute((const void*)k_wave128<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(k_wave128<2>, dim3(grid), dim3(256), lds, 0, a);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        std::vector<float> c((size_t)M * N);
        std::vector<_Float16> hA((size_t)M * K), hW((size_t)N * K);
        CK(hipMemcpy(c.data(), X, c.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hA.data(), A, hA.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hW.data(), W, hW.size() * 2, hipMemcpyDeviceToHost));
        int bad = 0;
        double worst = 0.0;
        uint32_t rs = 777u;
        auto chk = [&](int r, int cc) {
            double d = 0.0;
            for (int k = 0; k < K; ++k) d += (double)(float)hA[(size_t)r * K + k] * (double)(float)hW[(size_t)cc * K + k];
            const double err = fabs(d - c[(size_t)r * N + cc]);
            if (err > worst) worst = err;
            if (err > 2e-3) ++bad;
        };
        for (int t = 0; t < 20000; ++t) {
            rs = rs * 1664525u + 1013904223u;
            const int r = (int)((rs >> 8) % (uint32_t)M);
            rs = rs * 1664525u + 1013904223u;
            chk(r, (int)((rs >> 8) % (uint32_t)N));
        }
        for (int pnl = 0; pnl < panels; ++pnl)
            for (int tn = 0; tn < N / 256; ++tn) chk(pnl * 256 + (pnl * 7 + tn * 13) % 256, tn * 256 + (pnl * 5 + tn * 3) % 256);
        printf("{\"probe\": \"wave128_gemm\", \"check\": \"C vs fp64\", \"N\": %d, \"K\": %d, \"bad\": %d, \"worst_abs_err\": %.3e}\n", N, K, bad, worst);
        if (bad) return 3;
    }
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep)
        for (int m = 0; m < (ABL ? 1 : 3); ++m) {
            const int rc = m == 0 ? run<0>(a, grid, iters, &ms) : m == 1 ? run<1>(a, grid, iters, &ms) : run<2>(a, grid, iters, &ms);
            if (rc) return rc;
            printf("{\"probe\": \"wave128_gemm\", \"ablation\": %d, \"mode\": %d, \"rep\": %d, \"M\": %d, \"N\": %d, \"K\": %d, \"us\": %.2f, \"tflops\": %.1f}\n", ABL, m, rep, M, N, K,
                   ms * 1e3, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
