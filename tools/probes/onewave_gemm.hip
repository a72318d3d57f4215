// probe (round 6, VERDICT r5 "next" #1): the one in-CU overlap configuration that was never measured.
//   ONE wave per SIMD (256 threads, 512-register budget), wave tile 128 x 64 (the LDS reads per MFMA of the shipped 256 x 256 kernel),
//   TWO accumulator sets per wave (2 x 128 registers), macro tile 256 x 128, PERSISTENT workgroups (one per CU) that keep the operand
//   ring turning across tile seams; the epilogue of tile i (set S) is cut into 16 slices that are issued in the MFMA shadows of the 16
//   K tiles of tile i + 1 (set 1 - S).
// Gate (a): main loop only, workload-like operands, out-proj / FF1 shapes: go only if >= 1.10 PF.
// Gate (b): the same loop + a synthetic memory-bound residual epilogue (fp32 read-modify-write of x + the 16-bit operand copy:
//           640 KB per 256 x 256 of output) overlapped: complete out-proj launch <= 140 us (shipped kernel: 188-212 us).
//   mode 0  main loop only (both sets alternate and are sunk at the end)
//   mode 1  epilogue of tile i overlapped with the K loop of tile i + 1
//   mode 2  the same epilogue code run serially after each tile's K loop (what the overlap has to beat on the same main loop)
//   mode 3  overlapped, WITHOUT the read of x (x = gate * acc, 16-bit copy): stores only -- is the x load's latency the blocker?
//   mode 5  mode 3 with every store aimed at ONE 3 KB region per wave (always L2-resident, no HBM traffic): the instruction stream of
//           the overlapped epilogue without its memory system cost
//   mode 4  overlapped, x updated by global_atomic_add_f32 without return (the L2 does the read-modify-write, nothing comes back to
//           the wave) + the 16-bit store: the traffic of the real epilogue, no load in the MFMA waves' in-order vmcnt queue
// Geometry: LDS ring of 3 K tiles x (256 + 128 rows) x 64 k x 2 B = 144 KB; operands arrive by global_load_lds (lane-linear image,
// 16-byte XOR swizzle on the source chunk, as gemm256.hip); per K tile a wave issues 12 LDS-DMA pieces, 24 ds_read_b128 and 32 MFMAs;
// fragments double-buffered per 16-wide K step; ONE workgroup barrier per K tile ("boundary", in front of the last K step):
//   boundary g:  lgkmcnt(0) (every read of K tile g's slot has returned) -> vmcnt(X) (this wave's pieces of K tile g + 1 have landed)
//                -> s_barrier -> fragments (g + 1, 0) prefetched, pieces of K tile g + 3 issued into the slot of K tile g.
// X = 12 (the pieces of K tile g + 2, issued during the span behind boundary g - 1) + the epilogue's VMEM operations of one span (>= 16).
// Tile walk: workgroup b sits on XCD b % 8; the 32 workgroups of an XCD cover 4 row panels x 8 column tiles (2 MB of A + 2 MB of W
// per XCD and round = its L2), rounds of 32 row panels, bands of 1 024 columns.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/onewave_gemm.hip -o tools/probes/bin/onewave_gemm
// Run:   onewave_gemm <N> <mode> [iters]      (K = 1024, M = 235 x 256 = 60 160 rows)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16

constexpr int BK = 64, KDIM = 1024, KT = KDIM / BK;
constexpr int A_BYTES = 256 * BK * 2, B_BYTES = 128 * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;   // 32 + 16 = 48 KB
constexpr int ROWSTEP = 32 * KDIM * 2;     // bytes between two LDS-DMA pieces of one operand (32 rows)

struct Args {
    const _Float16* A;
    const _Float16* W;
    float* X;
    _Float16* X16;
    int panels, N, nbands;
    float gate;
};

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3); }
typedef __attribute__((address_space(1))) char gchar;          // global address space spelled out: an OPAQUE()d generic pointer
typedef __attribute__((address_space(1))) float gfloat;        // becomes flat_load / flat_store (counted on lgkmcnt as well)
typedef __attribute__((address_space(1))) _Float16 ghalf;
__device__ __forceinline__ void glds16(const gchar* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <typename T>
__device__ __forceinline__ T* uptr(T* ptr) {      // a pointer the compiler can see is wave-uniform (SGPR pair): saddr + 32-bit voffset addressing
    const uint64_t u = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}
// a 32-bit lane offset the compiler must take as given at the USE: its zero-extension then sits in the block of the access, where
// instruction selection can fold "uniform base + zext(offset)" into the saddr form (hoisted out of the block it becomes a 64-bit VALU add)
#define vhere(v_) ({ asm volatile("" : "+v"(v_)); v_; })
#define FENCE() { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); }

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_onewave(Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int bid = blockIdx.x, xcd = bid & 7, sl = bid >> 3, ct = sl & 7, pr = sl >> 3;
    const int frow = lane & 31, fk = lane >> 5, hi = lane >> 5, lcol = lane & 31;
    const int first_panel = xcd * 4 + pr;
    const int Rv = first_panel < p.panels ? (p.panels - first_panel + 31) / 32 : 0;
    const int nt = Rv * p.nbands;
    if (nt == 0) return;

    uint32_t fa[4], fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        fa[ks] = (uint32_t)swz_off(wm * 128 + frow, ks * 2 + fk) * 2u;
        fb[ks] = (uint32_t)A_BYTES + (uint32_t)swz_off(wn * 64 + frow, ks * 2 + fk) * 2u;
    }
    const int srow = tid >> 3;
    uint32_t voff[8];        // byte offset of this lane's 16 bytes inside piece i (32 rows further per piece): the same for A and W (ld = K)
#pragma unroll
    for (int i = 0; i < 8; ++i) voff[i] = (uint32_t)((srow + 32 * i) * KDIM + (((tid & 7) ^ ((srow >> 1) & 7)) << 3)) * 2u;
    const uint32_t wdst = (uint32_t)wave * 1024u;
    // a uniform pointer the compiler must take as given HERE (SGPR pair): without it every piece address of every unrolled K tile is
    // loop-invariant, gets hoisted to the head of the tile and spills (first build of this probe: 256 + 256 registers and scratch)
#define OPAQUE(p_) { asm volatile("" : "+s"(p_)); p_ = uptr(p_); }     // (+ readfirstlane: asm results count as divergent, and a
                                                                        // divergent base pointer costs a 64-bit VALU add per access)

    const gchar* Ab = (const gchar*)(p.A);
    const gchar* Wb = (const gchar*)(p.W);
#define TILE_AT(i_, m0_, n0_)                                   \
    {                                                           \
        const int band_ = (i_) / Rv, r_ = (i_) - band_ * Rv;    \
        m0_ = ((r_ * 8 + xcd) * 4 + pr) * 256;                  \
        n0_ = (band_ * 8 + ct) * 128;                           \
    }
    // one LDS-DMA piece i (0-7: A rows 32 i ..., 8-11: W rows 32 (i - 8) ...) of a K tile whose operand pointers are a_ / w_
#define PIECE(i_, a_, w_, slot_)                                                                                   \
    {                                                                                                              \
        if ((i_) < 8) glds16((a_) + vhere(voff[(i_)]), smem + (slot_) + (i_) * 4096 + wdst);                        \
        else glds16((w_) + vhere(voff[(i_) - 8]), smem + (slot_) + A_BYTES + ((i_) - 8) * 4096 + wdst);              \
    }

    f32x16 acc[2][4][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[s][i][j][e] = 0.0f;
    h8 FA[2][4], FB[2][2];

    // ---- prologue: K tiles 0 and 1 of the first tile, the first three pieces of K tile 2
    int m0c, n0c;
    TILE_AT(0, m0c, n0c);
    const gchar* Ac = Ab + (size_t)m0c * KDIM * 2;
    const gchar* Wc = Wb + (size_t)n0c * KDIM * 2;
    uint32_t s0 = 0, s1 = STAGE_BYTES, s2 = 2 * STAGE_BYTES;     // slots of K tiles (g, g + 1, g + 2) at the head of a tile; g - 1 = s2
#pragma unroll
    for (int i = 0; i < 12; ++i) PIECE(i, Ac, Wc, s0);
#pragma unroll
    for (int i = 0; i < 12; ++i) PIECE(i, Ac + 128, Wc + 128, s1);
#pragma unroll
    for (int i = 0; i < 3; ++i) PIECE(i, Ac + 256, Wc + 256, s2);
    asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FENCE();
#define FRAGS(dst_, slot_, ks_)                                                                                          \
    {                                                                                                                    \
        _Pragma("unroll") for (int mb = 0; mb < 4; ++mb) FA[dst_][mb] = *reinterpret_cast<const h8*>(smem + (slot_) + fa[ks_] + mb * 4096); \
        _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) FB[dst_][nb] = *reinterpret_cast<const h8*>(smem + (slot_) + fb[ks_] + nb * 4096); \
    }
    FRAGS(0, s0, 0);

    // ---- the synthetic residual epilogue of the PREVIOUS tile (set 1 - SET), slice J: rows mb = J >> 2, register quad J & 3
    // The first tile's "previous tile" is a private dummy tile behind the matrix (set 1 starts as zeros: x_dummy += 0), so the loop
    // carries no "is there a previous tile" branch: a branch ends the scheduling region and clusters the MFMAs behind it.
    float xv[2][2][4] = {};  // [parity of the slice][nb][i]
    int pm0 = (p.panels + first_panel) * 256, pn0 = ct * 128;
    const float gate = p.gate;
    uint32_t xoff[2][4];     // element offset of (nb, i) inside a slice: row i + 4 hi of the slice's 8 rows, column nb * 32 + lcol
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i) xoff[nb][i] = (uint32_t)((i + hi * 4) * p.N + nb * 32 + lcol) * 4u;      // bytes (fp32)
    uint32_t hoff[2][4];     // the same for the 16-bit copy
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i) hoff[nb][i] = xoff[nb][i] >> 1;
// elements e = nb * 4 + i in [E0_, E1_) of slice J_ (8 per slice); tile coordinates (tm_, tn_).
// The x loads are inline asm with HAND-COUNTED waits: a register load the compiler can see gets the compiler's wait at its use, and
// with LDS-DMA pieces outstanding (global_load_lds is a FLAT instruction that touches LDS and memory: "pending flat") that wait is
// always vmcnt(0) -- one full drain of the operand ring per K tile (second build of this probe).  Every K step issues 9 VMEM
// operations (3 pieces, 2 x loads, 4 stores); a value loaded in step s is used in step s + 4: at least 27 operations later.
#define EPI_LOAD(J_, E0_, E1_, tm_, tn_)                                                                                 \
    {                                                                                                                    \
        const gchar* xt_ = (const gchar*)(p.X + (size_t)((tm_) + wm * 128 + (J_) * 8) * p.N + (tn_) + wn * 64);          \
        OPAQUE(xt_);                                                                                                     \
        _Pragma("unroll") for (int e_ = (E0_); e_ < (E1_); ++e_)                                                         \
            asm volatile("global_load_dword %0, %1, %2" : "=v"(xv[(J_) & 1][e_ >> 2][e_ & 3]) : "v"(xoff[e_ >> 2][e_ & 3]), "s"(xt_)); \
    }
#define EPI_USE(SET_, J_, E0_, E1_, WAIT_)                                                                               \
    {                                                                                                                    \
        const size_t ro_ = (size_t)(pm0 + wm * 128 + (MODE == 5 ? 0 : (J_) * 8)) * p.N + pn0 + wn * 64;                   \
        gchar* xt_ = (gchar*)(p.X + ro_);                                                                                \
        gchar* ht_ = (gchar*)(p.X16 + ro_);                                                                              \
        OPAQUE(xt_);                                                                                                     \
        OPAQUE(ht_);                                                                                                     \
        if (WAIT_) asm volatile("s_waitcnt vmcnt(27)");                                                                  \
        _Pragma("unroll") for (int e_ = (E0_); e_ < (E1_); ++e_) {                                                       \
            float av_;     /* read from the accumulator file HERE: left to the compiler, the whole finished set is copied to VGPRs at   \
                              the head of the tile (128 registers) and everything else spills */                                  \
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(av_) : "a"(acc[SET_][(J_) >> 2][e_ >> 2][((J_) & 3) * 4 + (e_ & 3)]), "v"(xv[(J_) & 1][e_ >> 2][e_ & 3])); \
            const float o_ = (MODE == 3 || MODE == 4 || MODE == 5) ? gate * av_ : __builtin_fmaf(gate, av_, xv[(J_) & 1][e_ >> 2][e_ & 3]); \
            if (MODE == 4) asm volatile("global_atomic_add_f32 %0, %1, %2" ::"v"(xoff[e_ >> 2][e_ & 3]), "v"(o_), "s"(xt_) : "memory"); \
            else *(gfloat*)(xt_ + vhere(xoff[e_ >> 2][e_ & 3])) = o_;                                                    \
            *(ghalf*)(ht_ + vhere(hoff[e_ >> 2][e_ & 3])) = (_Float16)o_;                                                \
        }                                                                                                                \
    }
#define MMS(SET_, cur_, ZERO_)                                                                                           \
    {                                                                                                                    \
        _Pragma("unroll") for (int mb = 0; mb < 4; ++mb)                                                                 \
            _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                                           \
                if (ZERO_) {                                                                                             \
                    const f32x16 z_ = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   \
                    acc[SET_][mb][nb] = MFMA(FA[cur_][mb], FB[cur_][nb], z_, 0, 0, 0);                                   \
                } else {                                                                                                 \
                    acc[SET_][mb][nb] = MFMA(FA[cur_][mb], FB[cur_][nb], acc[SET_][mb][nb], 0, 0, 0);                    \
                }                                                                                                        \
            }                                                                                                            \
    }
    // interleave request: 8 x (1 MFMA, then up to 2 VALU instructions, 1 LDS read, 2 VMEM operations)
#define SCHED_STEP()                                                                    \
    {                                                                                   \
        _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                          \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                          \
            __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);                          \
        }                                                                               \
    }

    // K tile J of the tile in set SET_; ring slots of K tiles J - 1 / J / J + 1 are sp_ / sr_ / sn_
#define KTILE(SET_, J_, sp_, sr_, sn_)                                                                                           \
    {                                                                                                                            \
        const gchar* a2_ = (J_) + 2 < KT ? Ac + ((J_) + 2) * 128 : An + ((J_) + 2 - KT) * 128;                                    \
        const gchar* w2_ = (J_) + 2 < KT ? Wc + ((J_) + 2) * 128 : Wn + ((J_) + 2 - KT) * 128;                                    \
        const gchar* a3_ = (J_) + 3 < KT ? Ac + ((J_) + 3) * 128 : An + ((J_) + 3 - KT) * 128;                                    \
        const gchar* w3_ = (J_) + 3 < KT ? Wc + ((J_) + 3) * 128 : Wn + ((J_) + 3 - KT) * 128;                                    \
        if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) {     /* the finished set stays in the accumulator file (see EPI_USE) */        \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                     \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) asm volatile("" : "+a"(acc[1 - (SET_)][i_][j_]));               \
        }                                                                                                                        \
        OPAQUE(a2_);                                                                                                             \
        OPAQUE(w2_);                                                                                                             \
        OPAQUE(a3_);                                                                                                             \
        OPAQUE(w3_);                                                                                                             \
        _Pragma("unroll") for (int ks = 0; ks < 3; ++ks) {                                                                       \
            FRAGS((ks + 1) & 1, sr_, ks + 1);                                                                                    \
            _Pragma("unroll") for (int i = 0; i < 3; ++i) PIECE(3 + 3 * ks + i, a2_, w2_, sp_);                                  \
            if (MODE == 1) {     /* slice J of the previous tile (loaded one K tile ago), loads of the next slice */            \
                EPI_USE(1 - (SET_), J_, 2 * ks, 2 * ks + 2, true);                                                                     \
                if ((J_) + 1 < KT) EPI_LOAD((J_) + 1, 2 * ks, 2 * ks + 2, pm0, pn0)                                              \
                else EPI_LOAD(0, 2 * ks, 2 * ks + 2, m0c, n0c)                                                                   \
            }                                                                                                                    \
            if (MODE == 3 || MODE == 4 || MODE == 5) EPI_USE(1 - (SET_), J_, 2 * ks, 2 * ks + 2, false);                                      \
            MMS(SET_, ks & 1, ((J_) == 0 && ks == 0));                                                                           \
            SCHED_STEP();                                                                                                        \
            FENCE();                                                                                                             \
        }                                                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");     /* 12 pieces + 24 epilogue operations per span */   \
        else if (MODE == 3 || MODE == 4 || MODE == 5) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");     /* 12 + 16 */                     \
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                                                                   \
        __builtin_amdgcn_s_barrier();                                                                                            \
        FENCE();                                                                                                                 \
        FRAGS(0, sn_, 0);                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) PIECE(i, a3_, w3_, sr_);                                                   \
        if (MODE == 1) {                                                                                                         \
            EPI_USE(1 - (SET_), J_, 6, 8, true);                                                                                       \
            if ((J_) + 1 < KT) EPI_LOAD((J_) + 1, 6, 8, pm0, pn0)                                                                \
            else EPI_LOAD(0, 6, 8, m0c, n0c)                                                                                     \
        }                                                                                                                        \
        if (MODE == 3 || MODE == 4 || MODE == 5) EPI_USE(1 - (SET_), J_, 6, 8, false);                                                        \
        MMS(SET_, 1, false);                                                                                                     \
        SCHED_STEP();                                                                                                            \
        FENCE();                                                                                                                 \
    }
#define TILE(SET_)                                             \
    {                                                          \
        KTILE(SET_, 0, s2, s0, s1);                            \
        KTILE(SET_, 1, s0, s1, s2);                            \
        KTILE(SET_, 2, s1, s2, s0);                            \
        KTILE(SET_, 3, s2, s0, s1);                            \
        KTILE(SET_, 4, s0, s1, s2);                            \
        KTILE(SET_, 5, s1, s2, s0);                            \
        KTILE(SET_, 6, s2, s0, s1);                            \
        KTILE(SET_, 7, s0, s1, s2);                            \
        KTILE(SET_, 8, s1, s2, s0);                            \
        KTILE(SET_, 9, s2, s0, s1);                            \
        KTILE(SET_, 10, s0, s1, s2);                           \
        KTILE(SET_, 11, s1, s2, s0);                           \
        KTILE(SET_, 12, s2, s0, s1);                           \
        KTILE(SET_, 13, s0, s1, s2);                           \
        KTILE(SET_, 14, s1, s2, s0);                           \
        KTILE(SET_, 15, s2, s0, s1);                           \
        if (MODE == 0) {                                       \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)   \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) asm volatile("" ::"a"(acc[1 - (SET_)][i_][j_])); \
        }                                                      \
        {                                                      \
            const uint32_t t_ = s0;                            \
            s0 = s1;                                           \
            s1 = s2;                                           \
            s2 = t_;                                           \
        }                                                      \
    }
    // the epilogue of a finished set, serial (mode 2 after every tile, mode 1 after the last one)
#define EPI_SERIAL(SET_)                                                             \
    {                                                                                \
        _Pragma("unroll") for (int J = 0; J < 16; J += 2) {      /* batches of two slices: 16 loads, one wait, 32 stores */ \
            EPI_LOAD(J, 0, 8, pm0, pn0);                                             \
            EPI_LOAD(J + 1, 0, 8, pm0, pn0);                                         \
            asm volatile("s_waitcnt vmcnt(0)");                                      \
            EPI_USE(SET_, J, 0, 8, false);                                           \
            EPI_USE(SET_, J + 1, 0, 8, false);                                       \
        }                                                                            \
    }

    for (int it = 0; it < nt; it += 2) {
        int m0n, n0n;
        {
            const int inx = it + 1 < nt ? it + 1 : it;
            TILE_AT(inx, m0n, n0n);
        }
        const gchar* An = Ab + (size_t)m0n * KDIM * 2;
        const gchar* Wn = Wb + (size_t)n0n * KDIM * 2;
        TILE(0);
        if (MODE != 5) {
            pm0 = m0c;
            pn0 = n0c;
        }
        if (MODE == 2) EPI_SERIAL(0);
        if (it + 1 < nt) {
        m0c = m0n;
        n0c = n0n;
        Ac = An;
        Wc = Wn;
        {
            const int inx = it + 2 < nt ? it + 2 : it + 1;
            TILE_AT(inx, m0n, n0n);
        }
        {
            const gchar* An = Ab + (size_t)m0n * KDIM * 2;
            const gchar* Wn = Wb + (size_t)n0n * KDIM * 2;
            TILE(1);
            if (MODE != 5) {
                pm0 = m0c;
                pn0 = n0c;
            }
            if (MODE == 2) EPI_SERIAL(1);
            m0c = m0n;
            n0c = n0n;
            Ac = An;
            Wc = Wn;
        }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) {
        if ((nt & 1) == 1) EPI_SERIAL(0) else EPI_SERIAL(1)
    }
    if (MODE == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" ::"a"(acc[s][i][j]));
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
__global__ void fill_normal_f32(float* d, size_t n, float scale, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t a = (uint32_t)i * 2654435761u + seed;
        a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
        uint32_t c = a * 0x9e3779b9u + 0x7f4a7c15u;
        c ^= c >> 16; c *= 0x7feb352du; c ^= c >> 15; c *= 0x846ca68bu; c ^= c >> 16;
        const float u1 = ((a >> 8) + 1) * (1.0f / 16777217.0f), u2 = (c >> 8) * (1.0f / 16777216.0f);
        d[i] = scale * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
    }
}
#define CK(x) { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } }

template <int MODE>
static int run(const Args& a, int iters, float* ms_out) {
    const int lds = 3 * STAGE_BYTES;
    CK(hipFuncSetAttribute((const void*)k_onewave<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_onewave<MODE>, dim3(256), dim3(256), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_onewave<MODE>, dim3(256), dim3(256), lds, 0, a);
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
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const int iters = argc > 3 ? atoi(argv[3]) : 20;
    const int panels = 235, M = panels * 256;
    if (N % 1024 != 0) { fprintf(stderr, "N must be a multiple of 1024\n"); return 1; }
    _Float16 *A, *W, *X16;
    float* X;
    CK(hipMalloc(&A, (size_t)M * KDIM * 2));
    CK(hipMalloc(&W, (size_t)N * KDIM * 2));
    const size_t MX = (size_t)(panels + 32) * 256;          // + one private dummy tile row per workgroup slot (see the kernel)
    CK(hipMalloc(&X, MX * N * 4));
    CK(hipMalloc(&X16, MX * N * 2));
    CK(hipMemset(X, 0, MX * N * 4));
    fill_normal<<<2048, 256>>>(A, (size_t)M * KDIM, 1.0f, 1u);
    fill_normal<<<2048, 256>>>(W, (size_t)N * KDIM, 0.03125f, 2u);
    fill_normal_f32<<<2048, 256>>>(X, (size_t)M * N, 1.0f, 3u);
    CK(hipMemset(X16, 0, MX * N * 2));
    CK(hipDeviceSynchronize());
    Args a{A, W, X, X16, panels, N, N / 1024, 0.5f};
    const double flops = 2.0 * M * (double)N * KDIM;

    // correctness of the ring / the two sets / the overlapped epilogue: one launch of mode 1 and of mode 2 on a copy of x, sampled against fp64
    int bad_total = 0;
    double worst = 0.0;
    for (int vm : {1, 2, 4}) {
        std::vector<float> x0((size_t)M * N);
        CK(hipMemcpy(x0.data(), X, x0.size() * 4, hipMemcpyDeviceToHost));
        const int lds = 3 * STAGE_BYTES;
        if (vm == 1) {
            CK(hipFuncSetAttribute((const void*)k_onewave<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(k_onewave<1>, dim3(256), dim3(256), lds, 0, a);
        } else if (vm == 2) {
            CK(hipFuncSetAttribute((const void*)k_onewave<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(k_onewave<2>, dim3(256), dim3(256), lds, 0, a);
        } else {
            CK(hipFuncSetAttribute((const void*)k_onewave<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL(k_onewave<4>, dim3(256), dim3(256), lds, 0, a);
        }
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        std::vector<float> x1((size_t)M * N);
        std::vector<_Float16> hA((size_t)M * KDIM), hW((size_t)N * KDIM), h16((size_t)M * N);
        CK(hipMemcpy(x1.data(), X, x1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hA.data(), A, hA.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hW.data(), W, hW.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h16.data(), X16, h16.size() * 2, hipMemcpyDeviceToHost));
        int bad = 0;
        uint32_t rs = 12345u + vm;
        for (int t = 0; t < 20000; ++t) {
            rs = rs * 1664525u + 1013904223u;
            const int r = (int)((rs >> 8) % (uint32_t)M);
            rs = rs * 1664525u + 1013904223u;
            const int c = (int)((rs >> 8) % (uint32_t)N);
            double d = 0.0;
            for (int k = 0; k < KDIM; ++k) d += (double)(float)hA[(size_t)r * KDIM + k] * (double)(float)hW[(size_t)c * KDIM + k];
            const double want = x0[(size_t)r * N + c] + 0.5 * d;
            const double err = fabs(want - x1[(size_t)r * N + c]);
            const double err16 = fabs((double)(float)h16[(size_t)r * N + c] - x1[(size_t)r * N + c]);
            if (err > worst) worst = err;
            if (vm == 4 && t < 6) fprintf(stderr, "mode4 sample r=%d c=%d x0=%.5f delta=%.5f got=%.5f\n", r, c, x0[(size_t)r * N + c], 0.5 * d, x1[(size_t)r * N + c]);
            if (err > 2e-3 || (vm != 4 && err16 > 4e-3 * (1.0 + fabs(want)))) ++bad;     // (mode 4's 16-bit copy holds the update only)
        }
        // every row panel / column tile corner too (the tile walk covers the whole matrix exactly once)
        for (int pnl = 0; pnl < panels; ++pnl)
            for (int tn = 0; tn < N / 128; ++tn) {
                const int r = pnl * 256 + (pnl * 7 + tn * 13) % 256, c = tn * 128 + (pnl * 5 + tn * 3) % 128;
                double d = 0.0;
                for (int k = 0; k < KDIM; ++k) d += (double)(float)hA[(size_t)r * KDIM + k] * (double)(float)hW[(size_t)c * KDIM + k];
                const double want = x0[(size_t)r * N + c] + 0.5 * d;
                if (fabs(want - x1[(size_t)r * N + c]) > 2e-3) ++bad;
            }
        printf("{\"probe\": \"onewave_gemm\", \"check_mode\": %d, \"N\": %d, \"bad\": %d, \"worst_abs_err\": %.3e}\n", vm, N, bad, worst);
        bad_total += bad;
    }
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        for (int m = 0; m < 6; ++m) {
            if (mode >= 0 && mode != m && mode != 9) continue;
            int rc = m == 0 ? run<0>(a, iters, &ms) : m == 1 ? run<1>(a, iters, &ms) : m == 2 ? run<2>(a, iters, &ms) : m == 3 ? run<3>(a, iters, &ms) : m == 4 ? run<4>(a, iters, &ms) : run<5>(a, iters, &ms);
            if (rc) return rc;
            const double bytes = m == 0 ? 0.0 : (double)M * N * (m == 5 ? 0 : m == 3 ? 4 + 2 : 4 + 4 + 2) + (double)M * KDIM * 2;
            printf("{\"probe\": \"onewave_gemm\", \"mode\": %d, \"rep\": %d, \"M\": %d, \"N\": %d, \"K\": %d, \"us\": %.2f, \"tflops\": %.1f, \"epilogue_alg_tbs\": %.2f}\n",
                   m, rep, M, N, KDIM, ms * 1e3, flops / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e12);
        }
    }
    return bad_total ? 3 : 0;
}
