// Flash attention (non-causal, key-padding by per-batch prefix length) for gfx950, head_dim = 64.
// Reference semantics: mx.fast.scaled_dot_product_attention(q, k, v, scale, mask) at dit.py:166.
//
// Layout trick (all softmax state stays lane-local, no LDS round trip for P):
//   S^T = K * Q^T  via v_mfma_f32_32x32x16_bf16 with A = K tile rows (keys), B = Q rows (queries);
//         C layout => lane (l&31) owns ONE query column, its 16 accumulators are 16 of the 32 keys
//         of the block (the other 16 live in lane l^32).  Row max / row sum = in-lane reduction +
//         one exchange with lane^32.
//   O^T = V^T * P^T with A = V^T rows (dims) and B = P^T: the MFMA "k" index only has to be the SAME
//         key for A and B, so P is fed straight from the S accumulators (keys {0-3,8-11}+4*hi per
//         16-key step) and V^T is read from LDS with the matching key permutation.  V is kept
//         transposed in HBM ([b*h][64][npad], written by the QKV GEMM epilogue).
//   O^T's C layout again has query = lane&31, so the online-softmax rescale is a per-lane multiply.
//
// Block = 4 waves x 32 queries = 128 queries; KV tile = 64 keys, double-buffered in LDS (padded rows:
// K 144 B, V^T 136 B => conflict-free ds_read_b128 / ds_read_b64).  bf16x3 mode (HP) carries hi/lo
// for Q, K, V and splits P in registers: 3 MFMAs per product.
#include "attention.hpp"

#ifndef F5_LAB
#define F5_LAB 0
#endif
namespace F5_NS {

#define KLD 72   // K  tile row stride in elements (144 B)
#define VLD 68   // V^T tile row stride in elements (136 B)

// workgroup -> (query block, batch*head).  Workgroups are dealt to the 8 XCDs round-robin by their linear id, and every query
// block of a head streams that head's whole K / V^T: with a plain (x = query block, y = head) numbering the query blocks of
// a head land on different XCDs and K / V^T is fetched into several private L2s (FETCH_SIZE 1.12 GB per launch at 64 x 16 x
// 937 with 256-query blocks, against 0.37 GB of q + k + v; tools/gpu_pmc_ops.sh).  The launcher therefore uses a 1-D grid of
// 8 * ceil(B*H / 8) * nqb workgroups: XCD = id & 7, and on one XCD consecutive workgroups walk the query blocks of one head
// before moving to the next head (measured +3.5-6 % on the large-grid kernel, tools/attn_prio_bench.py).  A 2-D grid
// (f5_attn_variant bit 2, A/B only) keeps the plain numbering.
__device__ __forceinline__ bool attn_block_map(const F5AttnArgs& p, int qrows, int& bh, int& qblk) {
    if (gridDim.y != 1) {
        bh = blockIdx.y;
        qblk = blockIdx.x;
        return true;
    }
    const int nqb = (p.seq_len + qrows - 1) / qrows;
    const int s = blockIdx.x >> 3;
    qblk = s % nqb;
    bh = (s / nqb) * 8 + (blockIdx.x & 7);
    return bh < p.B * p.H;
}
#if F5_LAB   // round-1 register-staged kernel (f5_debug_set_attn_version 1)
template <bool HP>
__global__ __launch_bounds__(256) void f5_attn_kernel(F5AttnArgs p) {
    constexpr int NP = HP ? 2 : 1;
    __shared__ __attribute__((aligned(16))) op16_t sK[2][NP][64 * KLD];
    __shared__ __attribute__((aligned(16))) op16_t sV[2][NP][64 * VLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, lq = lane & 31;
    int bh, qblk;
    if (!attn_block_map(p, 128, bh, qblk)) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qblk * 128 + wave * 32;
    const int kvlen = p.kv_len ? p.kv_len[b] : p.seq_len;
    const int ntile = (kvlen + 63) >> 6;
    const size_t rowbase = (size_t)b * p.seq_len;

    // Q fragments (B operand): query row lq, dims ks*16 + hi*8 .. +8
    op16x8 qf[NP][4];
    {
        int qr = q0 + lq;
        if (qr > p.seq_len - 1) qr = p.seq_len - 1;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                qf[pp][ks] = *reinterpret_cast<const op16x8*>(p.qk[pp] + (rowbase + qr) * p.ldqk + h * 64 + ks * 16 + hi * 8);
    }

    // staging: 2 chunks of K and 2 chunks of V^T per thread (per precision part)
    u32x4 rk[NP][2], rv[NP][2];
    auto load_tile = [&](int j) {
        const int key0 = j * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qd = tid + 256 * i;
            const int r = qd >> 3, c = qd & 7;
            int key = key0 + r;
            if (key > p.seq_len - 1) key = p.seq_len - 1;
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                rk[pp][i] = *reinterpret_cast<const u32x4*>(p.qk[pp] + (rowbase + key) * p.ldqk + p.dmodel + h * 64 + c * 8);
                rv[pp][i] = *reinterpret_cast<const u32x4*>(p.vt[pp] + ((size_t)bh * 64 + r) * p.npad + key0 + c * 8);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int qd = tid + 256 * i;
            const int r = qd >> 3, c = qd & 7;
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                *reinterpret_cast<u32x4*>(&sK[buf][pp][r * KLD + c * 8]) = rk[pp][i];
                u32x2 lo2, hi2;
                lo2[0] = rv[pp][i][0]; lo2[1] = rv[pp][i][1];
                hi2[0] = rv[pp][i][2]; hi2[1] = rv[pp][i][3];
                *reinterpret_cast<u32x2*>(&sV[buf][pp][r * VLD + c * 8]) = lo2;
                *reinterpret_cast<u32x2*>(&sV[buf][pp][r * VLD + c * 8 + 4]) = hi2;
            }
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        o[0][e] = 0.0f;
        o[1][e] = 0.0f;
    }
    float m_run = -INFINITY, l_run = 0.0f;
    const float c2 = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int j = 0; j < ntile; ++j) {
        const int cur = j & 1;
        if (j + 1 < ntile) load_tile(j + 1);

        // ---- S^T = K Q^T -------------------------------------------------------------------
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = (kb * 32 + lq) * KLD + ks * 16 + hi * 8;
                const op16x8 a = *reinterpret_cast<const op16x8*>(&sK[cur][0][off]);
                s[kb] = F5_MFMA32(a, qf[0][ks], s[kb], 0, 0, 0);
                if (HP) {
                    const op16x8 al = *reinterpret_cast<const op16x8*>(&sK[cur][NP - 1][off]);
                    s[kb] = F5_MFMA32(al, qf[0][ks], s[kb], 0, 0, 0);
                    s[kb] = F5_MFMA32(a, qf[NP - 1][ks], s[kb], 0, 0, 0);
                }
            }
        }

        // ---- online softmax (per query column = per lane) ---------------------------------
        const int key0 = j * 64;
        if (key0 + 64 > kvlen) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= kvlen) s[kb][r] = -INFINITY;
                }
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[kb][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
        const float mc = m_new * c2;
        m_run = m_new;
        float psum = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[kb][r] * c2 - mc);
                s[kb][r] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            o[0][e] *= alpha;
            o[1][e] *= alpha;
        }

        // ---- O^T += V^T P^T ----------------------------------------------------------------
#pragma unroll
        for (int ks4 = 0; ks4 < 4; ++ks4) {
            const int kb = ks4 >> 1, sp = ks4 & 1;
            uint32_t pw[4], pwl[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p0 = s[kb][8 * sp + 2 * e], p1 = s[kb][8 * sp + 2 * e + 1];
                pw[e] = f5_pack2_bounded(p0, p1);
                if (HP) pwl[e] = f5_pack2_lo(p0, p1);
            }
            const op16x8 pb = __builtin_bit_cast(op16x8, u32x4{pw[0], pw[1], pw[2], pw[3]});
            op16x8 pbl = pb;
            if (HP) pbl = __builtin_bit_cast(op16x8, u32x4{pwl[0], pwl[1], pwl[2], pwl[3]});
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int off = (db * 32 + lq) * VLD + ks4 * 16 + 4 * hi;
                const op16x4 v0 = *reinterpret_cast<const op16x4*>(&sV[cur][0][off]);
                const op16x4 v1 = *reinterpret_cast<const op16x4*>(&sV[cur][0][off + 8]);
                const op16x8 a = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                o[db] = F5_MFMA32(a, pb, o[db], 0, 0, 0);
                if (HP) {
                    const op16x4 w0 = *reinterpret_cast<const op16x4*>(&sV[cur][NP - 1][off]);
                    const op16x4 w1 = *reinterpret_cast<const op16x4*>(&sV[cur][NP - 1][off + 8]);
                    const op16x8 al = __builtin_shufflevector(w0, w1, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[db] = F5_MFMA32(al, pb, o[db], 0, 0, 0);
                    o[db] = F5_MFMA32(a, pbl, o[db], 0, 0, 0);
                }
            }
        }

        if (j + 1 < ntile) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- normalise and store O[token][h*64 + d] ---------------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + lq;
    if (qr < p.seq_len) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = db * 32 + 8 * rg + 4 * hi;
                const float v0 = o[db][rg * 4 + 0] * inv, v1 = o[db][rg * 4 + 1] * inv;
                const float v2 = o[db][rg * 4 + 2] * inv, v3 = o[db][rg * 4 + 3] * inv;
                const size_t off = (rowbase + qr) * p.ldo + h * 64 + d;
                *reinterpret_cast<u32x2*>(p.out[0] + off) = u32x2{f5_pack2_bounded(v0, v1), f5_pack2_bounded(v2, v3)};
                if (HP && p.out[1])
                    *reinterpret_cast<u32x2*>(p.out[1] + off) = u32x2{f5_pack2_lo(v0, v1), f5_pack2_lo(v2, v3)};
            }
    }
}

#endif  // F5_LAB
// =================================================================================================
// v2: same math / layouts, but K and V^T tiles go HBM -> LDS with global_load_lds into a ring of NST
// tiles (3 for bf16: two tiles in flight while one is consumed; 2 for bf16x3), counted vmcnt, ONE barrier per
// tile, no staging VGPRs (=> 3 workgroups per CU for bf16).  LDS images are lane-linear 128-byte rows, so the
// 16-byte XOR swizzle chunk ^= (row>>1)&7 is applied to the SOURCE address and again on every read.  The
// O rescale is skipped when no lane's running max moved (exact: alpha == 1 for every lane).
// =================================================================================================
// O^T accumulators of one 32-query block -> MX-fp8: the 64 head dims are two MX blocks (db = 0, 1); a lane holds 16 values of
// each (the other 16 sit in lane ^ 32), 4 consecutive d per accumulator row group -> one 4-byte store each.
__device__ __forceinline__ void attn_store_f8(const F5AttnArgs& p, const f32x16 (&o)[2], float inv, size_t row, int h, int hi) {
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        float am = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) am = fmaxf(am, fabsf(o[db][e] * inv));
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        const int e8 = f5_mx_scale_byte(am);
        const float sc = inv * f5_mx_inv_scale(e8);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int d = db * 32 + 8 * rg + 4 * hi;
            *reinterpret_cast<uint32_t*>(p.out8 + row * p.ldo8 + h * 64 + d) =
                f5_pack4_fp8(o[db][rg * 4 + 0] * sc, o[db][rg * 4 + 1] * sc, o[db][rg * 4 + 2] * sc, o[db][rg * 4 + 3] * sc);
        }
        if (hi == 0) p.out8s[row * (size_t)(p.dmodel >> 5) + h * 2 + db] = (uint8_t)e8;
    }
}

// p = exp2(s * c2 - mc) for a 16-register accumulator block, in place, returning the partial row sum: the scale / subtract and the
// sum run as packed v_pk_fma_f32 / v_pk_add_f32 (two values per VALU issue; MFMA and VALU time add on a SIMD, so every VALU
// instruction saved is kernel time, tools/probes/coissue.hip)
typedef float attn_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ attn_f32x2 attn_exp_block(f32x16& s, float c2, float mc, attn_f32x2 sum2) {
    const attn_f32x2 c2v = {c2, c2}, mcv = {mc, mc};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        attn_f32x2 t = {s[r], s[r + 1]};
        t = t * c2v - mcv;
        t[0] = __builtin_amdgcn_exp2f(t[0]);
        t[1] = __builtin_amdgcn_exp2f(t[1]);
        s[r] = t[0];
        s[r + 1] = t[1];
        sum2 += t;
    }
    return sum2;
}

// The compiler does not see hand-counted `s_waitcnt vmcnt(N)` (inline asm).  Q fragments loaded from global memory before a tile
// loop are first USED inside the loop, so the compiler's own wait for them lands in the loop body -- as vmcnt(0) on every
// iteration, draining the K / V^T prefetch each time.  A use it can see, placed before the loop, keeps that wait in the prologue.
#define ATTN_PIN_Q(qf_, n0_, n1_)                                                                            \
    _Pragma("unroll") for (int a_ = 0; a_ < (n0_); ++a_)                                                     \
        _Pragma("unroll") for (int b_ = 0; b_ < (n1_); ++b_) asm volatile("" ::"v"((qf_)[a_][b_]));

__device__ __forceinline__ void attn_glds16(const op16_t* gptr, op16_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0);
}
__device__ __forceinline__ int attn_swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }
// K tile rows are stored PERMUTED: LDS row i of a 32-key block holds key (i&3) + 4*(i>>3) + 16*((i>>2)&1).  In the
// S^T = K Q^T accumulator layout lane-half hi then owns the 16 CONSECUTIVE keys 16*hi + r (r = register index), so
// the P operand of a 16-key MFMA step is 8 consecutive keys and the matching V^T fragment is ONE ds_read_b128
// (no bank conflicts, no register shuffles) instead of two ds_read_b64 halves.
__device__ __forceinline__ int attn_kperm(int i) { return (i & 32) | ((i & 3) + 4 * ((i >> 3) & 3) + 16 * ((i >> 2) & 1)); }

// ABL = timing-only ablation (results are wrong unless ABL == 0): 1 no exp2, 2 no barrier/vmcnt wait, 3 no PV MFMAs,
// 4 no S MFMAs, 5 no softmax VALU at all, 6 no K/V loads after the prologue, 7 no LDS fragment reads (debug hook f5_debug_set_attn_ablation; CDNA4 guide: ablate before optimising)
template <bool HP, int ABL>
__global__ __launch_bounds__(256, HP ? 2 : 3) void f5_attn2_kernel(F5AttnArgs p) {
    constexpr int NP = HP ? 2 : 1;
    constexpr int NST = HP ? 2 : 3;
    constexpr int TILE = 64 * 64;                       // elements per K or V^T tile image
    __shared__ __attribute__((aligned(16))) op16_t smem[NST * NP * 2 * TILE];   // [stage][part][K | V^T][64*64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    int bh, qblk;
    if (!attn_block_map(p, 128, bh, qblk)) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qblk * 128 + wave * 32;
    const int kvlen = p.kv_len ? p.kv_len[b] : p.seq_len;
    const int ntile = (kvlen + 63) >> 6;
    const size_t rowbase = (size_t)b * p.seq_len;

    op16x8 qf[NP][4];
    {
        int qr = q0 + lq;
        if (qr > p.seq_len - 1) qr = p.seq_len - 1;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                qf[pp][ks] = *reinterpret_cast<const op16x8*>(p.qk[pp] + (rowbase + qr) * p.ldqk + h * 64 + ks * 16 + hi * 8);
    }

    // staging pointers (advanced by one KV tile per issue: tiles are issued in increasing order) + wave-uniform LDS offsets
    const op16_t* kptr[NP][2];
    const op16_t* vptr[NP][2];
    int krow[2], kcol[2], ldsoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q_ = i * 256 + tid;
        const int srow = q_ >> 3;
        const int schunk = (q_ & 7) ^ ((srow >> 1) & 7);
        krow[i] = attn_kperm(srow);
        kcol[i] = p.dmodel + h * 64 + schunk * 8;
        ldsoff[i] = (i * 256 + wave * 64) * 8;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            kptr[pp][i] = p.qk[pp] + (rowbase + krow[i]) * p.ldqk + kcol[i];
            vptr[pp][i] = p.vt[pp] + ((size_t)bh * 64 + srow) * p.npad + schunk * 8;
        }
    }
    const size_t kstep = (size_t)64 * p.ldqk;
#define A2_ISSUE(j_)                                                                                         \
    {                                                                                                        \
        op16_t* st_ = smem + ((j_) % NST) * (NP * 2 * TILE);                                                 \
        const bool tail_ = ((j_) * 64 + 63) > p.seq_len - 1;                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
            _Pragma("unroll") for (int pp = 0; pp < NP; ++pp) {                                              \
                const op16_t* ks_ = kptr[pp][i];                                                             \
                if (tail_) {                                                                                 \
                    int key_ = (j_) * 64 + krow[i];                                                          \
                    if (key_ > p.seq_len - 1) key_ = p.seq_len - 1;                                          \
                    ks_ = p.qk[pp] + (rowbase + key_) * p.ldqk + kcol[i];                                    \
                }                                                                                            \
                attn_glds16(ks_, st_ + (pp * 2) * TILE + ldsoff[i]);                                         \
                attn_glds16(vptr[pp][i], st_ + (pp * 2 + 1) * TILE + ldsoff[i]);                             \
                kptr[pp][i] += kstep;                                                                        \
                vptr[pp][i] += 64;                                                                           \
            }                                                                                                \
        }                                                                                                    \
    }

    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        o[0][e] = 0.0f;
        o[1][e] = 0.0f;
    }
    float m_run = -INFINITY, l_run = 0.0f;
    const float c2 = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;

    A2_ISSUE(0);
    if (NST == 3 && ntile > 1) A2_ISSUE(1);
    ATTN_PIN_Q(qf, NP, 4);

    for (int j = 0; j < ntile; ++j) {
        // wait for tile j (own loads), then make every wave's part visible; tile j+1 may stay in flight (NST == 3)
        if (ABL != 2) {
            if (ABL == 6) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (NST == 3 && j + 1 < ntile) {
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (ABL != 6 && j + NST - 1 < ntile) A2_ISSUE(j + NST - 1);   // slot consumed in iteration j-1: every wave is past it

        const op16_t* st = smem + (j % NST) * (NP * 2 * TILE);
        const op16_t* sK = st;
        const op16_t* sV = st + TILE;
        const op16_t* sKl = st + (NP - 1) * 2 * TILE;
        const op16_t* sVl = st + (NP - 1) * 2 * TILE + TILE;

        f32x16 s[2];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kb][e] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int off = attn_swz(kb * 32 + lq, ks * 2 + hi);
                op16x8 a;
                if (ABL == 7) a = qf[0][ks ^ 1];
                else a = *reinterpret_cast<const op16x8*>(&sK[off]);
                if (ABL == 4) {
                    asm volatile("" ::"v"(a));
                    s[kb][ks] += (float)ks;
                } else {
                    s[kb] = F5_MFMA32(a, qf[0][ks], s[kb], 0, 0, 0);
                }
                if (HP) {
                    const op16x8 al = *reinterpret_cast<const op16x8*>(&sKl[off]);
                    s[kb] = F5_MFMA32(al, qf[0][ks], s[kb], 0, 0, 0);
                    s[kb] = F5_MFMA32(a, qf[NP - 1][ks], s[kb], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);

        const int key0 = j * 64;
        if (key0 + 64 > kvlen) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kb * 32 + 16 * hi + r;
                    if (key >= kvlen) s[kb][r] = -INFINITY;
                }
        }
        float tmax = -INFINITY;
        if (ABL != 5) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[kb][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        }
        if (ABL != 5 && __any(tmax > m_run)) {          // wave-uniform: rescale only when some lane's running max moved
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                o[0][e] *= alpha;
                o[1][e] *= alpha;
            }
        }
        const float mc = m_run * c2;
        float psum = 0.0f;
        if (ABL == 0 || ABL == 2 || ABL == 3 || ABL == 4 || ABL == 6 || ABL == 7) {
            attn_f32x2 ps2 = {0.0f, 0.0f};
            ps2 = attn_exp_block(s[0], c2, mc, ps2);
            ps2 = attn_exp_block(s[1], c2, mc, ps2);
            psum = ps2[0] + ps2[1];
        } else {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pv;
                    if (ABL == 5) pv = s[kb][r];
                    else pv = s[kb][r] * c2 - mc;
                    s[kb][r] = pv;
                    if (ABL != 5) psum += pv;
                }
        }
        l_run += psum;

#pragma unroll
        for (int ks4 = 0; ks4 < 4; ++ks4) {
            const int kb = ks4 >> 1, sp = ks4 & 1;
            uint32_t pw[4], pwl[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p0 = s[kb][8 * sp + 2 * e], p1 = s[kb][8 * sp + 2 * e + 1];
                pw[e] = f5_pack2_bounded(p0, p1);
                if (HP) pwl[e] = f5_pack2_lo(p0, p1);
            }
            const op16x8 pb = __builtin_bit_cast(op16x8, u32x4{pw[0], pw[1], pw[2], pw[3]});
            op16x8 pbl = pb;
            if (HP) pbl = __builtin_bit_cast(op16x8, u32x4{pwl[0], pwl[1], pwl[2], pwl[3]});
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int voff = attn_swz(db * 32 + lq, 4 * kb + 2 * hi + sp);
                op16x8 a;
                if (ABL == 7) a = qf[0][(ks4 + db) & 3];
                else a = *reinterpret_cast<const op16x8*>(&sV[voff]);
                if (ABL == 3) {
                    asm volatile("" ::"v"(a), "v"(pb));
                    o[db][ks4] += 1.0f;
                } else {
                    o[db] = F5_MFMA32(a, pb, o[db], 0, 0, 0);
                }
                if (HP) {
                    const op16x8 al = *reinterpret_cast<const op16x8*>(&sVl[voff]);
                    o[db] = F5_MFMA32(al, pb, o[db], 0, 0, 0);
                    o[db] = F5_MFMA32(a, pbl, o[db], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + lq;
    if (!HP && p.out8) {
        if (qr < p.seq_len) attn_store_f8(p, o, inv, rowbase + qr, h, hi);
    } else if (qr < p.seq_len) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = db * 32 + 8 * rg + 4 * hi;
                const float v0 = o[db][rg * 4 + 0] * inv, v1 = o[db][rg * 4 + 1] * inv;
                const float v2 = o[db][rg * 4 + 2] * inv, v3 = o[db][rg * 4 + 3] * inv;
                const size_t off = (rowbase + qr) * p.ldo + h * 64 + d;
                *reinterpret_cast<u32x2*>(p.out[0] + off) = u32x2{f5_pack2_bounded(v0, v1), f5_pack2_bounded(v2, v3)};
                if (HP && p.out[1])
                    *reinterpret_cast<u32x2*>(p.out[1] + off) = u32x2{f5_pack2_lo(v0, v1), f5_pack2_lo(v2, v3)};
            }
    }
}

#if F5_LAB   // v2w: per-tile maximum, priority variants (superseded by v2f; A/B only)
// =================================================================================================
// v2w (bf16, large grids): v2 with TWO 32-query blocks per wave (workgroup = 256 queries).  Every K / V^T fragment read
// from LDS and every staged tile now feeds twice the MFMAs: the v2 ablations (tools/attn_abl_b1.py) show the tile staging
// (-19 %), the LDS fragment reads (-15 %) and the barrier (-4 %) as additive costs next to the MFMAs and the softmax, and
// those three halve per flop here.  Price: ~230 VGPRs => 2 workgroups per CU instead of 3.  Measured 597 -> 649 TF at
// B = 32 (tools/attn_wide_bench.py).  The running-max rescale is branch-free here (+3 %).  A variant that software-pipelines the
// two query blocks against each other with sched_group_barrier (S(q1) under softmax(q0), PV(q0) under softmax(q1)) measured
// 515 TF (spills at 256 VGPRs, re-read fragments) and was dropped, like the in-wave pipelining of v3 / v4.
// =================================================================================================
// PRIO: which phase of a wave gets issue priority on its SIMD (two waves of DIFFERENT workgroups share a SIMD and drift in
// phase): 0 = the MFMA clusters (s_setprio 1 around them), 1 = no priority changes, 2 = the softmax VALU section (the wave
// sits at priority 1 and drops to 0 for its MFMA clusters, so a partner's transcendentals / VALU issue in the gaps of this
// wave's MFMAs instead of queueing behind them)
template <int PRIO, bool LAZY = true>
__global__ __launch_bounds__(256, 2) void f5_attn2w_kernel(F5AttnArgs p) {
    constexpr int NST = 3;
    constexpr int TILE = 64 * 64;
    __shared__ __attribute__((aligned(16))) op16_t smem[NST * 2 * TILE];   // [stage][K | V^T][64*64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    int bh, qblk;
    if (!attn_block_map(p, 256, bh, qblk)) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qblk * 256 + wave * 64;
    const int kvlen = p.kv_len ? p.kv_len[b] : p.seq_len;
    const int ntile = (kvlen + 63) >> 6;
    const size_t rowbase = (size_t)b * p.seq_len;

    op16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = q0 + qb * 32 + lq;
        if (qr > p.seq_len - 1) qr = p.seq_len - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[qb][ks] = *reinterpret_cast<const op16x8*>(p.qk[0] + (rowbase + qr) * p.ldqk + h * 64 + ks * 16 + hi * 8);
    }

    const op16_t* kptr[2];
    const op16_t* vptr[2];
    int krow[2], kcol[2], ldsoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q_ = i * 256 + tid;
        const int srow = q_ >> 3;
        const int schunk = (q_ & 7) ^ ((srow >> 1) & 7);
        krow[i] = attn_kperm(srow);
        kcol[i] = p.dmodel + h * 64 + schunk * 8;
        ldsoff[i] = (i * 256 + wave * 64) * 8;
        kptr[i] = p.qk[0] + (rowbase + krow[i]) * p.ldqk + kcol[i];
        vptr[i] = p.vt[0] + ((size_t)bh * 64 + srow) * p.npad + schunk * 8;
    }
    const size_t kstep = (size_t)64 * p.ldqk;
#define A2W_ISSUE(j_)                                                                                        \
    {                                                                                                        \
        op16_t* st_ = smem + ((j_) % NST) * (2 * TILE);                                                      \
        const bool tail_ = ((j_) * 64 + 63) > p.seq_len - 1;                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
            const op16_t* ks_ = kptr[i];                                                                     \
            if (tail_) {                                                                                     \
                int key_ = (j_) * 64 + krow[i];                                                              \
                if (key_ > p.seq_len - 1) key_ = p.seq_len - 1;                                              \
                ks_ = p.qk[0] + (rowbase + key_) * p.ldqk + kcol[i];                                         \
            }                                                                                                \
            attn_glds16(ks_, st_ + ldsoff[i]);                                                               \
            attn_glds16(vptr[i], st_ + TILE + ldsoff[i]);                                                    \
            kptr[i] += kstep;                                                                                \
            vptr[i] += 64;                                                                                   \
        }                                                                                                    \
    }

    f32x16 o[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            o[qb][0][e] = 0.0f;
            o[qb][1][e] = 0.0f;
        }
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.0f, 0.0f};
    const float c2 = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;
    const float lazy_margin = 8.0f / c2;                 // raw-score units: exponent of at most 8 in exp2 units

    A2W_ISSUE(0);
    if (ntile > 1) A2W_ISSUE(1);
    ATTN_PIN_Q(qf, 2, 4);

    for (int j = 0; j < ntile; ++j) {
        if (j + 1 < ntile) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j + NST - 1 < ntile) A2W_ISSUE(j + NST - 1);

        const op16_t* sK = smem + (j % NST) * (2 * TILE);
        const op16_t* sV = sK + TILE;

        f32x16 s[2][2];                                   // [query block][key block]
        if (PRIO == 0) __builtin_amdgcn_s_setprio(1);
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[qb][kb][e] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const op16x8 a = *reinterpret_cast<const op16x8*>(&sK[attn_swz(kb * 32 + lq, ks * 2 + hi)]);
                s[0][kb] = F5_MFMA32(a, qf[0][ks], s[0][kb], 0, 0, 0);
                s[1][kb] = F5_MFMA32(a, qf[1][ks], s[1][kb], 0, 0, 0);
            }
        }
        if (PRIO == 0) __builtin_amdgcn_s_setprio(0);
        if (PRIO == 2) __builtin_amdgcn_s_setprio(1);

        const int key0 = j * 64;
        if (key0 + 64 > kvlen) {
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = key0 + kb * 32 + 16 * hi + r;
                        if (key >= kvlen) s[qb][kb][r] = -INFINITY;
                    }
        }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float tmax = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[qb][kb][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            // Lazy rescale: m_run is the reference point of the exponentials, not necessarily the running maximum.  It only
            // moves (and O, l are only rescaled: 64 + 1 multiplies per query block) when some query's tile maximum exceeds it
            // by more than 2^8 -- softmax is shift invariant, so any reference within range gives the same normalised result;
            // P then lies in [0, 2^8], inside half's range, with the same RELATIVE rounding as P <= 1.  After the first
            // tile the branch is rarely taken (wave-uniform: `s_cbranch` on an SGPR mask, no divergence).
            if (!LAZY || __any(tmax > m_run[qb] + lazy_margin)) {
                const float m_new = fmaxf(m_run[qb], tmax);
                const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c2);
                m_run[qb] = m_new;
                l_run[qb] *= alpha;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    o[qb][0][e] *= alpha;
                    o[qb][1][e] *= alpha;
                }
            }
            const float mc = m_run[qb] * c2;
            attn_f32x2 ps2 = {0.0f, 0.0f};
            ps2 = attn_exp_block(s[qb][0], c2, mc, ps2);
            ps2 = attn_exp_block(s[qb][1], c2, mc, ps2);
            l_run[qb] += ps2[0] + ps2[1];
        }

#pragma unroll
        for (int ks4 = 0; ks4 < 4; ++ks4) {
            const int kb = ks4 >> 1, sp = ks4 & 1;
            op16x8 pb[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                uint32_t pw[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) pw[e] = f5_pack2_bounded(s[qb][kb][8 * sp + 2 * e], s[qb][kb][8 * sp + 2 * e + 1]);
                pb[qb] = __builtin_bit_cast(op16x8, u32x4{pw[0], pw[1], pw[2], pw[3]});
            }
            if (PRIO == 0) __builtin_amdgcn_s_setprio(1);
            if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const op16x8 a = *reinterpret_cast<const op16x8*>(&sV[attn_swz(db * 32 + lq, 4 * kb + 2 * hi + sp)]);
                o[0][db] = F5_MFMA32(a, pb[0], o[0][db], 0, 0, 0);
                o[1][db] = F5_MFMA32(a, pb[1], o[1][db], 0, 0, 0);
            }
            if (PRIO == 0) __builtin_amdgcn_s_setprio(0);
            if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
        }
    }
#undef A2W_ISSUE

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l_tot;
        const int qr = q0 + qb * 32 + lq;
        if (p.out8) {
            if (qr < p.seq_len) attn_store_f8(p, o[qb], inv, rowbase + qr, h, hi);
        } else if (qr < p.seq_len) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int d = db * 32 + 8 * rg + 4 * hi;
                    const float v0 = o[qb][db][rg * 4 + 0] * inv, v1 = o[qb][db][rg * 4 + 1] * inv;
                    const float v2 = o[qb][db][rg * 4 + 2] * inv, v3 = o[qb][db][rg * 4 + 3] * inv;
                    const size_t off = (rowbase + qr) * p.ldo + h * 64 + d;
                    *reinterpret_cast<u32x2*>(p.out[0] + off) = u32x2{f5_pack2_bounded(v0, v1), f5_pack2_bounded(v2, v3)};
                }
        }
    }
}

#endif  // F5_LAB
// =================================================================================================
// v2f (large grids, default): v2w with the softmax BOOKKEEPING taken off the per-tile path.  MFMA time and VALU time add on a
// SIMD (tools/probes/coissue.hip), and at head dim 64 a key costs more VALU than MFMA issue time, so every instruction removed
// from the softmax is kernel time.  Two things go:
//  (1) the running maximum.  v2w needs max(S) of every tile (32 v_max3 + a cross-lane exchange through the LDS pipe per
//      query block) only to find out that the reference point m_ref of the exponentials does not have to move.  Here the fast
//      path never looks at the scores: p = exp2(s - m_ref) is computed unconditionally and the ROW SUMS, which are needed
//      anyway, tell whether that was safe: if a lane's partial sum of a tile is <= 2^14, every p of it is <= 2^14, inside
//      half's range with the usual relative rounding; the O / l accumulators are fp32.  Only when some sum is larger (or inf:
//      a score more than 2^14 above the reference point), and on the first tile, the wave takes the slow path -- recompute
//      the scores of the tile (K is still in LDS), take the true maximum, move m_ref there, rescale O and l -- which is v2w's
//      arithmetic.  Softmax is shift invariant, so any reference point gives the same normalised result.
//  (2) the scale / subtract.  With q pre-multiplied by scale * log2(e) in the QKV epilogue (p.q_prescaled; PRE) the MFMA
//      result is already in exp2 units, and -m_ref enters as the C operand of the first QK^T MFMA of a block (16 registers per
//      query block holding the same per-lane value, rewritten only on the slow path): the accumulator leaves the matrix core
//      as s - m_ref and goes straight into v_exp_f32.  Without PRE the packed fma of v2w stays.
// Per 64-key tile and wave: 32 MFMAs next to 64 v_exp + 32 v_pk_add + 32 v_cvt_pk (+ 32 v_pk_fma without PRE), against
// v2w's additional 32 v_max3 + 10 v_max + 2 ds_bpermute round trips + the rescale test.
// =================================================================================================
__device__ __forceinline__ attn_f32x2 attn_exp_block_pre(f32x16& s, attn_f32x2 sum2) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        attn_f32x2 t;
        t[0] = __builtin_amdgcn_exp2f(s[r]);
        t[1] = __builtin_amdgcn_exp2f(s[r + 1]);
        s[r] = t[0];
        s[r + 1] = t[1];
        sum2 += t;
    }
    return sum2;
}

template <bool PRE>
__global__ __launch_bounds__(256, 2) void f5_attn2f_kernel(F5AttnArgs p) {
    constexpr int NST = 3;
    constexpr int TILE = 64 * 64;
    __shared__ __attribute__((aligned(16))) op16_t smem[NST * 2 * TILE];   // [stage][K | V^T][64*64]

    // the whole argument block in ONE scalar-load clause (left alone the compiler loads each field where it is first used: three or
    // four dependent s_load / s_waitcnt rounds in the prologue of a kernel that is one latency chain at batch 1)
    asm volatile("" ::"s"(p.qk[0]), "s"(p.vt[0]), "s"(p.out[0]), "s"(p.kv_len), "s"(p.B), "s"(p.H), "s"(p.seq_len), "s"(p.npad), "s"(p.ldqk),
                 "s"(p.ldo), "s"(p.dmodel), "s"(p.scale), "s"(p.out8));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    int bh, qblk;
    if (!attn_block_map(p, 256, bh, qblk)) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qblk * 256 + wave * 64;
    const int kvlen = p.kv_len ? p.kv_len[b] : p.seq_len;
    const int ntile = (kvlen + 63) >> 6;
    const size_t rowbase = (size_t)b * p.seq_len;
    // wave-uniform: at N = 937 one wave in 16 lies entirely past the sequence; it used to compute on clamped rows (6 % of the
    // launch's MFMA and softmax work, thrown away at the store -- on a power-limited part that is 6 % of the time)
    const bool live = q0 < p.seq_len;

    op16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = q0 + qb * 32 + lq;
        if (qr > p.seq_len - 1) qr = p.seq_len - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[qb][ks] = *reinterpret_cast<const op16x8*>(p.qk[0] + (rowbase + qr) * p.ldqk + h * 64 + ks * 16 + hi * 8);
    }

    const op16_t* kptr[2];
    const op16_t* vptr[2];
    int krow[2], kcol[2], ldsoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q_ = i * 256 + tid;
        const int srow = q_ >> 3;
        const int schunk = (q_ & 7) ^ ((srow >> 1) & 7);
        krow[i] = attn_kperm(srow);
        kcol[i] = p.dmodel + h * 64 + schunk * 8;
        ldsoff[i] = (i * 256 + wave * 64) * 8;
        kptr[i] = p.qk[0] + (rowbase + krow[i]) * p.ldqk + kcol[i];
        vptr[i] = p.vt[0] + ((size_t)bh * 64 + srow) * p.npad + schunk * 8;
    }
    const size_t kstep = (size_t)64 * p.ldqk;
#define A2F_ISSUE(j_)                                                                                        \
    {                                                                                                        \
        op16_t* st_ = smem + ((j_) % NST) * (2 * TILE);                                                      \
        const bool tail_ = ((j_) * 64 + 63) > p.seq_len - 1;                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
            const op16_t* ks_ = kptr[i];                                                                     \
            if (tail_) {                                                                                     \
                int key_ = (j_) * 64 + krow[i];                                                              \
                if (key_ > p.seq_len - 1) key_ = p.seq_len - 1;                                              \
                ks_ = p.qk[0] + (rowbase + key_) * p.ldqk + kcol[i];                                         \
            }                                                                                                \
            attn_glds16(ks_, st_ + ldsoff[i]);                                                               \
            attn_glds16(vptr[i], st_ + TILE + ldsoff[i]);                                                    \
            kptr[i] += kstep;                                                                                \
            vptr[i] += 64;                                                                                   \
        }                                                                                                    \
    }
    // S^T blocks of the current tile for both query blocks; USE_M: accumulate on top of -m_ref (PRE fast path)
#define A2F_QK(USE_M)                                                                                        \
    {                                                                                                        \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                   \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                               \
                const op16x8 a = *reinterpret_cast<const op16x8*>(&sK[attn_swz(kb * 32 + lq, ks * 2 + hi)]); \
                s[0][kb] = F5_MFMA32(a, qf[0][ks], ks == 0 ? ((USE_M) ? mneg[0] : zero16) : s[0][kb], 0, 0, 0); \
                s[1][kb] = F5_MFMA32(a, qf[1][ks], ks == 0 ? ((USE_M) ? mneg[1] : zero16) : s[1][kb], 0, 0, 0); \
            }                                                                                                \
        }                                                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        if (j * 64 + 64 > kvlen) {                                                                           \
            _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                                 \
                _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                             \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
                        const int key = j * 64 + kb * 32 + 16 * hi + r;                                      \
                        if (key >= kvlen) s[qb][kb][r] = -INFINITY;                                          \
                    }                                                                                        \
        }                                                                                                    \
    }

    f32x16 o[2][2], mneg[2], zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        zero16[e] = 0.0f;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            o[qb][0][e] = 0.0f;
            o[qb][1][e] = 0.0f;
            mneg[qb][e] = 0.0f;
        }
    }
    float m_ref[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.0f, 0.0f};   // m_ref in exp2 units (scores * c2)
    const float c2 = PRE ? 1.0f : p.scale * 1.4426950408889634f;
    constexpr float SUM_LIMIT = 16384.0f;

    A2F_ISSUE(0);
    if (ntile > 1) A2F_ISSUE(1);
    ATTN_PIN_Q(qf, 2, 4);

    for (int j = 0; j < ntile; ++j) {
        if (j + 1 < ntile) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j + NST - 1 < ntile) A2F_ISSUE(j + NST - 1);

        const op16_t* sK = smem + (j % NST) * (2 * TILE);
        const op16_t* sV = sK + TILE;

        if (!live) continue;                              // a wave entirely past the sequence only stages tiles and keeps the barriers
        f32x16 s[2][2];                                   // [query block][key block]
        float psum[2];
        bool slow = j == 0;
        if (!slow) {
            A2F_QK(PRE);
            bool bad = false;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                attn_f32x2 ps2 = {0.0f, 0.0f};
                if (PRE) {
                    ps2 = attn_exp_block_pre(s[qb][0], ps2);
                    ps2 = attn_exp_block_pre(s[qb][1], ps2);
                } else {
                    ps2 = attn_exp_block(s[qb][0], c2, m_ref[qb], ps2);
                    ps2 = attn_exp_block(s[qb][1], c2, m_ref[qb], ps2);
                }
                psum[qb] = ps2[0] + ps2[1];
                bad = bad || !(psum[qb] <= SUM_LIMIT);
            }
            slow = __any(bad);
        }
        if (slow) {                                       // wave-uniform: first tile, or a score far above the reference point
            A2F_QK(false);
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float tmax = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[qb][kb][r]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_ref[qb], tmax * c2);
                const float alpha = __builtin_amdgcn_exp2f(m_ref[qb] - m_new);
                m_ref[qb] = m_new;
                l_run[qb] *= alpha;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    o[qb][0][e] *= alpha;
                    o[qb][1][e] *= alpha;
                    mneg[qb][e] = -m_new;
                }
                attn_f32x2 ps2 = {0.0f, 0.0f};
                ps2 = attn_exp_block(s[qb][0], c2, m_new, ps2);
                ps2 = attn_exp_block(s[qb][1], c2, m_new, ps2);
                psum[qb] = ps2[0] + ps2[1];
            }
        }
        l_run[0] += psum[0];
        l_run[1] += psum[1];

#pragma unroll
        for (int ks4 = 0; ks4 < 4; ++ks4) {
            const int kb = ks4 >> 1, sp = ks4 & 1;
            op16x8 pb[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                uint32_t pw[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) pw[e] = f5_pack2_bounded(s[qb][kb][8 * sp + 2 * e], s[qb][kb][8 * sp + 2 * e + 1]);
                pb[qb] = __builtin_bit_cast(op16x8, u32x4{pw[0], pw[1], pw[2], pw[3]});
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const op16x8 a = *reinterpret_cast<const op16x8*>(&sV[attn_swz(db * 32 + lq, 4 * kb + 2 * hi + sp)]);
                o[0][db] = F5_MFMA32(a, pb[0], o[0][db], 0, 0, 0);
                o[1][db] = F5_MFMA32(a, pb[1], o[1][db], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }
#undef A2F_ISSUE
#undef A2F_QK

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l_tot;
        const int qr = q0 + qb * 32 + lq;
        if (p.out8) {
            if (qr < p.seq_len) attn_store_f8(p, o[qb], inv, rowbase + qr, h, hi);
        } else if (qr < p.seq_len) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int d = db * 32 + 8 * rg + 4 * hi;
                    const float v0 = o[qb][db][rg * 4 + 0] * inv, v1 = o[qb][db][rg * 4 + 1] * inv;
                    const float v2 = o[qb][db][rg * 4 + 2] * inv, v3 = o[qb][db][rg * 4 + 3] * inv;
                    const size_t off = (rowbase + qr) * p.ldo + h * 64 + d;
                    *reinterpret_cast<u32x2*>(p.out[0] + off) = u32x2{f5_pack2_bounded(v0, v1), f5_pack2_bounded(v2, v3)};
                }
        }
    }
}

// =================================================================================================
// v2p (large grids, q pre-multiplied): v2f's arithmetic, software-pipelined INSIDE the wave, ONE wave per SIMD.
//
// A wave's VALU / LDS instructions overlap matrix work only when they sit between the MFMAs of the SAME instruction stream: a
// 32-cycle v_mfma_f32_32x32x16 leaves ~8 issue slots, of which ~5 can be filled for free (MI355X_MICROARCH.md, "one wave per SIMD");
// two co-resident waves do not do that for each other (tools/probes/coissue.hip).  v2f runs S = K Q^T -> exp -> O += V^T P one after
// the other, so per 64-key tile its 32 MFMAs (1 024 cycles) and ~160 VALU + 16 LDS reads ADD: 2 720 cycles per wave tile.  The
// round-2 attempt at in-wave pipelining (v5, lab) kept 32 queries per wave -- one LDS fragment per MFMA -- and the per-tile maximum:
// 9+ fillers per MFMA, slower.  With v2f's softmax (no tile maximum, -m_ref as the C operand of the first QK^T MFMA, scores already
// in exp2 units) and 64 queries per wave (every K / V^T fragment feeds TWO MFMAs) a 64-key tile is 32 MFMAs next to
//     64 v_exp_f32 + 64 v_add_f32 + 32 v_cvt_pk + 16 ds_read_b128  =  5.5 fillers per MFMA.
// The pipeline step is HALF a tile (32 keys x 64 queries).  Half-step (j, kb) issues, as ONE stream of independent work,
//     MFMA:  S(next half) = K Q^T - m_ref      (8: the other key block of tile j, or key block 0 of tile j+1)
//            O^T += V^T(prev half) P(prev half) (8: P packed in the previous half-step)
//     VALU:  P(j, kb) = exp2(S(j, kb)), row sums (S computed in the previous half-step)
// as 16 half-slots of {2 exp, 2 add + 1 cvt_pk of the previous pair, 1 MFMA, every other one a fragment read three fragments ahead},
// pinned with sched_barrier.  S and P are double-buffered per key block (static roles: block 0 <-> buffer A, block 1 <-> buffer B).
//
// Registers.  VALU instructions cannot read the accumulator half of the register file, an MFMA's C and D must sit in the same
// half, and left to itself the compiler parks MFMA results in AGPRs as soon as a kernel needs more than 256 registers and copies
// them back one v_accvgpr_read at a time (the first build of this kernel: 130 reads + 226 writes per tile, 362 spilled registers).
// So the MFMAs are inline asm with the register class in the constraint: O^T (64), Q^T (32) and the K / V^T fragments (16) live in
// AGPRs ("a"); the two S buffers (64), -m_ref (32) and the two P buffers (32) in VGPRs ("v").  An asm MFMA is opaque to the
// compiler's hazard padding; the places where an MFMA result or operand is touched by another instruction class within the hazard
// window are padded by hand (A2P_PAD, s_nop 1).
//
// LDS: a K ring and a V^T ring of 4 stages each (2 x 32 KB), tile t in stage t & 3.  Iteration j reads K(j) (key block 1), K(j+1)
// (key block 0), V^T(j-1) (second half) and V^T(j) (first half); after its barrier it issues K(j+3) and V^T(j+2) (global_load_lds,
// uniform base + 32-bit lane offset), and the counted wait at its top lets the previous iteration's group stay in flight: every
// tile has two iterations to arrive (the first build issued one tile ahead behind vmcnt(0): each iteration then waited out a full
// HBM / L2 round trip, 4 800 cycles per tile).  One barrier per tile.  The tile loop is unrolled four times so that every fragment
// address is one loop-invariant VGPR plus an immediate.
// The slow path (a row sum above 2^14: some score far above the reference point) runs at the END of the half-step: S(j, kb) is
// still in registers (the exponentials are not taken in place), so the reference point moves by delta = max(S) and O, l, -m_ref,
// the already computed next S and this P follow.  The first half tile takes its true maximum in the prologue.  Same layouts
// (permuted K rows, XOR-swizzled 128-byte rows, P fed from the S accumulators) and the same results as v2f up to the order of the
// row-sum additions and the reference point of the rows.
// =================================================================================================
#if F5_F16
#define A2P_MFMA_OP "v_mfma_f32_32x32x16_f16"
#else
#define A2P_MFMA_OP "v_mfma_f32_32x32x16_bf16"
#endif
// 20 wait states: more than the 12 an 8-pass MFMA needs before another instruction class may touch its result
#define A2P_PAD "s_nop 15\n\ts_nop 3"
#define ATTN2P_TILE (64 * 64)
// The compiler may place a register copy (v_mov / v_accvgpr_mov / v_accvgpr_write, or the v_cvt_pk that produced a P word) directly in
// front of an asm statement; a VALU result needs 2 wait states before an MFMA may read it and the hazard recogniser does not look
// into asm (first build: stale O accumulators and P words wherever such a copy sat in front of an MFMA).  Every asm MFMA therefore
// opens with its own two states.
#define A2P_GUARD "s_nop 1\n\t"
// d (VGPRs) = a * b + c: first MFMA of a score block, c = -m_ref (VGPRs: C and D share one half of the register file)
__device__ __forceinline__ void a2p_mfma_first(f32x16& d, const op16x8& a, const op16x8& b, const f32x16& c) {
    asm volatile(A2P_GUARD A2P_MFMA_OP " %0, %1, %2, %3" : "=&v"(d) : "a"(a), "a"(b), "v"(c));
}
// d (VGPRs) = a * b (C = inline constant 0)
__device__ __forceinline__ void a2p_mfma_first0(f32x16& d, const op16x8& a, const op16x8& b) {
    asm volatile(A2P_GUARD A2P_MFMA_OP " %0, %1, %2, 0" : "=&v"(d) : "a"(a), "a"(b));
}
// d (VGPRs) += a * b
__device__ __forceinline__ void a2p_mfma_s(f32x16& d, const op16x8& a, const op16x8& b) {
    asm volatile(A2P_GUARD A2P_MFMA_OP " %0, %1, %2, %0" : "+v"(d) : "a"(a), "a"(b));
}
// d (AGPRs) += a * b, b = P (VGPRs)
__device__ __forceinline__ void a2p_mfma_o(f32x16& d, const op16x8& a, const op16x8& b) {
    asm volatile(A2P_GUARD A2P_MFMA_OP " %0, %1, %2, %0" : "+a"(d) : "a"(a), "v"(b));
}
// a wave-uniform pointer the compiler can see is uniform (SGPR pair): global_load_lds then uses the "SGPR base + 32-bit VGPR offset" form
__device__ __forceinline__ const char* attn_uniform_ptr(const void* ptr) {
    const uint64_t u = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

// One half-step.  s_cur: the scores of this 32-key block (both query blocks); s_nxt: the scores of the next block, taken from key
// block KBN of the K tile in ring stage ST_K; p_prev: P of the previous block = key block KBP of the tile whose V^T sits in stage
// ST_V; nvalid (MASK instantiations = last tile only): keys of THIS block that exist (>= 32: all).  pk / pv: LDS addresses of this
// lane's K / V^T fragments inside stage 0 (loop invariant; stage, key / d block are immediates).
// LDS fragment load in asm, straight into the accumulator file, with the wait counted by hand (the compiler's own lgkmcnt waits for
// its ds_reads were lgkmcnt(0) in front of the first MFMA that used one: a full LDS round trip, ~100 cycles, exposed twice per
// half-step; tools/probes/inwave_overlap.hip modes 11 / 13).  The destination is only valid after a2p_lds_wait.
__device__ __forceinline__ uint32_t a2p_lds_addr(const void* p) {
    return (uint32_t)reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)p);
}
template <int BYTES>
__device__ __forceinline__ void a2p_lds_read(op16x8& dst, uint32_t addr) {
    static_assert(BYTES >= 0 && BYTES < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(addr), "n"(BYTES));
}
__device__ __forceinline__ void a2p_lds_wait(op16x8 (&f)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(f[0]), "+a"(f[1]), "+a"(f[2]), "+a"(f[3]), "+a"(f[4]), "+a"(f[5]), "+a"(f[6]), "+a"(f[7]));
}
// the eight fragments of a half-step: K steps 0..3 of key block KBN of the K tile in ring stage ST_K, then V^T pieces (16-key step
// 2 KBP + (g >> 1), d block g & 1), g = 0..3, of the V^T tile in stage ST_V.  ak / av: this lane's LDS byte addresses inside stage 0.
template <int F, int ST_K, int KBN, int ST_V, int KBP>
__device__ __forceinline__ void a2p_load_frag(op16x8& dst, const uint32_t (&ak)[4], const uint32_t (&av)[4]) {
    if (F < 4) a2p_lds_read<(ST_K * ATTN2P_TILE + KBN * 2048) * 2>(dst, ak[F]);
    else a2p_lds_read<((4 + ST_V) * ATTN2P_TILE + ((F - 4) & 1) * 2048) * 2>(dst, av[2 * KBP + ((F - 4) >> 1)]);
}

// One half-step.  s_cur: the scores of this 32-key block (both query blocks); s_nxt: the scores of the next block; fr: the eight
// LDS fragments of THIS half-step (requested during the previous one, complete on entry); fn: those of the NEXT half-step, whose
// parameters are N_* -- requested in the first eight half-slots, waited for at the end; p_prev: P of the previous block;
// nvalid: keys of THIS block that exist (>= 32: all).
template <int N_ST_K, int N_KBN, int N_ST_V, int N_KBP>
__device__ __forceinline__ void attn2p_half(const uint32_t (&ak)[4], const uint32_t (&av)[4], const op16x8 (&qf)[2][4], op16x8 (&fr)[8],
                                            op16x8 (&fn)[8], f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2], f32x16 (&o)[2][2], f32x16 (&mneg)[2],
                                            const uint32_t (&p_prev)[2][8], uint32_t (&p_cur)[2][8], float (&l_run)[2], int hi,
                                            int nvalid) {
    constexpr float SUM_LIMIT = 16384.0f;
    if (__builtin_expect(nvalid < 32, 0)) {                  // wave-uniform; only the last tile of a sequence can be partial
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (16 * hi + r >= nvalid) s_cur[qb][r] = -INFINITY;
    }
    float sa[2] = {0.0f, 0.0f}, sb[2] = {0.0f, 0.0f};        // two partial row sums per query block (even / odd key of a pair)
    float y0 = 0.0f, y1 = 0.0f;                              // exponentials of the previous pair (summed / packed one half-slot later)
#define A2P_HALF_SLOT(H)                                                                                                \
    {                                                                                                                   \
        constexpr int h = (H), f = h >> 1, qb = h & 1;                                                                  \
        if (h < 8) a2p_load_frag<(h < 8 ? h : 0), N_ST_K, N_KBN, N_ST_V, N_KBP>(fn[h < 8 ? h : 0], ak, av);               \
        {                                                                                                               \
            constexpr int pq = h >> 3, pi = h & 7;           /* pair (query block, i): scores 2i, 2i + 1 of s_cur[.] */ \
            float x0 = __builtin_amdgcn_exp2f(s_cur[pq][2 * pi]);                                                       \
            float x1 = __builtin_amdgcn_exp2f(s_cur[pq][2 * pi + 1]);                                                   \
            asm volatile("" : "+v"(x0), "+v"(x1));                                                                      \
            if (h > 0) {                                                                                                \
                constexpr int qq = (h > 0 ? h - 1 : 0) >> 3, w = (h > 0 ? h - 1 : 0) & 7;                               \
                sa[qq] += y0;                                                                                           \
                sb[qq] += y1;                                                                                           \
                p_cur[qq][w] = f5_pack2_bounded(y0, y1);                                                                \
                asm volatile("" : "+v"(sa[qq]), "+v"(sb[qq]), "+v"(p_cur[qq][w]));                                      \
            }                                                                                                           \
            y0 = x0;                                                                                                    \
            y1 = x1;                                                                                                    \
        }                                                                                                               \
        if (f < 4) {                                                                                                    \
            if (f == 0) a2p_mfma_first(s_nxt[qb], fr[f], qf[qb][f < 4 ? f : 0], mneg[qb]);                              \
            else a2p_mfma_s(s_nxt[qb], fr[f], qf[qb][f < 4 ? f : 0]);                                                   \
        } else {                                                                                                        \
            constexpr int sp = (f - 4) >> 1 & 1, db = (f - 4) & 1;                                                      \
            const op16x8 pb = __builtin_bit_cast(op16x8, u32x4{p_prev[qb][4 * sp], p_prev[qb][4 * sp + 1], p_prev[qb][4 * sp + 2], \
                                                               p_prev[qb][4 * sp + 3]});                                \
            a2p_mfma_o(o[qb][db], fr[f], pb);                                                                           \
        }                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    }
    A2P_HALF_SLOT(0) A2P_HALF_SLOT(1) A2P_HALF_SLOT(2) A2P_HALF_SLOT(3) A2P_HALF_SLOT(4) A2P_HALF_SLOT(5) A2P_HALF_SLOT(6) A2P_HALF_SLOT(7)
    A2P_HALF_SLOT(8) A2P_HALF_SLOT(9) A2P_HALF_SLOT(10) A2P_HALF_SLOT(11) A2P_HALF_SLOT(12) A2P_HALF_SLOT(13) A2P_HALF_SLOT(14) A2P_HALF_SLOT(15)
#undef A2P_HALF_SLOT
    sa[1] += y0;
    sb[1] += y1;
    p_cur[1][7] = f5_pack2_bounded(y0, y1);
    a2p_lds_wait(fn);                                        // requested eight or more half-slots ago
    float psum[2] = {sa[0] + sb[0], sa[1] + sb[1]};
    if (__builtin_expect(__any(!(psum[0] <= SUM_LIMIT) || !(psum[1] <= SUM_LIMIT)) != 0, 0)) {
        // wave-uniform and rare: some score of this block lies more than 14 (exp2 units) above the reference point.  Move the
        // reference point of every row to max(old, this block's maximum): O and l shrink by alpha, -m_ref and the scores of the next
        // block (already computed against the old point) shift by delta, the block's P is taken again.
        asm volatile(A2P_PAD : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[1][0]), "+a"(o[1][1]));     // the last MFMAs of the half-step wrote O
        asm volatile("" : "+v"(s_nxt[0]), "+v"(s_nxt[1]));                                       // ... issued after the S MFMAs: covered
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float tmax = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s_cur[qb][r]);
            const float delta = fmaxf(tmax, __shfl_xor(tmax, 32, 64));       // >= 0, in exp2 units
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            l_run[qb] *= alpha;
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                o[qb][0][e] *= alpha;
                o[qb][1][e] *= alpha;
                mneg[qb][e] -= delta;
                s_nxt[qb][e] -= delta;
            }
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float e0 = __builtin_amdgcn_exp2f(s_cur[qb][2 * w] - delta);
                const float e1 = __builtin_amdgcn_exp2f(s_cur[qb][2 * w + 1] - delta);
                s0 += e0;
                s1 += e1;
                p_cur[qb][w] = f5_pack2_bounded(e0, e1);
            }
            psum[qb] = s0 + s1;
        }
        // the rewritten accumulators / -m_ref are MFMA operands of the next half-step: a VALU result needs 2 states
        asm volatile("s_nop 1" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[1][0]), "+a"(o[1][1]), "+v"(mneg[0]), "+v"(mneg[1]));
    }
    l_run[0] += psum[0];
    l_run[1] += psum[1];
}

__global__ __launch_bounds__(256, 1) void f5_attn2p_kernel(F5AttnArgs p) {
    constexpr int TILE = 64 * 64;
    __shared__ __attribute__((aligned(16))) op16_t smem[8 * TILE];       // K ring [4][64*64] then V^T ring [4][64*64]: 64 KB

    asm volatile("" ::"s"(p.qk[0]), "s"(p.vt[0]), "s"(p.out[0]), "s"(p.kv_len), "s"(p.B), "s"(p.H), "s"(p.seq_len), "s"(p.npad), "s"(p.ldqk),
                 "s"(p.ldo), "s"(p.dmodel), "s"(p.out8));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    int bh, qblk;
    if (!attn_block_map(p, 256, bh, qblk)) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qblk * 256 + wave * 64;
    const int kvlen = p.kv_len ? p.kv_len[b] : p.seq_len;
    const int T = (kvlen + 63) >> 6;
    const size_t rowbase = (size_t)b * p.seq_len;
    const bool live = q0 < p.seq_len;                       // a wave entirely past the sequence only stages tiles and keeps the barriers

    op16x8 qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = q0 + qb * 32 + lq;
        if (qr > p.seq_len - 1) qr = p.seq_len - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[qb][ks] = *reinterpret_cast<const op16x8*>(p.qk[0] + (rowbase + qr) * p.ldqk + h * 64 + ks * 16 + hi * 8);
    }
    // staging: 2 16-byte chunks of K and of V^T per thread per tile (thread q_ = i * 256 + tid fills LDS chunk q_ of the tile image);
    // source = wave-uniform tile base (SGPR pair) + a loop-invariant 32-bit lane offset
    const char* kbase = attn_uniform_ptr(p.qk[0] + rowbase * p.ldqk + p.dmodel + h * 64);     // key 0 of this head
    const char* vbase = attn_uniform_ptr(p.vt[0] + (size_t)bh * 64 * p.npad);                  // key 0, d 0
    const uint32_t kstep = (uint32_t)(64 * p.ldqk) * 2u;                                      // bytes per tile
    uint32_t koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q_ = i * 256 + tid;
        const int srow = q_ >> 3;
        const int schunk = (q_ & 7) ^ ((srow >> 1) & 7);
        koff[i] = ((uint32_t)attn_kperm(srow) * (uint32_t)p.ldqk + schunk * 8) * 2u;
        voff[i] = ((uint32_t)srow * (uint32_t)p.npad + schunk * 8) * 2u;
    }
    const int ldsw = wave * 64 * 8;                          // this wave's 1 KB piece of a 4 KB staging instruction (elements)
#define A2P_ISSUE_K(t_)                          /* K tile t_ -> K ring stage t_ & 3 */                   \
    {                                                                                                        \
        op16_t* dst_ = smem + ((t_) & 3) * TILE + ldsw;                                                      \
        const char* ks_ = kbase + (size_t)(t_) * kstep;                                                      \
        const bool tail_ = ((t_) * 64 + 63) > p.seq_len - 1;                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
            uint32_t ko_ = koff[i];                                                                          \
            if (tail_) {                                     /* rows past the sequence: re-read its last key */ \
                const int q_ = i * 256 + tid, srow_ = q_ >> 3;                                               \
                int key_ = (t_) * 64 + attn_kperm(srow_);                                                    \
                if (key_ > p.seq_len - 1) key_ = p.seq_len - 1;                                              \
                ko_ = ((uint32_t)(key_ - (t_) * 64) * (uint32_t)p.ldqk + ((q_ & 7) ^ ((srow_ >> 1) & 7)) * 8) * 2u; \
            }                                                                                                \
            attn_glds16(reinterpret_cast<const op16_t*>(ks_ + ko_), dst_ + i * 2048);                        \
        }                                                                                                    \
    }
#define A2P_ISSUE_V(t_)                          /* V^T tile t_ -> V ring stage t_ & 3 */                 \
    {                                                                                                        \
        op16_t* dst_ = smem + (4 + ((t_) & 3)) * TILE + ldsw;                                                \
        const char* vs_ = vbase + (size_t)(t_) * 128;                                                        \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                        \
            attn_glds16(reinterpret_cast<const op16_t*>(vs_ + voff[i]), dst_ + i * 2048);                    \
    }
    // issue group of iteration j (after its barrier): K(j+3) and V^T(j+2); the wait at the top of iteration j lets the group of
    // iteration j-1 -- K(j+2), V^T(j+1), 2 instructions each where the tile exists -- stay in flight
#define A2P_ISSUE_GROUP(j_)                                  \
    {                                                        \
        if ((j_) + 3 < T) A2P_ISSUE_K((j_) + 3);             \
        if ((j_) + 2 < T) A2P_ISSUE_V((j_) + 2);             \
    }
#define A2P_WAIT_TOP(j_)                                     \
    if ((j_) + 2 < T) {                                      \
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     \
    } else if ((j_) + 1 < T) {                               \
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     \
    } else {                                                 \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     \
    }
#define A2P_BARRIER()                              \
    {                                              \
        asm volatile("" ::: "memory");             \
        __builtin_amdgcn_s_barrier();              \
        asm volatile("" ::: "memory");             \
    }
    // this lane's fragment addresses (LDS bytes) inside a 64 x 64 tile image of stage 0: lane part of attn_swz; the key / d block adds 32 rows
    uint32_t ak[4], av[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ak[i] = a2p_lds_addr(smem + lq * 64 + (((i * 2 + hi) ^ ((lq >> 1) & 7)) << 3));                       // QK^T k-step i: chunk 2 i + hi
        av[i] = a2p_lds_addr(smem + lq * 64 + (((4 * (i >> 1) + 2 * hi + (i & 1)) ^ ((lq >> 1) & 7)) << 3));  // 16-key step i of the tile
    }

    // ---- prologue: K(0), V^T(0), K(1), then the group "of iteration -1": K(2), V^T(1).  P(-1, 1) = 0 multiplies the V^T stage of
    // "tile -1" (stage 3) in the first half-step: that stage is zeroed here (uninitialised LDS may hold NaN patterns).
    A2P_ISSUE_K(0);
    A2P_ISSUE_V(0);
    if (T > 1) A2P_ISSUE_K(1);
    if (T > 2) A2P_ISSUE_K(2);
    if (T > 1) A2P_ISSUE_V(1);
    {
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(smem + 7 * TILE + tid * 8) = z4;
        *reinterpret_cast<u32x4*>(smem + 7 * TILE + 2048 + tid * 8) = z4;
    }
    // Q^T fragments: waited for here (a use the compiler sees, as ATTN_PIN_Q) and from here on accumulator-file values -- every
    // MFMA takes them as "a" operands; defined as VGPR values they would be copied (4 v_accvgpr_write) in front of every use
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+a"(qf[qb][ks]));
    // K(0) has landed (the oldest 2 of the 4 + 2 [T > 1] + 4 [T > 2: 2, else V^T(1) only] requests)
    if (T > 2) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (T > 1) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    A2P_BARRIER();
    if (!live) {                                             // same barriers, same staging, no arithmetic
        for (int j = 0; j < T; ++j) {
            A2P_WAIT_TOP(j);
            A2P_BARRIER();
            A2P_ISSUE_GROUP(j);
        }
        return;
    }

    f32x16 o[2][2], mneg[2], s_a[2], s_b[2];                 // s_a / p_a: key block 0 of a tile, s_b / p_b: key block 1
    uint32_t p_a[2][8], p_b[2][8];
    op16x8 fr_a[8], fr_b[8];                                 // LDS fragments of the (j, 0) / (j, 1) half-steps: each set is requested during the other's half-step
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            o[qb][0][e] = 0.0f;
            o[qb][1][e] = 0.0f;
        }
#pragma unroll
        for (int w = 0; w < 8; ++w) p_b[qb][w] = 0u;         // P(-1, 1)
    }
    float l_run[2] = {0.0f, 0.0f};

    // ---- the reference point: the true maximum of S(0, block 0) per row; s_a = S(0, 0) - m_ref enters the loop like every other block
    {
        a2p_load_frag<0, 0, 0, 0, 0>(fr_b[0], ak, av);       // K(0) block 0, steps 0..3 (fr_b is free until half-step (0, 0) requests into it)
        a2p_load_frag<1, 0, 0, 0, 0>(fr_b[1], ak, av);
        a2p_load_frag<2, 0, 0, 0, 0>(fr_b[2], ak, av);
        a2p_load_frag<3, 0, 0, 0, 0>(fr_b[3], ak, av);
        // the fragments of half-step (0, 0): K(0) block 1; V^T(-1) second half = the zeroed stage 3
        a2p_load_frag<0, 0, 1, 3, 1>(fr_a[0], ak, av);
        a2p_load_frag<1, 0, 1, 3, 1>(fr_a[1], ak, av);
        a2p_load_frag<2, 0, 1, 3, 1>(fr_a[2], ak, av);
        a2p_load_frag<3, 0, 1, 3, 1>(fr_a[3], ak, av);
        a2p_load_frag<4, 0, 1, 3, 1>(fr_a[4], ak, av);
        a2p_load_frag<5, 0, 1, 3, 1>(fr_a[5], ak, av);
        a2p_load_frag<6, 0, 1, 3, 1>(fr_a[6], ak, av);
        a2p_load_frag<7, 0, 1, 3, 1>(fr_a[7], ak, av);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(fr_b[0]), "+a"(fr_b[1]), "+a"(fr_b[2]), "+a"(fr_b[3]));
        a2p_lds_wait(fr_a);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks == 0) {
                a2p_mfma_first0(s_a[0], fr_b[ks], qf[0][ks]);
                a2p_mfma_first0(s_a[1], fr_b[ks], qf[1][ks]);
            } else {
                a2p_mfma_s(s_a[0], fr_b[ks], qf[0][ks]);
                a2p_mfma_s(s_a[1], fr_b[ks], qf[1][ks]);
            }
        }
        asm volatile(A2P_PAD : "+v"(s_a[0]), "+v"(s_a[1]));
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, (16 * hi + r < kvlen) ? s_a[qb][r] : -INFINITY);
            const float m0 = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                mneg[qb][e] = -m0;
                s_a[qb][e] -= m0;
            }
        }
        asm volatile("s_nop 1" : "+v"(mneg[0]), "+v"(mneg[1]));
    }

    // ---- tiles.  Tile j = 4 g + PH (PH compile-time: four tiles per loop trip, a tile past the end is skipped by a scalar branch):
    //   TOP:    K(j+1) and V^T(j) have landed (counted wait), barrier, issue K(j+3) and V^T(j+2)
    //   (j, 0): exp of S(j, 0) in s_a -> p_a;  S(j, 1) -> s_b from K stage PH;            O += V^T(j-1, 1) p_b, V stage PH - 1
    //   (j, 1): exp of S(j, 1) in s_b -> p_b;  S(j+1, 0) -> s_a from K stage PH + 1;      O += V^T(j, 0) p_a,   V stage PH
    // Every tile runs the same code: keys past kvlen (last tile only) are masked under a wave-uniform branch, and the last tile's
    // second half-step computes an S(T, 0) nobody reads (K stage PH + 1 then holds an older tile or nothing: finite or not, unused).
    // After the loop: O += V^T(T-1, 1) p_b.
#define A2P_STEP(PH)                                                                                                    \
    if (j < T) {                                                                                                        \
        A2P_WAIT_TOP(j)                                                                                                 \
        A2P_BARRIER();                                                                                                  \
        A2P_ISSUE_GROUP(j)                                                                                              \
        const int nv_ = kvlen - j * 64;                                                                                 \
        /* (j, 0) requests the fragments of (j, 1): K(j+1) block 0, V^T(j) first half; (j, 1) those of (j+1, 0): K(j+1) block 1, V^T(j) second half */ \
        attn2p_half<((PH) + 1) & 3, 0, (PH), 0>(ak, av, qf, fr_a, fr_b, s_a, s_b, o, mneg, p_b, p_a, l_run, hi, nv_);       \
        attn2p_half<((PH) + 1) & 3, 1, (PH), 1>(ak, av, qf, fr_b, fr_a, s_b, s_a, o, mneg, p_a, p_b, l_run, hi, nv_ - 32);  \
        ++j;                                                                                                            \
    }
    {
        int j = 0;
        for (int g = (T + 3) >> 2; g > 0; --g) {
            A2P_STEP(0)
            A2P_STEP(1)
            A2P_STEP(2)
            A2P_STEP(3)
        }
    }
#undef A2P_STEP
    // O^T += V^T(T-1, 1) P(T-1, 1): the V ring stage of the last tile is a run-time value here (added to the lane address)
    {
        const uint32_t stv = (uint32_t)((T - 1) & 3) * (ATTN2P_TILE * 2);
        uint32_t avf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) avf[i] = av[i] + stv;
        a2p_load_frag<4, 0, 0, 0, 1>(fr_a[0], ak, avf);      // stage "0" + the run-time offset, second half of the tile (KBP = 1), pieces 0..3
        a2p_load_frag<5, 0, 0, 0, 1>(fr_a[1], ak, avf);
        a2p_load_frag<6, 0, 0, 0, 1>(fr_a[2], ak, avf);
        a2p_load_frag<7, 0, 0, 0, 1>(fr_a[3], ak, avf);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(fr_a[0]), "+a"(fr_a[1]), "+a"(fr_a[2]), "+a"(fr_a[3]));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int sp = g >> 1, db = g & 1;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const op16x8 pb = __builtin_bit_cast(op16x8, u32x4{p_b[qb][4 * sp], p_b[qb][4 * sp + 1], p_b[qb][4 * sp + 2], p_b[qb][4 * sp + 3]});
                a2p_mfma_o(o[qb][db], fr_a[g], pb);
            }
        }
    }
#undef A2P_BARRIER
#undef A2P_WAIT_TOP
#undef A2P_ISSUE_GROUP
#undef A2P_ISSUE_V
#undef A2P_ISSUE_K
    asm volatile(A2P_PAD : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[1][0]), "+a"(o[1][1]));

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l_tot;
        const int qr = q0 + qb * 32 + lq;
        if (p.out8) {
            if (qr < p.seq_len) attn_store_f8(p, o[qb], inv, rowbase + qr, h, hi);
        } else if (qr < p.seq_len) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int d = db * 32 + 8 * rg + 4 * hi;
                    const float v0 = o[qb][db][rg * 4 + 0] * inv, v1 = o[qb][db][rg * 4 + 1] * inv;
                    const float v2 = o[qb][db][rg * 4 + 2] * inv, v3 = o[qb][db][rg * 4 + 3] * inv;
                    const size_t off = (rowbase + qr) * p.ldo + h * 64 + d;
                    *reinterpret_cast<u32x2*>(p.out[0] + off) = u32x2{f5_pack2_bounded(v0, v1), f5_pack2_bounded(v2, v3)};
                }
        }
    }
}

#if F5_LAB   // role-split attention (round 3): measured 7-12 % slower than v2f (profiles/r03/attention_role_split_ab.txt)
// =================================================================================================
// v2r (large grids, round 3): v2f's arithmetic under a ROLE-SPLIT schedule.  In v2f two waves of different workgroups share a
// SIMD and drift freely: both may sit in their softmax (VALU, transcendentals) or both in their MFMA clusters at the same time,
// and the counters say so (MFMA busy 31 %, VALU 48 %, waits 31-37 %).  Here a workgroup is 8 waves = 2 groups x 4 waves of 64
// queries (512 queries), K / V^T tiles shared in LDS; per key tile j a wave alternates
//     MATRIX(j): O^T += V^T(j-1) P^T(j-1) [16 MFMAs], S^T(j) = K(j) Q^T [16 MFMAs]          (s_setprio 1)
//     VALU(j):   p = exp2(s - m_ref), row sums, pack P(j) to 16 bit (the slow path -- first tile, or a row sum above 2^14 --
//                recomputes the scores of the tile and moves the reference point; K(j) is still in LDS)
// each closed by a workgroup barrier, and group 1 runs ONE barrier behind group 0: in every interval one wave of each SIMD streams
// MFMAs while its partner runs the softmax of its own tile.  1 024 MFMA cycles against ~900 VALU cycles per tile and wave.
// Ring of 4 K / V^T tiles (64 KB), tile j+2 issued at the head of MATRIX(j) (the slot held tile j-2, whose V^T was last read in
// group 1's MATRIX(j-1), one barrier earlier), `vmcnt(2)` at the end of MATRIX(j) retires tile j+1 one barrier before its first
// reader.  The arithmetic is v2w's / v2f's without the -m_ref C operand: exp2(fma(s, c2, -m_ref)) (c2 = 1 when q is pre-multiplied).
// The first tile and the last PV are peeled so that the loop body is straight-line (with `if (j > 0)` / `if (live)` around the
// MFMA clusters the compiler moved all 64 accumulator registers through v_mov_b64 on every iteration); waves past the sequence
// compute on clamped rows.  QLDS: the Q^T fragments live in LDS and are re-read per tile (185 VGPRs) instead of in registers (216).
// RESULT (64 x 16 x 937, f16, interleaved runs on one box): v2f 305-316 us (754 TF), v2r 336-341 us (Q in registers) / 338-344 us
// (Q in LDS).  MFMA time and VALU time of a SIMD ADD on this chip whichever wave they come from (tools/probes/coissue.hip), so
// pairing one wave's MFMA cluster with its partner's softmax buys nothing, and the two extra workgroup barriers per tile cost.
// =================================================================================================
template <bool PRE, bool QLDS>
__global__ __launch_bounds__(512, 1) void f5_attn2r_kernel(F5AttnArgs p) {
    constexpr int NST = 4;
    constexpr int TILE = 64 * 64;
    // ONE array: with two, the LDS lowering tags the accesses with alias scopes and the compiler then orders every ds_read of the
    // ring behind the outstanding global_load_lds of the SAME array with its own vmcnt(0) -- the hand-counted prefetch is gone
    __shared__ __attribute__((aligned(16))) op16_t smem[NST * 2 * TILE + (QLDS ? 8 * 8 * 64 * 8 : 0)];   // [slot][K | V^T][64*64], then Q: 64 KB
    op16_t* const qsm = smem + NST * 2 * TILE;                                                // [wave][query block x k step][lane][8]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // 0..7; group = wave >> 2
    const int grp = wave >> 2;
    const int hi = lane >> 5, lq = lane & 31;
    int bh, qblk;
    if (!attn_block_map(p, 512, bh, qblk)) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qblk * 512 + wave * 64;
    const int kvlen = p.kv_len ? p.kv_len[b] : p.seq_len;
    const int ntile = (kvlen + 63) >> 6;
    const size_t rowbase = (size_t)b * p.seq_len;

    // Q^T fragments of the wave: 8 x 16 bytes per lane.  They live in LDS, lane-linear per fragment (conflict-free b128 reads), and
    // are re-read next to the K fragments of every tile: 32 registers that the score / probability / output tiles need (at two
    // waves per SIMD a wave has 256 registers; with Q resident the compiler spilled it to scratch inside the loop)
    op16_t* qs = qsm + wave * (8 * 64 * 8) + lane * 8;
    op16x8 qf[2][4];                                                      // !QLDS: the fragments stay in registers
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = q0 + qb * 32 + lq;
        if (qr > p.seq_len - 1) qr = p.seq_len - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[qb][ks] = *reinterpret_cast<const op16x8*>(p.qk[0] + (rowbase + qr) * p.ldqk + h * 64 + ks * 16 + hi * 8);
            if (QLDS) *reinterpret_cast<op16x8*>(qs + (qb * 4 + ks) * (64 * 8)) = qf[qb][ks];
        }
    }

    // staging: a K tile and a V^T tile are 512 16-byte chunks each = one per thread
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((srow >> 1) & 7);
    const int krow = attn_kperm(srow);
    const int kcol = p.dmodel + h * 64 + schunk * 8;
    const int ldsoff = wave * 64 * 8;
    const op16_t* kptr = p.qk[0] + (rowbase + krow) * p.ldqk + kcol;
    const op16_t* vptr = p.vt[0] + ((size_t)bh * 64 + srow) * p.npad + schunk * 8;
    const size_t kstep = (size_t)64 * p.ldqk;
#define A2R_ISSUE(j_)                                                                                        \
    {                                                                                                        \
        op16_t* st_ = smem + ((j_) & (NST - 1)) * (2 * TILE);                                                \
        const op16_t* ks_ = kptr;                                                                            \
        if (((j_) * 64 + 63) > p.seq_len - 1) {                                                              \
            int key_ = (j_) * 64 + krow;                                                                     \
            if (key_ > p.seq_len - 1) key_ = p.seq_len - 1;                                                  \
            ks_ = p.qk[0] + (rowbase + key_) * p.ldqk + kcol;                                                \
        }                                                                                                    \
        attn_glds16(ks_, st_ + ldsoff);                                                                      \
        attn_glds16(vptr, st_ + TILE + ldsoff);                                                              \
        kptr += kstep;                                                                                       \
        vptr += 64;                                                                                          \
    }
#define A2R_BARRIER()                          \
    {                                          \
        __builtin_amdgcn_sched_barrier(0);     \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_s_barrier();          \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
    }
    // S^T blocks of tile j_ for both query blocks (K of the tile in sK_), raw units (exp2 units when q is pre-multiplied).  v2f's
    // -m_ref C operand (32 more live registers) does not fit next to the packed probabilities that cross the barrier here: the
    // reference point is subtracted by the packed fma of the softmax instead (the VALU segment stays shorter than the MATRIX one)
#define A2R_QK(j_, sK_)                                                                               \
    {                                                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                   \
            const op16x8 qa = QLDS ? *reinterpret_cast<const op16x8*>(qs + ks * (64 * 8)) : qf[0][ks];       \
            const op16x8 qb_ = QLDS ? *reinterpret_cast<const op16x8*>(qs + (4 + ks) * (64 * 8)) : qf[1][ks]; \
            _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                               \
                const op16x8 a = *reinterpret_cast<const op16x8*>(&(sK_)[attn_swz(kb * 32 + lq, ks * 2 + hi)]); \
                s[0][kb] = F5_MFMA32(a, qa, ks == 0 ? zero16 : s[0][kb], 0, 0, 0);                           \
                s[1][kb] = F5_MFMA32(a, qb_, ks == 0 ? zero16 : s[1][kb], 0, 0, 0);                          \
            }                                                                                                \
        }                                                                                                    \
        if ((j_) * 64 + 64 > kvlen) {                                                                        \
            _Pragma("unroll") for (int qb = 0; qb < 2; ++qb)                                                 \
                _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                             \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
                        const int key = (j_) * 64 + kb * 32 + 16 * hi + r;                                   \
                        if (key >= kvlen) s[qb][kb][r] = -INFINITY;                                          \
                    }                                                                                        \
        }                                                                                                    \
    }
    // O^T += V^T P^T with the packed probabilities of the previous tile (V^T of that tile in sV_)
#define A2R_PV(sV_)                                                                                          \
    {                                                                                                        \
        _Pragma("unroll") for (int ks4 = 0; ks4 < 4; ++ks4) {                                                \
            const int kb = ks4 >> 1, sp = ks4 & 1;                                                           \
            _Pragma("unroll") for (int db = 0; db < 2; ++db) {                                               \
                const op16x8 a = *reinterpret_cast<const op16x8*>(&(sV_)[attn_swz(db * 32 + lq, 4 * kb + 2 * hi + sp)]); \
                o[0][db] = F5_MFMA32(a, __builtin_bit_cast(op16x8, pk[0][kb][sp]), o[0][db], 0, 0, 0);       \
                o[1][db] = F5_MFMA32(a, __builtin_bit_cast(op16x8, pk[1][kb][sp]), o[1][db], 0, 0, 0);       \
            }                                                                                                \
        }                                                                                                    \
    }

    // probabilities of query block qb_ -> 16-bit MFMA operands (the score registers of the block die here)
#define A2R_PACK(qb_)                                                                                        \
    {                                                                                                        \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                     \
            _Pragma("unroll") for (int sp = 0; sp < 2; ++sp) {                                               \
                uint32_t pw[4];                                                                              \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                \
                    pw[e] = f5_pack2_bounded(s[qb_][kb][8 * sp + 2 * e], s[qb_][kb][8 * sp + 2 * e + 1]);    \
                pk[qb_][kb][sp] = u32x4{pw[0], pw[1], pw[2], pw[3]};                                         \
            }                                                                                                \
    }
    f32x16 o[2][2], s[2][2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    u32x4 pk[2][2][2];                                  // packed probabilities [query block][key block][8-key half]
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            o[qb][0][e] = 0.0f;
            o[qb][1][e] = 0.0f;
        }
    }
    float m_ref[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.0f, 0.0f};   // m_ref in exp2 units (scores * c2)
    const float c2 = PRE ? 1.0f : p.scale * 1.4426950408889634f;
    constexpr float SUM_LIMIT = 16384.0f;

    A2R_ISSUE(0);
    if (ntile > 1) {
        A2R_ISSUE(1);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!QLDS) { ATTN_PIN_Q(qf, 2, 4); }
    A2R_BARRIER();
    if (grp == 1) A2R_BARRIER();                        // group 1 runs one barrier behind

    // softmax of tile j_: fast path p = exp2(s - m_ref) with the old reference point; the row sums say whether that was safe
#define A2R_SOFTMAX(j_, sK_, FIRST_)                                                                         \
    {                                                                                                        \
        float psum[2];                                                                                       \
        bool slow = (FIRST_);                                                                                \
        if (!(FIRST_)) {                                                                                     \
            bool bad = false;                                                                                \
            _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) {                                               \
                attn_f32x2 ps2 = {0.0f, 0.0f};                                                               \
                ps2 = attn_exp_block(s[qb][0], c2, m_ref[qb], ps2);                                          \
                ps2 = attn_exp_block(s[qb][1], c2, m_ref[qb], ps2);                                          \
                psum[qb] = ps2[0] + ps2[1];                                                                  \
                bad = bad || !(psum[qb] <= SUM_LIMIT);                                                       \
                A2R_PACK(qb);                                   /* redone by the slow path */                \
            }                                                                                                \
            slow = __any(bad);                                                                               \
            if (slow) A2R_QK(j_, sK_);                         /* rare: the raw scores again (K is still in LDS) */ \
        }                                                                                                    \
        if (slow) {                                            /* wave-uniform: first tile, or a score far above m_ref */ \
            _Pragma("unroll") for (int qb = 0; qb < 2; ++qb) {                                               \
                float tmax = -INFINITY;                                                                      \
                _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                             \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[qb][kb][r]);         \
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));                                                \
                const float m_new = fmaxf(m_ref[qb], tmax * c2);                                             \
                const float alpha = __builtin_amdgcn_exp2f(m_ref[qb] - m_new);                               \
                m_ref[qb] = m_new;                                                                           \
                l_run[qb] *= alpha;                                                                          \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                             \
                    o[qb][0][e] *= alpha;                                                                    \
                    o[qb][1][e] *= alpha;                                                                    \
                }                                                                                            \
                attn_f32x2 ps2 = {0.0f, 0.0f};                                                               \
                ps2 = attn_exp_block(s[qb][0], c2, m_new, ps2);                                              \
                ps2 = attn_exp_block(s[qb][1], c2, m_new, ps2);                                              \
                psum[qb] = ps2[0] + ps2[1];                                                                  \
                A2R_PACK(qb);                                                                                \
            }                                                                                                \
        }                                                                                                    \
        l_run[0] += psum[0];                                                                                 \
        l_run[1] += psum[1];                                                                                 \
    }
#define A2R_WAIT(j_)                                                                                         \
    if ((j_) + 2 < ntile) {                                                                                  \
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); /* tile j+1 landed (this wave's share), j+2 in flight */ \
    } else {                                                                                                 \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
    }

    // ---- tile 0 (peeled: no PV yet, the softmax takes the maximum) -- waves past the sequence compute on clamped rows
    {
        if (2 < ntile) A2R_ISSUE(2);
        __builtin_amdgcn_s_setprio(1);
        A2R_QK(0, smem);
        __builtin_amdgcn_s_setprio(0);
        A2R_WAIT(0);
        A2R_BARRIER();
        A2R_SOFTMAX(0, smem, true);
        A2R_BARRIER();
    }
    // ---- tiles 1 .. ntile-1: MATRIX(j) = PV(j-1), QK(j) | VALU(j) = softmax(j)
    for (int j = 1; j < ntile; ++j) {
        if (j + 2 < ntile) A2R_ISSUE(j + 2);
        const op16_t* sK = smem + (j & (NST - 1)) * (2 * TILE);
        const op16_t* sVp = smem + ((j - 1) & (NST - 1)) * (2 * TILE) + TILE;
        __builtin_amdgcn_s_setprio(1);
        A2R_PV(sVp);
        __builtin_amdgcn_sched_barrier(0);              // the packed probabilities die here, before the score registers are born
        A2R_QK(j, sK);
        __builtin_amdgcn_s_setprio(0);
        A2R_WAIT(j);
        A2R_BARRIER();
        A2R_SOFTMAX(j, sK, false);
        A2R_BARRIER();
    }
    // ---- the last PV
    {
        const op16_t* sVp = smem + ((ntile - 1) & (NST - 1)) * (2 * TILE) + TILE;
        A2R_PV(sVp);
    }
    if (grp == 0) A2R_BARRIER();                        // balance group 1's extra barrier
#undef A2R_ISSUE
#undef A2R_BARRIER
#undef A2R_QK
#undef A2R_PV
#undef A2R_PACK
#undef A2R_SOFTMAX
#undef A2R_WAIT
    if (q0 >= p.seq_len) return;

#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l_tot;
        const int qr = q0 + qb * 32 + lq;
        if (qr < p.seq_len) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int d = db * 32 + 8 * rg + 4 * hi;
                    const float v0 = o[qb][db][rg * 4 + 0] * inv, v1 = o[qb][db][rg * 4 + 1] * inv;
                    const float v2 = o[qb][db][rg * 4 + 2] * inv, v3 = o[qb][db][rg * 4 + 3] * inv;
                    const size_t off = (rowbase + qr) * p.ldo + h * 64 + d;
                    *reinterpret_cast<u32x2*>(p.out[0] + off) = u32x2{f5_pack2_bounded(v0, v1), f5_pack2_bounded(v2, v3)};
                }
        }
    }
}
#endif  // F5_LAB (v2r)

// =================================================================================================
// v2s: v2 with the KV range split across KS wave groups INSIDE the workgroup (small batches: B*H*ceil(N/128)
// workgroups of 4 waves leave the 256 CUs with one wave per SIMD and the whole kernel is one workgroup's latency
// chain over all KV tiles).  Group g (4 waves, the same 128 queries as the other groups) walks tiles g, g+KS, ...
// with its own LDS ring; barriers are shared, so every group runs ceil(ntile/KS) iterations.  At the end groups
// 1..KS-1 park (m, l, O^T) in LDS (the rings are dead by then) and group 0 merges them flash-decoding style:
// m = max m_g, l = sum l_g 2^((m_g-m)c), O = sum O_g 2^((m_g-m)c) — the same numbers as one pass up to fp32 rounding.
// =================================================================================================
template <bool HP, int KS, int NST, bool FAST = false>
__global__ __launch_bounds__(256 * KS, 1) void f5_attn2s_kernel(F5AttnArgs p) {
    constexpr int NP = HP ? 2 : 1;
    constexpr int TILE = 64 * 64;
    constexpr int RING = NST * NP * 2 * TILE;           // elements per group ring
    static_assert(KS * RING * 2 <= 160 * 1024, "LDS budget");
    static_assert((KS - 1) * 4 * 34 * 64 * 4 <= KS * RING * 2, "merge area must fit in the rings");
    __shared__ __attribute__((aligned(16))) op16_t smem_all[KS * RING];

    // the whole argument block in ONE scalar-load clause (left alone the compiler loads each field where it is first used: three or
    // four dependent s_load / s_waitcnt rounds in the prologue of a kernel that is one latency chain at batch 1)
    asm volatile("" ::"s"(p.qk[0]), "s"(p.vt[0]), "s"(p.out[0]), "s"(p.kv_len), "s"(p.B), "s"(p.H), "s"(p.seq_len), "s"(p.npad), "s"(p.ldqk),
                 "s"(p.ldo), "s"(p.dmodel), "s"(p.scale), "s"(p.out8));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave_all >> 2, wave = wave_all & 3;
    const int tg = tid & 255;
    op16_t* smem = smem_all + grp * RING;
    const int hi = lane >> 5, lq = lane & 31;
    int bh, qblk;
    if (!attn_block_map(p, 128, bh, qblk)) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qblk * 128 + wave * 32;
    const int kvlen = p.kv_len ? p.kv_len[b] : p.seq_len;
    const int ntile = (kvlen + 63) >> 6;
    const int nit = (ntile + KS - 1) / KS;              // iterations of group 0 (the most)
    const int ntg = ntile > grp ? (ntile - grp + KS - 1) / KS : 0;   // tiles of this group
    const size_t rowbase = (size_t)b * p.seq_len;

    op16x8 qf[NP][4];
    {
        int qr = q0 + lq;
        if (qr > p.seq_len - 1) qr = p.seq_len - 1;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                qf[pp][ks] = *reinterpret_cast<const op16x8*>(p.qk[pp] + (rowbase + qr) * p.ldqk + h * 64 + ks * 16 + hi * 8);
    }

    const op16_t* kptr[NP][2];
    const op16_t* vptr[NP][2];
    int krow[2], kcol[2], ldsoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q_ = i * 256 + tg;
        const int srow = q_ >> 3;
        const int schunk = (q_ & 7) ^ ((srow >> 1) & 7);
        krow[i] = attn_kperm(srow);
        kcol[i] = p.dmodel + h * 64 + schunk * 8;
        ldsoff[i] = (i * 256 + wave * 64) * 8;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            kptr[pp][i] = p.qk[pp] + (rowbase + grp * 64 + krow[i]) * p.ldqk + kcol[i];
            vptr[pp][i] = p.vt[pp] + ((size_t)bh * 64 + srow) * p.npad + grp * 64 + schunk * 8;
        }
    }
    const size_t kstep = (size_t)64 * KS * p.ldqk;
    // jj_ = group-local tile number; the global tile is jj_*KS + grp
#define A2S_ISSUE(jj_)                                                                                       \
    {                                                                                                        \
        op16_t* st_ = smem + ((jj_) % NST) * (NP * 2 * TILE);                                                \
        const int gt_ = (jj_) * KS + grp;                                                                    \
        const bool tail_ = (gt_ * 64 + 63) > p.seq_len - 1;                                                  \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
            _Pragma("unroll") for (int pp = 0; pp < NP; ++pp) {                                              \
                const op16_t* ks_ = kptr[pp][i];                                                             \
                if (tail_) {                                                                                 \
                    int key_ = gt_ * 64 + krow[i];                                                           \
                    if (key_ > p.seq_len - 1) key_ = p.seq_len - 1;                                          \
                    ks_ = p.qk[pp] + (rowbase + key_) * p.ldqk + kcol[i];                                    \
                }                                                                                            \
                attn_glds16(ks_, st_ + (pp * 2) * TILE + ldsoff[i]);                                         \
                attn_glds16(vptr[pp][i], st_ + (pp * 2 + 1) * TILE + ldsoff[i]);                             \
                kptr[pp][i] += kstep;                                                                        \
                vptr[pp][i] += 64 * KS;                                                                      \
            }                                                                                                \
        }                                                                                                    \
    }

    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        o[0][e] = 0.0f;
        o[1][e] = 0.0f;
    }
    float m_run = -INFINITY, l_run = 0.0f;
    const float c2 = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;

    if (ntg > 0) A2S_ISSUE(0);
    if (NST == 3 && ntg > 1) A2S_ISSUE(1);
    ATTN_PIN_Q(qf, NP, 4);

    for (int jj = 0; jj < nit; ++jj) {
        if (NST == 3 && jj + 1 < ntg) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (jj + NST - 1 < ntg) A2S_ISSUE(jj + NST - 1);
        if (jj >= ntg) continue;                          // wave-uniform: this group has no tile left (still meets the barrier)

        const op16_t* st = smem + (jj % NST) * (NP * 2 * TILE);
        const op16_t* sK = st;
        const op16_t* sV = st + TILE;
        const op16_t* sKl = st + (NP - 1) * 2 * TILE;
        const op16_t* sVl = st + (NP - 1) * 2 * TILE + TILE;

        f32x16 s[2];
        const int key0 = (jj * KS + grp) * 64;
        // scores of the tile (raw units) with the key-padding tail masked
#define A2S_QK()                                                                                             \
        {                                                                                                    \
            __builtin_amdgcn_s_setprio(1);                                                                   \
            _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                               \
                _Pragma("unroll") for (int e = 0; e < 16; ++e) s[kb][e] = 0.0f;                              \
                _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                           \
                    const int off = attn_swz(kb * 32 + lq, ks * 2 + hi);                                     \
                    const op16x8 a = *reinterpret_cast<const op16x8*>(&sK[off]);                             \
                    s[kb] = F5_MFMA32(a, qf[0][ks], s[kb], 0, 0, 0);                                         \
                    if (HP) {                                                                                \
                        const op16x8 al = *reinterpret_cast<const op16x8*>(&sKl[off]);                       \
                        s[kb] = F5_MFMA32(al, qf[0][ks], s[kb], 0, 0, 0);                                    \
                        s[kb] = F5_MFMA32(a, qf[NP - 1][ks], s[kb], 0, 0, 0);                                \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
            __builtin_amdgcn_s_setprio(0);                                                                   \
            if (key0 + 64 > kvlen) {                                                                         \
                _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                             \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
                        const int key = key0 + kb * 32 + 16 * hi + r;                                        \
                        if (key >= kvlen) s[kb][r] = -INFINITY;                                              \
                    }                                                                                        \
            }                                                                                                \
        }
        A2S_QK();
        // FAST (see f5_attn2f_kernel): after a group's first tile the exponentials are taken against the standing reference point
        // m_run without looking for the tile maximum -- at this grid size the kernel is one workgroup's latency chain, and
        // max -> cross-lane exchange (an LDS round trip) -> rescale test sits in front of every tile's exponentials.  The row
        // sums tell whether that was safe (every p <= 2^14); if not, the tile is redone the slow way.
        float psum = 0.0f;
        bool slow = !FAST || HP || jj == 0;
        if (!slow) {
            const float mc = m_run * c2;
            attn_f32x2 ps2 = {0.0f, 0.0f};
            ps2 = attn_exp_block(s[0], c2, mc, ps2);
            ps2 = attn_exp_block(s[1], c2, mc, ps2);
            psum = ps2[0] + ps2[1];
            slow = __any(!(psum <= 16384.0f));
            if (slow) A2S_QK();
        }
        if (slow) {
            float tmax = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[kb][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            if (__any(tmax > m_run)) {
                const float m_new = fmaxf(m_run, tmax);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    o[0][e] *= alpha;
                    o[1][e] *= alpha;
                }
            }
            const float mc = m_run * c2;
            attn_f32x2 ps2 = {0.0f, 0.0f};
            ps2 = attn_exp_block(s[0], c2, mc, ps2);
            ps2 = attn_exp_block(s[1], c2, mc, ps2);
            psum = ps2[0] + ps2[1];
        }
        l_run += psum;
#undef A2S_QK

#pragma unroll
        for (int ks4 = 0; ks4 < 4; ++ks4) {
            const int kb = ks4 >> 1, sp = ks4 & 1;
            uint32_t pw[4], pwl[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p0 = s[kb][8 * sp + 2 * e], p1 = s[kb][8 * sp + 2 * e + 1];
                pw[e] = f5_pack2_bounded(p0, p1);
                if (HP) pwl[e] = f5_pack2_lo(p0, p1);
            }
            const op16x8 pb = __builtin_bit_cast(op16x8, u32x4{pw[0], pw[1], pw[2], pw[3]});
            op16x8 pbl = pb;
            if (HP) pbl = __builtin_bit_cast(op16x8, u32x4{pwl[0], pwl[1], pwl[2], pwl[3]});
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int voff = attn_swz(db * 32 + lq, 4 * kb + 2 * hi + sp);
                const op16x8 a = *reinterpret_cast<const op16x8*>(&sV[voff]);
                o[db] = F5_MFMA32(a, pb, o[db], 0, 0, 0);
                if (HP) {
                    const op16x8 al = *reinterpret_cast<const op16x8*>(&sVl[voff]);
                    o[db] = F5_MFMA32(al, pb, o[db], 0, 0, 0);
                    o[db] = F5_MFMA32(a, pbl, o[db], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
    }
#undef A2S_ISSUE

    // ---- merge the KS partial states (same lane of the same wave index in every group holds the same elements)
    __syncthreads();                                     // every ring is dead
    float* mrg = reinterpret_cast<float*>(smem_all);     // [KS-1][4 waves][34][64 lanes]
    if (grp > 0) {
        float* dst = mrg + (size_t)((grp - 1) * 4 + wave) * 34 * 64 + lane;
        dst[0] = m_run;
        dst[64] = l_run;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[(2 + db * 16 + e) * 64] = o[db][e];
    }
    __syncthreads();
    if (grp > 0) return;
    float m_all = m_run;
#pragma unroll
    for (int g = 1; g < KS; ++g) m_all = fmaxf(m_all, mrg[(size_t)((g - 1) * 4 + wave) * 34 * 64 + lane]);
    {
        const float a0 = __builtin_amdgcn_exp2f((m_run - m_all) * c2);
        l_run *= a0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            o[0][e] *= a0;
            o[1][e] *= a0;
        }
    }
#pragma unroll
    for (int g = 1; g < KS; ++g) {
        const float* src = mrg + (size_t)((g - 1) * 4 + wave) * 34 * 64 + lane;
        const float ag = __builtin_amdgcn_exp2f((src[0] - m_all) * c2);    // empty group: m = -inf -> 0
        l_run += src[64] * ag;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[db][e] += src[(2 + db * 16 + e) * 64] * ag;
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + lq;
    if (!HP && p.out8) {
        if (qr < p.seq_len) attn_store_f8(p, o, inv, rowbase + qr, h, hi);
    } else if (qr < p.seq_len) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = db * 32 + 8 * rg + 4 * hi;
                const float v0 = o[db][rg * 4 + 0] * inv, v1 = o[db][rg * 4 + 1] * inv;
                const float v2 = o[db][rg * 4 + 2] * inv, v3 = o[db][rg * 4 + 3] * inv;
                const size_t off = (rowbase + qr) * p.ldo + h * 64 + d;
                *reinterpret_cast<u32x2*>(p.out[0] + off) = u32x2{f5_pack2_bounded(v0, v1), f5_pack2_bounded(v2, v3)};
                if (HP && p.out[1])
                    *reinterpret_cast<u32x2*>(p.out[1] + off) = u32x2{f5_pack2_lo(v0, v1), f5_pack2_lo(v2, v3)};
            }
    }
}

#if F5_LAB   // in-wave software pipelining experiments (v3 / v4 / v5 / v6), all measured slower
// =================================================================================================
// v3 (bf16 only): v2 + software pipelining inside the wave.  Ablations of v2 (tools/attn_ablate.py) show that
// QK^T MFMAs, softmax VALU and PV MFMAs each cost ~1/3 of the time and do not overlap: co-resident waves run the
// same phase at the same time.  Here the 8 MFMAs of S(j+1) = K(j+1) Q^T are issued in the same basic block as the
// exp2 / sum / pack of tile j (independent registers), so the matrix pipe works under the softmax of every wave.
// Ring of 3 K/V tiles: slot j%3 feeds PV(j), slot (j+1)%3 feeds S(j+1), slot (j+2)%3 is being loaded.
// =================================================================================================
template <int WPS>
__global__ __launch_bounds__(256, WPS) void f5_attn3_kernel(F5AttnArgs p) {
    constexpr int TILE = 64 * 64;
    __shared__ __attribute__((aligned(16))) op16_t smem[3 * 2 * TILE];   // [stage][K | V^T][64*64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    int bh, qblk;
    if (!attn_block_map(p, 128, bh, qblk)) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qblk * 128 + wave * 32;
    const int kvlen = p.kv_len ? p.kv_len[b] : p.seq_len;
    const int ntile = (kvlen + 63) >> 6;
    const size_t rowbase = (size_t)b * p.seq_len;

    op16x8 qf[4];
    {
        int qr = q0 + lq;
        if (qr > p.seq_len - 1) qr = p.seq_len - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[ks] = *reinterpret_cast<const op16x8*>(p.qk[0] + (rowbase + qr) * p.ldqk + h * 64 + ks * 16 + hi * 8);
    }
    // staging pointers (advanced by one KV tile per issue) and wave-uniform LDS offsets
    const op16_t* kptr[2];
    const op16_t* vptr[2];
    int krow[2], kcol[2], ldsoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q_ = i * 256 + tid;
        const int srow = q_ >> 3;
        const int schunk = (q_ & 7) ^ ((srow >> 1) & 7);
        krow[i] = attn_kperm(srow);
        kcol[i] = p.dmodel + h * 64 + schunk * 8;
        kptr[i] = p.qk[0] + (rowbase + krow[i]) * p.ldqk + kcol[i];
        vptr[i] = p.vt[0] + ((size_t)bh * 64 + srow) * p.npad + schunk * 8;
        ldsoff[i] = (i * 256 + wave * 64) * 8;
    }
    const size_t kstep = (size_t)64 * p.ldqk;
    // issue KV tile j_ (tiles are issued in increasing order: pointers advance by one tile per issue)
#define A3_ISSUE(j_)                                                                                         \
    {                                                                                                        \
        op16_t* st_ = smem + ((j_) % 3) * (2 * TILE);                                                        \
        const bool tail_ = ((j_) * 64 + 63) > p.seq_len - 1;                                                 \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
            const op16_t* ks_ = kptr[i];                                                                     \
            if (tail_) {                                                                                     \
                int key_ = (j_) * 64 + krow[i];                                                              \
                if (key_ > p.seq_len - 1) key_ = p.seq_len - 1;                                              \
                ks_ = p.qk[0] + (rowbase + key_) * p.ldqk + kcol[i];                                         \
            }                                                                                                \
            attn_glds16(ks_, st_ + ldsoff[i]);                                                               \
            attn_glds16(vptr[i], st_ + TILE + ldsoff[i]);                                                    \
            kptr[i] += kstep;                                                                                \
            vptr[i] += 64;                                                                                   \
        }                                                                                                    \
    }
#define A3_SCORES(dst_, slot_)                                                                               \
    {                                                                                                        \
        const op16_t* sK_ = smem + (slot_) * (2 * TILE);                                                     \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                   \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) dst_[kb][e] = 0.0f;                               \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                               \
                const op16x8 a_ = *reinterpret_cast<const op16x8*>(&sK_[koff[kb][ks]]);                      \
                dst_[kb] = F5_MFMA32(a_, qf[ks], dst_[kb], 0, 0, 0);          \
            }                                                                                                \
        }                                                                                                    \
    }
    // loop-invariant LDS fragment offsets
    int koff[2][4], voff[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            koff[kb][ks] = attn_swz(kb * 32 + lq, ks * 2 + hi);
            voff[kb][ks] = attn_swz(kb * 32 + lq, ks * 2 + hi);   // [db][ks4]: row db*32+lq, chunk 4*(ks4>>1) + 2*hi + (ks4&1)
        }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int ks4 = 0; ks4 < 4; ++ks4) voff[db][ks4] = attn_swz(db * 32 + lq, 4 * (ks4 >> 1) + 2 * hi + (ks4 & 1));

    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        o[0][e] = 0.0f;
        o[1][e] = 0.0f;
    }
    float m_run = -INFINITY, l_run = 0.0f;
    const float c2 = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;

    A3_ISSUE(0);
    if (ntile > 1) {
        A3_ISSUE(1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    f32x16 sa[2], sb[2];
    A3_SCORES(sa, 0);

    // one KV tile: softmax + PV of tile j_ from scores cur_, while the scores of tile j_+1 are produced into nxt_
#define A3_BODY(j_, cur_, nxt_)                                                                              \
    {                                                                                                        \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
        asm volatile("" ::: "memory");                                                                       \
        __builtin_amdgcn_s_barrier();                                                                        \
        asm volatile("" ::: "memory");                                                                       \
        if ((j_) + 2 < ntile) A3_ISSUE((j_) + 2);                                                            \
        const int key0_ = (j_) * 64;                                                                         \
        if (key0_ + 64 > kvlen) {                                                                            \
            _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                 \
                _Pragma("unroll") for (int r = 0; r < 16; ++r)                                               \
                    if (key0_ + kb * 32 + 16 * hi + r >= kvlen) cur_[kb][r] = -INFINITY;                     \
        }                                                                                                    \
        float tmax_ = -INFINITY;                                                                             \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                     \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) tmax_ = fmaxf(tmax_, cur_[kb][r]);                \
        tmax_ = fmaxf(tmax_, __shfl_xor(tmax_, 32, 64));                                                     \
        if (__any(tmax_ > m_run)) {                                                                          \
            const float m_new_ = fmaxf(m_run, tmax_);                                                        \
            const float alpha_ = __builtin_amdgcn_exp2f((m_run - m_new_) * c2);                              \
            m_run = m_new_;                                                                                  \
            l_run *= alpha_;                                                                                 \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                 \
                o[0][e] *= alpha_;                                                                           \
                o[1][e] *= alpha_;                                                                           \
            }                                                                                                \
        }                                                                                                    \
        const float mc_ = m_run * c2;                                                                        \
        A3_SCORES(nxt_, ((j_) + 1) % 3);                                                                     \
        float psum_ = 0.0f;                                                                                  \
        uint32_t pw_[2][8];                                                                                  \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                     \
            _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                              \
                const float p0_ = __builtin_amdgcn_exp2f(cur_[kb][r] * c2 - mc_);                            \
                const float p1_ = __builtin_amdgcn_exp2f(cur_[kb][r + 1] * c2 - mc_);                        \
                psum_ += p0_ + p1_;                                                                          \
                pw_[kb][r >> 1] = f5_pack2_bounded(p0_, p1_);                                                        \
            }                                                                                                \
        l_run += psum_;                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
            __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);                                              \
        }                                                                                                    \
        const op16_t* sV_ = smem + ((j_) % 3) * (2 * TILE) + TILE;                                           \
        _Pragma("unroll") for (int ks4 = 0; ks4 < 4; ++ks4) {                                                \
            const int kb = ks4 >> 1, sp = ks4 & 1;                                                           \
            const op16x8 pb_ = __builtin_bit_cast(                                                           \
                op16x8, u32x4{pw_[kb][4 * sp], pw_[kb][4 * sp + 1], pw_[kb][4 * sp + 2], pw_[kb][4 * sp + 3]}); \
            _Pragma("unroll") for (int db = 0; db < 2; ++db) {                                               \
                const op16x8 a_ = *reinterpret_cast<const op16x8*>(&sV_[voff[db][ks4]]);                     \
                o[db] = F5_MFMA32(a_, pb_, o[db], 0, 0, 0);                    \
            }                                                                                                \
        }                                                                                                    \
    }

    // two tiles per trip so that the score registers ping-pong without copies
    for (int j = 0; j < ntile; j += 2) {
        A3_BODY(j, sa, sb);
        if (j + 1 < ntile) A3_BODY(j + 1, sb, sa);
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + lq;
    if (qr < p.seq_len) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = db * 32 + 8 * rg + 4 * hi;
                const float v0 = o[db][rg * 4 + 0] * inv, v1 = o[db][rg * 4 + 1] * inv;
                const float v2 = o[db][rg * 4 + 2] * inv, v3 = o[db][rg * 4 + 3] * inv;
                const size_t off = (rowbase + qr) * p.ldo + h * 64 + d;
                *reinterpret_cast<u32x2*>(p.out[0] + off) = u32x2{f5_pack2_bounded(v0, v1), f5_pack2_bounded(v2, v3)};
            }
    }
}

// =================================================================================================
// v5 (one-pass modes, large grids): software pipeline INSIDE the wave.  On this chip a wave's VALU / LDS / DMA instructions
// only overlap matrix work when they sit between the MFMAs of the SAME instruction stream (<= ~5 issue slots per 32-cycle MFMA
// gap; two co-resident waves do not interleave MFMA with VALU for each other -- tools/probes/coissue.hip, and the round-2
// issue-priority experiment tools/attn_prio_bench.py moved nothing).  v2 runs S = K Q^T -> softmax -> O += V^T P strictly one
// after the other per tile, so its MFMAs and its ~9 VALU per MFMA add up.  Here iteration j issues, as ONE block of
// independent work,
//     MFMA:  O^T += V^T(j-1) P(j-1)      (P of the previous tile, packed last iteration)
//            S(j+1) = K(j+1) Q^T          (scores of the next tile, consumed next iteration)
//     VALU:  softmax of S(j) -> P(j)      (scores computed last iteration)
// and only the running-max rescale of O (a per-lane multiply by alpha(j)) follows it.  The source interleaves one MFMA with a
// slice of the softmax; SCHED pins those slices with sched_barrier so the order survives instruction scheduling.
// Workgroup = 8 waves x 32 queries (256 queries), <= 256 VGPRs => 2 waves per SIMD; K / V^T tiles of 64 keys arrive by
// global_load_lds into 3-stage rings two iterations ahead (K(j+3), V^T(j+1) issued in iteration j), counted vmcnt, one barrier
// per iteration.  Layouts (permuted K rows, XOR-swizzled 128-byte rows, P fed from the S accumulators) are v2's.
// =================================================================================================
template <bool HAS_PV, bool HAS_QK, bool SCHED, bool PK>
__device__ __forceinline__ void attn5_body(const op16_t* sKn, const op16_t* sVp, const op16x8 (&qf)[4], f32x16 (&s_cur)[2],
                                           f32x16 (&s_nxt)[2], f32x16 (&o)[2], const uint32_t (&pk_prev)[16], uint32_t (&pk_cur)[16],
                                           float& m_run, float& l_run, float c2, int lq, int hi, int nmask) {
    // nmask: keys >= nmask of this tile are padding (64 = none).  Only the LAST tile of a sequence can be partial, i.e. only the
    // instantiations without a next tile carry the masking code (in the middle iterations it would be if-converted into 64
    // unconditional compare / select instructions per tile)
    if (!HAS_QK && nmask < 64) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kb * 32 + 16 * hi + r >= nmask) s_cur[kb][r] = -INFINITY;
    }
    // MFMA slots of this iteration, in order: PV k-steps 0..3 (2 MFMAs each: d blocks 0, 1), then QK k-steps 0..3 (2 each: key
    // blocks 0, 1).  Slot i's LDS fragment is read TWO slots ahead (fr[i & 3] rotates), so a ds_read_b128 has ~2 x (MFMA + its
    // fillers) to land before the s_waitcnt in front of its MFMA.
    constexpr int NPV = HAS_PV ? 8 : 0, NQK = HAS_QK ? 8 : 0, NSLOT = NPV + NQK;
    op16x8 fr[4];
    auto frag = [&](int i) -> op16x8 {                       // fragment of slot i (compile-time i after unrolling)
        if (i < NPV) {
            const int ks4 = i >> 1, db = i & 1;
            return *reinterpret_cast<const op16x8*>(&sVp[attn_swz(db * 32 + lq, 4 * (ks4 >> 1) + 2 * hi + (ks4 & 1))]);
        }
        const int q = i - NPV, ks = q >> 1, kb = q & 1;
        return *reinterpret_cast<const op16x8*>(&sKn[attn_swz(kb * 32 + lq, ks * 2 + hi)]);
    };
    if (NSLOT > 0) fr[0] = frag(0);
    if (NSLOT > 1) fr[1] = frag(1);
    if (HAS_QK) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s_nxt[0][e] = 0.0f;
            s_nxt[1][e] = 0.0f;
        }
    }
#define A5_SLOT(i)                                                                                                      \
    if ((i) < NSLOT) {                                                                                                  \
        if ((i) + 2 < NSLOT) fr[((i) + 2) & 3] = frag((i) + 2);                                                         \
        if ((i) < NPV) {                                                                                                \
            const int ks4_ = (i) >> 1;                                                                                  \
            const op16x8 pb_ = __builtin_bit_cast(op16x8, u32x4{pk_prev[4 * ks4_], pk_prev[4 * ks4_ + 1], pk_prev[4 * ks4_ + 2], \
                                                                pk_prev[4 * ks4_ + 3]});                                \
            o[(i) & 1] = F5_MFMA32(fr[(i) & 3], pb_, o[(i) & 1], 0, 0, 0);                                              \
        } else {                                                                                                        \
            const int q_ = (i) - NPV;                                                                                   \
            s_nxt[q_ & 1] = F5_MFMA32(fr[(i) & 3], qf[q_ >> 1], s_nxt[q_ & 1], 0, 0, 0);                                \
        }                                                                                                               \
    }                                                                                                                   \
    if (SCHED) __builtin_amdgcn_sched_barrier(0);
    // ---- VALU part 1 (slots 0..3): scale the scores by c2 = scale * log2(e) with packed multiplies -- their results are
    // ordinary VALU outputs, so the maxima need no canonicalising v_max in front of every MFMA output -- and reduce them to the
    // tile maximum; lanes l and l + 32 own the two halves of a query's keys: one v_permlane32_swap
    // PK: packed f32 multiplies / subtracts / adds (half the instruction count, but a v_pk_*_f32 costs more than its issue
    // slot beside MFMAs); !PK: the same arithmetic as single-issue v_mul / v_sub / v_add
    attn_f32x2 t2[2][8];
    const attn_f32x2 c2v = {c2, c2};
#define A5_SCALE(kb)                                                                                                    \
    _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                                     \
        if (PK) {                                                                                                       \
            t2[kb][r] = attn_f32x2{s_cur[kb][2 * r], s_cur[kb][2 * r + 1]} * c2v;                                       \
        } else {                                                                                                        \
            float a_ = s_cur[kb][2 * r] * c2, b_ = s_cur[kb][2 * r + 1] * c2;                                           \
            asm volatile("" : "+v"(a_), "+v"(b_));                                                                      \
            t2[kb][r] = attn_f32x2{a_, b_};                                                                             \
        }                                                                                                               \
    }
    A5_SCALE(0)
    A5_SLOT(0)
    A5_SCALE(1)
#undef A5_SCALE
    A5_SLOT(1)
    float tm[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r0 = q * 4;
        tm[q] = fmaxf(fmaxf(fmaxf(t2[0][r0][0], t2[0][r0][1]), fmaxf(t2[0][r0 + 1][0], t2[0][r0 + 1][1])),
                      fmaxf(fmaxf(t2[0][r0 + 2][0], t2[0][r0 + 2][1]), fmaxf(t2[0][r0 + 3][0], t2[0][r0 + 3][1])));
    }
    A5_SLOT(2)
#pragma unroll
    for (int q = 2; q < 4; ++q) {
        const int r0 = (q & 1) * 4;
        tm[q] = fmaxf(fmaxf(fmaxf(t2[1][r0][0], t2[1][r0][1]), fmaxf(t2[1][r0 + 1][0], t2[1][r0 + 1][1])),
                      fmaxf(fmaxf(t2[1][r0 + 2][0], t2[1][r0 + 2][1]), fmaxf(t2[1][r0 + 3][0], t2[1][r0 + 3][1])));
    }
    const float tmax_own = fmaxf(fmaxf(tm[0], tm[1]), fmaxf(tm[2], tm[3]));
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax_own), __float_as_uint(tmax_own), false, false);
    // after the swap (sw[0], sw[1]) = (own, partner) in the lower half-wave and (partner, own) in the upper one
    const float mc = fmaxf(m_run, fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));   // running maximum, in exp2 units
    const float alpha = __builtin_amdgcn_exp2f(m_run - mc);       // m_run = -inf on the first tile: alpha = 0 (o = l = 0 then)
    m_run = mc;
    A5_SLOT(3)
    // ---- VALU part 2 (slots 4..15): p = exp2(t - m), row sum, pack to the operand type, 1-2 pairs per MFMA slot.  Every packed
    // P dword passes through an empty asm so that the slice stays HERE (it is consumed next iteration and would be sunk there)
    attn_f32x2 ps2 = {0.0f, 0.0f};
    const attn_f32x2 mcv = {mc, mc};
#define A5_EXP(kb, r)                                                                                                   \
    {                                                                                                                   \
        float e0_, e1_;                                                                                                 \
        if (PK) {                                                                                                       \
            const attn_f32x2 t = t2[kb][r] - mcv;                                                                       \
            e0_ = __builtin_amdgcn_exp2f(t[0]);                                                                         \
            e1_ = __builtin_amdgcn_exp2f(t[1]);                                                                         \
            ps2 += attn_f32x2{e0_, e1_};                                                                                \
        } else {                                                                                                        \
            float a_ = t2[kb][r][0] - mc, b_ = t2[kb][r][1] - mc;                                                       \
            asm volatile("" : "+v"(a_), "+v"(b_));                                                                      \
            e0_ = __builtin_amdgcn_exp2f(a_);                                                                           \
            e1_ = __builtin_amdgcn_exp2f(b_);                                                                           \
            ps2[0] += e0_;                                                                                              \
            asm volatile("" : "+v"(ps2[0]));                                                                            \
            ps2[1] += e1_;                                                                                              \
            asm volatile("" : "+v"(ps2[1]));                                                                            \
        }                                                                                                               \
        pk_cur[(kb) * 8 + (r)] = f5_pack2_bounded(e0_, e1_);                                                            \
        asm volatile("" : "+v"(pk_cur[(kb) * 8 + (r)]));                                                                \
    }
    A5_EXP(0, 0) A5_SLOT(4)
    A5_EXP(0, 1) A5_SLOT(5)
    A5_EXP(0, 2) A5_SLOT(6)
    A5_EXP(0, 3) A5_SLOT(7)
    A5_EXP(0, 4) A5_EXP(0, 5) A5_SLOT(8)
    A5_EXP(0, 6) A5_SLOT(9)
    A5_EXP(0, 7) A5_EXP(1, 0) A5_SLOT(10)
    A5_EXP(1, 1) A5_SLOT(11)
    A5_EXP(1, 2) A5_EXP(1, 3) A5_SLOT(12)
    A5_EXP(1, 4) A5_SLOT(13)
    A5_EXP(1, 5) A5_EXP(1, 6) A5_SLOT(14)
    A5_EXP(1, 7) A5_SLOT(15)
#undef A5_EXP
#undef A5_SLOT
    l_run = l_run * alpha + (ps2[0] + ps2[1]);
    // O (now including tile j-1) moves from the old to the new running maximum; exact skip when no lane's maximum moved
    if (HAS_PV && __any(alpha != 1.0f)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            o[0][e] *= alpha;
            o[1][e] *= alpha;
        }
    }
}

// NW waves x 32 queries per workgroup: NW = 8 (one workgroup per CU, 256 queries) or 4 (two per CU, independent barriers)
template <bool PK, int NW>
__global__ __launch_bounds__(64 * NW, 2) void f5_attn5_kernel(F5AttnArgs p) {
    constexpr bool SCHED = true;
    constexpr int NST = 3;
    constexpr int TILE = 64 * 64;
    constexpr int NCH = 8 / NW;                                            // 16-byte chunks of K (and of V^T) per thread per tile
    __shared__ __attribute__((aligned(16))) op16_t smem[NST * 2 * TILE];   // [stage][K | V^T][64 x 64]: 48 KB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    int bh, qblk;
    if (!attn_block_map(p, 32 * NW, bh, qblk)) return;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qblk * (32 * NW) + wave * 32;
    const int kvlen = p.kv_len ? p.kv_len[b] : p.seq_len;
    const int nt = (kvlen + 63) >> 6;
    const size_t rowbase = (size_t)b * p.seq_len;

    op16x8 qf[4];
    {
        int qr = q0 + lq;
        if (qr > p.seq_len - 1) qr = p.seq_len - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[ks] = *reinterpret_cast<const op16x8*>(p.qk[0] + (rowbase + qr) * p.ldqk + h * 64 + ks * 16 + hi * 8);
    }
    // staging: NCH 16-byte chunks of K and of V^T per thread per tile (64 * NW threads x NCH x 16 B = one 8 KB tile image)
    int krow[NCH], ldsoff[NCH];
    const op16_t* kbase[NCH];
    const op16_t* vbase[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int q_ = i * (64 * NW) + tid;
        const int srow = q_ >> 3;
        const int schunk = (q_ & 7) ^ ((srow >> 1) & 7);
        krow[i] = attn_kperm(srow);
        kbase[i] = p.qk[0] + rowbase * p.ldqk + p.dmodel + h * 64 + schunk * 8;
        vbase[i] = p.vt[0] + ((size_t)bh * 64 + srow) * p.npad + schunk * 8;
        ldsoff[i] = (i * (64 * NW) + wave * 64) * 8;
    }
    auto issue_k = [&](int t) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            int key = t * 64 + krow[i];
            if (key > p.seq_len - 1) key = p.seq_len - 1;
            attn_glds16(kbase[i] + (size_t)key * p.ldqk, smem + (t % NST) * (2 * TILE) + ldsoff[i]);
        }
    };
    auto issue_v = [&](int t) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) attn_glds16(vbase[i] + t * 64, smem + (t % NST) * (2 * TILE) + TILE + ldsoff[i]);
    };
#define A5_BARRIER()                               \
    {                                              \
        asm volatile("" ::: "memory");             \
        __builtin_amdgcn_s_barrier();              \
        asm volatile("" ::: "memory");             \
    }

    f32x16 o[2], s_a[2], s_b[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        o[0][e] = 0.0f;
        o[1][e] = 0.0f;
        s_a[0][e] = 0.0f;
        s_a[1][e] = 0.0f;
    }
    uint32_t pk_a[16], pk_b[16];
    float m_run = -INFINITY, l_run = 0.0f;
    const float c2 = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;

    // ---- prologue: K(0..2), V^T(0); S(0) = K(0) Q^T
    issue_k(0);
    if (nt > 1) issue_k(1);
    if (nt > 2) issue_k(2);
    issue_v(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    A5_BARRIER();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const op16x8 k0 = *reinterpret_cast<const op16x8*>(&smem[attn_swz(lq, ks * 2 + hi)]);
        const op16x8 k1 = *reinterpret_cast<const op16x8*>(&smem[attn_swz(32 + lq, ks * 2 + hi)]);
        s_a[0] = F5_MFMA32(k0, qf[ks], s_a[0], 0, 0, 0);
        s_a[1] = F5_MFMA32(k1, qf[ks], s_a[1], 0, 0, 0);
    }
    if (nt > 3) A5_BARRIER();        // iteration 0 refills K(0)'s slot with K(3): every wave must be done reading K(0)

    // ---- main loop.  Iteration j: S(j) in s_cur, P(j-1) in pk_prev; reads K(j+1), V^T(j-1); issues K(j+3), V^T(j+1).
    // (s_a, pk_a) hold the even tiles, (s_b, pk_b) the odd ones: the loop runs two iterations per trip so that the roles
    // alternate without register copies; the first and the last iteration are peeled (no P yet / no next tile).
#define A5_WAIT(n_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_) : "memory")
#define A5_ITER(j_, HAS_PV_, HAS_QK_, s_cur_, s_nxt_, pk_prev_, pk_cur_)                                                \
    {                                                                                                                   \
        const int jt = (j_);   /* not `j`: the argument may be a variable of that name */                               \
        if (jt + 3 < nt) issue_k(jt + 3);                                                                               \
        if (jt + 1 < nt) issue_v(jt + 1);                                                                               \
        const op16_t* sKn = smem + ((jt + 1) % NST) * (2 * TILE);                                                       \
        const op16_t* sVp = smem + ((jt + NST - 1) % NST) * (2 * TILE) + TILE;                                          \
        const int nmask = kvlen - jt * 64 < 64 ? kvlen - jt * 64 : 64;                                                  \
        attn5_body<HAS_PV_, HAS_QK_, SCHED, PK>(sKn, sVp, qf, s_cur_, s_nxt_, o, pk_prev_, pk_cur_, m_run, l_run, c2, lq, hi, nmask); \
        /* next iteration reads K(j+2), V^T(j): everything but the group just issued must have landed */              \
        const int pend = (jt + 3 < nt ? 1 : 0) + (jt + 1 < nt ? 1 : 0);                                                 \
        if (pend == 2) {                                                                                                \
            A5_WAIT(2 * NCH);                                                                                           \
        } else if (pend == 1) {                                                                                         \
            A5_WAIT(NCH);                                                                                               \
        } else {                                                                                                        \
            A5_WAIT(0);                                                                                                 \
        }                                                                                                               \
        A5_BARRIER();                                                                                                   \
    }
    // O += V^T(nt-1) P(nt-1) after the last iteration (its P sits in pk_a for an even last tile, pk_b for an odd one; the two
    // arrays must never be selected between at run time, or they are demoted from registers to scratch memory)
#define A5_FINAL_PV(pk_)                                                                                                \
    {                                                                                                                   \
        const op16_t* sVl = smem + ((nt - 1) % NST) * (2 * TILE) + TILE;                                                \
        _Pragma("unroll") for (int ks4 = 0; ks4 < 4; ++ks4) {                                                           \
            const op16x8 pb = __builtin_bit_cast(op16x8, u32x4{pk_[4 * ks4], pk_[4 * ks4 + 1], pk_[4 * ks4 + 2], pk_[4 * ks4 + 3]}); \
            const int ch = 4 * (ks4 >> 1) + 2 * hi + (ks4 & 1);                                                         \
            o[0] = F5_MFMA32(*reinterpret_cast<const op16x8*>(&sVl[attn_swz(lq, ch)]), pb, o[0], 0, 0, 0);              \
            o[1] = F5_MFMA32(*reinterpret_cast<const op16x8*>(&sVl[attn_swz(32 + lq, ch)]), pb, o[1], 0, 0, 0);         \
        }                                                                                                               \
    }
    if (nt == 1) {
        A5_ITER(0, false, false, s_a, s_b, pk_b, pk_a);
        A5_FINAL_PV(pk_a);
    } else {
        A5_ITER(0, false, true, s_a, s_b, pk_b, pk_a);
        int j = 1;
        for (; j + 2 < nt; j += 2) {                     // middle iterations j (odd), j + 1 (even), both < nt - 1
            A5_ITER(j, true, true, s_b, s_a, pk_a, pk_b);
            A5_ITER(j + 1, true, true, s_a, s_b, pk_b, pk_a);
        }
        if (j + 1 < nt) {                                // one middle iteration left (odd), the last one is even
            A5_ITER(j, true, true, s_b, s_a, pk_a, pk_b);
            A5_ITER(j + 1, true, false, s_a, s_b, pk_b, pk_a);
            A5_FINAL_PV(pk_a);
        } else {                                         // the last iteration is odd
            A5_ITER(j, true, false, s_b, s_a, pk_a, pk_b);
            A5_FINAL_PV(pk_b);
        }
    }
#undef A5_FINAL_PV
#undef A5_ITER
#undef A5_WAIT
#undef A5_BARRIER
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qr = q0 + lq;
    if (qr < p.seq_len) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = db * 32 + 8 * rg + 4 * hi;
                const size_t off = (rowbase + qr) * p.ldo + h * 64 + d;
                *reinterpret_cast<u32x2*>(p.out[0] + off) =
                    u32x2{f5_pack2_bounded(o[db][rg * 4 + 0] * inv, o[db][rg * 4 + 1] * inv),
                          f5_pack2_bounded(o[db][rg * 4 + 2] * inv, o[db][rg * 4 + 3] * inv)};
            }
    }
}

#endif  // F5_LAB
#if F5_LAB
int f5_attn_version = 2;   // 1 = register-staged, 2 = global_load_lds ring (default), 3/4 = ring + in-wave software pipelining (measured slower); 5 / 6 = pipelined experiment
int f5_attn_variant = 0;   // experiment bits (see the launcher): 1, 2 = pipelined kernel variants, 4 = plain 2-D block numbering, 8 = eager O rescale in the large-grid kernel
int f5_attn_ablation = 0;  // timing experiments only
int f5_attn_prio = 0;      // wide kernel: which phase holds issue priority (0 MFMA clusters, 1 none, 2 softmax section)
#else
static constexpr int f5_attn_version = 2, f5_attn_variant = 0, f5_attn_ablation = 0, f5_attn_prio = 0;   // the shipped configuration
#endif
// test hooks that choose among the SHIPPED kernels (the shape heuristics below decide otherwise)
int f5_attn_wide = -1;     // -1 auto (>= 512 workgroups of 256 queries), 0 off, 1 force: 256-query workgroups, two query blocks per wave; lab: 2 = role-split v2r
int f5_attn_kvsplit = -1;  // -1 auto, 1 / 2 / 4 = force the in-workgroup KV split
int f5_attn_pipe = 0;      // process default of F5AttnArgs::pipe (-1): 1 = in-wave software-pipelined v2p (one wave per SIMD), 0 = v2f

// 1-D XCD-aware grid (attn_block_map); the lab build's f5_attn_variant bit 2 asks for the plain 2-D numbering
static dim3 attn_grid(const F5AttnArgs& a, int qrows) {
    const int nqb = f5_cdiv(a.seq_len, qrows);
    if (f5_attn_variant & 4) return dim3(nqb, a.B * a.H);
    return dim3(nqb * 8 * f5_cdiv(a.B * a.H, 8), 1);
}

int f5_launch_attention(const F5AttnArgs& a, hipStream_t stream) {
    F5_REQUIRE(a.B > 0 && a.H > 0 && a.seq_len > 0, "attention: bad shape");
    F5_REQUIRE(a.npad % 64 == 0 && a.npad >= a.seq_len, "attention: npad must be a multiple of 64 and >= seq_len");
    F5_REQUIRE(a.ldqk % 8 == 0 && a.ldo % 4 == 0, "attention: bad leading dims");
    F5_REQUIRE(a.qk[0] && a.vt[0] && (a.out[0] || a.out8), "attention: null pointer");
    F5_REQUIRE(!a.out8 || (!a.hp && a.out8s && f5_attn_version == 2 && f5_attn_ablation == 0 && a.ldo8 % 4 == 0),
               "attention: fp8 output needs the bf16 ring kernels");
    const dim3 grid = attn_grid(a, 128);
    const long wgs128 = (long)f5_cdiv(a.seq_len, 128) * a.B * a.H;
    // small batches: fewer workgroups than ~2 per CU -> split the KV range over 2 or 4 wave groups inside the workgroup
    int ks = f5_attn_kvsplit;
    if (ks < 0) {
        const long wgs = wgs128;
        const int ntile = f5_cdiv(a.seq_len, 64);
        // measured (tools/attn_split_bench.py, N = 937, 16 heads): 128 WGs 20.5 / 17.4 / 16.2 us for 1 / 2 / 4 groups,
        // 256 WGs 21.4 / 19.1 / 20.1, 512 WGs 31.8 / 37.3 / 38.9
        ks = (wgs <= 160 && ntile >= 8) ? 4 : ((wgs <= 320 && ntile >= 4) ? 2 : 1);
    }
#if F5_LAB
    // one-pass modes: in-wave software-pipelined kernel (256 queries per workgroup, 8 waves)
    if ((f5_attn_version == 5 || f5_attn_version == 6) && !a.hp && !a.out8 && f5_attn_ablation == 0) {
        // experiment matrix.  waves per workgroup: 8 (256 queries) or 4 (128); packed or single-issue softmax arithmetic;
        // f5_attn_variant bit 0: single-issue VALU, bit 1: 40 KB of unused dynamic LDS so that only ONE 4-wave workgroup fits
        // a CU (one wave per SIMD)
        const bool pk = !(f5_attn_variant & 1);
        const size_t dyn = (f5_attn_variant & 2) ? 40 * 1024 : 0;
        if (f5_attn_version == 5) {
            const dim3 g = attn_grid(a, 256);
            if (pk) hipLaunchKernelGGL((f5_attn5_kernel<true, 8>), g, dim3(512), dyn, stream, a);
            else hipLaunchKernelGGL((f5_attn5_kernel<false, 8>), g, dim3(512), dyn, stream, a);
        } else {
            const dim3 g = attn_grid(a, 128);
            if (pk) hipLaunchKernelGGL((f5_attn5_kernel<true, 4>), g, dim3(256), dyn, stream, a);
            else hipLaunchKernelGGL((f5_attn5_kernel<false, 4>), g, dim3(256), dyn, stream, a);
        }
        F5_LAUNCH_CHECK();
        return 0;
    }
#endif
    // large grids (one-pass modes): two query blocks per wave (256 queries per workgroup), no per-tile maximum
    if (f5_attn_version == 2 && f5_attn_ablation == 0 && !a.hp && ks <= 1 &&
        (f5_attn_wide >= 1 || (f5_attn_wide < 0 && (long)f5_cdiv(a.seq_len, 256) * a.B * a.H >= 512))) {
        const dim3 gw = attn_grid(a, 256);
#if F5_LAB
        if (f5_attn_prio != 0 || (f5_attn_variant & (8 | 16))) {       // A/B: the kernel with a per-tile maximum (v2w) and its priority variants
            if (f5_attn_prio == 1) hipLaunchKernelGGL(f5_attn2w_kernel<1>, gw, dim3(256), 0, stream, a);
            else if (f5_attn_prio == 2) hipLaunchKernelGGL(f5_attn2w_kernel<2>, gw, dim3(256), 0, stream, a);
            else if (f5_attn_variant & 8) hipLaunchKernelGGL((f5_attn2w_kernel<0, false>), gw, dim3(256), 0, stream, a);   // eager rescale
            else hipLaunchKernelGGL(f5_attn2w_kernel<0>, gw, dim3(256), 0, stream, a);
            F5_LAUNCH_CHECK();
            return 0;
        }
#endif
#if F5_LAB
        if (f5_attn_wide == 2 && !a.out8) {              // role-split schedule: 512-query workgroups of 8 waves
            const dim3 gr = attn_grid(a, 512);
            if (!(f5_attn_variant & 32)) {              // variant bit 5: Q fragments in LDS instead of registers
                if (a.q_prescaled) hipLaunchKernelGGL((f5_attn2r_kernel<true, false>), gr, dim3(512), 0, stream, a);
                else hipLaunchKernelGGL((f5_attn2r_kernel<false, false>), gr, dim3(512), 0, stream, a);
            } else {
                if (a.q_prescaled) hipLaunchKernelGGL((f5_attn2r_kernel<true, true>), gr, dim3(512), 0, stream, a);
                else hipLaunchKernelGGL((f5_attn2r_kernel<false, true>), gr, dim3(512), 0, stream, a);
            }
            F5_LAUNCH_CHECK();
            return 0;
        }
#endif
        // f5_attn_pipe: 1 = the in-wave software-pipelined kernel (v2p, one wave per SIMD), 0 = v2f; q must be pre-multiplied
        if ((a.pipe < 0 ? f5_attn_pipe : a.pipe) && a.q_prescaled) hipLaunchKernelGGL(f5_attn2p_kernel, gw, dim3(256), 0, stream, a);
        else if (a.q_prescaled) hipLaunchKernelGGL(f5_attn2f_kernel<true>, gw, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(f5_attn2f_kernel<false>, gw, dim3(256), 0, stream, a);
        F5_LAUNCH_CHECK();
        return 0;
    }
    if (f5_attn_version == 2 && f5_attn_ablation == 0 && ks > 1) {
        if (a.hp) {
            F5_REQUIRE(a.qk[1] && a.vt[1] && a.out[1], "attention: bf16x3 needs lo buffers");
            hipLaunchKernelGGL((f5_attn2s_kernel<true, 2, 2>), grid, dim3(512), 0, stream, a);
#if F5_LAB
        } else if (f5_attn_variant & 16) {               // A/B: tile maximum on every tile
            if (ks >= 4) hipLaunchKernelGGL((f5_attn2s_kernel<false, 4, 2>), grid, dim3(1024), 0, stream, a);
            else hipLaunchKernelGGL((f5_attn2s_kernel<false, 2, 3>), grid, dim3(512), 0, stream, a);
#endif
        } else if (ks >= 4) {
            hipLaunchKernelGGL((f5_attn2s_kernel<false, 4, 2, true>), grid, dim3(1024), 0, stream, a);
        } else {
            hipLaunchKernelGGL((f5_attn2s_kernel<false, 2, 3, true>), grid, dim3(512), 0, stream, a);
        }
        F5_LAUNCH_CHECK();
        return 0;
    }
    if (a.hp) {
        F5_REQUIRE(a.qk[1] && a.vt[1] && a.out[1], "attention: bf16x3 needs lo buffers");
#if F5_LAB
        if (f5_attn_version < 2) {
            hipLaunchKernelGGL((f5_attn_kernel<true>), grid, dim3(256), 0, stream, a);
            F5_LAUNCH_CHECK();
            return 0;
        }
#endif
        hipLaunchKernelGGL((f5_attn2_kernel<true, 0>), grid, dim3(256), 0, stream, a);
    } else {
#if F5_LAB
        if (f5_attn_version == 3) {
            hipLaunchKernelGGL((f5_attn3_kernel<3>), grid, dim3(256), 0, stream, a);
        } else if (f5_attn_version == 4) {
            hipLaunchKernelGGL((f5_attn3_kernel<2>), grid, dim3(256), 0, stream, a);
        } else if (f5_attn_version == 2) {
            switch (f5_attn_ablation) {
                case 1: hipLaunchKernelGGL((f5_attn2_kernel<false, 1>), grid, dim3(256), 0, stream, a); break;
                case 2: hipLaunchKernelGGL((f5_attn2_kernel<false, 2>), grid, dim3(256), 0, stream, a); break;
                case 3: hipLaunchKernelGGL((f5_attn2_kernel<false, 3>), grid, dim3(256), 0, stream, a); break;
                case 4: hipLaunchKernelGGL((f5_attn2_kernel<false, 4>), grid, dim3(256), 0, stream, a); break;
                case 5: hipLaunchKernelGGL((f5_attn2_kernel<false, 5>), grid, dim3(256), 0, stream, a); break;
                case 6: hipLaunchKernelGGL((f5_attn2_kernel<false, 6>), grid, dim3(256), 0, stream, a); break;
                case 7: hipLaunchKernelGGL((f5_attn2_kernel<false, 7>), grid, dim3(256), 0, stream, a); break;
                default: hipLaunchKernelGGL((f5_attn2_kernel<false, 0>), grid, dim3(256), 0, stream, a); break;
            }
        }
        else hipLaunchKernelGGL((f5_attn_kernel<false>), grid, dim3(256), 0, stream, a);
#else
        hipLaunchKernelGGL((f5_attn2_kernel<false, 0>), grid, dim3(256), 0, stream, a);
#endif
    }
    F5_LAUNCH_CHECK();
    return 0;
}
}  // namespace F5_NS
