// probe (round 4, second): the half-step of attention v2p as a pure register loop (no LDS, no barriers, no memory), with its real data
// flow: exp of S_cur (MFMA results of the previous half-step) -> P_cur words; 8 "QK" MFMAs into S_nxt (VGPR accumulators, two chains,
// an MFMA and the next one on its accumulator two MFMAs apart) then 8 "PV" MFMAs into O (AGPR accumulators, four chains) with
// B = P_prev (VALU results of the previous half-step).  Variants switch one dependency off at a time:
//   V = 0 replica   1 exps read loop-invariant registers (not MFMA results)   2 PV B operand loop invariant (not the P words)
//   3 = 1 + 2       4 QK MFMAs on four accumulators (chain distance 4)        5 QK MFMAs with AGPR accumulators (exps read constants)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/inwave_dataflow.hip -o tools/probes/bin/inwave_dataflow
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define MFMA "v_mfma_f32_32x32x16_f16"

template <int V, int L>
__device__ __forceinline__ void half_step(f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2], f32x16 (&s_x)[2], f32x16 (&o)[2][2], const uint32_t (&p_prev)[2][8],
                                          uint32_t (&p_cur)[2][8], h8 (&fr)[8], h8 (&fn)[8], uint32_t laddr, const h8 (&qf)[2][4], const f32x16 (&mneg)[2], const h8& bconst,
                                          float (&cst)[4], float& acc_sum) {
    float sa = 0.f, sb = 0.f, y0 = 0.f, y1 = 0.f;
#pragma unroll
    for (int h = 0; h < 16; ++h) {
        const int f = h >> 1, qb = h & 1, pq = h >> 3, pi = h & 7;
        float x0, x1;
        if (L >= 1 && h < 8) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(fn[h]) : "v"(laddr), "n"(4096 * (h < 8 ? h : 0)));
        if (V == 1 || V == 3 || V == 5) {
            asm volatile("v_exp_f32 %0, %1" : "=v"(x0) : "v"(cst[0]));
            asm volatile("v_exp_f32 %0, %1" : "=v"(x1) : "v"(cst[1]));
        } else {
            asm volatile("v_exp_f32 %0, %1" : "=v"(x0) : "v"(s_cur[pq][2 * pi]));
            asm volatile("v_exp_f32 %0, %1" : "=v"(x1) : "v"(s_cur[pq][2 * pi + 1]));
        }
        if (h > 0) {
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(sa) : "v"(y0));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(sb) : "v"(y1));
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p_cur[(h - 1) >> 3][(h - 1) & 7]) : "v"(y0), "v"(y1));
        }
        y0 = x0; y1 = x1;
        if (f < 4) {
            if (V == 5) {
                asm volatile("s_nop 1\n\t" MFMA " %0, %1, %2, %0" : "+a"(o[qb][f & 1]) : "a"(fr[f]), "a"(qf[qb][f]));
            } else if (V == 4) {
                f32x16& d = (f & 1) ? s_x[qb] : s_nxt[qb];
                if (f < 2) asm volatile("s_nop 1\n\t" MFMA " %0, %1, %2, %3" : "=&v"(d) : "a"(fr[f]), "a"(qf[qb][f]), "v"(mneg[qb]));
                else asm volatile("s_nop 1\n\t" MFMA " %0, %1, %2, %0" : "+v"(d) : "a"(fr[f]), "a"(qf[qb][f]));
            } else {
                if (f == 0) asm volatile("s_nop 1\n\t" MFMA " %0, %1, %2, %3" : "=&v"(s_nxt[qb]) : "a"(fr[f]), "a"(qf[qb][f]), "v"(mneg[qb]));
                else asm volatile("s_nop 1\n\t" MFMA " %0, %1, %2, %0" : "+v"(s_nxt[qb]) : "a"(fr[f]), "a"(qf[qb][f]));
            }
        } else {
            const int sp = ((f - 4) >> 1) & 1, db = (f - 4) & 1;
            if (V == 2 || V == 3) {
                asm volatile("s_nop 1\n\t" MFMA " %0, %1, %2, %0" : "+a"(o[qb][db]) : "a"(fr[f]), "v"(bconst));
            } else {
                const h8 pb = __builtin_bit_cast(h8, u32x4{p_prev[qb][4 * sp], p_prev[qb][4 * sp + 1], p_prev[qb][4 * sp + 2], p_prev[qb][4 * sp + 3]});
                asm volatile("s_nop 1\n\t" MFMA " %0, %1, %2, %0" : "+a"(o[qb][db]) : "a"(fr[f]), "v"(pb));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p_cur[1][7]) : "v"(y0), "v"(y1));
    if (L >= 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+a"(fn[0]), "+a"(fn[1]), "+a"(fn[2]), "+a"(fn[3]), "+a"(fn[4]), "+a"(fn[5]), "+a"(fn[6]), "+a"(fn[7]));
    acc_sum += sa + sb;
}

template <int V, int L>
__global__ __launch_bounds__(256, 1) void k(uint64_t* out, int iters, float seed) {
    __shared__ float pad[24 * 1024];
    if (seed == 123.0f) pad[threadIdx.x] = seed;
    f32x16 s_a[2], s_b[2], s_x[2], o[2][2], mneg[2];
    uint32_t p_a[2][8], p_b[2][8];
    h8 fr[8], fn[8], qf[2][4], bconst;
    const uint32_t laddr = (threadIdx.x & 63) * 16;
    for (int i = threadIdx.x; i < 24 * 1024; i += 256) pad[i] = seed * 1e-3f * (i & 255);
    float cst[4] = {seed, seed * 2, seed * 3, seed * 4}, acc_sum = 0.f;
    for (int q = 0; q < 2; ++q) {
        for (int e = 0; e < 16; ++e) { s_a[q][e] = seed * e; s_b[q][e] = seed * (e + 1); s_x[q][e] = 0.f; mneg[q][e] = -seed; o[q][0][e] = 0.f; o[q][1][e] = 0.f; }
        for (int w = 0; w < 8; ++w) { p_a[q][w] = 0x3c003c00u; p_b[q][w] = 0x3c003c00u; }
        for (int ks = 0; ks < 4; ++ks) for (int e = 0; e < 8; ++e) qf[q][ks][e] = (_Float16)(seed * (e + ks));
    }
    for (int f = 0; f < 8; ++f) for (int e = 0; e < 8; ++e) { fr[f][e] = (_Float16)(seed * (f + e) * 0.01f); fn[f][e] = fr[f][e]; }
    for (int e = 0; e < 8; ++e) bconst[e] = (_Float16)(seed * e);
    for (int f = 0; f < 8; ++f) asm volatile("" : "+a"(fr[f]), "+a"(fn[f]));
    for (int q = 0; q < 2; ++q) { for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+a"(qf[q][ks])); asm volatile("" : "+a"(o[q][0]), "+a"(o[q][1])); asm volatile("" : "+v"(mneg[q]), "+v"(s_a[q]), "+v"(s_b[q]), "+v"(s_x[q])); }
    asm volatile("" : "+v"(bconst));
    __syncthreads();
    const uint64_t c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (L == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        half_step<V, L>(s_a, s_b, s_x, o, p_b, p_a, fr, fn, laddr, qf, mneg, bconst, cst, acc_sum);
        half_step<V, L>(s_b, s_a, s_x, o, p_a, p_b, fn, fr, laddr, qf, mneg, bconst, cst, acc_sum);
    }
    const uint64_t c1 = clock64();
    float s = acc_sum;
    for (int q = 0; q < 2; ++q) { s += s_a[q][3] + s_b[q][5] + s_x[q][1] + o[q][0][2] + o[q][1][7] + (float)p_a[q][3] + (float)p_b[q][1] + (float)fn[q][1] + (float)fr[q][2]; }
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = c1 - c0; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = (uint64_t)(s != 1.5f); }
}
template <int V, int L = 0>
static void run(const char* name) {
    const int grid = 256, iters = 4000;
    uint64_t* d; (void)hipMalloc(&d, grid * 4 * 2 * 8);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<V, L>), dim3(grid), dim3(256), 0, 0, d, iters, 0.001f); (void)hipDeviceSynchronize(); }
    uint64_t* h = (uint64_t*)malloc(grid * 4 * 2 * 8);
    (void)hipMemcpy(h, d, grid * 4 * 2 * 8, hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < grid * 4; ++i) c += h[i * 2];
    printf("%-92s %7.2f ticks per half-slot (MFMA)\n", name, c / (grid * 4) / (iters * 32.0));
    (void)hipFree(d); free(h);
}
int main() {
    run<0>("V0 replica of the v2p half-step data flow");
    run<1>("V1 exps read loop-invariant registers instead of the MFMA results");
    run<2>("V2 PV B operand loop invariant instead of the P words");
    run<3>("V3 = V1 + V2");
    run<4>("V4 QK MFMAs on four VGPR accumulators (chain distance 4)");
    run<5>("V5 QK MFMAs with AGPR accumulators, exps read constants");
    run<0, 1>("V0 + 8 asm ds_read_b128 -> AGPR per half-step (next set), lgkmcnt(0) at its end");
    run<0, 2>("V0 + the LDS reads + s_barrier per tile (4 waves)");
    return 0;
}
