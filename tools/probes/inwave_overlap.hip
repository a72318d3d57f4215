// probe (round 4): does VALU work placed BETWEEN the MFMAs of one wave's instruction stream overlap the matrix pipe on gfx950?
// One wave per SIMD (256-thread blocks, one per CU: 96 KB of LDS each).  Every variant runs `iters` slots; a slot is one
// v_mfma_f32_32x32x16_f16 and / or NV VALU instructions; cycles per slot from s_memtime (clock64), averaged over all waves.
//   mode 0  MFMA only, accumulators in AGPRs          mode 1  VALU only: 5 v_fma_f32
//   mode 2  MFMA (AGPR acc) + 5 v_fma_f32             mode 3  MFMA (VGPR acc) + 5 v_fma_f32
//   mode 4  MFMA (AGPR acc) + 2 v_exp_f32 + 3 v_add   mode 5  VALU only: 2 v_exp_f32 + 3 v_add
//   mode 6  MFMA (VGPR acc) + 2 v_exp_f32 + 3 v_add   mode 7  MFMA (AGPR acc, B operand in VGPRs) + 2 v_exp + 3 v_add
//   mode 8  MFMA (AGPR acc) + 3 v_fma                 mode 9  MFMA (AGPR acc) + 8 v_fma
//   modes 10-13  the half-slot of the attention kernel v2p (see the code)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/inwave_overlap.hip -o tools/probes/bin/inwave_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA "v_mfma_f32_32x32x16_f16"

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(uint64_t* out, int iters, float seed) {
    __shared__ float pad[24 * 1024];
    if (seed == 123.0f) pad[threadIdx.x] = seed;      // keep the LDS allocation (one block per CU)
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = seed * (e + i);
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed + e); b[e] = (_Float16)(seed * e); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = seed * 0.001f + 0.0001f * e + 1e-6f * threadIdx.x;
    if (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 8 || MODE == 9) { for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(acc[i])); asm volatile("" : "+a"(a), "+a"(b)); }
    if (MODE == 7) { for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(acc[i])); asm volatile("" : "+a"(a)); asm volatile("" : "+v"(b)); }
    if (MODE == 3 || MODE == 6) { for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(acc[i])); asm volatile("" : "+a"(a), "+a"(b)); }
    uint32_t pw[4] = {0, 0, 0, 0};
    h8 la = a, bv = b;
    f32x16 accv[2];
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) accv[i][e] = seed * e;
    _Float16* lds16 = reinterpret_cast<_Float16*>(pad);
    if (MODE >= 10) {
        for (int i = threadIdx.x; i < 8 * 512; i += 256) lds16[i] = (_Float16)(seed * i);
        for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(acc[i]));
        asm volatile("" : "+a"(la), "+a"(b));
        asm volatile("" : "+v"(bv), "+v"(accv[0]), "+v"(accv[1]));
    }
    __syncthreads();
    const uint64_t c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 8 || MODE == 9) asm volatile(MFMA " %0, %1, %2, %0" : "+a"(acc[i]) : "a"(a), "a"(b));
            if (MODE == 7) asm volatile(MFMA " %0, %1, %2, %0" : "+a"(acc[i]) : "a"(a), "v"(b));
            if (MODE == 3 || MODE == 6) asm volatile(MFMA " %0, %1, %2, %0" : "+v"(acc[i]) : "a"(a), "a"(b));
            if (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 8 || MODE == 9) {
                constexpr int NV = MODE == 8 ? 3 : (MODE == 9 ? 8 : 5);
#pragma unroll
                for (int e = 0; e < NV; ++e) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[e]) : "v"(1.0001f), "v"(0.5f));
            }
            if (MODE == 4 || MODE == 5 || MODE == 6 || MODE == 7) {
                asm volatile("v_exp_f32 %0, %1" : "=v"(v[4 + (i & 1) * 2]) : "v"(v[0]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(v[5 + (i & 1) * 2]) : "v"(v[1]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[0]) : "v"(v[2]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[1]) : "v"(v[3]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[2]) : "v"(v[3]));
            }
            if (MODE >= 10 && MODE <= 13) {
                // the half-slot of attention v2p: 2 exp, 2 add, cvt_pk, (s_nop 1), MFMA; A operand from LDS every second slot (11, 13)
                float x0, x1;
                asm volatile("v_exp_f32 %0, %1" : "=v"(x0) : "v"(v[0]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(x1) : "v"(v[1]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[2]) : "v"(v[4]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[3]) : "v"(v[5]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pw[i]) : "v"(v[4]), "v"(v[5]));
                v[4] = x0; v[5] = x1;
                if ((MODE == 11 || MODE == 13) && (i & 1) == 0) { la = *reinterpret_cast<const h8*>(&lds16[((it * 2 + (i >> 1)) & 7) * 512 + (threadIdx.x & 63) * 8]); }
                if (MODE == 12 || MODE == 13) {
                    if (i < 2) asm volatile("s_nop 1\n\t" MFMA " %0, %1, %2, %0" : "+v"(accv[i]) : "a"(la), "a"(b));
                    else asm volatile("s_nop 1\n\t" MFMA " %0, %1, %2, %0" : "+a"(acc[i]) : "a"(la), "v"(bv));
                } else {
                    asm volatile("s_nop 1\n\t" MFMA " %0, %1, %2, %0" : "+a"(acc[i]) : "a"(la), "a"(b));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint64_t c1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int e = 0; e < 8; ++e) s += v[e];
    for (int i = 0; i < 4; ++i) s += (float)pw[i];
    for (int i = 0; i < 2; ++i) s += accv[i][3];
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = c1 - c0; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = (uint64_t)(s != 1.5f); }
}
template <int MODE>
static void run(const char* name) {
    const int grid = 256, iters = 20000;
    uint64_t* d; hipMalloc(&d, grid * 4 * 2 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int r = 0; r < 2; ++r) { hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 0.001f); hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1); }
    uint64_t* h = (uint64_t*)malloc(grid * 4 * 2 * 8);
    hipMemcpy(h, d, grid * 4 * 2 * 8, hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < grid * 4; ++i) c += h[i * 2];
    const double cyc = c / (grid * 4) / (iters * 4.0);
    printf("%-62s %7.2f s_memtime ticks per slot   kernel %.3f ms = %.2f ns per slot\n", name, cyc, ms, ms * 1e6 / (iters * 4.0));
    hipFree(d); free(h);
}
int main() {
    run<0>("mode 0  MFMA only (AGPR acc)");
    run<1>("mode 1  5 v_fma only");
    run<2>("mode 2  MFMA (AGPR acc) + 5 v_fma");
    run<3>("mode 3  MFMA (VGPR acc) + 5 v_fma");
    run<5>("mode 5  2 v_exp + 3 v_add only");
    run<4>("mode 4  MFMA (AGPR acc) + 2 v_exp + 3 v_add");
    run<6>("mode 6  MFMA (VGPR acc) + 2 v_exp + 3 v_add");
    run<7>("mode 7  MFMA (AGPR acc, B in VGPRs) + 2 v_exp + 3 v_add");
    run<8>("mode 8  MFMA (AGPR acc) + 3 v_fma");
    run<9>("mode 9  MFMA (AGPR acc) + 8 v_fma");
    run<10>("mode 10 v2p half-slot: 2 exp 2 add cvt_pk s_nop MFMA(AGPR acc)");
    run<11>("mode 11 = 10 + A operand from LDS (ds_read_b128 -> AGPR) per 2 slots");
    run<12>("mode 12 = 10, two MFMAs VGPR acc / two AGPR acc with B in VGPRs");
    run<13>("mode 13 = 12 + A operand from LDS");
    return 0;
}
