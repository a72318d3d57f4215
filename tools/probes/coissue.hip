// probe: do MFMA and VALU work of two waves on the SAME SIMD overlap?  One 512-thread workgroup per CU: waves 0-3 run an
// MFMA chain, waves 4-7 (same SIMDs, round-robin placement) run fp32 FMAs / exp2 / nothing.  clock64 per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// mode_lo: work of waves 0-3, mode_hi: work of waves 4-7; 0 = idle (exit), 1 = MFMA, 2 = v_fma_f32, 3 = v_exp_f32 + mul
__global__ __launch_bounds__(512) void k(uint64_t* out, int iters, float seed, int mode_lo, int mode_hi) {
    const int wave = threadIdx.x >> 6;
    const int mode = wave < 4 ? mode_lo : mode_hi;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = seed * (e + i);
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed * e); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = seed + e + threadIdx.x;
    __shared__ unsigned int lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = i * 2654435761u;
    __syncthreads();
    const uint64_t c0 = clock64();
    if (mode == 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    } else if (mode == 4) {          // accumulators pinned to AGPRs
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    } else if (mode == 5) {          // 16x16x32 MFMAs (4 passes)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 c4[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) c4[i][e] = seed * e;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) c4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) v[i] += c4[i][0];
    } else if (mode == 7) {          // scalar ALU stream
        int sacc = __builtin_amdgcn_readfirstlane(iters + 3);
        for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("s_mul_i32 %0, %0, 1664525\n\ts_add_i32 %0, %0, 1013904223" : "+s"(sacc));
        }
        if (sacc == 12345) v[0] += 1.0f;
    } else if (mode == 6) {          // ds_read_b128 stream (conflict-free, 16 B per lane), results folded into v[]
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* src = reinterpret_cast<const u32x4*>(lds) + (threadIdx.x & 63);
        unsigned int fold = 0;
        for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const u32x4 t = src[e * 64 + (it & 1) * 512];
                fold ^= t[0] ^ t[3];
            }
        }
        v[0] += (float)fold;
    } else if (mode == 2) {
        for (int it = 0; it < iters * 6; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * 1.0001f + 0.5f;
    } else if (mode == 3) {
        for (int it = 0; it < iters * 2; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_exp2f(v[e]) * 0.5f;
    }
    const uint64_t c1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int e = 0; e < 8; ++e) s += v[e];
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 8 + wave) * 2] = c1 - c0; out[(blockIdx.x * 8 + wave) * 2 + 1] = (uint64_t)(s != 1.5f); }
}
static void run(const char* name, int mlo, int mhi) {
    const int grid = 256, iters = 20000;
    uint64_t* d; hipMalloc(&d, grid * 8 * 2 * 8);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, d, iters, 0.001f, mlo, mhi); hipDeviceSynchronize(); }
    uint64_t* h = (uint64_t*)malloc(grid * 8 * 2 * 8);
    hipMemcpy(h, d, grid * 8 * 2 * 8, hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += h[(b * 8 + w) * 2];
    printf("%-44s waves 0-3: %9.0f cycles   waves 4-7: %9.0f cycles\n", name, lo / (grid * 4), hi / (grid * 4));
    hipFree(d); free(h);
}
int main() {
    run("MFMA alone", 1, 0);
    run("v_fma alone (on waves 4-7)", 0, 2);
    run("v_exp alone (on waves 4-7)", 0, 3);
    run("MFMA + v_fma on the same SIMDs", 1, 2);
    run("MFMA + v_exp on the same SIMDs", 1, 3);
    run("MFMA(AGPR acc) alone", 4, 0);
    run("MFMA(AGPR acc) + v_fma on the same SIMDs", 4, 2);
    run("MFMA(AGPR acc) + v_exp on the same SIMDs", 4, 3);
    run("MFMA 16x16x32 alone (2x count)", 5, 0);
    run("MFMA 16x16x32 + v_fma", 5, 2);
    run("SALU alone (on waves 4-7)", 0, 7);
    run("MFMA + SALU on the same SIMDs", 1, 7);
    run("ds_read_b128 alone (on waves 4-7)", 0, 6);
    run("MFMA + ds_read_b128 on the same SIMDs", 1, 6);
    run("MFMA + MFMA", 1, 1);
    run("v_fma + v_fma", 2, 2);
    return 0;
}
