// probe: shader clock under load.  clock64() (s_memtime, shader clock) vs wall_clock64() (100 MHz) over a fixed amount of
// work on every CU: bf16 MFMA chain / fp32 VALU chain / v_exp_f32 chain / idle-ish (1 workgroup).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t* out, int iters, float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = seed * (e + i);
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed * e); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = seed + e + threadIdx.x;
    const uint64_t c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * 1.0001f + 0.5f;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_exp2f(v[e]) * 0.5f;
        }
    }
    const uint64_t c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int e = 0; e < 8; ++e) s += v[e];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 3 + 0] = c1 - c0;
        out[blockIdx.x * 3 + 1] = w1 - w0;
        out[blockIdx.x * 3 + 2] = (uint64_t)(s != 12345.0f);
    }
}
template <int MODE>
static void run(const char* name, int grid, int iters, double ops_per_iter_per_wave) {
    uint64_t* d;
    hipMalloc(&d, grid * 3 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, d, iters, 0.001f);
        hipDeviceSynchronize();
    }
    uint64_t* h = (uint64_t*)malloc(grid * 3 * 8);
    hipMemcpy(h, d, grid * 3 * 8, hipMemcpyDeviceToHost);
    double c = 0, w = 0;
    for (int i = 0; i < grid; ++i) { c += h[i * 3]; w += h[i * 3 + 1]; }
    c /= grid; w /= grid;
    const double us = w / 100.0;   // 100 MHz
    printf("%-28s grid=%4d  shader cycles=%.0f  wall=%.1f us  => %.0f MHz;  cycles per op per wave = %.2f\n", name, grid, c, us,
           c / us, c / (iters * ops_per_iter_per_wave));
    hipFree(d); free(h);
}
int main() {
    run<0>("mfma 32x32x16 bf16, 1 WG", 1, 20000, 4);
    run<0>("mfma, 256 WG (1/CU)", 256, 20000, 4);
    run<0>("mfma, 1024 WG (4/CU)", 1024, 20000, 4);
    run<0>("mfma, 256 WG, 60 ms", 256, 1000000, 4);
    run<0>("mfma, 256 WG, 60 ms again", 256, 1000000, 4);
    run<0>("mfma, 1 WG after load", 1, 20000, 4);
    run<1>("v_fma_f32, 1 WG", 1, 20000, 8);
    run<1>("v_fma_f32, 1024 WG", 1024, 20000, 8);
    run<2>("v_exp_f32+mul, 1 WG", 1, 20000, 16);
    run<2>("v_exp_f32+mul, 1024 WG", 1024, 20000, 16);
    return 0;
}
