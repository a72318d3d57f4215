// probe: issue rate of v_mfma_f32_32x32x16_bf16 / v_mfma_scale_f32_32x32x64_f8f6f4 as a function of the number of independent
// accumulator chains (1, 2, 4) in a back-to-back stream, one wave per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int NACC, int F8>
__global__ __launch_bounds__(256) void k(uint64_t* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = seed * (e + i);
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed * e); }
    i32x8 a8, b8;
    for (int e = 0; e < 8; ++e) { a8[e] = 0x38383838 + e; b8[e] = 0x30303030; }
    const uint64_t c0 = clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 4 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (F8) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i], 0, 0, 0, 127, 0, 127);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            }
    const uint64_t c1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if ((threadIdx.x & 63) == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = (uint64_t)(s != 1.5f); }
}
template <int NACC, int F8>
static void run(const char* name) {
    const int grid = 256, iters = 20000;
    uint64_t* d; hipMalloc(&d, grid * 16);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<NACC, F8>), dim3(grid), dim3(256), 0, 0, d, iters, 0.001f); hipDeviceSynchronize(); }
    uint64_t h[512]; hipMemcpy(h, d, grid * 16, hipMemcpyDeviceToHost);
    double c = 0; for (int b = 0; b < grid; ++b) c += h[b * 2];
    printf("%-40s %6.1f cycles per MFMA\n", name, c / grid / (iters * 4.0));
    hipFree(d);
}
int main() {
    run<1, 0>("bf16 32x32x16, 1 accumulator chain");
    run<2, 0>("bf16 32x32x16, 2 chains alternating");
    run<4, 0>("bf16 32x32x16, 4 chains");
    run<1, 1>("mx-fp8 32x32x64, 1 chain");
    run<2, 1>("mx-fp8 32x32x64, 2 chains alternating");
    run<4, 1>("mx-fp8 32x32x64, 4 chains");
    return 0;
}
