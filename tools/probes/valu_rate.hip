// probe (round 5): issue cost of one wave64 VALU instruction on gfx950, plain vs packed f32 vs transcendental, with 1 / 2 waves per SIMD
// -- the GEMM epilogues are VALU-issue-bound (profiles/r05), so "cycles per instruction" and "does v_pk_fma_f32 do two elements for the
// price of one" decide what an epilogue can cost.  s_memtime around N dependent-free instructions (8 independent chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(uint64_t* out, float seed, int iters) {
    float a[8];
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = f32x2{a[i], a[i] * 0.5f}; }
    const float m = 1.0001f, c = 0.5f;
    const f32x2 m2 = {m, m}, c2 = {c, c};
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
                if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
                if (MODE == 4) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if (MODE == 5) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (MODE == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 7) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1];
    if (s == 123.456f) out[1000] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE>
void run(const char* name, uint64_t* d, int threads) {
    const int iters = 2000;
    uint64_t h[8];
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, 1.0f, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, d, 1.0f, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"probe\": \"valu_rate\", \"instr\": \"%s\", \"waves_per_simd\": %d, \"memtime_ticks_per_instr_wave0\": %.2f}\n", name, threads / 256,
           (double)h[0] / (iters * 32.0));
}
int main() {
    uint64_t* d;
    hipMalloc(&d, 8192 * 8);
    for (int threads : {256, 512}) {
        run<0>("v_fma_f32", d, threads);
        run<1>("v_pk_fma_f32", d, threads);
        run<3>("v_pk_mul_f32", d, threads);
        run<4>("v_med3_f32", d, threads);
        run<5>("v_cvt_pk_f16_f32", d, threads);
        run<2>("v_exp_f32", d, threads);
        run<6>("v_rcp_f32", d, threads);
        run<7>("v_pk_fma_f16", d, threads);
    }
    // s_memtime runs at a constant 100 MHz on gfx950; print the shader clock too
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("{\"probe\": \"valu_rate\", \"device_clock_khz\": %d, \"note\": \"s_memtime ticks: compare instructions with each other; v_fma_f32 is the unit\"}\n", clk);
    return 0;
}
