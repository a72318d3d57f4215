// Hazard (a), which packed instruction goes wrong next to another wave's MFMAs (see pk_f32_vs_mfma.hip for the setting).
// Waves 0-3 of every workgroup loop over register-resident data (no memory traffic inside the loop) and compare, in the kernel,
// ONE packed-f32 instruction form per kernel against the same arithmetic done with plain VALU instructions (which never failed);
// waves 4-7 (their SIMD partners) stream v_mfma_f32_32x32x16_f16, or idle.  s = (s0, s1), a = (a0, a1), b = (b0, b1):
//   0  v_pk_mul_f32 d, s, a                                  lo = s0 a0   hi = s1 a1      plain selects
//   1  v_pk_mul_f32 d, s, a op_sel:[0,1] op_sel_hi:[0,0]     lo = s0 a1   hi = s0 a0      the compiler's rotation form
//   2  v_pk_mul_f32 d, s, a op_sel:[0,1]                     lo = s0 a1   hi = s1 a1      lo result reads a HI register
//   3  v_pk_mul_f32 d, s, a op_sel_hi:[0,0]                  lo = s0 a0   hi = s0 a0      hi result reads LO registers
//   4  v_pk_mul_f32 d, s, a op_sel:[1,0]                     lo = s1 a0   hi = s1 a1
//   5  v_pk_mul_f32 d, s, a op_sel_hi:[1,0]                  lo = s0 a0   hi = s1 a0      src1.lo broadcast (attention softmax form)
//   6  v_pk_add_f32 d, s, a op_sel:[0,1]                     lo = s0 + a1 hi = s1 + a1
//   7  v_pk_fma_f32 d, s, a, b op_sel:[0,0,1]                lo = s0 a0 + b1
//   8  v_pk_fma_f32 d, s, a, b op_sel_hi:[0,1,1]             hi = s0 a1 + b1              src0.lo broadcast (rotation form)
//   9  v_pk_fma_f32 d, s, a, b                               plain selects
//  10  v_mov_b64 d, s                                        (the attention kernel copies accumulator pairs with it)
//  11  v_pk_add_f32 d, s, a op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]   lo = s0 - a0, hi = s1 - a0   (the attention softmax form, literally)
//  12  v_pk_mul_f32 d, s, a op_sel:[1,1]                     lo = s1 a1   both sources hi -> lo
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/pk_f32_vs_mfma2.hip -o tools/probes/bin/pk_f32_vs_mfma2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float umul(float x, float y) {
    float r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ float uadd(float x, float y) {
    float r;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ float ufma(float x, float y, float z) {
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}

__device__ unsigned rec_n;
__device__ float rec[8];

template <int TEST, int MFMA>
__global__ __launch_bounds__(512) void k_mix(unsigned long long* __restrict__ cnt, float* __restrict__ sink, int iters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) {
        if (MFMA) {
            f32x16 acc0 = {0}, acc1 = {0};
            h16x8 x, y;
            for (int e = 0; e < 8; ++e) {
                x[e] = (_Float16)(0.01f * (lane + e));
                y[e] = (_Float16)(0.02f * (lane - e));
            }
            for (int it = 0; it < iters; ++it) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, acc1, 0, 0, 0);
            }
            sink[(size_t)blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc1[3];
        }
        return;
    }
    unsigned bad0 = 0, bad1 = 0;
    float seed = 0.37f * (float)(lane + 1) + 0.011f * (float)(blockIdx.x & 1023);
    for (int it = 0; it < iters; ++it) {
        seed = ufma(seed, 1.0009765625f, 0.123f);
        if (seed > 64.0f) seed = umul(seed, 0.015625f);
        f32x2 a = {uadd(seed, -1.5f), umul(seed, 0.75f)}, b = {umul(seed, 0.01f), uadd(seed, -0.3f)}, s = {umul(seed, -0.31f), umul(seed, 0.19f)};
        asm volatile("" : "+v"(a), "+v"(b), "+v"(s));
        f32x2 d;
        float w0, w1;
        if (TEST == 0) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(d) : "v"(s), "v"(a)); w0 = umul(s[0], a[0]); w1 = umul(s[1], a[1]); }
        if (TEST == 1) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=&v"(d) : "v"(s), "v"(a)); w0 = umul(s[0], a[1]); w1 = umul(s[0], a[0]); }
        if (TEST == 2) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(d) : "v"(s), "v"(a)); w0 = umul(s[0], a[1]); w1 = umul(s[1], a[1]); }
        if (TEST == 3) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,0]" : "=&v"(d) : "v"(s), "v"(a)); w0 = umul(s[0], a[0]); w1 = umul(s[0], a[0]); }
        if (TEST == 4) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=&v"(d) : "v"(s), "v"(a)); w0 = umul(s[1], a[0]); w1 = umul(s[1], a[1]); }
        if (TEST == 5) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(d) : "v"(s), "v"(a)); w0 = umul(s[0], a[0]); w1 = umul(s[1], a[0]); }
        if (TEST == 6) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(d) : "v"(s), "v"(a)); w0 = uadd(s[0], a[1]); w1 = uadd(s[1], a[1]); }
        if (TEST == 7) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=&v"(d) : "v"(s), "v"(a), "v"(b)); w0 = ufma(s[0], a[0], b[1]); w1 = ufma(s[1], a[1], b[1]); }
        if (TEST == 8) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=&v"(d) : "v"(s), "v"(a), "v"(b)); w0 = ufma(s[0], a[0], b[0]); w1 = ufma(s[0], a[1], b[1]); }
        if (TEST == 9) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(s), "v"(a), "v"(b)); w0 = ufma(s[0], a[0], b[0]); w1 = ufma(s[1], a[1], b[1]); }
        if (TEST == 10) { asm volatile("v_mov_b64 %0, %1" : "=&v"(d) : "v"(s)); w0 = s[0]; w1 = s[1]; }
        if (TEST == 11) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(d) : "v"(s), "v"(a)); w0 = uadd(s[0], -a[0]); w1 = uadd(s[1], -a[0]); }
        if (TEST == 12) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1]" : "=&v"(d) : "v"(s), "v"(a)); w0 = umul(s[1], a[1]); w1 = umul(s[1], a[1]); }
        if (d[0] != w0) {
            if (atomicAdd(&rec_n, 1u) == 0) { rec[0] = d[0]; rec[1] = s[0]; rec[2] = s[1]; rec[3] = a[0]; rec[4] = a[1]; rec[5] = (float)lane; rec[6] = d[1]; rec[7] = w0; }
            ++bad0;
        }
        bad1 += d[1] != w1;
    }
    const int q = lane >> 4;
    if (bad0) atomicAdd(&cnt[q], (unsigned long long)bad0);
    if (bad1) atomicAdd(&cnt[4 + q], (unsigned long long)bad1);
}

static const char* names[13] = {"pk_mul plain", "pk_mul op_sel:[0,1] op_sel_hi:[0,0]", "pk_mul op_sel:[0,1]", "pk_mul op_sel_hi:[0,0]", "pk_mul op_sel:[1,0]",
                                "pk_mul op_sel_hi:[1,0]", "pk_add op_sel:[0,1]", "pk_fma op_sel:[0,0,1]", "pk_fma op_sel_hi:[0,1,1]", "pk_fma plain", "v_mov_b64", "pk_add op_sel_hi:[1,0] neg", "pk_mul op_sel:[1,1]"};

template <int TEST>
static void run(unsigned long long* dcnt, float* dsink, int blocks, int iters) {
    for (int mfma = 0; mfma < 2; ++mfma) {
        unsigned z = 0;
        CK(hipMemset(dcnt, 0, 8 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(rec_n), &z, 4));
        for (int rep = 0; rep < 5; ++rep) {
            if (mfma) hipLaunchKernelGGL((k_mix<TEST, 1>), dim3(blocks), dim3(512), 0, 0, dcnt, dsink, iters);
            else hipLaunchKernelGGL((k_mix<TEST, 0>), dim3(blocks), dim3(512), 0, 0, dcnt, dsink, iters);
            CK(hipDeviceSynchronize());
        }
        unsigned long long h[8];
        CK(hipMemcpy(h, dcnt, 64, hipMemcpyDeviceToHost));
        printf("%-36s partners %-11s lo wrong by lane quarter %llu %llu %llu %llu   hi wrong %llu %llu %llu %llu\n", names[TEST], mfma ? "stream MFMA" : "idle", h[0], h[1],
               h[2], h[3], h[4], h[5], h[6], h[7]);
        unsigned rn;
        float r[8];
        CK(hipMemcpyFromSymbol(&rn, HIP_SYMBOL(rec_n), 4));
        CK(hipMemcpyFromSymbol(r, HIP_SYMBOL(rec), 32));
        if (rn) printf("      first wrong lo (lane %d): got %.9g, want %.9g; s0 %.9g s1 %.9g a0 %.9g a1 %.9g  (s0*a0 %.9g, s0*a1 %.9g, s1*a0 %.9g, s1*a1 %.9g); hi got %.9g\n", (int)r[5],
                       r[0], r[7], r[1], r[2], r[3], r[4], r[1] * r[3], r[1] * r[4], r[2] * r[3], r[2] * r[4], r[6]);
    }
}

int main() {
    unsigned long long* dcnt;
    float* dsink;
    const int blocks = 4096, iters = 8192;
    CK(hipMalloc(&dcnt, 64));
    CK(hipMalloc(&dsink, (size_t)blocks * 512 * 4));
    printf("%.2e trials per line\n", 5.0 * blocks * 256 * iters);
    run<0>(dcnt, dsink, blocks, iters);
    run<1>(dcnt, dsink, blocks, iters);
    run<2>(dcnt, dsink, blocks, iters);
    run<3>(dcnt, dsink, blocks, iters);
    run<4>(dcnt, dsink, blocks, iters);
    run<5>(dcnt, dsink, blocks, iters);
    run<6>(dcnt, dsink, blocks, iters);
    run<7>(dcnt, dsink, blocks, iters);
    run<8>(dcnt, dsink, blocks, iters);
    run<9>(dcnt, dsink, blocks, iters);
    run<10>(dcnt, dsink, blocks, iters);
    run<11>(dcnt, dsink, blocks, iters);
    run<12>(dcnt, dsink, blocks, iters);
    return 0;
}
