// Hazard (a), the trigger: packed-f32 VALU instructions of one wave next to MFMAs of ANOTHER wave on the same SIMD.
// The failing kernels of the round-2 experiment were the 4-wave GEMMs: several workgroups share a CU, so the LDS-free epilogue
// of one workgroup (SLP-packed v_pk_add / v_pk_mul / v_pk_fma_f32) ran while another workgroup on the same SIMDs was in its MFMA
// loop.  The 8-wave kernels (one workgroup per CU, all waves in the same phase) never failed.  The isolated packed group alone
// never fails either (pk_f32_hazard.hip).  Here: 8 waves per workgroup; waves 0-3 (one per SIMD) run the packed rotation group on
// data from memory and store the results, waves 4-7 (their SIMD partners) either idle (mode 0), stream independent
// v_mfma_f32_32x32x16_f16 (mode 1) or stream plain v_fma_f32 (mode 2).  `packed` = 0 runs the same arithmetic un-packed.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/pk_f32_vs_mfma.hip -o tools/probes/bin/pk_f32_vs_mfma
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

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

template <int MODE, int PACKED>
__global__ __launch_bounds__(512) void k_mix(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ cs,
                                             float* __restrict__ out, float* __restrict__ sink, int per_wave) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) {                                   // SIMD partners of waves 0-3
        if (MODE == 1) {
            f32x16 acc0 = {0}, acc1 = {0};
            h16x8 x, y;
            for (int e = 0; e < 8; ++e) {
                x[e] = (_Float16)(0.01f * (lane + e));
                y[e] = (_Float16)(0.02f * (lane - e));
            }
            for (int it = 0; it < per_wave * 6; ++it) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, acc1, 0, 0, 0);
            }
            sink[(size_t)blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc1[3];
        } else if (MODE == 2) {
            float f0 = 0.1f * lane, f1 = 0.2f, f2 = 0.3f, f3 = 0.4f;
            for (int it = 0; it < per_wave * 40; ++it) {
                f0 = fmaf(f0, 0.999f, f1);
                f1 = fmaf(f1, 0.998f, f2);
                f2 = fmaf(f2, 0.997f, f3);
                f3 = fmaf(f3, 0.996f, f0);
            }
            sink[(size_t)blockIdx.x * 512 + threadIdx.x] = f0 + f1 + f2 + f3;
        }
        return;
    }
    for (int it = 0; it < per_wave; ++it) {
        const size_t i = (((size_t)blockIdx.x * 4 + wave) * per_wave + it) * 64 + lane;
        f32x2 ab = *reinterpret_cast<const f32x2*>(a + 2 * i), bb = *reinterpret_cast<const f32x2*>(b + 2 * i);
        const float c0 = cs[2 * i], s0 = cs[2 * i + 1];
        float o0, o1;
        if (PACKED) {
            f32x2 sp = {s0, __int_as_float(0x5000)}, cp = {c0, __int_as_float(0x6000)}, r;
            asm volatile(
                "v_pk_add_f32 %1, %1, %4\n\t"
                "v_pk_mul_f32 %2, %2, %1 op_sel:[0,1] op_sel_hi:[0,0]\n\t"
                "v_pk_fma_f32 %0, %3, %1, %2 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                "v_pk_fma_f32 %3, %3, %1, %2 op_sel_hi:[0,1,1]"
                : "=&v"(r), "+v"(ab), "+v"(sp), "+v"(cp)
                : "v"(bb));
            o0 = r[0];
            o1 = cp[1];
        } else {
            float a0, a1, t0, t1;
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(a0) : "v"(ab[0]), "v"(bb[0]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(a1) : "v"(ab[1]), "v"(bb[1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(s0), "v"(a1));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(s0), "v"(a0));
            asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(o0) : "v"(c0), "v"(a0), "v"(t0));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(o1) : "v"(c0), "v"(a1), "v"(t1));
        }
        out[2 * i] = o0;
        out[2 * i + 1] = o1;
    }
}

template <int MODE, int PACKED>
static void run(const float* da, const float* db, const float* dcs, float* dout, float* dsink, const std::vector<float>& ha, const std::vector<float>& hb,
                const std::vector<float>& hcs, int blocks, int per_wave) {
    const size_t n = (size_t)blocks * 4 * per_wave * 64;
    std::vector<float> ho(2 * n);
    long bad_lo = 0, bad_hi = 0, q[4] = {0, 0, 0, 0};
    int launches_bad = 0;
    for (int rep = 0; rep < 10; ++rep) {
        CK(hipMemset(dout, 0xff, 2 * n * 4));
        hipLaunchKernelGGL((k_mix<MODE, PACKED>), dim3(blocks), dim3(512), 0, 0, da, db, dcs, dout, dsink, per_wave);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ho.data(), dout, 2 * n * 4, hipMemcpyDeviceToHost));
        long bcount = 0;
        for (size_t i = 0; i < n; ++i) {
            const float a0 = ha[2 * i] + hb[2 * i], a1 = ha[2 * i + 1] + hb[2 * i + 1], c0 = hcs[2 * i], s0 = hcs[2 * i + 1];
            const float w0 = fmaf(c0, a0, -(s0 * a1)), w1 = fmaf(c0, a1, s0 * a0);
            const bool e0 = fabsf(ho[2 * i] - w0) > 1e-5f * (1.0f + fabsf(w0)), e1 = fabsf(ho[2 * i + 1] - w1) > 1e-5f * (1.0f + fabsf(w1));
            if (e0 || e1) {
                ++bcount;
                bad_lo += e0;
                bad_hi += e1;
                ++q[(i & 63) >> 4];
            }
        }
        launches_bad += bcount != 0;
    }
    printf("partner waves %-22s arithmetic %-9s: launches with wrong results %2d / 10, wrong 'c a0 - s a1' %ld, wrong 'c a1 + s a0' %ld, by lane quarter = %ld %ld %ld %ld\n",
           MODE == 0 ? "idle" : (MODE == 1 ? "stream MFMA 32x32x16" : "stream v_fma_f32"), PACKED ? "packed" : "un-packed", launches_bad, bad_lo, bad_hi, q[0], q[1],
           q[2], q[3]);
}

int main() {
    const int blocks = 2048, per_wave = 64;
    const size_t n = (size_t)blocks * 4 * per_wave * 64;
    std::vector<float> ha(2 * n), hb(2 * n), hcs(2 * n);
    srand(1);
    for (size_t i = 0; i < 2 * n; ++i) {
        ha[i] = (float)(rand() % 4000) / 1000.0f - 2.0f;
        hb[i] = (float)(rand() % 2000) / 10000.0f - 0.1f;
        hcs[i] = (float)(rand() % 2000) / 1000.0f - 1.0f;
    }
    float *da, *db, *dcs, *dout, *dsink;
    CK(hipMalloc(&da, 2 * n * 4));
    CK(hipMalloc(&db, 2 * n * 4));
    CK(hipMalloc(&dcs, 2 * n * 4));
    CK(hipMalloc(&dout, 2 * n * 4));
    CK(hipMalloc(&dsink, (size_t)blocks * 512 * 4));
    CK(hipMemcpy(da, ha.data(), 2 * n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), 2 * n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcs, hcs.data(), 2 * n * 4, hipMemcpyHostToDevice));
    run<0, 1>(da, db, dcs, dout, dsink, ha, hb, hcs, blocks, per_wave);
    run<2, 1>(da, db, dcs, dout, dsink, ha, hb, hcs, blocks, per_wave);
    run<1, 1>(da, db, dcs, dout, dsink, ha, hb, hcs, blocks, per_wave);
    run<1, 0>(da, db, dcs, dout, dsink, ha, hb, hcs, blocks, per_wave);
    return 0;
}
