// Round-2 verdict hazard (a), isolated.  The LDS-free QKV epilogue experiment (tools/experiments/qkv_direct_epilogue.patch) returned
// wrong elements "on lanes 48-63 only and differently on every launch".  Round 3 (tools/r3_probe11.py, r3_probe12.py) showed:
//   * the K loop is not involved (constant accumulators fail the same way), waits are not involved (vmcnt(0) everywhere: same);
//   * only the "x cos - y sin" outputs fail, their "sin" product comes out as 0, on lanes 48-63;
//   * built with -fno-slp-vectorize (no v_pk_*_f32 in the epilogue) the kernel is exact and deterministic.
// The compiler had SLP-packed the rotation into packed-f32 VALU instructions whose halves read each other's operands through
// op_sel / op_sel_hi.  This probe executes exactly that 3-instruction group (inline asm, same operand selects, same overlap of
// destination and source pairs) on data that arrives from global memory, many waves per SIMD, and checks every lane.
// Variants: 0 = the group as the compiler emitted it, 1 = the packed multiply with a destination pair that overlaps no source,
// 2 = an s_nop between the multiply and the first fma, 3 = the hi halves of the scalar-broadcast pairs (never selected) set to 0
// instead of stale integers, 4 = variant 0 behind the EXEC restore / save-and-mask pair the compiler wraps around `if (rowok)`.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/pk_f32_hazard.hip -o tools/probes/bin/pk_f32_hazard
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

__device__ __forceinline__ bool a0_ok(f32x2 v) { return v[0] > -100.0f; }      // always true, but the compiler cannot know

template <int VAR>
__global__ __launch_bounds__(256) void k_rot(const float* __restrict__ a, const float* __restrict__ cs, float* __restrict__ out, int n,
                                             int iters) {
    for (int it = 0; it < iters; ++it) {
        const size_t i = ((size_t)it * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        if (i >= (size_t)n) return;
        f32x2 av = *reinterpret_cast<const f32x2*>(a + 2 * i);     // (a0, a1)
        const float c0 = cs[2 * i], s0 = cs[2 * i + 1];
        // pairs whose hi register is never selected: the compiler left whatever the register held before (an LDS address)
        const float junk = VAR == 3 ? 0.0f : __int_as_float(0x5000 + (int)threadIdx.x);
        f32x2 sv = {s0, junk}, cv = {c0, junk}, o_lo, o_hi;
        if (VAR == 1) {
            f32x2 t;
            asm volatile(
                "v_pk_mul_f32 %0, %3, %5 op_sel:[0,1] op_sel_hi:[0,0]\n\t"
                "v_pk_fma_f32 %1, %4, %5, %0 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                "v_pk_fma_f32 %2, %4, %5, %0 op_sel_hi:[0,1,1]"
                : "=&v"(t), "=&v"(o_lo), "=&v"(o_hi)
                : "v"(sv), "v"(cv), "v"(av));
        } else if (VAR == 2) {
            asm volatile(
                "v_pk_mul_f32 %2, %2, %4 op_sel:[0,1] op_sel_hi:[0,0]\n\t"
                "s_nop 1\n\t"
                "v_pk_fma_f32 %0, %3, %4, %2 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                "v_pk_fma_f32 %1, %3, %4, %2 op_sel_hi:[0,1,1]"
                : "=&v"(o_lo), "=&v"(o_hi), "+v"(sv)
                : "v"(cv), "v"(av));
        } else if (VAR == 5) {
            // variant 0 in the REGISTERS of the failing kernel (v[84:85] sin pair, v[44:45] operands, v[80:81] cos pair, v[16:17] result)
            float o0, o1;
            asm volatile(
                "v_mov_b32 v84, %2\n\tv_mov_b32 v85, %4\n\tv_mov_b32 v80, %3\n\tv_mov_b32 v81, %4\n\tv_mov_b32 v44, %5\n\tv_mov_b32 v45, %6\n\t"
                "v_pk_mul_f32 v[84:85], v[84:85], v[44:45] op_sel:[0,1] op_sel_hi:[0,0]\n\t"
                "v_pk_fma_f32 v[16:17], v[80:81], v[44:45], v[84:85] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                "v_pk_fma_f32 v[80:81], v[80:81], v[44:45], v[84:85] op_sel_hi:[0,1,1]\n\t"
                "v_mov_b32 %0, v16\n\tv_mov_b32 %1, v81"
                : "=&v"(o0), "=&v"(o1)
                : "v"(s0), "v"(c0), "v"(junk), "v"(av[0]), "v"(av[1])
                : "v16", "v17", "v44", "v45", "v80", "v81", "v84", "v85");
            o_lo = f32x2{o0, 0.0f};
            o_hi = f32x2{0.0f, o1};
        } else if (VAR == 4) {
            // the group behind the EXEC juggling the compiler puts around `if (rowok)`: restore EXEC, AND it with a (full) lane mask
            unsigned long long saved, mask = __ballot(a0_ok(av));
            asm volatile(
                "s_or_b64 exec, exec, %5\n\t"
                "s_and_saveexec_b64 %3, %5\n\t"
                "v_pk_mul_f32 %1, %1, %4 op_sel:[0,1] op_sel_hi:[0,0]\n\t"
                "v_pk_fma_f32 %0, %2, %4, %1 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                "v_pk_fma_f32 %2, %2, %4, %1 op_sel_hi:[0,1,1]\n\t"
                "s_or_b64 exec, exec, %3"
                : "=&v"(o_lo), "+v"(sv), "+v"(cv), "=&s"(saved)
                : "v"(av), "s"(mask));
            o_hi = cv;
        } else {
            // exactly the compiler's group: multiply in place (dst pair == src0 pair), second fma in place on the cos pair
            asm volatile(
                "v_pk_mul_f32 %1, %1, %3 op_sel:[0,1] op_sel_hi:[0,0]\n\t"
                "v_pk_fma_f32 %0, %2, %3, %1 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                "v_pk_fma_f32 %2, %2, %3, %1 op_sel_hi:[0,1,1]"
                : "=&v"(o_lo), "+v"(sv), "+v"(cv)
                : "v"(av));
            o_hi = cv;
        }
        out[2 * i] = o_lo[0];        // c0 a0 - s0 a1
        out[2 * i + 1] = o_hi[1];    // c0 a1 + s0 a0
    }
}

template <int VAR>
static void run(const float* da, const float* dcs, float* dout, const std::vector<float>& ha, const std::vector<float>& hcs, int n) {
    std::vector<float> ho(2 * (size_t)n);
    long bad_lo = 0, bad_hi = 0, bad_lane[4] = {0, 0, 0, 0};
    int launches_bad = 0;
    for (int rep = 0; rep < 20; ++rep) {
        CK(hipMemset(dout, 0xff, 2 * (size_t)n * 4));
        hipLaunchKernelGGL(k_rot<VAR>, dim3(4096), dim3(256), 0, 0, da, dcs, dout, n, (n + 4096 * 256 - 1) / (4096 * 256));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ho.data(), dout, 2 * (size_t)n * 4, hipMemcpyDeviceToHost));
        long b = 0;
        for (int i = 0; i < n; ++i) {
            const float a0 = ha[2 * i], a1 = ha[2 * i + 1], c0 = hcs[2 * i], s0 = hcs[2 * i + 1];
            const float w0 = fmaf(c0, a0, -(s0 * a1)), w1 = fmaf(c0, a1, s0 * a0);
            const bool e0 = fabsf(ho[2 * i] - w0) > 1e-5f * (1.0f + fabsf(w0)), e1 = fabsf(ho[2 * i + 1] - w1) > 1e-5f * (1.0f + fabsf(w1));
            if (e0 || e1) {
                ++b;
                bad_lo += e0;
                bad_hi += e1;
                ++bad_lane[(i & 63) >> 4];
            }
        }
        launches_bad += b != 0;
    }
    printf("variant %d: launches with wrong results %d / 20, wrong 'c a0 - s a1' %ld, wrong 'c a1 + s a0' %ld, by lane quarter [0-15 16-31 32-47 48-63] = %ld %ld %ld %ld\n",
           VAR, launches_bad, bad_lo, bad_hi, bad_lane[0], bad_lane[1], bad_lane[2], bad_lane[3]);
}

int main() {
    const int n = 1 << 23;
    std::vector<float> ha(2 * (size_t)n), hcs(2 * (size_t)n);
    srand(1);
    for (size_t i = 0; i < 2 * (size_t)n; ++i) {
        ha[i] = (float)rand() / RAND_MAX * 4.0f - 2.0f;
        hcs[i] = (float)rand() / RAND_MAX * 2.0f - 1.0f;
    }
    float *da, *dcs, *dout;
    CK(hipMalloc(&da, 2 * (size_t)n * 4));
    CK(hipMalloc(&dcs, 2 * (size_t)n * 4));
    CK(hipMalloc(&dout, 2 * (size_t)n * 4));
    CK(hipMemcpy(da, ha.data(), 2 * (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcs, hcs.data(), 2 * (size_t)n * 4, hipMemcpyHostToDevice));
    run<0>(da, dcs, dout, ha, hcs, n);
    run<1>(da, dcs, dout, ha, hcs, n);
    run<2>(da, dcs, dout, ha, hcs, n);
    run<3>(da, dcs, dout, ha, hcs, n);
    run<4>(da, dcs, dout, ha, hcs, n);
    run<5>(da, dcs, dout, ha, hcs, n);
    return 0;
}
