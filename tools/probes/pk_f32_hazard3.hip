// Hazard (a), third isolation step.  tools/r3_probe13.py caught the failing lanes in the act: every operand register is right when
// re-read a few instructions later, the hi half of the packed group is right, the LO half behaves as if the just-loaded "sin"
// register still held its old content on lanes 48-63 -- and it is always the FIRST packed instruction after an s_waitcnt vmcnt.
// This probe puts load, wait and consumer back to back in one asm block:
//     global_load_dword s, ...    (s arrives from memory)
//     s_waitcnt vmcnt(0)
//     [pad]                       variant-dependent: nothing / s_nop 0 / s_nop 1 / v_nop
//     consumer                    packed: v_pk_mul_f32 d[0:1], s[0:1], a[0:1] op_sel:[0,1] op_sel_hi:[0,0]   (lo = s * a1, hi = s * a0)
//                                 plain:  v_mul_f32 d0, s, a1 ; v_mul_f32 d1, s, a0
// and counts wrong lo / hi products per lane quarter.  The register that receives s is pre-set to a stale pattern.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/pk_f32_hazard3.hip -o tools/probes/bin/pk_f32_hazard3
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

template <int VAR>
__global__ __launch_bounds__(256) void k_probe(const float* __restrict__ a, const float* __restrict__ s, float* __restrict__ out, int n, int iters,
                                               int stride) {
    for (int it = 0; it < iters; ++it) {
        const size_t i = ((size_t)it * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        if (i >= (size_t)n) return;
        f32x2 av = *reinterpret_cast<const f32x2*>(a + 2 * i);
        const float* sp = s + (i * (size_t)stride) % (size_t)n;      // stride > 1: scattered lines, the load takes longer and lands unevenly
        f32x2 d;
        asm volatile("" : "+v"(av));                                  // operands in registers before the block
        // v100 receives s (pre-set to a stale pattern), v101 is the never-selected hi half of the pair
#define LOAD_WAIT "v_mov_b32 v100, 0x5000\n\tv_mov_b32 v101, 0x6000\n\tglobal_load_dword v100, %1, off\n\ts_waitcnt vmcnt(0)\n\t"
        if (VAR == 0)
            asm volatile(LOAD_WAIT "v_pk_mul_f32 %0, v[100:101], %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=&v"(d) : "v"(sp), "v"(av) : "v100", "v101", "memory");
        else if (VAR == 1)
            asm volatile(LOAD_WAIT "s_nop 0\n\tv_pk_mul_f32 %0, v[100:101], %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=&v"(d) : "v"(sp), "v"(av) : "v100", "v101", "memory");
        else if (VAR == 2)
            asm volatile(LOAD_WAIT "s_nop 3\n\tv_pk_mul_f32 %0, v[100:101], %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=&v"(d) : "v"(sp), "v"(av) : "v100", "v101", "memory");
        else if (VAR == 3) {
            float d0, d1;
            asm volatile(LOAD_WAIT "v_mul_f32 %0, v100, %4\n\tv_mul_f32 %1, v100, %3" : "=&v"(d0), "=&v"(d1) : "v"(sp), "v"(av[0]), "v"(av[1]) : "v100", "v101", "memory");
            d = f32x2{d0, d1};
        } else if (VAR == 4) {   // packed, in place on the loaded pair (as the compiler emitted it)
            float d0, d1;
            asm volatile(LOAD_WAIT "v_pk_mul_f32 v[100:101], v[100:101], %3 op_sel:[0,1] op_sel_hi:[0,0]\n\tv_mov_b32 %0, v100\n\tv_mov_b32 %1, v101"
                         : "=&v"(d0), "=&v"(d1) : "v"(sp), "v"(av) : "v100", "v101", "memory");
            d = f32x2{d0, d1};
        }
#undef LOAD_WAIT
        out[2 * i] = d[0];
        out[2 * i + 1] = d[1];
    }
}

template <int VAR>
static void run(const float* da, const float* ds, float* dout, const std::vector<float>& ha, const std::vector<float>& hs, int n, int stride) {
    std::vector<float> ho(2 * (size_t)n);
    long bad_lo = 0, bad_hi = 0, q[4] = {0, 0, 0, 0};
    int launches_bad = 0;
    for (int rep = 0; rep < 10; ++rep) {
        CK(hipMemset(dout, 0xff, 2 * (size_t)n * 4));
        hipLaunchKernelGGL(k_probe<VAR>, dim3(2048), dim3(256), 0, 0, da, ds, dout, n, (n + 2048 * 256 - 1) / (2048 * 256), stride);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ho.data(), dout, 2 * (size_t)n * 4, hipMemcpyDeviceToHost));
        long b = 0;
        for (int i = 0; i < n; ++i) {
            const float sv = hs[((size_t)i * stride) % (size_t)n];
            const float w0 = sv * ha[2 * i + 1], w1 = sv * ha[2 * i];
            const bool e0 = ho[2 * i] != w0, e1 = ho[2 * i + 1] != w1;
            if (e0 || e1) {
                ++b;
                bad_lo += e0;
                bad_hi += e1;
                ++q[(i & 63) >> 4];
            }
        }
        launches_bad += b != 0;
    }
    printf("variant %d stride %4d: launches with wrong products %2d / 10, wrong lo %ld, wrong hi %ld, by lane quarter = %ld %ld %ld %ld\n", VAR, stride,
           launches_bad, bad_lo, bad_hi, q[0], q[1], q[2], q[3]);
}

int main() {
    const int n = 1 << 22;
    std::vector<float> ha(2 * (size_t)n), hs(n);
    srand(1);
    for (size_t i = 0; i < 2 * (size_t)n; ++i) ha[i] = (float)(rand() % 2000) / 500.0f - 2.0f;
    for (size_t i = 0; i < (size_t)n; ++i) hs[i] = (float)(rand() % 2000) / 1000.0f - 1.0f;
    float *da, *ds, *dout;
    CK(hipMalloc(&da, 2 * (size_t)n * 4));
    CK(hipMalloc(&ds, (size_t)n * 4));
    CK(hipMalloc(&dout, 2 * (size_t)n * 4));
    CK(hipMemcpy(da, ha.data(), 2 * (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, hs.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    for (int stride : {1, 937}) {
        run<0>(da, ds, dout, ha, hs, n, stride);
        run<4>(da, ds, dout, ha, hs, n, stride);
        run<1>(da, ds, dout, ha, hs, n, stride);
        run<2>(da, ds, dout, ha, hs, n, stride);
        run<3>(da, ds, dout, ha, hs, n, stride);
    }
    return 0;
}
