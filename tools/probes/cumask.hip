// Can two halves of the chip run two independent kernel streams side by side?  hipExtStreamCreateWithCUMask creates a stream whose
// kernels are confined to a subset of CUs.  Questions (MI355X, ROCm 7.x):
//   1. which mask bit is which XCD (hypotheses: bit i -> XCD i % 8, or bit i -> XCD i / 32)?
//   2. do two masked streams execute concurrently (an MFMA-bound kernel on one half, an HBM-bound kernel on the other)?
//   3. does a kernel node captured from a masked stream keep the mask when the graph is replayed (on the same masked stream)?
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/cumask.hip -o /tmp/cumask
#include <hip/hip_runtime.h>
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

__global__ void where_kernel(int* xcc_hist, int* se_hist) {
    if (threadIdx.x == 0) {
        // HW_REG_XCC_ID = 20, bits [3:0]; HW_REG_HW_ID = 4: SE_ID bits [15:13] (3 bits), CU_ID [11:8]
        const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
        atomicAdd(&xcc_hist[xcc & 15], 1);
        const unsigned hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);
        atomicAdd(&se_hist[((hw >> 13) & 7) * 16 + ((hw >> 8) & 15)], 1);
    }
    // hold the CU for a moment so that the blocks spread over every CU the stream may use
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void mfma_kernel(float* out, int iters) {
    f32x16 acc0 = {}, acc1 = {};
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(0.01f * (threadIdx.x % 13 + i));
        b[i] = (_Float16)(0.02f * (threadIdx.x % 7 + i));
    }
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
    }
    if (acc0[0] + acc1[3] == 12345.0f) out[0] = acc0[0];
}

__global__ __launch_bounds__(256) void rmw_kernel(float4* x, size_t n_per_block) {
    float4* p = x + (size_t)blockIdx.x * n_per_block;
    for (size_t i = threadIdx.x; i < n_per_block; i += 256) {
        float4 v = p[i];
        v.x += 1.0f; v.y += 1.0f; v.z += 1.0f; v.w += 1.0f;
        p[i] = v;
    }
}

static void print_hist(const char* tag, int* d_x, int* d_s) {
    int hx[16], hs[128];
    CK(hipMemcpy(hx, d_x, sizeof(hx), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs, d_s, sizeof(hs), hipMemcpyDeviceToHost));
    printf("%-34s XCC:", tag);
    for (int i = 0; i < 8; ++i) printf(" %5d", hx[i]);
    int cus = 0;
    for (int i = 0; i < 128; ++i) cus += hs[i] > 0;
    printf("   distinct (SE,CU) ids seen: %d\n", cus);
    CK(hipMemset(d_x, 0, sizeof(hx)));
    CK(hipMemset(d_s, 0, sizeof(hs)));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    int *d_x, *d_s;
    CK(hipMalloc(&d_x, 16 * 4));
    CK(hipMalloc(&d_s, 128 * 4));
    CK(hipMemset(d_x, 0, 64));
    CK(hipMemset(d_s, 0, 512));

    std::vector<uint32_t> m_mod_lo(words, 0), m_mod_hi(words, 0), m_div_lo(words, 0), m_div_hi(words, 0);
    for (int i = 0; i < ncu; ++i) {
        ((i % 8) < 4 ? m_mod_lo : m_mod_hi)[i / 32] |= 1u << (i % 32);
        ((i / 32) < 4 ? m_div_lo : m_div_hi)[i / 32] |= 1u << (i % 32);
    }
    hipStream_t s_plain, s_mod_lo, s_mod_hi, s_div_lo, s_div_hi;
    CK(hipStreamCreate(&s_plain));
    CK(hipExtStreamCreateWithCUMask(&s_mod_lo, words, m_mod_lo.data()));
    CK(hipExtStreamCreateWithCUMask(&s_mod_hi, words, m_mod_hi.data()));
    CK(hipExtStreamCreateWithCUMask(&s_div_lo, words, m_div_lo.data()));
    CK(hipExtStreamCreateWithCUMask(&s_div_hi, words, m_div_hi.data()));

    // ---- 1. placement
    struct { const char* n; hipStream_t s; } cases[] = {{"plain stream", s_plain}, {"mask bits i%8<4", s_mod_lo}, {"mask bits i%8>=4", s_mod_hi},
                                                         {"mask bits i/32<4", s_div_lo}, {"mask bits i/32>=4", s_div_hi}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(where_kernel, dim3(4096), dim3(64), 0, c.s, d_x, d_s);
        CK(hipStreamSynchronize(c.s));
        print_hist(c.n, d_x, d_s);
    }

    // ---- 3. graph capture from a masked stream, replay on the same stream and on a plain stream
    for (int which = 0; which < 2; ++which) {
        hipStream_t ms = which == 0 ? s_mod_lo : s_div_lo;
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(ms, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(where_kernel, dim3(4096), dim3(64), 0, ms, d_x, d_s);
        CK(hipStreamEndCapture(ms, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, ms));
        CK(hipStreamSynchronize(ms));
        print_hist(which == 0 ? "graph replay on masked (i%8<4)" : "graph replay on masked (i/32<4)", d_x, d_s);
        CK(hipGraphLaunch(ge, s_plain));
        CK(hipStreamSynchronize(s_plain));
        print_hist("  same graph on the plain stream", d_x, d_s);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }

    // ---- 2. concurrency: MFMA-bound kernel and HBM read-modify-write kernel, alone on the full chip, alone on half, side by side
    float* d_out;
    CK(hipMalloc(&d_out, 4));
    const int rmw_blocks = 2048;
    const size_t n_per_block = 16384;                  // 256 KB per block -> 512 MB
    float4* d_buf;
    CK(hipMalloc(&d_buf, (size_t)rmw_blocks * n_per_block * sizeof(float4)));
    CK(hipMemset(d_buf, 0, (size_t)rmw_blocks * n_per_block * sizeof(float4)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto wall = [&](auto&& fn) {
        fn();
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s_plain));
            CK(hipStreamWaitEvent(s_mod_lo, e0, 0));
            CK(hipStreamWaitEvent(s_mod_hi, e0, 0));
            CK(hipStreamWaitEvent(s_div_lo, e0, 0));
            CK(hipStreamWaitEvent(s_div_hi, e0, 0));
            fn();
            hipEvent_t j[4];
            hipStream_t ss[4] = {s_mod_lo, s_mod_hi, s_div_lo, s_div_hi};
            for (int k = 0; k < 4; ++k) {
                CK(hipEventCreate(&j[k]));
                CK(hipEventRecord(j[k], ss[k]));
                CK(hipStreamWaitEvent(s_plain, j[k], 0));
            }
            CK(hipEventRecord(e1, s_plain));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            for (int k = 0; k < 4; ++k) CK(hipEventDestroy(j[k]));
        }
        return best;
    };
    const int mf_iters = 40000;
    auto mf = [&](hipStream_t s, int blocks) { hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(512), 0, s, d_out, mf_iters); };
    auto rm = [&](hipStream_t s, int blocks) { hipLaunchKernelGGL(rmw_kernel, dim3(blocks), dim3(256), 0, s, d_buf, n_per_block); };
    printf("mfma kernel, 256 blocks, plain stream        : %.3f ms\n", wall([&] { mf(s_plain, 256); }));
    printf("rmw 512 MB, plain stream                     : %.3f ms\n", wall([&] { rm(s_plain, rmw_blocks); }));
    printf("mfma then rmw, one plain stream              : %.3f ms\n", wall([&] { mf(s_plain, 256); rm(s_plain, rmw_blocks); }));
    for (int which = 0; which < 2; ++which) {
        hipStream_t lo = which == 0 ? s_mod_lo : s_div_lo, hi = which == 0 ? s_mod_hi : s_div_hi;
        const char* tag = which == 0 ? "i%8" : "i/32";
        printf("[%s] mfma 128 blocks on the low half alone      : %.3f ms\n", tag, wall([&] { mf(lo, 128); }));
        printf("[%s] rmw 512 MB on the high half alone          : %.3f ms\n", tag, wall([&] { rm(hi, rmw_blocks); }));
        printf("[%s] mfma(128, low) || rmw(high)                : %.3f ms\n", tag, wall([&] { mf(lo, 128); rm(hi, rmw_blocks); }));
        printf("[%s] mfma(128, low) || mfma(128, high)          : %.3f ms\n", tag, wall([&] { mf(lo, 128); mf(hi, 128); }));
        printf("[%s] rmw(256 MB, low) || rmw(256 MB, high)      : %.3f ms\n", tag, wall([&] { rm(lo, rmw_blocks / 2); hipLaunchKernelGGL(rmw_kernel, dim3(rmw_blocks / 2), dim3(256), 0, hi, d_buf + (size_t)(rmw_blocks / 2) * n_per_block, n_per_block); }));
    }
    return 0;
}
