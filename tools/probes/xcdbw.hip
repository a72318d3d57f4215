// How much HBM bandwidth can the workgroups of nx XCDs (of 8) pull on their own?  Decides whether de-phasing the GEMM epilogues
// XCD by XCD can shorten them: if one XCD's fabric port sustains much more than 1/8 of the chip's bandwidth, an epilogue that
// runs while the other XCDs are in their main loops finishes faster than one that all 256 CUs enter together.
// Workgroup b runs on XCD b % 8 (observed placement, MI355X_MICROARCH.md); blocks of XCDs >= nx exit at once.
// mode 0: fp32 read-modify-write (the residual-update epilogue), 1: read only, 2: write only.  16 B per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void stream_kernel(float4* x, size_t n_per_block, int nx, int mode, float* sink) {
    const int xcd = blockIdx.x & 7;
    if (xcd >= nx) return;
    const size_t blk = (size_t)(blockIdx.x >> 3) * nx + xcd;
    float4* p = x + blk * n_per_block;
    float acc = 0.0f;
    for (size_t i = threadIdx.x; i < n_per_block; i += 256) {
        if (mode == 2) {
            p[i] = make_float4(1.0f, 2.0f, 3.0f, (float)i);
        } else {
            float4 v = p[i];
            if (mode == 0) {
                v.x += 1.0f; v.y += 1.0f; v.z += 1.0f; v.w += 1.0f;
                p[i] = v;
            } else {
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (mode == 1 && acc == 12345.678f) sink[0] = acc;
}

int main() {
    const int blocks = 4096;                       // 16 per CU in dispatch order
    const size_t n_per_block = 16384;              // float4 per block = 256 KB
    float4* x;
    float* sink;
    hipMalloc(&x, (size_t)blocks * n_per_block * sizeof(float4));   // 1 GiB
    hipMalloc(&sink, 4);
    hipMemset(x, 0, (size_t)blocks * n_per_block * sizeof(float4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[3] = {"rmw", "read", "write"};
    for (int mode = 0; mode < 3; ++mode)
        for (int nx = 1; nx <= 8; nx *= 2) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, 0, x, n_per_block, nx, mode, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double bytes = (double)blocks * nx / 8 * n_per_block * 16 * (mode == 0 ? 2 : 1);
            printf("%-5s nx=%d  %.3f ms  %.2f TB/s total  %.2f TB/s per active XCD\n", names[mode], nx, best, bytes / best / 1e9,
                   bytes / best / 1e9 / nx);
        }
    return 0;
}
