// Round-2 hazard (DESIGN.md "The sampling path enqueues kernels only"): with hipMemsetAsync / hipMemcpyAsync nodes inside the
// captured sample() graph, back-to-back replays returned garbage on some boxes -- the memory operations did not seem to be
// ordered against their neighbouring KERNEL nodes.  This probe is the 80-line version of that pattern, outside the engine:
//
//   capture:  memset(A, 0)  ->  k_add(A += 1)  ->  memcpy D2D (B <- A)  ->  k_check(B == 1 ? ok : ++errors)  ->  k_poison(A = 7, B = 9)
//
// and N back-to-back replays with no host synchronisation in between.  Every stage depends on the previous one through stream
// order only.  If a memset / memcpy node can overtake or lag a kernel node, k_check sees 7, 8, 9 or 10 instead of 1.
// Variants: sizes from 4 KB to 64 MB (small memsets may run as kernels, large ones on the copy engines), a second variant where
// the memory operations are replaced by kernels (what the engine does since round 2), graph replay vs plain stream order.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/graph_memops.hip -o tools/probes/bin/graph_memops
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

__global__ void k_add(float* a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] += 1.0f;
}
__global__ void k_check(const float* b, size_t n, unsigned long long* errors, float* first_bad) {
    unsigned long long bad = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (b[i] != 1.0f) {
            if (bad == 0 && atomicAdd(errors + 1, 1ull) == 0) *first_bad = b[i];
            ++bad;
        }
    if (bad) atomicAdd(errors, bad);
}
__global__ void k_poison(float* a, float* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        a[i] = 7.0f;
        b[i] = 9.0f;
    }
}
__global__ void k_zero(float* a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = 0.0f;
}
__global__ void k_copy(float* b, const float* a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

static void enqueue(hipStream_t s, float* A, float* B, size_t n, unsigned long long* err, float* bad, bool memops) {
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (memops) CK(hipMemsetAsync(A, 0, n * 4, s));
    else hipLaunchKernelGGL(k_zero, dim3(blocks), dim3(256), 0, s, A, n);
    hipLaunchKernelGGL(k_add, dim3(blocks), dim3(256), 0, s, A, n);
    if (memops) CK(hipMemcpyAsync(B, A, n * 4, hipMemcpyDeviceToDevice, s));
    else hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s, B, A, n);
    hipLaunchKernelGGL(k_check, dim3(blocks), dim3(256), 0, s, B, n, err, bad);
    hipLaunchKernelGGL(k_poison, dim3(blocks), dim3(256), 0, s, A, B, n);
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    unsigned long long* err;
    float* bad;
    CK(hipMalloc(&err, 16));
    CK(hipMalloc(&bad, 4));
    const size_t sizes[] = {1024, 64 * 1024, 1024 * 1024, 16 * 1024 * 1024};
    const int replays = 300;
    for (size_t n : sizes) {
        float *A, *B;
        CK(hipMalloc(&A, n * 4));
        CK(hipMalloc(&B, n * 4));
        for (int memops = 1; memops >= 0; --memops)
            for (int graph = 1; graph >= 0; --graph) {
                CK(hipMemset(err, 0, 16));
                CK(hipMemset(bad, 0, 4));
                hipLaunchKernelGGL(k_poison, dim3(256), dim3(256), 0, s, A, B, n);
                CK(hipStreamSynchronize(s));
                if (graph) {
                    hipGraph_t g;
                    hipGraphExec_t ge;
                    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                    for (int rep = 0; rep < 4; ++rep) enqueue(s, A, B, n, err, bad, memops);       // 4 rounds per graph: 20 nodes
                    CK(hipStreamEndCapture(s, &g));
                    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                    for (int r = 0; r < replays / 4; ++r) CK(hipGraphLaunch(ge, s));              // back to back, no host sync
                    CK(hipStreamSynchronize(s));
                    CK(hipGraphExecDestroy(ge));
                    CK(hipGraphDestroy(g));
                } else {
                    for (int r = 0; r < replays; ++r) enqueue(s, A, B, n, err, bad, memops);
                    CK(hipStreamSynchronize(s));
                }
                unsigned long long h[2];
                float hb;
                CK(hipMemcpy(h, err, 16, hipMemcpyDeviceToHost));
                CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
                printf("%8zu KB  %-22s %-12s wrong elements %llu (first wrong value %g)\n", n * 4 / 1024,
                       memops ? "memset/memcpy nodes" : "kernels only", graph ? "graph replay" : "stream", h[0], hb);
            }
        CK(hipFree(A));
        CK(hipFree(B));
    }
    return 0;
}
