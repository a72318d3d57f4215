// Round-2 hazard (b), re-read.  Until commit fe99108 f5_sample() staged its small host-side tables (time grid, step sizes, lens,
// duration pairs, cfg) with hipMemcpyAsync(..., hipMemcpyHostToDevice, stream) FROM LOCAL std::vectors / the caller's argument struct, and
// those calls were captured into the sample() graph.  Replays then returned garbage "on some boxes", which round 2 wrote up as
// memset / memcpy nodes being mis-ordered against kernel nodes.  tools/probes/graph_memops.hip shows that ordering is fine.  What a
// captured host-to-device copy actually records is the host ADDRESS: every replay reads that address again.  By then the vectors
// were gone and the heap block held whatever came next -- layout- and box-dependent garbage, and none on the eager path.
// This probe: capture an H2D copy from a vector that dies at the end of the scope, reuse the heap, replay, look at the device.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/graph_h2d_capture.hip -o tools/probes/bin/graph_h2d_capture
#include <hip/hip_runtime.h>
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

__global__ void k_scale(float* d, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] *= 2.0f;
}

static void report(const char* what, float* dev, int n) {
    std::vector<float> h(n);
    CK(hipMemcpy(h.data(), dev, n * 4, hipMemcpyDeviceToHost));
    int ok = 0;
    for (int i = 0; i < n; ++i) ok += h[i] == 2.0f;
    printf("%-64s %d of %d elements are 2.0 (first values %g %g %g)\n", what, ok, n, h[0], h[1], h[n - 1]);
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int n : {8, 1024, 65536}) {
        float* dev;
        CK(hipMalloc(&dev, n * 4));
        hipGraph_t g;
        hipGraphExec_t ge;
        {
            std::vector<float> staged(n, 1.0f);              // like `tnfe`, `dts`, `dur2` in the old f5_sample()
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            CK(hipMemcpyAsync(dev, staged.data(), n * 4, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_scale, dim3((n + 255) / 256), dim3(256), 0, s, dev, n);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s));                       // first replay while the vector is alive
            CK(hipStreamSynchronize(s));
            char what[96];
            snprintf(what, sizeof what, "n = %6d  replay while the host vector is alive:", n);
            report(what, dev, n);
        }                                                    // the vector dies here
        std::vector<float> next(n, 7.0f);                    // the next allocation of that size: very likely the same heap block
        CK(hipMemset(dev, 0, n * 4));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        char what[96];
        snprintf(what, sizeof what, "n = %6d  replay after the vector died and the heap was reused:", n);
        report(what, dev, n);
        {                                                    // the same two calls without a graph: the copy is staged at call time
            std::vector<float> staged(n, 1.0f);
            CK(hipMemcpyAsync(dev, staged.data(), n * 4, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_scale, dim3((n + 255) / 256), dim3(256), 0, s, dev, n);
        }
        std::vector<float> next2(n, 9.0f);
        CK(hipStreamSynchronize(s));
        snprintf(what, sizeof what, "n = %6d  plain stream, vector destroyed right after the call:", n);
        report(what, dev, n);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
        CK(hipFree(dev));
    }
    return 0;
}
