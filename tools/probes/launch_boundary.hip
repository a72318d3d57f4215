// What does a kernel boundary cost in a stream vs in a replayed hipGraph?  bench.py's batch-1 sub-record `b1_eager` is 2 % FASTER than
// the graph replay of the same 5 100 launches (71.5 vs 73.0 ms): 0.3 us per launch.  This probe times chains of N dependent
// launches of (a) an empty kernel, (b) a kernel that reads what its predecessor wrote (256 blocks x 16 KB: a boundary has to make
// the data visible), each as plain stream launches and as a captured graph replayed back to back.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/launch_boundary.hip -o tools/probes/bin/launch_boundary
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

__global__ void k_empty() {}
__global__ __launch_bounds__(256) void k_chain(const float4* __restrict__ in, float4* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 v = in[base + i * 256];
        v.x += 1.0f;
        out[base + i * 256] = v;
    }
}

static float time_ms(hipStream_t s, hipEvent_t e0, hipEvent_t e1, int reps, void (*fn)(hipStream_t, void*), void* ctx) {
    fn(s, ctx);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) fn(s, ctx);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

struct Ctx {
    int n, chain;
    float4 *a, *b;
    hipGraphExec_t ge;
};
static void eager(hipStream_t s, void* c_) {
    Ctx* c = (Ctx*)c_;
    for (int i = 0; i < c->n; ++i) {
        if (c->chain) hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, s, (i & 1) ? c->b : c->a, (i & 1) ? c->a : c->b);
        else hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s);
    }
}
static void replay(hipStream_t s, void* c_) { CK(hipGraphLaunch(((Ctx*)c_)->ge, s)); }

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    Ctx c;
    c.n = 1000;
    CK(hipMalloc(&c.a, 256 * 1024 * 16));
    CK(hipMalloc(&c.b, 256 * 1024 * 16));
    CK(hipMemset(c.a, 0, 256 * 1024 * 16));
    for (int chain = 0; chain < 2; ++chain) {
        c.chain = chain;
        hipGraph_t g;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        eager(s, &c);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&c.ge, g, nullptr, nullptr, 0));
        for (int round = 0; round < 3; ++round) {
            const float te = time_ms(s, e0, e1, 5, eager, &c), tg = time_ms(s, e0, e1, 5, replay, &c);
            printf("%-34s stream launches %.3f us / launch   graph replay %.3f us / launch\n", chain ? "kernel reading its predecessor's output" : "empty kernel", te * 1e3 / c.n,
                   tg * 1e3 / c.n);
        }
        CK(hipGraphExecDestroy(c.ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
