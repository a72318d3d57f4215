// Hazard (a), second isolation step: the epilogue FUNCTION of the experiment (direct_epilogue_qk of
// tools/experiments/qkv_direct_epilogue.patch, MB = 1, NB = 2, fp16 outputs) compiled as the compiler likes (SLP-packed f32 VALU),
// driven by constant accumulators, launched like the failing 64 x 128 kernel (4 waves, 480 workgroups), repeated, compared
// run to run and against a scalar host evaluation.  -DNOSLP=1 builds the same source with -fno-slp-vectorize semantics
// (the function is then compiled under `#pragma clang attribute` optnone-free scalar code via volatile barriers).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/pk_f32_hazard2.hip -o tools/probes/bin/pk_f32_hazard2
//        hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/probes/pk_f32_hazard2.hip -o tools/probes/bin/pk_f32_hazard2_noslp
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
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

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

struct Args {
    int M, seq_len, dmodel, ldob, rope_ldt;
    const float *rope_cos_tq, *rope_sin_tq, *rope_cos_tk, *rope_sin_tk, *bias;
    _Float16* out;
};

__device__ __forceinline__ float sat(float f) { return __builtin_amdgcn_fmed3f(f, -65504.0f, 65504.0f); }
__device__ inline uint32_t pack2(float a, float b) {
    const f32x2 v = {sat(a), sat(b)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h16x2));
}

template <int MBW, int NBW>
__device__ __forceinline__ void direct_epilogue_qk(const Args& p, f32x16 (&acc)[MBW][NBW], int row0, int colbase, int lane) {
    const int hi = lane >> 5, lcol = lane & 31;
    const bool isq = colbase < p.dmodel;
    const float* ct = isq ? p.rope_cos_tq : p.rope_cos_tk;
    const float* st = isq ? p.rope_sin_tq : p.rope_sin_tk;
    constexpr int RGB = 4;
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
        const int row = row0 + mb * 32 + lcol;
        const bool rowok = row < p.M;
        const int n = rowok ? row % p.seq_len : 0;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int rg0 = 0; rg0 < 4; rg0 += RGB) {
                float c0[RGB], c1[RGB], s0[RGB], s1[RGB];
                f32x4 b4[RGB];
#pragma unroll
                for (int i = 0; i < RGB; ++i) {
                    const int c = colbase + nb * 32 + (rg0 + i) * 8 + hi * 4;
                    const int j0 = (c & 63) >> 1;
                    c0[i] = ct[(size_t)j0 * p.rope_ldt + n];
                    c1[i] = ct[(size_t)(j0 + 1) * p.rope_ldt + n];
                    s0[i] = st[(size_t)j0 * p.rope_ldt + n];
                    s1[i] = st[(size_t)(j0 + 1) * p.rope_ldt + n];
                    b4[i] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int i = 0; i < RGB; ++i) {
                    const int rg = rg0 + i;
                    const int c = colbase + nb * 32 + rg * 8 + hi * 4;
                    const float a0 = acc[mb][nb][rg * 4 + 0] + b4[i][0], a1 = acc[mb][nb][rg * 4 + 1] + b4[i][1];
                    const float a2 = acc[mb][nb][rg * 4 + 2] + b4[i][2], a3 = acc[mb][nb][rg * 4 + 3] + b4[i][3];
                    const float o0 = a0 * c0[i] - a1 * s0[i], o1 = a1 * c0[i] + a0 * s0[i];
                    const float o2 = a2 * c1[i] - a3 * s1[i], o3 = a3 * c1[i] + a2 * s1[i];
                    if (rowok) {
                        const size_t off = (size_t)row * p.ldob + c;
                        *reinterpret_cast<u32x2*>(p.out + off) = u32x2{pack2(o0, o1), pack2(o2, o3)};
                    }
                }
            }
    }
}

__host__ __device__ __forceinline__ float acc_value(int row_in_tile_lane, int e, int nb) { return 0.25f + 0.001f * (float)(row_in_tile_lane + 3 * e + 7 * nb); }

__global__ __launch_bounds__(256) void k_epi(Args p, int tiles_n) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * 64, n0 = tile_n * 128;
    f32x16 acc[1][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][nb][e] = acc_value(lane, e, nb);
    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]));
    direct_epilogue_qk<1, 2>(p, acc, m0 + wm * 32, n0 + wn * 64, lane);
}

int main() {
    const int B = 2, N = 937, M = B * N, D = 1024, W = 2 * D;
    std::vector<float> hct((size_t)32 * N), hst((size_t)32 * N), hbias(W);
    for (int j = 0; j < 32; ++j)
        for (int n = 0; n < N; ++n) {
            const double th = pow(10000.0, -2.0 * j / 64.0) * n;
            hct[(size_t)j * N + n] = (float)cos(th);
            hst[(size_t)j * N + n] = (float)sin(th);
        }
    srand(3);
    for (int c = 0; c < W; ++c) hbias[c] = ((float)rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    float *dct, *dst, *dbias;
    _Float16* dout;
    CK(hipMalloc(&dct, hct.size() * 4));
    CK(hipMalloc(&dst, hst.size() * 4));
    CK(hipMalloc(&dbias, W * 4));
    CK(hipMalloc(&dout, (size_t)M * W * 2));
    CK(hipMemcpy(dct, hct.data(), hct.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dst, hst.data(), hst.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), W * 4, hipMemcpyHostToDevice));
    Args p{M, N, D, W, N, dct, dst, dct, dst, dbias, dout};
    const int tiles_n = W / 128, tiles_m = (M + 63) / 64;
    std::vector<uint16_t> h0((size_t)M * W), h((size_t)M * W);
    // host expectation (scalar, same expression order; fp16 rounding compared with a tolerance of 2 ulp)
    long total_bad = 0, runs_bad = 0, runs_diff = 0, quarter[4] = {0, 0, 0, 0}, elem[4] = {0, 0, 0, 0};
    for (int rep = 0; rep < 30; ++rep) {
        CK(hipMemset(dout, 0x11, (size_t)M * W * 2));
        hipLaunchKernelGGL(k_epi, dim3(tiles_m * tiles_n), dim3(256), 0, 0, p, tiles_n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), dout, (size_t)M * W * 2, hipMemcpyDeviceToHost));
        if (rep == 0) h0 = h;
        else if (memcmp(h0.data(), h.data(), h.size() * 2) != 0) ++runs_diff;
        long bad = 0;
        for (int row = 0; row < M; ++row) {
            const int n = row % N, lcol = row & 31;
            for (int c = 0; c < W; c += 4) {
                const int hi = (c & 7) >> 2, lane = lcol + 32 * hi;
                const int nbk = (c & 63) >> 5, rg = (c & 31) >> 3, j0 = (c & 63) >> 1;
                float a[4], o[4];
                for (int e = 0; e < 4; ++e) a[e] = acc_value(lane, rg * 4 + e, nbk) + hbias[c + e];
                const float c0 = hct[(size_t)j0 * N + n], c1 = hct[(size_t)(j0 + 1) * N + n], s0 = hst[(size_t)j0 * N + n], s1 = hst[(size_t)(j0 + 1) * N + n];
                o[0] = a[0] * c0 - a[1] * s0; o[1] = a[1] * c0 + a[0] * s0; o[2] = a[2] * c1 - a[3] * s1; o[3] = a[3] * c1 + a[2] * s1;
                for (int e = 0; e < 4; ++e) {
                    const float got = (float)__builtin_bit_cast(_Float16, h[(size_t)row * W + c + e]);
                    if (fabsf(got - o[e]) > 4e-3f * (1.0f + fabsf(o[e]))) {
                        ++bad;
                        ++quarter[lane >> 4];
                        ++elem[e];
                    }
                }
            }
        }
        total_bad += bad;
        runs_bad += bad != 0;
    }
    printf("epilogue-only kernel: runs with wrong elements %ld / 30, runs that differ from run 0: %ld, wrong elements %ld, by lane quarter = %ld %ld %ld %ld, by element of the lane's 4 = %ld %ld %ld %ld\n",
           runs_bad, runs_diff, total_bad, quarter[0], quarter[1], quarter[2], quarter[3], elem[0], elem[1], elem[2], elem[3]);
    return 0;
}
