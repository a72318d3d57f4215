// probe: operand layout of v_mfma_(scale_)f32_32x32x64_f8f6f4 with fp8 (e4m3, OCP) operands, scale semantics, fp8 conversion
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// OCP e4m3fn decode on the host
static float e4m3_to_float(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) r = ldexpf((float)m, -9);
    else r = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -r : r;
}

// A [32][64] bytes, B [32(n)][64] bytes (both K-contiguous), hypothesised layout: lane l holds row (l&31), k = 32*(l>>5) + j
template <int SCALED>
__global__ void mm(const uint8_t* A, const uint8_t* B, float* C, int sa, int sb) {
    const int l = threadIdx.x;
    i32x8 a, b;
    const int* ap = reinterpret_cast<const int*>(A + (l & 31) * 64 + 32 * (l >> 5));
    const int* bp = reinterpret_cast<const int*>(B + (l & 31) * 64 + 32 * (l >> 5));
    for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    if (SCALED) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
    // C/D layout of the 32x32 family: col = l&31 (B row index n), row = (r&3) + 8*(r>>2) + 4*(l>>5) (A row index m)
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void cvt(const float* x, uint32_t* out, int n) {
    const int i = threadIdx.x;
    if (i < n / 2) {
        int v = 0;
        v = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], v, false);
        out[i] = (uint32_t)v;
    }
}
int main() {
    uint8_t hA[32 * 64], hB[32 * 64];
    srand(1);
    // small integers and halves, exactly representable: codes for 0, +-0.5, +-1, +-1.5, +-2, +-3, +-4
    const uint8_t codes[] = {0x00, 0x30, 0xB0, 0x38, 0xB8, 0x3C, 0xBC, 0x40, 0xC0, 0x44, 0xC4, 0x48, 0xC8};
    for (int i = 0; i < 32 * 64; ++i) { hA[i] = codes[rand() % 13]; hB[i] = codes[rand() % 13]; }
    float ref[32 * 32];
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            float s = 0;
            for (int k = 0; k < 64; ++k) s += e4m3_to_float(hA[m * 64 + k]) * e4m3_to_float(hB[n * 64 + k]);
            ref[m * 32 + n] = s;
        }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    float hC[1024];
    struct { int scaled, sa, sb; float expect; const char* name; } cases[] = {
        {0, 0, 0, 1.0f, "scale operands 0 (unscaled form)"},
        {1, 127, 127, 1.0f, "E8M0 127 x 127 (1.0)"},
        {1, 128, 127, 2.0f, "E8M0 128 x 127 (2.0)"},
        {1, 126, 125, 0.125f, "E8M0 126 x 125 (1/8)"},
        {1, 0x7F7F807F, 127, -1.0f, "byte 0 = 127, other bytes differ (opsel 0 reads byte 0?)"},
    };
    for (auto& cs : cases) {
        if (cs.scaled) hipLaunchKernelGGL(mm<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, cs.sa, cs.sb);
        else hipLaunchKernelGGL(mm<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, 0, 0);
        hipDeviceSynchronize();
        hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
        double maxerr = 0, ratio = 0; int cnt = 0;
        for (int i = 0; i < 1024; ++i) {
            if (fabsf(ref[i]) > 1.0f) { ratio += hC[i] / ref[i]; ++cnt; }
            const float want = ref[i] * (cs.expect > 0 ? cs.expect : 1.0f);
            maxerr = fmax(maxerr, fabs(hC[i] - want));
        }
        printf("%-60s  mean C/ref = %.4f   max |C - expect*ref| = %.4g\n", cs.name, ratio / cnt, maxerr);
    }
    // conversion
    float hx[16] = {0.0f, 1.0f, -1.0f, 0.5f, 448.0f, 449.0f, 500.0f, 1e9f, 0.001953125f, 0.0009765625f, 0.0146f, 0.0156f, 3.3f, -3.7f, 240.0f, INFINITY};
    float* dx; uint32_t* dout; uint32_t hout[8];
    hipMalloc(&dx, 64); hipMalloc(&dout, 32);
    hipMemcpy(dx, hx, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, dx, dout, 16);
    hipMemcpy(hout, dout, 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) {
        const uint8_t b = (hout[i / 2] >> (8 * (i & 1))) & 0xFF;
        printf("cvt_pk_fp8_f32(%g) = 0x%02X -> %g\n", hx[i], b, e4m3_to_float(b));
    }
    return 0;
}
