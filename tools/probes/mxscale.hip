// probe: which of a lane's 32 operand bytes does ITS E8M0 scale apply to in v_mfma_scale_f32_32x32x64_f8f6f4?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// A: every byte 1.0 (0x38).  scale_a = 2^3 (130) on lane `sl` only, 1.0 elsewhere.  B: 1.0 in register positions
// (half hb, bytes [jb, jb+16)) of every row, 0 elsewhere.  C[m][n] = sum over the non-zero positions of scaleA.
__global__ void k(float* C, int sl, int hb, int jb, int scale_on_b) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0; }
    if ((l >> 5) == hb) for (int i = 0; i < 4; ++i) b[jb / 4 + i] = 0x38383838;
    int sa = 127, sb = 127;
    if (l == sl) { if (scale_on_b) sb = 130; else sa = 130; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void k2(float* C, int opsel_case) {   // opsel check: scale VGPR = bytes {127,128,129,130}; which byte does opsel pick?
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
    const int sa = 127 | (128 << 8) | (129 << 16) | (130 << 24);
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.0f;
    if (opsel_case == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, 127);
    if (opsel_case == 1) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, sa, 0, 127);
    if (opsel_case == 2) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 2, sa, 0, 127);
    if (opsel_case == 3) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 3, sa, 0, 127);
    if (l == 0) C[0] = c[0];
}
int main() {
    float* d; float h[1024];
    hipMalloc(&d, 4096);
    for (int onb = 0; onb < 2; ++onb)
        for (int sl : {5, 37})
            for (int hb = 0; hb < 2; ++hb)
                for (int jb : {0, 16}) {
                    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sl, hb, jb, onb);
                    hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
                    // row 5 (scaled lane's row when scaling A) col 9; when scaling B: row 9, col 5
                    const float v = onb ? h[9 * 32 + 5] : h[5 * 32 + 9];
                    const float other = h[11 * 32 + 13];
                    printf("scale on %s lane %2d (row 5, half %d); B non-zero at half %d bytes %2d-%2d:  C = %5.0f (unscaled rows: %3.0f)\n",
                           onb ? "B" : "A", sl, sl >> 5, hb, jb, jb + 15, v, other);
                }
    for (int o = 0; o < 4; ++o) {
        hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d, o);
        hipMemcpy(h, d, 4, hipMemcpyDeviceToHost);
        printf("opsel %d: C[0][0] = %.0f  (64 x 2^(byte-127): byte0 -> 64, byte1 -> 128, byte2 -> 256, byte3 -> 512)\n", o, h[0]);
    }
    return 0;
}
