// probe: are 16-byte global stores / loads at 2-byte-aligned addresses legal and correct on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* buf, int off) {
    const int l = threadIdx.x;
    u32x4 v = {0x00010000u + l, 0x00030002u, 0x00050004u, 0x00070006u};
    // each lane writes 8 halfwords starting at halfword index off + l*8  (off odd -> 2-byte aligned only)
    u32x4* p = (u32x4*)(buf + off + l * 8);
    __builtin_nontemporal_store(v, p);
}
__global__ void r(const uint16_t* buf, int off, uint32_t* out) {
    const int l = threadIdx.x;
    const u32x4* p = (const u32x4*)(buf + off + l * 8);
    u32x4 v = __builtin_nontemporal_load(p);
    out[l] = v[0] ^ v[1] ^ v[2] ^ v[3];
}
int main() {
    uint16_t* d; uint32_t* o;
    hipMalloc(&d, 65536); hipMalloc(&o, 1024);
    hipMemset(d, 0xff, 65536);
    for (int off : {0, 1, 3, 5}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, off);
        hipError_t e = hipDeviceSynchronize();
        uint16_t h[600];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
            uint16_t want = (j == 0) ? (uint16_t)l : (j == 1 ? 1 : (uint16_t)(j));
            if (j >= 2) want = j;  // 2..7
            if (h[off + l * 8 + j] != want) ++bad;
        }
        hipLaunchKernelGGL(r, dim3(1), dim3(64), 0, 0, d, off, o);
        hipError_t e2 = hipDeviceSynchronize();
        printf("off=%d store_err=%s load_err=%s mismatches=%d pre=%04x\n", off, hipGetErrorString(e), hipGetErrorString(e2), bad, off ? h[off - 1] : 0);
        hipMemset(d, 0xff, 65536);
    }
    return 0;
}
