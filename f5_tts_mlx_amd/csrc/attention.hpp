#pragma once
#include "common.hpp"

struct F5AttnArgs {
    const bf16_t* qk[2];  // hi, lo: [B*seq_len][ldqk]; q at col h*64, k at col dmodel + h*64 (RoPE applied)
    const bf16_t* vt[2];  // hi, lo: [B*H][64][npad]  (V transposed, pad columns are zero)
    bf16_t* out[2];       // hi, lo: [B*seq_len][ldo], col h*64 + d
    const int* kv_len;    // [B] valid key prefix per batch element, or null (= seq_len)
    int B, H, seq_len, npad, ldqk, ldo, dmodel;
    int hp;               // 0 bf16, 1 bf16x3
    float scale;
    // MX-fp8 output (bf16 kernels only): e4m3 [B*seq_len][ldo8] + E8M0 [B*seq_len][dmodel/32] (one scale per head half)
    uint8_t* out8;
    uint8_t* out8s;
    int ldo8;
};

int f5_launch_attention(const F5AttnArgs& a, hipStream_t stream);
