#pragma once
#include "common.hpp"

struct F5ConvPosArgs {
    const bf16_t* in[2];  // hi, lo: [B*seq_len][ld] channels-last
    const bf16_t* W[2];   // hi, lo: [C][taps*64]  (reference layout (out, k, in/groups) flattened)
    const float* bias;    // [C]
    int B, seq_len, C, groups, taps, ld;
    int nseg;             // 1 bf16, 3 bf16x3
    int mode;             // 0: out_bf = bf16(mish(.)); 1: out_f32 += mish(.)
    bf16_t* out_bf[2];
    float* out_f32;
    int ldo;
};

int f5_launch_convpos(const F5ConvPosArgs& a, hipStream_t stream);
