#include "op16.hpp"

// one body per operand build (this header is included once per F5_F16 value)
#if F5_F16
#ifndef F5_ATTENTION_HPP_F16
#define F5_ATTENTION_HPP_F16
#define F5_ATTENTION_HPP_BODY
#endif
#else
#ifndef F5_ATTENTION_HPP_BF16
#define F5_ATTENTION_HPP_BF16
#define F5_ATTENTION_HPP_BODY
#endif
#endif
#ifdef F5_ATTENTION_HPP_BODY
#undef F5_ATTENTION_HPP_BODY
namespace F5_NS {

struct F5AttnArgs {
    const op16_t* qk[2];  // hi, lo: [B*seq_len][ldqk]; q at col h*64, k at col dmodel + h*64 (RoPE applied)
    const op16_t* vt[2];  // hi, lo: [B*H][64][npad]  (V transposed, pad columns are zero)
    op16_t* out[2];       // hi, lo: [B*seq_len][ldo], col h*64 + d
    const int* kv_len;    // [B] valid key prefix per batch element, or null (= seq_len)
    int B, H, seq_len, npad, ldqk, ldo, dmodel;
    int hp;               // 0 bf16, 1 bf16x3
    float scale;
    int q_prescaled;      // 1: q was multiplied by scale * log2(e) before rounding (QKV epilogue, F5GemmArgs::q_premul): scores are in exp2 units
    int pipe;             // large grids, q pre-scaled: -1 = the process default (f5_debug_set_attn_pipe), 1 = in-wave software-pipelined kernel (v2p), 0 = v2f
    // MX-fp8 output (bf16 kernels only): e4m3 [B*seq_len][ldo8] + E8M0 [B*seq_len][dmodel/32] (one scale per head half)
    uint8_t* out8;
    uint8_t* out8s;
    int ldo8;
};

int f5_launch_attention(const F5AttnArgs& a, hipStream_t stream);
}  // namespace F5_NS
#endif
