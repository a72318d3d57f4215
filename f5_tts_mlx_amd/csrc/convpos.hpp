#include "op16.hpp"

// one body per operand build (this header is included once per F5_F16 value)
#if F5_F16
#ifndef F5_CONVPOS_HPP_F16
#define F5_CONVPOS_HPP_F16
#define F5_CONVPOS_HPP_BODY
#endif
#else
#ifndef F5_CONVPOS_HPP_BF16
#define F5_CONVPOS_HPP_BF16
#define F5_CONVPOS_HPP_BODY
#endif
#endif
#ifdef F5_CONVPOS_HPP_BODY
#undef F5_CONVPOS_HPP_BODY
namespace F5_NS {

struct F5ConvPosArgs {
    const op16_t* in[2];  // hi, lo: [B*seq_len][ld] channels-last
    const op16_t* W[2];   // hi, lo: [C][taps*64]  (reference layout (out, k, in/groups) flattened)
    const float* bias;    // [C]
    int B, seq_len, C, groups, taps, ld;
    int nseg;             // 1 bf16, 3 bf16x3
    int mode;             // 0: out_bf = bf16(mish(.)); 1: out_f32 += mish(.)
    op16_t* out_bf[2];
    float* out_f32;
    int ldo;
    int* sat_flag;        // where mode 0's 16-bit pack reports saturation (fp16 build, op16.hpp); null = rowops.hpp f5_sat_flag_host
};

int f5_launch_convpos(const F5ConvPosArgs& a, hipStream_t stream);
}  // namespace F5_NS
#endif
