// HBM-bound row / elementwise kernels of the sampling path (declarations).
#include "op16.hpp"

// one body per operand build (this header is included once per F5_F16 value)
#if F5_F16
#ifndef F5_ROWOPS_HPP_F16
#define F5_ROWOPS_HPP_F16
#define F5_ROWOPS_HPP_BODY
#endif
#else
#ifndef F5_ROWOPS_HPP_BF16
#define F5_ROWOPS_HPP_BF16
#define F5_ROWOPS_HPP_BODY
#endif
#endif
#ifdef F5_ROWOPS_HPP_BODY
#undef F5_ROWOPS_HPP_BODY
namespace F5_NS {

extern thread_local int* f5_sat_flag_host;   // rowops.hip (per host thread): where the 16-bit packers of the next launches report saturation (fp16 build), or null

// y = LN(x) * (1 + scale) + shift, LN without affine, eps (dit.py:270,289,321). One wave per row.
// mean_out (optional, [rows]): the row means -- the shift of the first folded operand that follows (gemm.hpp x16_shift)
int f5_launch_ln_modulate(const float* x, const float* scale, const float* shift, op16_t* out_hi, op16_t* out_lo,
                          int rows, int dim, float eps, hipStream_t s, float* mean_out = nullptr);

// ConvNeXtV2 front half (convnext_v2.py:46-48): depthwise Conv1d(k=7,pad=3)+bias -> LayerNorm(affine).
int f5_launch_dwconv_ln(const float* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                        op16_t* out_hi, op16_t* out_lo, int nbatch, int seq_len, int dim, float eps, hipStream_t s);

// GRN (convnext_v2.py:15-18) over the SEQUENCE axis: three deterministic passes.
int f5_launch_grn(const float* g, const float* gamma, const float* beta, float* partial, float* nx, op16_t* out_hi,
                  op16_t* out_lo, int nbatch, int seq_len, int dim, hipStream_t s);
size_t f5_grn_partial_floats(int nbatch, int seq_len, int dim);

// TextEmbedding index path + gather (dit.py:196-222): ids (+1, pad 0, mask, drop) -> emb + pos.
int f5_launch_text_embed(const int* text, int nt, const float* table, const float* pos_table, int max_pos, float* out,
                         int* ids_out, uint8_t* keep_out, int B, int seq_len, int dim, int mask_padding, hipStream_t s);

// pack A operand of the hoisted input projection: [cond(128, zero padded) | text_embed(dt)] for both branches
int f5_launch_pack_cond_text(const float* cond, const int* lens, const float* text_emb, op16_t* out_hi, op16_t* out_lo,
                             int B, int seq_len, int mel_dim, int dt, int null_keeps_cond, hipStream_t s);

// sinusoidal time embedding (dit.py:61-67)
int f5_launch_time_sinus(const float* t, float* out, int n, int dim, hipStream_t s);

// out[m][n] = act_out(sum_k act_in(a[m][k]) * w[n][k] + b[n]) in fp32 for small M (time MLP, adaLN tables)
int f5_launch_skinny_gemm(const float* a, const float* w, const float* b, float* out, int M, int N, int K, int silu_in,
                          int silu_out, hipStream_t s);

// rotary cos/sin table [seq_len][dim_head/2] (rope.py:38-60) and text positional table (rope.py:63-73)
int f5_launch_rope_table(float* cos_t, float* sin_t, int seq_len, int dim_head, hipStream_t s);
int f5_launch_rope_table_g4(float* tq, float* tk, int seq_len, int dim_head, float qscale, hipStream_t s);
int f5_launch_text_pos_table(float* table, int max_pos, int dim, hipStream_t s);

// y (fp32 [rows][mel]) -> bf16 [rows][128] zero padded (A operand of the per-step x projection)
int f5_launch_pack_x(const float* y, op16_t* out_hi, op16_t* out_lo, int rows, int mel_dim, hipStream_t s);

// CFG combine + ODE stage (cfm.py:38-122, :364).  k = pred + (pred - null) * cfg  (pred only when !has_null)
//   mode 0: out = base + a * k                         (optionally k -> kstore)
//   mode 1: out = base + a * (k1 + 2*k2 + 2*k3 + k)    (RK4 final)
// also writes the bf16 padded copy of `out` (next DiT input) when xin_hi != null.
struct F5OdeArgs {
    const float* pred;
    const float* null_pred;  // may be null
    float cfg;               // guidance scale when cfg_ptr is null
    const float* cfg_ptr;    // device scalar (the engine stages it per call: a by-value scalar would be frozen into a hipGraph)
    const float* base;
    const float* dt_ptr;     // device scalar dt of this step
    float coef;
    float divisor;           // a = (coef * dt) / divisor
    int mode;
    float* kstore;
    const float* k1;
    const float* k2;
    const float* k3;
    float* out;
    op16_t* xin_hi;
    op16_t* xin_lo;
    int rows, mel_dim;
    int* sat_flag;           // null = f5_sat_flag_host (op16.hpp f5_sat_commit)
};
int f5_launch_ode_stage(const F5OdeArgs& a, hipStream_t s);

// out = where(n < lens[b], cond, y)  (cfm.py:395-397)
int f5_launch_splice(const float* cond, const float* y, const int* lens, float* out, int B, int seq_len, int mel_dim,
                     hipStream_t s);

// per-call staging as kernels (no memcpy / memset API on the sampling path): host words -> device through kernel arguments,
// device -> device word copy, zeroing of the V^T pad columns
int f5_launch_stage_words(const uint32_t* host_words, size_t nwords, uint32_t* dst, hipStream_t s);
int f5_launch_copy_words(const void* src, void* dst, size_t nwords, hipStream_t s);
int f5_launch_zero_vt_pad(op16_t* vt, size_t rows, int seq_len, int npad, hipStream_t s);
// MFMA rate yardstick: blocks x 512 threads, iters x 16 v_mfma_f32_32x32x16 per wave; operands nullptr = lane-constant registers
int f5_launch_mfma_peak(const op16_t* operands, int blocks, int iters, float* sink, double* flops, hipStream_t s);

// rowkeep[b*seq+n] = n < dur[b]  for 2 branches ([nb][seq])
int f5_launch_rowkeep(const int* dur, uint8_t* keep, int nbatch, int seq_len, hipStream_t s);

// LayerNorm with affine weight/bias over the last axis (dim = 256..1024), fp32 out and/or bf16 (hi, lo) out
int f5_launch_layernorm(const float* x, const float* w, const float* b, float* out_f32, op16_t* out_hi, op16_t* out_lo,
                        int rows, int dim, float eps, hipStream_t s);
// im2col for Conv1d(k=7, pad=3) on channels-last input (c <= 128): out[b*n][7*128] bf16, tap-major, zero padded
int f5_launch_im2col7(const float* x, op16_t* out_hi, op16_t* out_lo, int nbatch, int seq_len, int channels, hipStream_t s);

// duration predictor helpers
int f5_launch_pack_bf16(const float* src, const uint8_t* rowkeep, op16_t* out_hi, op16_t* out_lo, int rows, int cols, int ld,
                        int col0, hipStream_t s);
int f5_launch_duration_head(const float* x, const float* g, const float* w, const uint8_t* mask, float* out, int B, int seq_len,
                            int dim, float eps, hipStream_t s);
// MX-fp8 variants (engine precision mxfp8): LN + modulation straight to e4m3 + E8M0 scales; bf16 rows -> MX-fp8 (weights)
int f5_launch_ln_modulate_f8(const float* x, const float* scale, const float* shift, uint8_t* q, uint8_t* qs, int rows, int dim,
                             float eps, hipStream_t s);
int f5_launch_quantize_mx_bf16(const op16_t* x, int ldx, uint8_t* q, int ldq, uint8_t* sc, int rows, int cols, hipStream_t stream);
}  // namespace F5_NS
#endif
