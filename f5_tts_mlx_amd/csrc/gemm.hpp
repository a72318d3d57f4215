// bf16 MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogues) for gfx950.
#include "op16.hpp"

// one body per operand build (this header is included once per F5_F16 value)
#if F5_F16
#ifndef F5_GEMM_HPP_F16
#define F5_GEMM_HPP_F16
#define F5_GEMM_HPP_BODY
#endif
#else
#ifndef F5_GEMM_HPP_BF16
#define F5_GEMM_HPP_BF16
#define F5_GEMM_HPP_BODY
#endif
#endif
#ifdef F5_GEMM_HPP_BODY
#undef F5_GEMM_HPP_BODY
namespace F5_NS {


struct F5GemmArgs {
    const op16_t* A[2];   // hi, lo   [a_rows][lda]
    const op16_t* W[2];   // hi, lo   [>=ceil128(N)][ldw]  (row = output feature, K contiguous)
    int lda, ldw;
    int M, N, K;          // K % 64 == 0
    int nseg;             // 1 = bf16, 3 = bf16x3 (hi*hi + lo*hi + hi*lo)
    int a_row_mod;        // >0: A row index = row % a_row_mod
    const float* bias;    // [N] or null
    float* out_f32;
    int ldo;
    op16_t* out_bf[2];    // hi, lo (lo may be null)
    int ldob;
    const float* gate;        // [N]
    const uint8_t* rowkeep;   // [M] or null
    const float* addrows;     // [M][ldadd]
    int ldadd;
    const float* resid;       // [M][ldres]
    int ldres;
    const float* rope_cos;    // [seq_len][dim_head/2]
    const float* rope_sin;
    int seq_len, npad, heads, dmodel;
    op16_t* vt[2];            // [B*heads][64][npad]
    float q_premul;           // EPI_QKV_ROPE: != 0 -> the q columns (col < dmodel) are multiplied by it before rounding (softmax scale * log2 e)
    // EPI_QKV_ROPE on the staged kernels: GROUP-major rotation tables [dim_head / 4][seq_len][4] -- for rotation pairs (2g, 2g + 1) of
    // position n the four factors (cos_2g, cos_2g+1, sin_2g, sin_2g+1) are ONE 16-byte element, positions contiguous: a lane that holds
    // token n and 4 consecutive features reads its factors with one global_load_dwordx4, 32 consecutive tokens read 512 contiguous
    // bytes (round 6; the pair-major [dim_head/2][positions] tables before it took four 4-byte loads).  One table for the q columns
    // (already carrying q_premul) and one for the k columns.  Both set = the q / k column tiles are accumulated TRANSPOSED (lane =
    // token: rotation pairs in-lane, 8-byte staging writes; gemm_dev.hpp staged_epilogue_tr_rope); other kernels and the V tiles keep
    // using the token-major tables / the straight tile.
    const float* rope_g4q;
    const float* rope_g4k;
    int debug_flags;          // bit 0: skip the epilogue (timing experiments only)
    int nband;                // 256x256 kernel: > 0 = tiles numbered band-major, bands of nband column tiles (set by the launcher)
    // ---- EPI_RESID_GATE only, small-tile kernels only (f5_gemm_resid_ln_fusable): LN-modulate of the NEXT sub-layer fused
    // behind the residual update.  The workgroup that completes a row block (last of its tiles_n column tiles to finish, found
    // with one agent-scope atomic counter per row block) re-reads the block's rows of x and writes
    // ln_out = LN(x) * (1 + ln_scale) + ln_shift.  Needs N == ldo == the LN width (256 / 512 / 768 / 1024).
    int* ln_counter;          // [>= ceil(M / 64)] ints, zero before the first launch (self re-arming), or null = not fused
    const float* ln_scale;    // [N]
    const float* ln_shift;    // [N]
    op16_t* ln_out[2];        // hi, lo (lo may be null): [M][N]
    float ln_eps;
    // ---- LN-modulate folded into the GEMMs around it (round 4; 256x256 and role-split 128x256 kernels, one-pass operand modes):
    //   (LN(x) (1 + s) + b) W^T + bias  =  rstd (((x - m) (1 + s)) W^T) - rstd (mu - m) c1 + c2,    c1 = W (1 + s),  c2 = W b + bias
    // for ANY per-row shift m (round 5).  LayerNorm removes the row mean mu; the 16-bit operand must not carry it: its rounding error
    // is relative to |x - m|, the exact path's to |x - mu|, so m = the row's mean at the PREVIOUS LayerNorm (x16_shift: written by
    // the LN kernel in front of block 0 and kept up to date by f5_launch_fold_rows) keeps the folded operand as accurate as the
    // unfolded one however large the mean is, as long as one residual update moves it by less than a few sigma.
    // PRODUCER (EPI_RESID_GATE): besides x it writes (x - m)(1 + s) in the 16-bit operand type (the next GEMM's A operand; s = the
    // scale of the LN that follows, x16_scale) and, per row and 64-column slice, (sum d, sum (d - slice mean)^2) of d = x - m.
    // f5_launch_fold_rows merges the slices (Chan's pairwise update: no E[d^2] - E[d]^2 cancellation) into the row factors
    // (rstd, rstd (mu - m)) and stores the new mean.  CONSUMER (EPI_QKV_ROPE with transposed q / k tiles, EPI_GELU_TANH): A = that
    // operand, W unchanged, and the epilogue applies the row factors before everything else; `bias` is ignored (it is inside
    // fold_c2, f5_launch_fold_consts).
    op16_t* x16_out;          // [M][ldx16] or null
    int ldx16;
    const float* x16_scale;   // [N]: s (the kernel adds the 1)
    const float* x16_shift;   // [M]: m, or null = 0
    float* stats_out;         // [N / 64][stats_ld][2] (sum d, centred sum of squares), slice-major, or null; x16_out and stats_out: both or neither
    int stats_ld;             // rows per slice of stats_out (>= M)
    int* x16_overflow;        // or null: bit 0 is set when a value of x (1 + s) does not fit the operand type (fp16 build: |v| > 65 504 or
                              // not finite -- the un-normalised residual stream is the one operand producer without a natural bound)
    int* sat_flag;            // or null: F5_STATUS_SATURATED (4) is ORed in when a 16-bit OUTPUT of this launch (q / k / v, the GELU output, a plain
                              // 16-bit tile) went through the fp16 clamp (op16.hpp f5_sat_commit; f5_launch_gemm fills in rowops.hpp
                              // f5_sat_flag_host when this is null)
    const float* fold_rowf;   // [M][2] (rstd, rstd * (mean - m)) or null = plain GEMM
    // ... or, INSTEAD of fold_rowf (round 6): the producer's slice statistics themselves -- the consumer merges them into its rows' factors
    // before its K loop (the arithmetic of f5_fold_rows_kernel in the same order: the same bits), so no row-factor launch sits between
    // producer and consumer: what makes the fold pay at batch 1, where a launch costs what it removes.  K must be 1024 (16 slices).
    const float* fold_stats;  // [K / 64][fold_stats_ld][2] (sum d, centred sum of squares), as stats_out of the producer; or null
    int fold_stats_ld;
    const float* fold_shift;  // [M]: the shift m the producer subtracted (its x16_shift), or null = 0
    float* fold_mean_out;     // [M]: written with m + mean(d) = the row's mean (the NEXT producer's x16_shift; must not alias fold_shift:
                              // other workgroups still read it), by the workgroups of column tile 0 only; or null
    float fold_eps;
    const float* fold_c1;     // [N], 16-byte aligned
    const float* fold_c2;     // [N], 16-byte aligned
    // ---- MX-fp8 path (f5_launch_gemm_f8): e4m3 operands with one E8M0 scale per 32 consecutive K elements
    const uint8_t* A8;        // [a_rows][lda8] bytes
    const uint8_t* W8;        // [>=ceil256(N)][ldw8] bytes
    const uint8_t* As;        // [a_rows][K/32] E8M0
    const uint8_t* Ws;        // [>=ceil256(N)][K/32] E8M0
    int lda8, ldw8;
    uint8_t* out8;            // EPI_GELU_TANH: fp8 output [M][ldo8] + scales [M][N/32]
    uint8_t* out8s;
    int ldo8;
};

int f5_launch_gemm(const F5GemmArgs& a, int epi, hipStream_t stream);
// true when f5_launch_gemm runs this launch on a kernel with the LDS-staged epilogues (256x256 / role-split 128x256): the only ones
// that implement the x16_out / stats_out / fold_* fields (f5_launch_gemm fails loudly for the others)
bool f5_gemm_runs_staged(const F5GemmArgs& a, int epi);
// true when f5_launch_gemm runs this launch on the batch-1-sized (single-round) kernel that implements the LN fold for its role:
// producer (EPI_RESID_GATE), or consumer in the statistics form (EPI_GELU_TANH, EPI_QKV_ROPE with group-major rotation tables = qkv_tr)
bool f5_gemm_fold_small(const F5GemmArgs& a, int epi, bool qkv_tr);
// row factors of the fold: rowf[r] = (rstd, rstd * (mean - m)) of row r from its nslice slice statistics (stats[slice][ld][2] =
// (sum d, centred sum of squares) of d = x - m; width = 64 nslice).  row_shift (optional, [M]): on entry m (what the producer
// subtracted; null = 0), on exit the row's mean -- the shift of the next folded operand.
int f5_launch_fold_rows(const float* stats, int ld, int nslice, int M, float eps, float* rowf, float* row_shift, hipStream_t stream);
// Constants of the fold for `nvec` modulation vectors at once: c1[v][n] = sum_k W[n][k] (1 + scale_v[k]), c2[v][n] = sum_k W[n][k]
// shift_v[k] + bias[n], fp32 sums over the operand-typed weights the GEMM multiplies by.  scale_v = scale + v * vec_stride (floats),
// likewise shift_v; c1 / c2 rows are out_stride floats apart.  K % 256 == 0, K <= 2048.
// `batch` (optional): count equally shaped problems in ONE launch -- problem i reads w + i * w_stride (elements), bias + i * bias_stride,
// scale / shift + i * mod_stride and writes c1 / c2 + i * out_blk_stride (floats): the same projection of every DiT block.
struct F5FoldBatch {
    int count;
    size_t w_stride, bias_stride, mod_stride, out_blk_stride;
};
int f5_launch_fold_consts(const op16_t* w, int ldw, const float* bias, const float* scale, const float* shift, size_t vec_stride, int nvec,
                          float* c1, float* c2, size_t out_stride, int N, int K, hipStream_t stream, const F5FoldBatch* batch = nullptr);
// the large-shape kernel (gemm256.hip: 256 x 256 tiles, one workgroup per CU); f5_launch_gemm routes N % 256 == 0, >= 512-tile shapes here
int f5_launch_gemm256(const F5GemmArgs& a, int epi, hipStream_t stream);
// batch-1-sized shapes: role-split 128 x 256 tiles, one round of 8-wave workgroups (gemm_rs128.hip)
int f5_launch_gemm_rs128(const F5GemmArgs& a, int epi, hipStream_t stream);
#if defined(F5_LAB) && F5_LAB
int f5_launch_gemm128(const F5GemmArgs& a, int epi, hipStream_t stream);   // gemm128.hip: 128 x 256 tiles, two workgroups per CU (experiment)
extern int f5_gemm128_pad_lds;
#endif
// true when f5_launch_gemm(a, EPI_RESID_GATE) would run a small-tile kernel that implements the fused LN tail (a.ln_* unset or set)
bool f5_gemm_resid_ln_fusable(const F5GemmArgs& a);

#if defined(F5_LAB) && F5_LAB
// stream-K schedule of the lock-step 256x256 kernel (gemm_lab.hip): device scratch for partial tiles; must be initialised outside
// of a stream capture.  f5_gemm_streamk_error() != 0 means a consumer timed out (results invalid).
int f5_gemm_streamk_init();
int f5_gemm_streamk_error();
extern int f5_gemm_streamk;
#endif

// MX-fp8 GEMM (256x256x128 tiles, v_mfma_scale_f32_32x32x64_f8f6f4): M >= 1, N % 256 == 0, K % 128 == 0.
// epi: EPI_F32, EPI_BF16, EPI_GELU_TANH (fp8 + scales out), EPI_RESID_GATE, EPI_QKV_ROPE.
int f5_launch_gemm_f8(const F5GemmArgs& a, int epi, hipStream_t stream);
// rows of fp32 -> e4m3 + E8M0 block scales (scale = 2^ceil(log2(amax/448)), round to nearest even, no saturation needed)
int f5_launch_quantize_mx(const float* x, int ldx, uint8_t* q, int ldq, uint8_t* sc, int rows, int cols, hipStream_t stream);
}  // namespace F5_NS
#endif
