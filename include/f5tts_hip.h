/*
 * f5tts_hip.h -- C ABI of the MI355X-native (gfx950) F5-TTS flow-matching sampling engine.
 *
 * The reference (lucasnewman/f5-tts-mlx) has no FFI layer: its boundary for this path is the Python
 * API `F5TTS.sample` (f5_tts_mlx/cfm.py:264-402) -> `DiT.__call__` (f5_tts_mlx/dit.py:374-401) with
 * MLX arrays.  This header is what a Python/ctypes (or cgo/JNI/...) host binds instead; every entry
 * point cites the reference code it replaces.  Conventions:
 *   - plain pointers and sizes only; "dev" pointers are HIP device pointers, "host" pointers are
 *     ordinary host memory; the caller owns every buffer (weights arena, workspace, inputs, outputs)
 *   - every function returns 0 on success, non-zero on error; `f5_last_error()` (thread local)
 *     describes the last failure.  Nothing throws across the ABI.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *     unless stated.  An engine handle is not re-entrant (one handle per device / stream at a time).
 */
#ifndef F5TTS_HIP_H
#define F5TTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct f5_engine f5_engine;

/* Constructor arguments of the reference DiT (dit.py:332-346) + derived sizes. */
typedef struct f5_config {
    int32_t dim;             /* 1024 */
    int32_t depth;           /* 22 */
    int32_t heads;           /* 16 */
    int32_t dim_head;        /* 64 (only 64 is supported by the attention kernel) */
    int32_t ff_dim;          /* dim * ff_mult = 2048 */
    int32_t mel_dim;         /* 100 */
    int32_t text_num_embeds; /* 2545 (embedding table has +1 rows, dit.py:184) */
    int32_t text_dim;        /* 512 */
    int32_t text_ff_dim;     /* text_dim * conv_mult = 1024 */
    int32_t conv_layers;     /* 4; 0 = plain embedding: no positional table, no ConvNeXt blocks, no masking (dit.py:188-194) */
    int32_t conv_pos_kernel; /* 31 */
    int32_t conv_pos_groups; /* 16 (dim / groups must be 64) */
    int32_t freq_embed_dim;  /* 256 */
    int32_t text_max_pos;    /* 4096 */
    int32_t text_mask_padding; /* 1 (dit.py:186): zero the text embedding at filler / padded positions around every text block */
} f5_config;

/* MFMA operand encoding.  The reference computes in fp32 throughout (dit.py, cfm.py); measured mel-L1 drift of a full 32-point
 * sample against the fp32 oracle: BF16 2-4e-3 (outside the 1e-3 gate), F16 (IEEE half operands, same MFMA rate, 11 significand
 * bits, producers saturate at +-65504) inside the gate, BF16X3 (hi/lo split, 3 MFMA passes) ~5e-6; MXFP8: the four block GEMMs on
 * MX-fp8 (e4m3 + E8M0 per 32), rest bf16 -- a reduced-precision mode. */
enum { F5_PREC_BF16 = 0, F5_PREC_BF16X3 = 1, F5_PREC_MXFP8 = 2, F5_PREC_F16 = 3 };
enum { F5_GRAPH_OFF = 0, F5_GRAPH_ON = 1, F5_GRAPH_AUTO = 2 };  /* f5_sample_args.use_graph */
enum { F5_EULER = 0, F5_MIDPOINT = 1, F5_RK4 = 2 };       /* cfm.py:38-122 */

const char* f5_last_error(void);
int f5_version(void);

/* ---- engine lifetime ------------------------------------------------------------------------ */
int f5_engine_create(const f5_config* cfg, int precision, f5_engine** out);
void f5_engine_destroy(f5_engine* e);   /* also destroys every cached hipGraphExec */
/* at most `max_graphs` (default 8) captured sample graphs are kept per engine; the least recently used one is destroyed */
int f5_engine_set_graph_cache(f5_engine* e, int max_graphs);
int f5_engine_graph_count(f5_engine* e);
/* Per-engine launch options (two engines of one process keep their own): "q_premul" (1: q leaves the QKV epilogue multiplied by
 * softmax_scale * log2 e), "qkv_transposed" (1: transposed q / k tiles in the 256x256 QKV kernel), "ln_fusion" (0; 1 = LN-modulate
 * fused behind small-tile residual GEMMs, measured slower), "gemm_flags" (0; F5GemmArgs debug bits of this engine's launches),
 * "attn_pipe" (-1 = the process default of f5_debug_set_attn_pipe, 0 = large-grid attention kernel v2f, 1 = in-wave software-pipelined v2p),
 * "null_keeps_cond" (0; 1 = the second branch of f5_dit_forward / f5_sample keeps the audio conditioning and drops only the text:
 * DiT.__call__(drop_audio_cond=False, drop_text=True), dit.py:374-401), "ln_fold" (LN-modulate folded into the epilogues of the block
 * GEMMs around it, see f5_debug_set_op_fold_producer: -1 (default) = where it is measured faster: >= 22 000 rows (batch >= 12 at the
 * 335M shape and 937 frames) with the four block GEMMs on the 256x256 / role-split 128x256 kernels, f16 / bf16 modes; 0 = never;
 * 1 = wherever it can run (batch >= 4 at that shape), f5_sample fails where it cannot).
 * Part of the hipGraph cache key.  New engines start from the process defaults (f5_debug_set_ln_fusion / _qkv_transposed /
 * _q_premul). */
int f5_engine_set_option(f5_engine* e, const char* name, int value);
int f5_engine_get_option(f5_engine* e, const char* name, int* value);

/* ---- weights: replaces F5TTS.load_weights / from_pretrained upload (cfm.py:475-518) ------------
 * The caller allocates `f5_weights_bytes` of device memory (one contiguous arena, so that a single
 * RCCL broadcast replicates the model), hands it over, then loads every tensor by its reference
 * (MLX-layout) name, e.g. "transformer.transformer_blocks.3.attn.to_q.weight".  host_data is fp32,
 * C-contiguous, shape as in the reference.  f5_finalize_weights checks completeness and builds the
 * derived tables (text positional table, rope.py:63-73). */
int f5_weights_bytes(f5_engine* e, size_t* bytes);
int f5_set_weights_arena(f5_engine* e, void* dev_arena, size_t bytes, void* stream);
int f5_load_tensor(f5_engine* e, const char* name, const float* host_data, int ndim, const int64_t* shape);
int f5_finalize_weights(f5_engine* e, void* stream);
/* A second handle on `owner`'s finalised arena (same f5_config and precision; nothing is copied or written, the caller keeps the arena
 * alive for both).  Handles are not re-entrant, but two handles may run at once: a host that drives two of them from two threads on two
 * streams, each with half of a large batch, fills the partly empty last rounds of each other's launches (-5 % at 32 utterances;
 * f5_tts_mlx_amd/engine.py Engine._sample_split does this, INTEGRATION.md describes it).  Own workspace, hipGraphs and status word per
 * handle. */
int f5_share_weights(f5_engine* e, const f5_engine* owner);
/* Multi-GPU start-up (one process per GPU, utterances sharded, no per-step communication): replicate rank `root`'s arena over an
 * RCCL communicator (`nccl_comm` = ncclComm_t) with ONE ncclBroadcast enqueued on `stream`; ranks other than root are marked
 * loaded and then call f5_finalize_weights.  librccl.so is resolved with dlopen at the first call.  (A torch.distributed host
 * does the same with dist.broadcast on the arena tensor, f5_tts_mlx_amd/dist.py.) */
int f5_broadcast_weights(f5_engine* e, void* nccl_comm, int root, int rank, void* stream);
/* for ranks that received the arena by broadcast instead of f5_load_tensor */
int f5_mark_weights_loaded(f5_engine* e);

/* ---- F5TTS.sample hot path (cfm.py:312-397) ---------------------------------------------------
 * The Python wrapper keeps the reference's host logic (text -> ids, lens / duration clamp, time
 * grid); the engine runs everything from the masks to the final splice on the GPU.             */
typedef struct f5_sample_args {
    int32_t B;                 /* utterances                                                     */
    int32_t N;                 /* max_duration in frames (cfm.py:319); the engine needs N >= 4   */
    int32_t nt;                /* text columns                                                   */
    const int32_t* text;       /* dev  [B][nt] token ids, -1 padded (utils.py:124-133)           */
    const float* cond;         /* dev  [B][N][mel] reference mel, zero padded to N (cfm.py:321)  */
    const int32_t* lens;       /* host [B] conditioning length per utterance (cfm.py:301-303)    */
    const int32_t* durations;  /* host [B] total frames per utterance (cfm.py:317-318)           */
    const float* y0;           /* dev  [B][N][mel] initial noise, zero beyond durations[b]       */
    const float* t;            /* host [steps] time grid (cfm.py:379-381)                        */
    int32_t steps;             /* number of grid POINTS                                          */
    int32_t method;            /* F5_EULER / F5_MIDPOINT / F5_RK4                                */
    float cfg_strength;        /* < 1e-5 disables the null branch (cfm.py:352)                   */
    int32_t use_mask;          /* key-padding mask + output row mask; reference: batch > 1       */
    int32_t use_graph;         /* F5_GRAPH_ON: capture/replay the whole call as one hipGraph (cached per shape signature,
                                * LRU bounded); F5_GRAPH_AUTO: eager on the first sighting of a signature, graph from the
                                * second on (mx.compile re-traces per call shape in the reference, cfm.py:392)       */
    float* out;                /* dev  [B][N][mel] where(cond_mask, cond, y_final)               */
    float* trajectory;         /* dev  [steps][B][N][mel] or NULL                                */
    void* workspace;           /* dev, f5_workspace_bytes                                        */
    size_t workspace_bytes;
} f5_sample_args;

int f5_workspace_bytes(f5_engine* e, int B, int N, int nt, int steps, int method, size_t* bytes);
int f5_sample(f5_engine* e, const f5_sample_args* args, void* stream);

/* Status word of the last f5_sample / f5_dit_forward on the workspace of `args`.  It is the FIRST 32-bit word of the workspace for every
 * shape and solver (a caller may copy it itself).  bit 1 (F5_STATUS_FOLD_RAN) = the LN fold (engine option "ln_fold") ran in that call;
 * bit 0 (F5_STATUS_FOLD_OVERFLOW) = a value of the folded LayerNorm operand (x - m)(1 + scale) did not fit fp16 (precision f16): the
 * output is saturated there -- rerun with ln_fold = 0 or in bf16 (f5_tts_mlx_amd.engine.Engine does that by itself);
 * bit 2 (F5_STATUS_SATURATED, precision f16, engine option "sat_check" = 1, the default) = some OTHER producer of a 16-bit MFMA operand
 * -- LN-modulate, q / k / v behind the rotation, the GELU output of FF1, conv-pos, the packed ODE state, the text path -- clamped a
 * value beyond +-65 504 somewhere in the call (every block of every evaluation is covered, at every batch size): the result is finite
 * and wrong there; this checkpoint / input needs precision bf16x3 (fp32-class) or bf16 (Engine re-runs the call in bf16x3 by itself).
 * f5_sample_status synchronises `stream`; f5_sample_status_async only
 * enqueues the 4-byte device-to-host copy behind the call (`flags` should be pinned host memory, valid once the caller has
 * synchronised `stream` or an event recorded after it): no host block per call.  No reference counterpart (the reference has no
 * reduced-precision operands). */
#define F5_STATUS_FOLD_OVERFLOW 1
#define F5_STATUS_FOLD_RAN 2
#define F5_STATUS_SATURATED 4
int f5_sample_status(f5_engine* e, const f5_sample_args* args, int* flags, void* stream);
int f5_sample_status_async(f5_engine* e, const f5_sample_args* args, int* flags, void* stream);
/* *active = 1 when f5_sample with these arguments runs the LN fold (the only configuration that can set bit 0 of the status word): a
 * host-side question, no GPU work, so a caller only reads the word where it can say something.  For f5_dit_forward pass steps = 2,
 * method = F5_EULER. */
int f5_engine_ln_fold_active(f5_engine* e, const f5_sample_args* args, int* active);

/* One DiT forward (dit.py:374-401) for tests/diagnostics: same inputs as f5_sample, evaluates the
 * velocity field at time `t` for state `x` (dev [B][N][mel]); writes pred (and null when
 * cfg_strength >= 1e-5) as dev [B][N][mel] each. */
int f5_dit_forward(f5_engine* e, const f5_sample_args* args, const float* x, float t, float* pred, float* null_pred,
                   void* stream);

/* ---- per-op entry points (exported for the parity tests; all pointers are dev) ------------------
 * 16-bit operand buffers (a_hi, w_hi, qk_hi, out_hi ...) hold bf16 by default; f5_op_set_operand_type(1) switches every
 * f5_op_* call of the process to IEEE fp16 operands (the kernels are built for both), 0 switches back. */
int f5_op_set_operand_type(int fp16);
int f5_op_get_operand_type(void);   /* current setting: 0 bf16, 1 fp16 (callers that switch temporarily restore this) */
/* C = A * W^T (+epilogue), bf16 MFMA; epi codes in csrc/gemm.hpp (0 = fp32 out + bias, 1 = bf16 out) */
int f5_op_gemm(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias, float* out_f32,
               void* out_bf_hi, void* out_bf_lo, int M, int N, int K, int lda, int ldw, int ldo, int nseg, int epi,
               void* stream);
/* dit.py:136-166: qk [B*n][2*dmodel] (RoPE'd q | k), vt [B*H][64][npad] -> out [B*n][dmodel] */
int f5_op_attention(const void* qk_hi, const void* qk_lo, const void* vt_hi, const void* vt_lo, void* out_hi, void* out_lo,
                    const int32_t* kv_len, int B, int H, int seq_len, int npad, int dmodel, float scale, int hp, void* stream);
/* QKV projection + bias + RoPE + head split (dit.py:136-158); B > 1 needs seq_len >= 4 */
int f5_op_qkv_rope(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                   const float* rope_cos, const float* rope_sin, void* qk_hi, void* qk_lo, void* vt_hi, void* vt_lo, int B,
                   int seq_len, int npad, int heads, int dmodel, int nseg, void* stream);
int f5_op_rope_table(float* cos_t, float* sin_t, int seq_len, int dim_head, void* stream);
/* GROUP-major twins of the rotation tables: tq / tk are [dim_head/4][seq_len][4] floats (16 seq_len dim_head/4 bytes each, 16-byte
 * aligned), element (g, n) = (cos, cos, sin, sin) of rotation pairs 2g, 2g + 1 at position n, the q table multiplied by qscale (1 = plain
 * q): with them the staged QKV kernels accumulate the q / k column tiles transposed (rotation pairs in-lane, one 16-byte load per lane
 * and 4 features); sample() builds its own */
int f5_op_rope_table_g4(float* tq, float* tk, int seq_len, int dim_head, float qscale, void* stream);
/* dit.py:29-50 one grouped conv + Mish; mode 0 -> bf16 out, mode 1 -> out_f32 += */
int f5_op_convpos(const void* in_hi, const void* in_lo, const void* w_hi, const void* w_lo, const float* bias, void* out_hi,
                  void* out_lo, float* out_f32, int B, int seq_len, int C, int groups, int taps, int nseg, int mode,
                  void* stream);
/* dit.py:270 */
int f5_op_ln_modulate(const float* x, const float* scale, const float* shift, void* out_hi, void* out_lo, int rows, int dim,
                      void* stream);
/* MFMA rate yardstick (no reference counterpart; BASELINE.md section 4: "print the measured MFMA micro-benchmark peak you divide by"):
 * blocks x 8 waves x iters x 16 v_mfma_f32_32x32x16 of the current operand type on register operands -- operands NULL = lane-constant
 * registers, else >= 16 x 64 x 8 16-bit values of workload-like data that are rotated through the MFMAs; *flops = flops of the launch */
int f5_op_mfma_peak(const void* operands, int blocks, int iters, float* sink, double* flops, void* stream);
/* convnext_v2.py:46-48 */
int f5_op_dwconv_ln(const float* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b, void* out_hi,
                    void* out_lo, int nbatch, int seq_len, int dim, void* stream);
/* convnext_v2.py:15-18; scratch: f5_op_grn_scratch_floats floats */
size_t f5_op_grn_scratch_floats(int nbatch, int seq_len, int dim);
int f5_op_grn(const float* g, const float* gamma, const float* beta, float* scratch, void* out_hi, void* out_lo, int nbatch,
              int seq_len, int dim, void* stream);
/* dit.py:196-222 (index path is bit exact) */
int f5_op_text_embed(const int32_t* text, int nt, const float* table, const float* pos_table, int max_pos, float* out,
                     int32_t* ids_out, uint8_t* keep_out, int B, int seq_len, int dim, void* stream);
/* same with mask_padding=False (duration.py:116-118): filler positions keep their embedding */
int f5_op_text_embed_nomask(const int32_t* text, int nt, const float* table, const float* pos_table, int max_pos, float* out,
                            int32_t* ids_out, uint8_t* keep_out, int B, int seq_len, int dim, void* stream);
int f5_op_text_pos_table(float* table, int max_pos, int dim, void* stream);
/* out = (resid + A W^T + bias) * keep[row]  (convnext_v2.py:53-54, dit.py:225); keep may be NULL */
int f5_op_gemm_resid_keep(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                          const float* resid, const uint8_t* rowkeep, float* out, int M, int N, int K, int lda, int ldw, int ldo,
                          int nseg, void* stream);
/* fp32 [rows][cols] (rows with keep==0 zeroed) -> bf16 (hi, lo) columns [col0, col0+cols) of a [rows][ld] operand matrix */
int f5_op_pack_bf16(const float* src, const uint8_t* rowkeep, void* out_hi, void* out_lo, int rows, int cols, int ld, int col0,
                    void* stream);
/* duration.py:137,188-190 + utils.py:82-90: RMSNorm -> masked mean over n -> Linear(dim->1) -> Softplus; out [B] seconds */
int f5_op_duration_head(const float* x, const float* g, const float* w, const uint8_t* mask, float* out, int B, int seq_len,
                        int dim, float eps, void* stream);
/* dit.py:61-82, 267 small fp32 GEMM */
int f5_op_time_sinus(const float* t, float* out, int n, int dim, void* stream);
int f5_op_skinny_gemm(const float* a, const float* w, const float* b, float* out, int M, int N, int K, int silu_in,
                      int silu_out, void* stream);
/* cfm.py:56,364: out = base + (coef*dt/divisor) * cfg(pred, null) */
int f5_op_cfg_axpy(const float* pred, const float* null_pred, float cfg, const float* base, const float* dt_dev, float coef,
                   float divisor, float* out, void* xin_hi, void* xin_lo, int rows, int mel_dim, void* stream);

/* Initial noise of F5TTS.sample (cfm.py:369-375) on the device: for each batch element the channel-major draw
 * `mx.random.seed(seed); mx.random.normal((mel, dur))`, zero padded to N frames, laid out [B][N][mel].  The generator follows
 * the published MLX algorithm as restated in f5_tts_mlx_amd/rng.py (threefry2x32 key split and counters: bit-exact integer path;
 * float32 uniform mapping step by step; sqrt(2) * erfinv in float64, one final rounding) -- unverifiable against MLX itself here.
 * seeds: one per element (the reference uses the same seed for all); durations: host; scratch_words: >= 3 * B device words. */
int f5_noise_normal(const uint64_t* seeds, int B, const int32_t* durations, int N, int mel, float* y0, void* scratch_words,
                    void* stream);

/* x += gate[col] * ((A W^T + bias) * keep[row])  (dit.py:319,323; also Vocos' layer-scale + residual) */
int f5_op_gemm_resid_gate(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                          const float* gate, const uint8_t* rowkeep, float* x, int M, int N, int K, int lda, int ldw, int ldx,
                          int nseg, void* stream);
/* the same with the LN-modulate that follows it in a DiT block (dit.py:321 after :319; the next block's dit.py:270 after :323)
 * fused into the launch: h = LN(x_new) * (1 + ln_scale) + ln_shift, bit-identical to f5_op_gemm_resid_gate followed by
 * f5_op_ln_modulate.  Small-tile shapes only (batch-1-sized M; an error otherwise); N = 256 / 512 / 768 / 1024, x is [M][N];
 * counters: >= ceil(M / 64) ints, zero on entry, zero again on exit. */
int f5_op_gemm_resid_gate_ln(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                             const float* gate, const uint8_t* rowkeep, float* x, const float* ln_scale, const float* ln_shift,
                             void* h_hi, void* h_lo, int* counters, int M, int N, int K, int lda, int ldw, int nseg, void* stream);
/* out_f32 = A[row % a_row_mod] W^T + addrows[row]; (out_hi, out_lo) = the same as 16-bit operands: the split input projection of
 * InputEmbedding (dit.py:250), x part per step + hoisted cond / text part, x rows shared by the two CFG branches */
int f5_op_gemm_addrows(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* addrows,
                       int a_row_mod, float* out_f32, void* out_hi, void* out_lo, int M, int N, int K, int lda, int ldw, int ldo,
                       int nseg, void* stream);
/* nn.LayerNorm with affine parameters, eps 1e-6; fp32 and/or bf16 (hi, lo) outputs (any may be NULL) */
int f5_op_layernorm(const float* x, const float* w, const float* b, float* out_f32, void* out_hi, void* out_lo, int rows,
                    int dim, void* stream);
/* im2col for Conv1d(k=7, pad=3), channels-last input with <= 128 channels -> bf16 [rows][7*128] */
int f5_op_im2col7(const float* x, void* out_hi, void* out_lo, int nbatch, int seq_len, int channels, void* stream);
/* Vocos ISTFT head (vocos_mlx Vocos.decode tail, cfm.py:399-400): x [nframes][ldx>=1026] (log-mag | phase) ->
 * wave [hop*(nframes-1)]; frames_scratch [nframes][1024] */
int f5_op_istft(const float* x, int ldx, const float* window, float* frames_scratch, float* wave, int nframes, int n_fft,
                int hop, void* stream);

/* host-side float <-> 16-bit operand conversions used by f5_load_tensor (exported for the CPU tests): fp16 / bf16 round to
 * nearest even; fp16 saturates at +-65504 instead of overflowing to inf */
uint16_t f5_debug_f2h_bits(float f);
float f5_debug_h_bits2f(uint16_t h);
uint16_t f5_debug_f2bf_bits(float f);
/* ---- test hooks of the product library.  Process-wide (set them once, before the calls they should affect); they choose among
 * kernels that sample() itself reaches by shape, so that small test shapes can exercise every shipped tile path.  Hooks that select
 * experiments (superseded kernels, ablations, stream-K, priorities) exist in the lab build only: include/f5tts_hip_lab.h,
 * `F5_LAB=1 bash f5_tts_mlx_amd/csrc/build.sh`; f5_lab_build() tells which library is loaded. */
int f5_lab_build(void);
/* force the GEMM block tile: 0 auto, 1 = 128x128, 2 = 64x128, 3 = 64x64 (register-staged), 4 = 256x256 (gemm256.hip), 5 / 6 = ring
 * 64x128 / 64x64, 8 / 9 = 8-wave ring 128x192 / 128x128, 10 / 11 = in-workgroup split-K 64x128 / 128x128, 12 / 13 = 8-wave ring
 * 128x256 with 64x64 / 32x128 wave tiles (7 = the lab build's 128x256 two-per-CU kernel) */
int f5_debug_set_gemm_tile(int sel);
/* GEMM flag bits, OR-ed into every GEMM launch of the process (an engine's own: f5_engine_set_option "gemm_flags"):
 * bit 0: skip the epilogue of the 256x256 kernel (timing only; results are garbage);
 * bit 1: small-tile kernels use the direct (2-byte store) epilogue instead of the LDS-staged one;
 * bit 8 (256): the small-tile ring kernels load x / bias / gate / keep of the residual update in the epilogue instead of
 *            requesting them before the K loop (A/B of the default; identical bits);
 * bit 12 (4096): the role-split 128x256 kernel numbers its tiles as an XCD-chunked list (M fastest) instead of dealing a 2 x 4
 *            grid of tile blocks to the XCDs (A/B of the default; identical bits);
 * bit 14 (16384): the 256x256 kernel accumulates 16-bit-output tiles (FF1, plain 16-bit, q / k of QKV) in the straight order with
 *            2-byte staging writes instead of transposed with 8-byte ones (A/B; identical bits for FF1 / plain);
 * lab build only: bit 3 (8) residual update by no-return L2 atomics, bits 9-11 x-tile prefetch, bits 4-7 ring-loop ablations */
int f5_debug_set_gemm_flags(int v);
/* ring GEMM tile numbering: 0 auto (band-major one-round launches whose A fits an L2, else m / n fastest), 1 n fastest, 2 m fastest,
 * 3 band-major (bands of 4 column tiles) wherever the tile grid allows */
int f5_debug_set_gemm_order(int v);
/* small-tile GEMM staging: 1 = global_load_lds ring (default), 0 = register-staged double buffer */
int f5_debug_set_gemm_ring(int v);
int f5_debug_set_gemm_qkv_tile(int sel);    /* small-M QKV projection with pair-major tables: 0 = auto (role-split 128x256 tiles when one round of >= 176), 1 = small tiles, 12 / 13 = lock-step 8-wave 128x256 ring, 14 = role-split whenever one round */
int f5_debug_set_gemm_nband(int n);         /* 256x256 GEMM: tiles numbered in bands of n column tiles (0 = n fastest) */
int f5_debug_set_convpos_tps(int taps);     /* conv-pos kernel: weight slabs per pipeline step, 0 = auto (4 for small grids), 1 / 2 / 4 */
int f5_debug_set_convpos_xcd_map(int on);   /* conv-pos kernel: 1 (default) = groups dealt to XCDs, 0 = plain 3-D block numbering */
/* 256-query attention workgroups with two query blocks per wave (one-pass modes, large grids): -1 auto, 0 off, 1 force */
int f5_debug_set_attn_wide(int v);
/* in-workgroup KV split of the attention kernel: -1 auto (by grid size), 1 none, 2 / 4 wave groups */
int f5_debug_set_attn_kvsplit(int v);
/* large-grid attention kernel (q pre-multiplied): 1 = in-wave software-pipelined kernel, one wave per SIMD (v2p), 0 = v2f */
int f5_debug_set_attn_pipe(int v);
/* process DEFAULTS of the per-engine options (f5_engine_set_option); engines that already exist keep their own values */
int f5_debug_set_ln_fusion(int on);         /* 1: LN-modulate fused behind the residual GEMMs of small-M launches (default 0: measured slower) */
int f5_debug_set_qkv_transposed(int on);    /* 1 (default): sample() hands the pair-major rotation tables to the QKV projection (256x256 kernel: transposed q / k tiles) */
int f5_debug_set_q_premul(int on);          /* 1 (default): sample() multiplies q by softmax_scale * log2(e) in the QKV epilogue (single-segment operand modes) */
/* LN-modulate folded into the GEMMs around it (dit.py:270 / :321 between the residual updates :319 / :323 and the projections;
 * csrc/gemm.hpp fold_*; engine option "ln_fold"), for any per-row shift m:
 *   (LN(x)(1 + s) + b) W^T + bias = rstd (((x - m)(1 + s)) W^T) - rstd (mean - m) c1 + c2.
 * Op-level twins: with a producer set, f5_op_gemm_resid_gate also writes (x - row_shift)(1 + next_scale) as [M][N] 16-bit operands and
 * the slice statistics [N / 64][M][2] (slice-major: sum of d = x - row_shift over the slice's 64 columns, sum of squares about the
 * slice mean); row_shift [M] or NULL = 0.  f5_op_fold_rows merges those into the row factors [M][2] = (rstd, rstd * (mean - m)),
 * eps 1e-6; its row_shift (NULL = 0) is m on entry and the row's mean on exit (the next producer's shift).  f5_debug_set_op_ln_mean_out:
 * f5_op_ln_modulate also writes the row means [rows] (the first shift of a forward).  With a consumer set, f5_op_gemm (epi 2) and
 * f5_op_qkv_rope take the operands as A and finish the LN in their epilogues (bias ignored: it is inside c2).  Shapes must run on the
 * 256x256 / role-split 128x256 kernels.  NULLs = off. */
int f5_debug_set_op_fold_producer(const float* next_scale, void* x16_out, float* stats_out, const float* row_shift);
int f5_op_fold_rows(const float* stats, int nslice, int M, float* rowf, float* row_shift, void* stream);
int f5_debug_set_op_ln_mean_out(float* mean_out);
int f5_debug_set_op_fold_consumer(const float* rowf, const float* c1, const float* c2);
/* the statistics form of the consumer (round 6): with rowf = NULL above, f5_op_gemm (epi 2) / f5_op_qkv_rope merge the producer's slice
 * statistics (its stats_out, [16][ld][2]; K = 1024) into their rows' factors themselves -- no f5_op_fold_rows launch in between; shift = the
 * producer's row_shift on ENTRY (or NULL = 0); mean_out (or NULL) receives the rows' means from the workgroups of column tile 0 and must
 * not alias shift.  NULL stats = off. */
int f5_debug_set_op_fold_stats(const float* stats, int ld, const float* shift, float* mean_out);
int f5_debug_set_op_sat_flag(int* flag);   /* device word the 16-bit packers of the f5_op_* launches that follow OR F5_STATUS_SATURATED into (fp16 operand type); NULL = off */
int f5_debug_set_op_fold_overflow_flag(int* flag);   /* device word the producer ORs bit 0 into when (x - m)(1 + s) leaves the fp16 range; NULL = off */
/* c1[v][n] = sum_k W[n][k] (1 + scale_v[k]), c2[v][n] = sum_k W[n][k] shift_v[k] + bias[n] for nvec modulation vectors (vec_stride
 * floats apart; result rows out_stride floats apart); K % 256 == 0, K <= 2048 */
int f5_op_fold_consts(const void* w_hi, int ldw, const float* bias, const float* scale, const float* shift, size_t vec_stride, int nvec,
                      float* c1, float* c2, size_t out_stride, int N, int K, void* stream);
/* op-level twins for f5_op_qkv_rope / f5_op_attention */
int f5_debug_set_op_rope_tables_g4(const float* tq, const float* tk); /* NULLs = off */
int f5_debug_set_op_q_premul(float factor); /* f5_op_qkv_rope scales q by factor, f5_op_attention expects q pre-scaled; 0 = off (default) */

/* ---- MX-fp8 path (BASELINE configs[4]; gfx950 v_mfma_scale_f32_32x32x64_f8f6f4): OCP e4m3 elements, one E8M0 scale per 32
 * consecutive K elements (scale = 2^ceil(log2(amax/448))).  No reference counterpart (the reference's reduced-precision mode
 * is MLX int4/int8 weight quantisation, cfm.py:510-515). ---------------------------------------------------------------- */
/* rows of fp32 [rows][ldx] -> e4m3 [rows][ldq] + E8M0 [rows][cols/32]; cols % 32 == 0 */
int f5_op_quantize_mx(const float* x, int ldx, void* q, int ldq, void* scales, int rows, int cols, void* stream);
/* C = A W^T on MX-fp8 operands (N % 256 == 0, K % 128 == 0).  epi 0: out_f32 = acc + bias; 1: out_bf = bf16(acc + bias);
 * 2: (out8, out8_scales) = MX-fp8(gelu_tanh(acc + bias)); 4: out_f32 += gate * ((acc + bias) * keep[row]) */
int f5_op_gemm_f8(const void* a8, const void* a_scales, const void* w8, const void* w_scales, const float* bias,
                  const float* gate, const uint8_t* rowkeep, float* out_f32, void* out_bf, void* out8, void* out8_scales,
                  int M, int N, int K, int lda8, int ldw8, int ldo, int epi, void* stream);

/* ---- audio front-end (audio.py:115-230) ---------------------------------------------------------- */
/* log-mel spectrogram (audio.py:162-210: zero centre padding, periodic Hann, |rfft|, HTK filterbank, log(max(., 1e-5))) of a
 * batch of equally long waveforms in ONE launch: wave dev [B][L] fp32 -> out dev [B][L / hop][n_mels].  The reference loops
 * over the batch in Python (audio.py:195). */
int f5_mel_spectrogram_batch(const float* wave, int B, int64_t L, const float* window, const float* filterbank, int n_fft, int hop,
                             int n_mels, float* out, void* stream);
/* the same for one waveform: wave dev [L] -> out dev [L / hop][n_mels] */
int f5_mel_spectrogram(const float* wave, int64_t L, const float* window, const float* filterbank, int n_fft, int hop,
                       int n_mels, float* out, void* stream);

/* ---- vocoder: Vocos mel-24khz behind one call (replaces `self._vocoder(out)`, cfm.py:399-400; wiring cfm.py:446,471) ---------
 * The reference delegates to the third-party `vocos_mlx` package (not in its repository); this is the published Vocos
 * architecture (ConvNeXt backbone + ISTFT head) on the same kernels as the DiT.  Same ownership rules as the engine: the caller
 * allocates the weights arena and the workspace; tensors are loaded by their upstream state-dict names
 * ("backbone.embed.weight", "backbone.convnext.0.pwconv1.weight", ..., "head.out.bias"), conv weights in the PyTorch (out, in, k)
 * or the MLX (out, k, in) layout.  PARITY UNPINNED against vocos_mlx (no vector, no checkpoint reachable offline). */
typedef struct f5_vocoder f5_vocoder;
typedef struct f5_vocos_config {
    int32_t n_mels;            /* 100 */
    int32_t dim;               /* 512 */
    int32_t intermediate_dim;  /* 1536 */
    int32_t num_layers;        /* 8 */
    int32_t n_fft;             /* 1024 (only value supported) */
    int32_t hop_length;        /* 256 */
} f5_vocos_config;
int f5_vocoder_create(const f5_vocos_config* cfg, int precision, f5_vocoder** out);   /* F5_PREC_BF16 / _BF16X3 / _F16 */
void f5_vocoder_destroy(f5_vocoder* v);
int f5_vocoder_weights_bytes(f5_vocoder* v, size_t* bytes);
int f5_vocoder_set_weights_arena(f5_vocoder* v, void* dev_arena, size_t bytes, void* stream);
int f5_vocoder_load_tensor(f5_vocoder* v, const char* name, const float* host_data, int ndim, const int64_t* shape);
int f5_vocoder_mark_weights_loaded(f5_vocoder* v);     /* arena content arrived by a broadcast */
int f5_vocoder_finalize(f5_vocoder* v, void* stream);
int f5_vocoder_workspace_bytes(f5_vocoder* v, int B, int N, size_t* bytes);
/* mel dev [B][N][n_mels] fp32 -> wave dev [B][hop * (N - 1)] fp32; use_graph != 0: captured per (B, N, workspace), <= 8 graphs kept */
int f5_vocode(f5_vocoder* v, const float* mel, int B, int N, float* wave, void* workspace, size_t workspace_bytes, int use_graph,
              void* stream);
/* ISTFT head alone (per-op entry for tests): x dev [B * nframes][ldx >= n_fft + 2] (log-magnitude | phase) ->
 * wave dev [B][hop * (nframes - 1)]; frames_scratch dev [B * nframes][n_fft]; two launches for the whole batch */
int f5_op_istft_batch(const float* x, int ldx, const float* window, float* frames_scratch, float* wave, int B, int nframes,
                      int n_fft, int hop, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* F5TTS_HIP_H */
