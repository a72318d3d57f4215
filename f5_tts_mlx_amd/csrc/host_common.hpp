// Host-side plumbing shared by the engine (engine.hip) and the vocoder (vocoder.hip): declarations of both kernel builds, the
// operand-type dispatch, the weights-arena bump allocator and the fp32 -> 16-bit upload.
#pragma once
#include <string>
#include <vector>

// the kernel sources are built twice (bf16 operands: namespace f5bf, fp16 operands: namespace f5hf, op16.hpp); this file is
// built once and sees both sets of declarations
#define F5_F16 0
#include "attention.hpp"
#include "convpos.hpp"
#include "gemm.hpp"
#include "rowops.hpp"
#undef F5_F16
#define F5_F16 1
#include "attention.hpp"
#include "convpos.hpp"
#include "gemm.hpp"
#include "rowops.hpp"
#undef F5_F16

// Host-side view: 16-bit operand buffers are opaque storage (typed as the bf16 build's op16_t), argument structs are the
// bf16 build's; the fp16 build's structs have the same layout (only the pointee type of the operand pointers differs).
typedef f5bf::op16_t op16_t;
using f5bf::F5AttnArgs;
using f5bf::F5ConvPosArgs;
using f5bf::F5GemmArgs;
using f5bf::F5OdeArgs;
// launches that do not touch 16-bit operands (or are bf16-only by definition: the MX-fp8 mode) come from the bf16 build
using f5bf::f5_grn_partial_floats;
using f5bf::f5_launch_duration_head;
using f5bf::f5_launch_gemm_f8;
using f5bf::f5_launch_ln_modulate_f8;
using f5bf::f5_launch_quantize_mx;
using f5bf::f5_launch_quantize_mx_bf16;
using f5bf::f5_launch_rope_table;
using f5bf::f5_launch_rope_table_g4;
using f5bf::f5_launch_rowkeep;
using f5bf::f5_launch_stage_words;
using f5bf::f5_launch_copy_words;
using f5bf::f5_launch_skinny_gemm;
using f5bf::f5_launch_splice;
using f5bf::f5_launch_text_embed;
using f5bf::f5_launch_text_pos_table;
using f5bf::f5_launch_time_sinus;

// Kernels of one operand type.  `h` selects the fp16 build.
struct Ops {
    bool h = false;
    static f5hf::op16_t* H(op16_t* p) { return reinterpret_cast<f5hf::op16_t*>(p); }
    int gemm(const F5GemmArgs& a, int epi, hipStream_t s) const {
        return h ? f5hf::f5_launch_gemm(reinterpret_cast<const f5hf::F5GemmArgs&>(a), epi, s) : f5bf::f5_launch_gemm(a, epi, s);
    }
    bool gemm_resid_ln_fusable(const F5GemmArgs& a) const {
        return h ? f5hf::f5_gemm_resid_ln_fusable(reinterpret_cast<const f5hf::F5GemmArgs&>(a)) : f5bf::f5_gemm_resid_ln_fusable(a);
    }
    bool gemm_runs_staged(const F5GemmArgs& a, int epi) const {
        return h ? f5hf::f5_gemm_runs_staged(reinterpret_cast<const f5hf::F5GemmArgs&>(a), epi) : f5bf::f5_gemm_runs_staged(a, epi);
    }
    bool gemm_fold_small(const F5GemmArgs& a, int epi, bool qkv_tr) const {
        return h ? f5hf::f5_gemm_fold_small(reinterpret_cast<const f5hf::F5GemmArgs&>(a), epi, qkv_tr) : f5bf::f5_gemm_fold_small(a, epi, qkv_tr);
    }
    int fold_rows(const float* stats, int ld, int nslice, int M, float eps, float* rowf, float* row_shift, hipStream_t s) const {
        return h ? f5hf::f5_launch_fold_rows(stats, ld, nslice, M, eps, rowf, row_shift, s)
                 : f5bf::f5_launch_fold_rows(stats, ld, nslice, M, eps, rowf, row_shift, s);
    }
    // batch: {count, w_stride, bias_stride, mod_stride, out_blk_stride} (gemm.hpp F5FoldBatch), count == 1 = a single problem
    int fold_consts(const op16_t* w, int ldw, const float* bias, const float* scale, const float* shift, size_t vec_stride, int nvec, float* c1,
                    float* c2, size_t out_stride, int N, int K, hipStream_t s, int count = 1, size_t w_stride = 0, size_t bias_stride = 0,
                    size_t mod_stride = 0, size_t out_blk_stride = 0) const {
        if (h) {
            const f5hf::F5FoldBatch bt = {count, w_stride, bias_stride, mod_stride, out_blk_stride};
            return f5hf::f5_launch_fold_consts(reinterpret_cast<const f5hf::op16_t*>(w), ldw, bias, scale, shift, vec_stride, nvec, c1, c2,
                                               out_stride, N, K, s, &bt);
        }
        const f5bf::F5FoldBatch bt = {count, w_stride, bias_stride, mod_stride, out_blk_stride};
        return f5bf::f5_launch_fold_consts(w, ldw, bias, scale, shift, vec_stride, nvec, c1, c2, out_stride, N, K, s, &bt);
    }
    int attention(const F5AttnArgs& a, hipStream_t s) const {
        return h ? f5hf::f5_launch_attention(reinterpret_cast<const f5hf::F5AttnArgs&>(a), s) : f5bf::f5_launch_attention(a, s);
    }
    int convpos(const F5ConvPosArgs& a, hipStream_t s) const {
        return h ? f5hf::f5_launch_convpos(reinterpret_cast<const f5hf::F5ConvPosArgs&>(a), s) : f5bf::f5_launch_convpos(a, s);
    }
    int ode_stage(const F5OdeArgs& a, hipStream_t s) const {
        return h ? f5hf::f5_launch_ode_stage(reinterpret_cast<const f5hf::F5OdeArgs&>(a), s) : f5bf::f5_launch_ode_stage(a, s);
    }
    int ln_modulate(const float* x, const float* scale, const float* shift, op16_t* hi, op16_t* lo, int rows, int dim, float eps,
                    hipStream_t s, float* mean_out = nullptr) const {
        return h ? f5hf::f5_launch_ln_modulate(x, scale, shift, H(hi), H(lo), rows, dim, eps, s, mean_out)
                 : f5bf::f5_launch_ln_modulate(x, scale, shift, hi, lo, rows, dim, eps, s, mean_out);
    }
    int dwconv_ln(const float* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b, op16_t* hi, op16_t* lo,
                  int nbatch, int seq_len, int dim, float eps, hipStream_t s) const {
        return h ? f5hf::f5_launch_dwconv_ln(x, dw_w, dw_b, ln_w, ln_b, H(hi), H(lo), nbatch, seq_len, dim, eps, s)
                 : f5bf::f5_launch_dwconv_ln(x, dw_w, dw_b, ln_w, ln_b, hi, lo, nbatch, seq_len, dim, eps, s);
    }
    int grn(const float* g, const float* gamma, const float* beta, float* partial, float* nx, op16_t* hi, op16_t* lo, int nbatch,
            int seq_len, int dim, hipStream_t s) const {
        return h ? f5hf::f5_launch_grn(g, gamma, beta, partial, nx, H(hi), H(lo), nbatch, seq_len, dim, s)
                 : f5bf::f5_launch_grn(g, gamma, beta, partial, nx, hi, lo, nbatch, seq_len, dim, s);
    }
    int pack_cond_text(const float* cond, const int* lens, const float* text_emb, op16_t* hi, op16_t* lo, int B, int seq_len,
                       int mel_dim, int dt, int null_keeps_cond, hipStream_t s) const {
        return h ? f5hf::f5_launch_pack_cond_text(cond, lens, text_emb, H(hi), H(lo), B, seq_len, mel_dim, dt, null_keeps_cond, s)
                 : f5bf::f5_launch_pack_cond_text(cond, lens, text_emb, hi, lo, B, seq_len, mel_dim, dt, null_keeps_cond, s);
    }
    int pack_x(const float* y, op16_t* hi, op16_t* lo, int rows, int mel_dim, hipStream_t s) const {
        return h ? f5hf::f5_launch_pack_x(y, H(hi), H(lo), rows, mel_dim, s) : f5bf::f5_launch_pack_x(y, hi, lo, rows, mel_dim, s);
    }
    int layernorm(const float* x, const float* w, const float* b, float* out_f32, op16_t* hi, op16_t* lo, int rows, int dim,
                  float eps, hipStream_t s) const {
        return h ? f5hf::f5_launch_layernorm(x, w, b, out_f32, H(hi), H(lo), rows, dim, eps, s)
                 : f5bf::f5_launch_layernorm(x, w, b, out_f32, hi, lo, rows, dim, eps, s);
    }
    int im2col7(const float* x, op16_t* hi, op16_t* lo, int nbatch, int seq_len, int channels, hipStream_t s) const {
        return h ? f5hf::f5_launch_im2col7(x, H(hi), H(lo), nbatch, seq_len, channels, s)
                 : f5bf::f5_launch_im2col7(x, hi, lo, nbatch, seq_len, channels, s);
    }
    int zero_vt_pad(op16_t* vt, size_t rows, int seq_len, int npad, hipStream_t s) const {
        return h ? f5hf::f5_launch_zero_vt_pad(H(vt), rows, seq_len, npad, s) : f5bf::f5_launch_zero_vt_pad(vt, rows, seq_len, npad, s);
    }
    int pack_bf16(const float* src, const uint8_t* rowkeep, op16_t* hi, op16_t* lo, int rows, int cols, int ld, int col0,
                  hipStream_t s) const {
        return h ? f5hf::f5_launch_pack_bf16(src, rowkeep, H(hi), H(lo), rows, cols, ld, col0, s)
                 : f5bf::f5_launch_pack_bf16(src, rowkeep, hi, lo, rows, cols, ld, col0, s);
    }
};


#define RC(expr)            \
    do {                    \
        int _rc = (expr);   \
        if (_rc) return _rc; \
    } while (0)

// ------------------------------------------------------------------------------------------------
// arena / workspace bump allocator
// ------------------------------------------------------------------------------------------------
struct Bump {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    }
};

struct MatBF {          // bf16 matrix in the arena (hi + optional lo)
    size_t hi = 0, lo = 0;
    int rows = 0, ld = 0;
};

struct TensorDst {      // where (part of) a reference tensor goes
    int kind;           // 0: fp32 copy at off (count elements); 1: bf16 matrix placement
    size_t off = 0;     // kind 0
    MatBF mat;          // kind 1
    int row0 = 0;       // first destination row
    int src_rows = 0, src_cols = 0;
    int c0 = 0, c1 = 0, dst_c0 = 0;  // source column range -> destination column offset
    std::vector<int64_t> shape;
    bool loaded = false;
};

static inline MatBF alloc_mat(Bump& b, int rows, int ld, int np) {
    MatBF m;
    m.rows = rows;
    m.ld = ld;
    const int rows_pad = (rows + 127) / 128 * 128;  // GEMM reads whole 128-row weight tiles
    m.hi = b.take((size_t)rows_pad * ld * 2);
    m.lo = np == 2 ? b.take((size_t)rows_pad * ld * 2) : 0;
    return m;
}


// upload (part of) one reference tensor: fp32 copy, or rows x column range of a matrix rounded to the 16-bit operand type
// (bf16, or fp16 when f16; np == 2 adds the bf16 residual copy of the 3-pass mode)
static inline int f5_upload_tensor(char* arena, const TensorDst& d, const float* host, size_t count, int np, bool f16) {
    if (d.kind == 0) {
        F5_HIP_CHECK(hipMemcpy(arena + d.off, host, count * 4, hipMemcpyHostToDevice));
        return 0;
    }
    const int ncol = d.c1 - d.c0;
    std::vector<u16> hi((size_t)d.src_rows * ncol), lo;
    if (np == 2) lo.resize(hi.size());
    for (int r = 0; r < d.src_rows; ++r)
        for (int cc = 0; cc < ncol; ++cc) {
            const float v = host[(size_t)r * d.src_cols + d.c0 + cc];
            const u16 h = f16 ? f5_f2h_bits(v) : f5_f2bf_bits(v);
            hi[(size_t)r * ncol + cc] = h;
            if (np == 2) lo[(size_t)r * ncol + cc] = f5_f2bf_bits(v - f5_bf_bits2f(h));
        }
    const size_t dst_off = ((size_t)d.row0 * d.mat.ld + d.dst_c0) * 2;
    F5_HIP_CHECK(hipMemcpy2D(arena + d.mat.hi + dst_off, (size_t)d.mat.ld * 2, hi.data(), (size_t)ncol * 2, (size_t)ncol * 2,
                             d.src_rows, hipMemcpyHostToDevice));
    if (np == 2)
        F5_HIP_CHECK(hipMemcpy2D(arena + d.mat.lo + dst_off, (size_t)d.mat.ld * 2, lo.data(), (size_t)ncol * 2, (size_t)ncol * 2,
                                 d.src_rows, hipMemcpyHostToDevice));
    return 0;
}
