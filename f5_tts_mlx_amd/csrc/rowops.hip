// HBM-bound row / elementwise kernels of the F5-TTS sampling path (gfx950, wave64).
// Everything here is fp32 arithmetic; bf16 only appears as the (hi, lo) operand encoding handed to
// the MFMA kernels.  Loads/stores are 16 B per lane wherever the layout allows it.
#include "rowops.hpp"
#include "lnrow.hpp"

namespace F5_NS {

// The status word the 16-bit packers of the NEXT launches report saturation to (op16.hpp f5_sat_commit), or null.  Host-side, set by
// the engine around a call (engine.hip SatScope) and by f5_debug_set_op_sat_flag for the op-level tests; the launchers of this file,
// of convpos.hip and f5_launch_gemm pass it to their kernels by value (a captured graph keeps the pointer it was captured with: the
// status word of its workspace).  Per host THREAD: two engine handles may be driven from two threads at once (two half batches on two
// streams, engine.py Engine._sample_split), each call with its own status word.
thread_local int* f5_sat_flag_host = nullptr;


// =================================================================================================
// LayerNorm (no affine) + adaLN modulation: one wave per row, NV float4 per lane (dim = NV*256)
// =================================================================================================
// WITH_MEAN: also writes the row means (LN fold: the shift of the first folded operand of a forward, gemm.hpp x16_shift).  A separate
// instantiation, so that the plain kernel keeps the code it shares bit for bit with the LN tail fused into gemm.hip (lnrow.hpp).
template <int NV, bool WITH_MEAN>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, op16_t* __restrict__ out_hi,
                                                          op16_t* __restrict__ out_lo, int rows, float eps, float* __restrict__ mean_out,
                                                          int* sat_flag) {
    constexpr int DIM = NV * 256;
    asm volatile("" ::"s"(x), "s"(scale), "s"(shift), "s"(out_hi), "s"(out_lo), "s"(rows), "s"(eps), "s"(mean_out), "s"(sat_flag));    // one scalar-load clause
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * DIM;
    f32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f32x4*>(xr + i * 256 + lane * 4);
    if (WITH_MEAN) {
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i) sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        const float mean = f5_wave_sum(sum) * (1.0f / DIM);
        if (lane == 0) mean_out[row] = mean;
    }
    f5_ln_modulate_row<NV>(v, scale, shift, out_hi, out_lo, (size_t)row, lane, eps, sat_flag);
}

int f5_launch_ln_modulate(const float* x, const float* scale, const float* shift, op16_t* out_hi, op16_t* out_lo,
                          int rows, int dim, float eps, hipStream_t s, float* mean_out) {
    F5_REQUIRE(dim % 256 == 0 && dim >= 256 && dim <= 1024, "ln_modulate: dim must be 256/512/768/1024 (got %d)", dim);
    const dim3 grid(f5_cdiv(rows, 4)), block(256);
#define LNM_LAUNCH(NV_)                                                                                                               \
    if (mean_out != nullptr)                                                                                                          \
        hipLaunchKernelGGL((ln_modulate_kernel<NV_, true>), grid, block, 0, s, x, scale, shift, out_hi, out_lo, rows, eps, mean_out, f5_sat_flag_host); \
    else                                                                                                                              \
        hipLaunchKernelGGL((ln_modulate_kernel<NV_, false>), grid, block, 0, s, x, scale, shift, out_hi, out_lo, rows, eps, mean_out, f5_sat_flag_host);
    switch (dim / 256) {
        case 1: LNM_LAUNCH(1); break;
        case 2: LNM_LAUNCH(2); break;
        case 3: LNM_LAUNCH(3); break;
        default: LNM_LAUNCH(4); break;
    }
#undef LNM_LAUNCH
    F5_LAUNCH_CHECK();
    return 0;
}

// MX-fp8 variant: same math, output as e4m3 + one E8M0 scale per 32 columns (8 lanes x 4 columns = one block)
template <int NV>
__global__ __launch_bounds__(256) void ln_modulate_f8_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, uint8_t* __restrict__ q,
                                                             uint8_t* __restrict__ qs, int rows, float eps) {
    constexpr int DIM = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * DIM;
    f32x4 v[NV];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(xr + i * 256 + lane * 4);
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = f5_wave_sum(sum) * (1.0f / DIM);
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            sq += d * d;
        }
    const float var = f5_wave_sum(sq) * (1.0f / DIM);
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * (1.0f + sc[e]) + sh[e];
        float am = fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3])));
        am = fmaxf(am, __shfl_xor(am, 1, 64));
        am = fmaxf(am, __shfl_xor(am, 2, 64));
        am = fmaxf(am, __shfl_xor(am, 4, 64));
        const int e8 = f5_mx_scale_byte(am);
        const float inv = f5_mx_inv_scale(e8);
        *reinterpret_cast<uint32_t*>(q + (size_t)row * DIM + c) = f5_pack4_fp8(y[0] * inv, y[1] * inv, y[2] * inv, y[3] * inv);
        if ((lane & 7) == 0) qs[(size_t)row * (DIM / 32) + (c >> 5)] = (uint8_t)e8;
    }
}
int f5_launch_ln_modulate_f8(const float* x, const float* scale, const float* shift, uint8_t* q, uint8_t* qs, int rows, int dim,
                             float eps, hipStream_t s) {
    F5_REQUIRE(dim % 256 == 0 && dim >= 256 && dim <= 1024, "ln_modulate_f8: dim must be 256/512/768/1024 (got %d)", dim);
    const dim3 grid(f5_cdiv(rows, 4)), block(256);
    switch (dim / 256) {
        case 1: hipLaunchKernelGGL((ln_modulate_f8_kernel<1>), grid, block, 0, s, x, scale, shift, q, qs, rows, eps); break;
        case 2: hipLaunchKernelGGL((ln_modulate_f8_kernel<2>), grid, block, 0, s, x, scale, shift, q, qs, rows, eps); break;
        case 3: hipLaunchKernelGGL((ln_modulate_f8_kernel<3>), grid, block, 0, s, x, scale, shift, q, qs, rows, eps); break;
        default: hipLaunchKernelGGL((ln_modulate_f8_kernel<4>), grid, block, 0, s, x, scale, shift, q, qs, rows, eps); break;
    }
    F5_LAUNCH_CHECK();
    return 0;
}
// bf16 rows -> MX-fp8 (weights at finalize time; the bf16 copy is the source so both precisions see the same rounding)
__global__ __launch_bounds__(256) void quantize_mx_bf16_kernel(const op16_t* __restrict__ x, int ldx, uint8_t* __restrict__ q, int ldq,
                                                               uint8_t* __restrict__ sc, int rows, int cols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    for (int c0 = lane * 4; c0 < cols; c0 += 256) {
        const op16x4 raw = *reinterpret_cast<const op16x4*>(x + (size_t)row * ldx + c0);
        const float v0 = (float)raw[0], v1 = (float)raw[1], v2 = (float)raw[2], v3 = (float)raw[3];
        float am = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3)));
        am = fmaxf(am, __shfl_xor(am, 1, 64));
        am = fmaxf(am, __shfl_xor(am, 2, 64));
        am = fmaxf(am, __shfl_xor(am, 4, 64));
        const int e8 = f5_mx_scale_byte(am);
        const float inv = f5_mx_inv_scale(e8);
        *reinterpret_cast<uint32_t*>(q + (size_t)row * ldq + c0) = f5_pack4_fp8(v0 * inv, v1 * inv, v2 * inv, v3 * inv);
        if ((lane & 7) == 0) sc[(size_t)row * (cols >> 5) + (c0 >> 5)] = (uint8_t)e8;
    }
}
int f5_launch_quantize_mx_bf16(const op16_t* x, int ldx, uint8_t* q, int ldq, uint8_t* sc, int rows, int cols, hipStream_t stream) {
    F5_REQUIRE(rows > 0 && cols > 0 && cols % 32 == 0 && ldx % 4 == 0 && ldq % 4 == 0, "quantize_mx_bf16: cols must be a multiple of 32");
    hipLaunchKernelGGL(quantize_mx_bf16_kernel, dim3(f5_cdiv(rows, 4)), dim3(256), 0, stream, x, ldx, q, ldq, sc, rows, cols);
    F5_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// depthwise conv (k=7, pad=3, per batch element zero padding) + bias + LayerNorm(affine)
// one wave per token; lane owns dim/64 channels as float4 chunks (dim = NV*256)
// =================================================================================================
template <int NV>
__global__ __launch_bounds__(256) void dwconv_ln_kernel(const float* __restrict__ x, const float* __restrict__ dw_w,
                                                        const float* __restrict__ dw_b, const float* __restrict__ ln_w,
                                                        const float* __restrict__ ln_b, op16_t* __restrict__ out_hi,
                                                        op16_t* __restrict__ out_lo, int rows, int seq_len, float eps, int* sat_flag) {
    constexpr int DIM = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / seq_len, n = row - b * seq_len;
    f5_sat_t trk;
    float y[NV][4];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[i][e] = dw_b[c + e];
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int nn = n + t - 3;
            if (nn >= 0 && nn < seq_len) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(x + ((size_t)b * seq_len + nn) * DIM + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[i][e] += xv[e] * dw_w[(c + e) * 7 + t];
            }
        }
        sum += (y[i][0] + y[i][1]) + (y[i][2] + y[i][3]);
    }
    const float mean = f5_wave_sum(sum) * (1.0f / DIM);
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = y[i][e] - mean;
            sq += d * d;
        }
    const float rstd = rsqrtf(f5_wave_sum(sq) * (1.0f / DIM) + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        float z[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = (y[i][e] - mean) * rstd * ln_w[c + e] + ln_b[c + e];
        *reinterpret_cast<u32x2*>(out_hi + (size_t)row * DIM + c) = u32x2{f5_pack2(z[0], z[1], trk), f5_pack2(z[2], z[3], trk)};
        if (out_lo)
            *reinterpret_cast<u32x2*>(out_lo + (size_t)row * DIM + c) =
                u32x2{f5_pack2_lo(z[0], z[1]), f5_pack2_lo(z[2], z[3])};
    }
    f5_sat_commit(trk, sat_flag);
}

int f5_launch_dwconv_ln(const float* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                        op16_t* out_hi, op16_t* out_lo, int nbatch, int seq_len, int dim, float eps, hipStream_t s) {
    F5_REQUIRE(dim % 256 == 0 && dim >= 256 && dim <= 1024, "dwconv_ln: dim must be 256/512/768/1024 (got %d)", dim);
    const int rows = nbatch * seq_len;
    const dim3 grid(f5_cdiv(rows, 4)), block(256);
#define DW_ARGS x, dw_w, dw_b, ln_w, ln_b, out_hi, out_lo, rows, seq_len, eps, f5_sat_flag_host
    switch (dim / 256) {
        case 1: hipLaunchKernelGGL((dwconv_ln_kernel<1>), grid, block, 0, s, DW_ARGS); break;
        case 2: hipLaunchKernelGGL((dwconv_ln_kernel<2>), grid, block, 0, s, DW_ARGS); break;
        case 3: hipLaunchKernelGGL((dwconv_ln_kernel<3>), grid, block, 0, s, DW_ARGS); break;
        default: hipLaunchKernelGGL((dwconv_ln_kernel<4>), grid, block, 0, s, DW_ARGS); break;
    }
#undef DW_ARGS
    F5_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// GRN: Gx[b][c] = ||g[b,:,c]||_2 over the sequence; Nx = Gx / (mean_c Gx + 1e-6);
//      out = gamma * (g * Nx) + beta + g.   Deterministic: per-chunk partial sums, fixed-order finish.
// =================================================================================================
#define GRN_CHUNK 32
size_t f5_grn_partial_floats(int nbatch, int seq_len, int dim) {
    return (size_t)nbatch * f5_cdiv(seq_len, GRN_CHUNK) * dim;
}

__global__ __launch_bounds__(256) void grn_partial_kernel(const float* __restrict__ g, float* __restrict__ partial,
                                                          int seq_len, int dim, int nchunk) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int ch = blockIdx.y, b = blockIdx.z;
    if (c >= dim) return;
    const int n0 = ch * GRN_CHUNK;
    const int n1 = min(n0 + GRN_CHUNK, seq_len);
    float acc = 0.0f;
    for (int n = n0; n < n1; ++n) {
        const float v = g[((size_t)b * seq_len + n) * dim + c];
        acc += v * v;
    }
    partial[((size_t)b * nchunk + ch) * dim + c] = acc;
}

// one block per batch element: Gx (fixed summation order) -> mean over channels -> Nx
__global__ __launch_bounds__(256) void grn_finish_kernel(const float* __restrict__ partial, float* __restrict__ nx, int dim,
                                                         int nchunk) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float local = 0.0f;
    for (int c = threadIdx.x; c < dim; c += 256) {
        float ss = 0.0f;
        for (int ch = 0; ch < nchunk; ++ch) ss += partial[((size_t)b * nchunk + ch) * dim + c];
        const float gx = sqrtf(ss);
        nx[(size_t)b * dim + c] = gx;
        local += gx;
    }
    local = f5_wave_sum(local);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)dim;
    const float inv = 1.0f / (mean + 1e-6f);
    for (int c = threadIdx.x; c < dim; c += 256) nx[(size_t)b * dim + c] *= inv;
}

__global__ __launch_bounds__(256) void grn_apply_kernel(const float* __restrict__ g, const float* __restrict__ nx,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        op16_t* __restrict__ out_hi, op16_t* __restrict__ out_lo, int seq_len,
                                                        int dim, size_t total4, int* sat_flag) {
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= total4) return;
    const size_t idx = i4 * 4;
    const int c = (int)(idx % dim);
    const size_t row = idx / dim;
    const int b = (int)(row / seq_len);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + idx);
    const f32x4 nv = *reinterpret_cast<const f32x4*>(nx + (size_t)b * dim + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c);
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = ga[e] * (gv[e] * nv[e]) + be[e] + gv[e];
    f5_sat_t trk;
    *reinterpret_cast<u32x2*>(out_hi + idx) = u32x2{f5_pack2(y[0], y[1], trk), f5_pack2(y[2], y[3], trk)};
    if (out_lo) *reinterpret_cast<u32x2*>(out_lo + idx) = u32x2{f5_pack2_lo(y[0], y[1]), f5_pack2_lo(y[2], y[3])};
    f5_sat_commit(trk, sat_flag);
}

int f5_launch_grn(const float* g, const float* gamma, const float* beta, float* partial, float* nx, op16_t* out_hi,
                  op16_t* out_lo, int nbatch, int seq_len, int dim, hipStream_t s) {
    F5_REQUIRE(dim % 4 == 0, "grn: dim must be a multiple of 4");
    const int nchunk = f5_cdiv(seq_len, GRN_CHUNK);
    hipLaunchKernelGGL(grn_partial_kernel, dim3(f5_cdiv(dim, 256), nchunk, nbatch), dim3(256), 0, s, g, partial, seq_len, dim,
                       nchunk);
    hipLaunchKernelGGL(grn_finish_kernel, dim3(nbatch), dim3(256), 0, s, partial, nx, dim, nchunk);
    const size_t total4 = (size_t)nbatch * seq_len * dim / 4;
    hipLaunchKernelGGL(grn_apply_kernel, dim3(f5_cdiv((long)total4, 256)), dim3(256), 0, s, g, nx, gamma, beta, out_hi, out_lo,
                       seq_len, dim, total4, f5_sat_flag_host);
    F5_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// TextEmbedding index path + gather.  out layout [2 branches][B][seq][dim]; ids_out [2][B][seq];
// keep_out [2][B][seq] (1 where the ORIGINAL id != 0, i.e. not filler; same for both branches) -- bit exact.
// =================================================================================================
__global__ __launch_bounds__(128) void text_embed_kernel(const int* __restrict__ text, int nt, const float* __restrict__ table,
                                                         const float* __restrict__ pos_table, int max_pos,
                                                         float* __restrict__ out, int* __restrict__ ids_out,
                                                         uint8_t* __restrict__ keep_out, int B, int seq_len, int dim,
                                                         int mask_padding) {
    const int n = blockIdx.x, b = blockIdx.y, br = blockIdx.z;
    int id = 0;
    if (n < nt) id = text[(size_t)b * nt + n] + 1;   // text + 1, curtailed to seq_len, right-padded with 0
    const bool keep = (id != 0) || !mask_padding;     // text_mask = (text == 0), taken BEFORE the drop; none when mask_padding=False
    const int emb_id = (br == 1) ? 0 : id;            // drop_text -> all-zero ids
    const int pos = n < max_pos ? n : max_pos - 1;
    if (threadIdx.x == 0) {
        ids_out[((size_t)br * B + b) * seq_len + n] = emb_id;
        keep_out[((size_t)br * B + b) * seq_len + n] = keep ? 1 : 0;
    }
    float* o = out + (((size_t)br * B + b) * seq_len + n) * dim;
    for (int c = threadIdx.x * 4; c < dim; c += 128 * 4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (keep) {
            v = *reinterpret_cast<const f32x4*>(table + (size_t)emb_id * dim + c);
            if (pos_table) v = v + *reinterpret_cast<const f32x4*>(pos_table + (size_t)pos * dim + c);   // null: conv_layers == 0
        }
        *reinterpret_cast<f32x4*>(o + c) = v;
    }
}

int f5_launch_text_embed(const int* text, int nt, const float* table, const float* pos_table, int max_pos, float* out,
                         int* ids_out, uint8_t* keep_out, int B, int seq_len, int dim, int mask_padding, hipStream_t s) {
    F5_REQUIRE(dim % 4 == 0, "text_embed: dim must be a multiple of 4");
    hipLaunchKernelGGL(text_embed_kernel, dim3(seq_len, B, 2), dim3(128), 0, s, text, nt, table, pos_table, max_pos, out,
                       ids_out, keep_out, B, seq_len, dim, mask_padding);
    F5_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// A operand of the hoisted input projection: rows [2][B*seq], cols [cond padded to 128 | text dt]
// branch 0: cond masked by n < lens[b] (step_cond, cfm.py:331); branch 1: cond dropped (dit.py:249) unless null_keeps_cond
// =================================================================================================
__global__ __launch_bounds__(256) void pack_cond_text_kernel(const float* __restrict__ cond, const int* __restrict__ lens,
                                                             const float* __restrict__ text_emb, op16_t* __restrict__ out_hi,
                                                             op16_t* __restrict__ out_lo, int B, int seq_len, int mel_dim,
                                                             int dt, int null_keeps_cond, int* sat_flag) {
    const int n = blockIdx.x, b = blockIdx.y, br = blockIdx.z;
    const int ld = 128 + dt;
    const size_t orow = (((size_t)br * B + b) * seq_len + n) * ld;
    // null_keeps_cond: the second branch is (drop_audio_cond = False, drop_text = True) instead of (True, True), dit.py:245-247 / 209-210
    const bool use_cond = (br == 0 || null_keeps_cond) && (n < lens[b]);
    f5_sat_t trk;
    for (int c = threadIdx.x; c < ld; c += 256) {
        float v = 0.0f;
        if (c < 128) {
            if (use_cond && c < mel_dim) v = cond[((size_t)b * seq_len + n) * mel_dim + c];
        } else {
            v = text_emb[(((size_t)br * B + b) * seq_len + n) * dt + (c - 128)];
        }
        op16_t h, l;
        f5_split(v, h, l, trk);
        out_hi[orow + c] = h;
        if (out_lo) out_lo[orow + c] = l;
    }
    f5_sat_commit(trk, sat_flag);
}

int f5_launch_pack_cond_text(const float* cond, const int* lens, const float* text_emb, op16_t* out_hi, op16_t* out_lo,
                             int B, int seq_len, int mel_dim, int dt, int null_keeps_cond, hipStream_t s) {
    F5_REQUIRE(mel_dim <= 128, "pack_cond_text: mel_dim must be <= 128");
    hipLaunchKernelGGL(pack_cond_text_kernel, dim3(seq_len, B, 2), dim3(256), 0, s, cond, lens, text_emb, out_hi, out_lo, B,
                       seq_len, mel_dim, dt, null_keeps_cond, f5_sat_flag_host);
    F5_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// time embedding pieces
// =================================================================================================
__global__ void time_sinus_kernel(const float* __restrict__ t, float* __restrict__ out, int n, int dim) {
    const int i = blockIdx.x;
    const int half = dim / 2;
    const float step = logf(10000.0f) / (float)(half - 1);
    for (int j = threadIdx.x; j < half; j += blockDim.x) {
        const float e = expf((float)j * -step);
        const float a = (1000.0f * t[i]) * e;
        out[(size_t)i * dim + j] = sinf(a);
        out[(size_t)i * dim + half + j] = cosf(a);
    }
}
int f5_launch_time_sinus(const float* t, float* out, int n, int dim, hipStream_t s) {
    hipLaunchKernelGGL(time_sinus_kernel, dim3(n), dim3(128), 0, s, t, out, n, dim);
    F5_LAUNCH_CHECK();
    return 0;
}

// out[m][n] = act_out(sum_k act_in(a[m][k]) * w[n][k] + b[n]) for M <= 128 rows in exact fp32 on the matrix cores:
// v_mfma_f32_32x32x2_f32 (fp32 in / fp32 accumulate, bitwise an fmaf chain).  One wave owns 32 output columns and all
// rows (ceil(M/32) accumulator blocks); every lane streams 16 B of its W row per step, i.e. 8 k-values per 4 MFMAs:
// MFMA e pairs k = k0 + e (lanes 0-31) with k = k0 + 4 + e (lanes 32-63) on both operands.  W (the 562 MB of adaLN
// weights) is read exactly once; the small A operand stays in L2.
typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <int MBLK>
__global__ __launch_bounds__(256) void skinny_gemm_mfma_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ out, int M,
                                                               int N, int K, int silu_in, int silu_out) {
    const int lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    if (n0 >= N) return;
    const int hi = lane >> 5, lj = lane & 31;
    int wn = n0 + lj;
    if (wn > N - 1) wn = N - 1;
    const float* wrow = w + (size_t)wn * K + hi * 4;
    const float* arow[MBLK];
#pragma unroll
    for (int mb = 0; mb < MBLK; ++mb) {
        int m = mb * 32 + lj;
        if (m > M - 1) m = M - 1;
        arow[mb] = a + (size_t)m * K + hi * 4;
    }
    f32x16_t acc[MBLK];
#pragma unroll
    for (int mb = 0; mb < MBLK; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mb][e] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 8) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + k0);
#pragma unroll
        for (int mb = 0; mb < MBLK; ++mb) {
            f32x4 av = *reinterpret_cast<const f32x4*>(arow[mb] + k0);
            if (silu_in) {
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = f5_silu(av[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], wv[e], acc[mb], 0, 0, 0);
        }
    }
    const int col = n0 + lj;
    if (col < N) {
        const float bn = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int mb = 0; mb < MBLK; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < M) {
                    float v = acc[mb][r] + bn;
                    if (silu_out) v = f5_silu(v);
                    out[(size_t)m * N + col] = v;
                }
            }
    }
}
int f5_launch_skinny_gemm(const float* a, const float* w, const float* b, float* out, int M, int N, int K, int silu_in,
                          int silu_out, hipStream_t s) {
    F5_REQUIRE(K % 8 == 0, "skinny_gemm: K must be a multiple of 8 (got %d)", K);
    F5_REQUIRE(M >= 1 && N >= 1, "skinny_gemm: empty problem (M=%d N=%d)", M, N);
    const dim3 grid(f5_cdiv(f5_cdiv(N, 32), 4)), block(256);
    for (int m0 = 0; m0 < M; m0 += 128) {       // 128 rows (4 accumulator blocks) per pass over W
        const int mc = M - m0 < 128 ? M - m0 : 128;
        const float* ap = a + (size_t)m0 * K;
        float* op = out + (size_t)m0 * N;
        switch ((mc + 31) / 32) {
            case 1: hipLaunchKernelGGL((skinny_gemm_mfma_kernel<1>), grid, block, 0, s, ap, w, b, op, mc, N, K, silu_in, silu_out); break;
            case 2: hipLaunchKernelGGL((skinny_gemm_mfma_kernel<2>), grid, block, 0, s, ap, w, b, op, mc, N, K, silu_in, silu_out); break;
            case 3: hipLaunchKernelGGL((skinny_gemm_mfma_kernel<3>), grid, block, 0, s, ap, w, b, op, mc, N, K, silu_in, silu_out); break;
            default: hipLaunchKernelGGL((skinny_gemm_mfma_kernel<4>), grid, block, 0, s, ap, w, b, op, mc, N, K, silu_in, silu_out); break;
        }
        F5_LAUNCH_CHECK();
    }
    return 0;
}

// =================================================================================================
// positional tables
// =================================================================================================
__global__ void rope_table_kernel(float* __restrict__ cos_t, float* __restrict__ sin_t, int seq_len, int dim_head) {
    const int n = blockIdx.x;
    const int half = dim_head / 2;
    for (int j = threadIdx.x; j < half; j += blockDim.x) {
        const float inv = 1.0f / powf(10000.0f, (float)(2 * j) / (float)dim_head);  // rope.py:23
        const float a = (float)n * inv;                                              // fp32 product, rope.py:45
        cos_t[(size_t)n * half + j] = cosf(a);
        sin_t[(size_t)n * half + j] = sinf(a);
    }
}
// GROUP-major twins of the tables above for the transposed q / k tiles of the staged QKV kernels (gemm_dev.hpp staged_epilogue_tr_rope):
// [dim_head / 4][seq_len][4] with element (g, n) = (cos_2g, cos_2g+1, sin_2g, sin_2g+1) of position n -- one 16-byte load per lane and
// 4 features, 512 contiguous bytes per 32 consecutive tokens.  Same fp32 expression for the angle; the q table carries the factor the
// engine folds into q (softmax scale * log2 e, or 1).
__global__ void rope_table_g4_kernel(float* __restrict__ tq, float* __restrict__ tk, int seq_len, int dim_head, float qscale) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;                                                    // rotation pair
    if (n >= seq_len) return;
    const float inv = 1.0f / powf(10000.0f, (float)(2 * j) / (float)dim_head);  // rope.py:23
    const float a = (float)n * inv;                                              // fp32 product, rope.py:45
    const float c = cosf(a), s = sinf(a);
    const size_t o = ((size_t)(j >> 1) * seq_len + n) * 4 + (j & 1);
    tk[o] = c;
    tk[o + 2] = s;
    tq[o] = c * qscale;
    tq[o + 2] = s * qscale;
}
int f5_launch_rope_table_g4(float* tq, float* tk, int seq_len, int dim_head, float qscale, hipStream_t s) {
    F5_REQUIRE(dim_head % 4 == 0, "rope_table_g4: dim_head must be a multiple of 4");
    hipLaunchKernelGGL(rope_table_g4_kernel, dim3(f5_cdiv(seq_len, 64), dim_head / 2), dim3(64), 0, s, tq, tk, seq_len, dim_head, qscale);
    F5_LAUNCH_CHECK();
    return 0;
}
int f5_launch_rope_table(float* cos_t, float* sin_t, int seq_len, int dim_head, hipStream_t s) {
    hipLaunchKernelGGL(rope_table_kernel, dim3(seq_len), dim3(64), 0, s, cos_t, sin_t, seq_len, dim_head);
    F5_LAUNCH_CHECK();
    return 0;
}

__global__ void text_pos_table_kernel(float* __restrict__ table, int max_pos, int dim) {
    const int n = blockIdx.x;
    const int half = dim / 2;
    for (int j = threadIdx.x; j < half; j += blockDim.x) {
        const float f = 1.0f / powf(10000.0f, (float)(2 * j) / (float)dim);  // rope.py:66-68
        const float a = (float)n * f;                                        // rope.py:70
        table[(size_t)n * dim + j] = cosf(a);                                // [cos | sin], rope.py:71-73
        table[(size_t)n * dim + half + j] = sinf(a);
    }
}
int f5_launch_text_pos_table(float* table, int max_pos, int dim, hipStream_t s) {
    hipLaunchKernelGGL(text_pos_table_kernel, dim3(max_pos), dim3(256), 0, s, table, max_pos, dim);
    F5_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// ODE state plumbing
// =================================================================================================
__global__ __launch_bounds__(256) void pack_x_kernel(const float* __restrict__ y, op16_t* __restrict__ out_hi,
                                                     op16_t* __restrict__ out_lo, int rows, int mel_dim, int* sat_flag) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * 128) return;
    const size_t row = i >> 7;
    const int c = (int)(i & 127);
    const float v = (c < mel_dim) ? y[row * mel_dim + c] : 0.0f;
    op16_t h, l;
    f5_sat_t trk;
    f5_split(v, h, l, trk);
    out_hi[i] = h;
    if (out_lo) out_lo[i] = l;
    f5_sat_commit(trk, sat_flag);
}
int f5_launch_pack_x(const float* y, op16_t* out_hi, op16_t* out_lo, int rows, int mel_dim, hipStream_t s) {
    hipLaunchKernelGGL(pack_x_kernel, dim3(f5_cdiv((long)rows * 128, 256)), dim3(256), 0, s, y, out_hi, out_lo, rows, mel_dim, f5_sat_flag_host);
    F5_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void ode_stage_kernel(F5OdeArgs p) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over rows * 128 (padded columns)
    if (i >= (size_t)p.rows * 128) return;
    const size_t row = i >> 7;
    const int c = (int)(i & 127);
    float o = 0.0f;
    if (c < p.mel_dim) {
        const size_t idx = row * p.mel_dim + c;
        const float pr = p.pred[idx];
        float k = pr;
        if (p.null_pred) k = pr + (pr - p.null_pred[idx]) * (p.cfg_ptr ? p.cfg_ptr[0] : p.cfg);     // cfm.py:364
        if (p.kstore) p.kstore[idx] = k;
        const float a = (p.coef * p.dt_ptr[0]) / p.divisor;
        float upd = k;
        if (p.mode == 1) upd = ((p.k1[idx] + 2.0f * p.k2[idx]) + 2.0f * p.k3[idx]) + k;  // cfm.py:117
        o = p.base[idx] + a * upd;
        p.out[idx] = o;
    }
    if (p.xin_hi) {
        op16_t h, l;
        f5_sat_t trk;
        f5_split(o, h, l, trk);
        p.xin_hi[i] = h;
        if (p.xin_lo) p.xin_lo[i] = l;
        f5_sat_commit(trk, p.sat_flag);
    }
}
int f5_launch_ode_stage(const F5OdeArgs& a, hipStream_t s) {
    F5_REQUIRE(a.mel_dim <= 128, "ode_stage: mel_dim must be <= 128");
    F5OdeArgs ab = a;
    if (ab.sat_flag == nullptr) ab.sat_flag = f5_sat_flag_host;
    hipLaunchKernelGGL(ode_stage_kernel, dim3(f5_cdiv((long)a.rows * 128, 256)), dim3(256), 0, s, ab);
    F5_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void splice_kernel(const float* __restrict__ cond, const float* __restrict__ y,
                                                     const int* __restrict__ lens, float* __restrict__ out, int seq_len,
                                                     int mel_dim, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t row = i / mel_dim;
    const int b = (int)(row / seq_len);
    const int n = (int)(row - (size_t)b * seq_len);
    out[i] = (n < lens[b]) ? cond[i] : y[i];
}
int f5_launch_splice(const float* cond, const float* y, const int* lens, float* out, int B, int seq_len, int mel_dim,
                     hipStream_t s) {
    const size_t total = (size_t)B * seq_len * mel_dim;
    hipLaunchKernelGGL(splice_kernel, dim3(f5_cdiv((long)total, 256)), dim3(256), 0, s, cond, y, lens, out, seq_len, mel_dim,
                       total);
    F5_LAUNCH_CHECK();
    return 0;
}

// ---- per-call staging without memcpy / memset API calls --------------------------------------------------------------------
// Everything f5_sample enqueues is a KERNEL on the caller's stream: host scalars travel as kernel arguments (copied at launch
// time: no host buffer to keep alive, no host synchronisation), device inputs / outputs are moved by a copy kernel, the V^T pad
// columns are zeroed by a kernel.  hipMemcpyAsync / hipMemsetAsync go through the copy engines (and, captured, become graph
// memcpy / memset nodes): their ordering against neighbouring kernel nodes of back-to-back graph launches was observed to fail
// on some boxes of the pool (a late memset of V^T zeroing freshly written keys' values).
struct F5StageWords {
    uint32_t w[960];
};
__global__ __launch_bounds__(256) void stage_words_kernel(F5StageWords s, uint32_t* __restrict__ dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = s.w[i];
}
int f5_launch_stage_words(const uint32_t* host_words, size_t nwords, uint32_t* dst, hipStream_t s) {
    for (size_t o = 0; o < nwords; o += 960) {
        F5StageWords sw;
        const int n = (int)(nwords - o < 960 ? nwords - o : 960);
        memcpy(sw.w, host_words + o, (size_t)n * 4);
        hipLaunchKernelGGL(stage_words_kernel, dim3(f5_cdiv(n, 256)), dim3(256), 0, s, sw, dst + o, n);
    }
    F5_LAUNCH_CHECK();
    return 0;
}
__global__ __launch_bounds__(256) void copy_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n4, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<u32x4*>(dst)[i] = reinterpret_cast<const u32x4*>(src)[i];
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void copy_words_scalar_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
// device -> device copy of n 32-bit words (16-byte accesses when both pointers allow it)
int f5_launch_copy_words(const void* src, void* dst, size_t nwords, hipStream_t s) {
    if (nwords == 0) return 0;
    const bool al = (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    const size_t work = al ? (nwords + 3) / 4 : nwords;
    long blocks = f5_cdiv((long)work, 256);
    if (blocks > 4096) blocks = 4096;
    if (al)
        hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, nwords / 4, nwords);
    else
        hipLaunchKernelGGL(copy_words_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t*)src, (uint32_t*)dst, nwords);
    F5_LAUNCH_CHECK();
    return 0;
}
// V^T [rows][npad]: zero the pad columns [seq_len, npad) (the attention kernel multiplies them by P = 0: they must be finite;
// columns < seq_len are rewritten by every QKV epilogue)
__global__ __launch_bounds__(256) void zero_vt_pad_kernel(op16_t* __restrict__ vt, int seq_len, int npad, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int pw = npad - seq_len;
    const size_t row = i / pw;
    const int c = seq_len + (int)(i - row * pw);
    vt[row * npad + c] = static_cast<op16_t>(0.0f);
}
int f5_launch_zero_vt_pad(op16_t* vt, size_t rows, int seq_len, int npad, hipStream_t s) {
    if (npad <= seq_len || rows == 0) return 0;
    const size_t total = rows * (size_t)(npad - seq_len);
    hipLaunchKernelGGL(zero_vt_pad_kernel, dim3(f5_cdiv((long)total, 256)), dim3(256), 0, s, vt, seq_len, npad, total);
    F5_LAUNCH_CHECK();
    return 0;
}

// MFMA rate yardstick (bench.py prints the rate it measures next to the data-sheet peak it divides by): every wave
// streams v_mfma_f32_32x32x16 on EIGHT independent accumulators in the operand order of the shipped K step (consecutive pairs share
// the B fragment), no memory traffic in the loop.  operands == nullptr: lane-constant operand registers (what a zero / constant-filled
// benchmark sees); otherwise eight A and eight B fragments per lane are loaded once from `operands` (>= 16 x 64 x 8 values of
// workload-like data) and rotated, so consecutive MFMAs see different data like a K loop does -- the delivered clock depends on how
// many operand bits toggle (MI355X_MICROARCH.md "DVFS give-back").
// Round 6 correction: until round 5 this loop had FOUR accumulators visited 0,1,1,2,2,3,3,0: every other MFMA waited for its
// predecessor's result, and the loop measured that chain (1 336 TF at 2.0 GHz and 1 050 W -- below every limit of the chip) instead
// of the pipe; with independent accumulators the same operand values run at 1 656 TF (1.72-1.79 GHz, 92 % pipe time, 1 240-1 270 W);
// tools/probes/mfma_energy.hip keeps both patterns, profiles/r06/mfma_energy_probe.jsonl has the numbers.
__global__ __launch_bounds__(512) void mfma_peak_kernel(const op16_t* __restrict__ operands, int iters, float* __restrict__ sink) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    op16x8 a[8], b[8];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (operands) {
            a[i] = *reinterpret_cast<const op16x8*>(operands + ((size_t)i * 64 + lane) * 8);
            b[i] = *reinterpret_cast<const op16x8*>(operands + ((size_t)(8 + i) * 64 + lane) * 8);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a[i][e] = static_cast<op16_t>(0.5f);
                b[i][e] = static_cast<op16_t>(0.03125f);
            }
        }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[(2 * i) & 7] = F5_MFMA32(a[(2 * i) & 7], b[i], acc[(2 * i) & 7], 0, 0, 0);
            acc[(2 * i + 1) & 7] = F5_MFMA32(a[(2 * i + 1) & 7], b[i], acc[(2 * i + 1) & 7], 0, 0, 0);
        }
    }
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][7];
    if (t == 123456.789f) sink[0] = t;                     // never true: keeps the accumulators alive
}
// returns the number of MFMA flops of the launch in *flops (blocks x 8 waves x iters x 16 MFMAs x 32 x 32 x 16 x 2)
int f5_launch_mfma_peak(const op16_t* operands, int blocks, int iters, float* sink, double* flops, hipStream_t s) {
    F5_REQUIRE(blocks > 0 && iters > 0 && sink, "mfma_peak: bad arguments");
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(512), 0, s, operands, iters, sink);
    F5_LAUNCH_CHECK();
    if (flops) *flops = (double)blocks * 8.0 * iters * 16.0 * 32768.0;
    return 0;
}

__global__ void rowkeep_kernel(const int* __restrict__ dur, uint8_t* __restrict__ keep, int seq_len, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / seq_len);
    const int n = (int)(i - (size_t)b * seq_len);
    keep[i] = n < dur[b] ? 1 : 0;
}
int f5_launch_rowkeep(const int* dur, uint8_t* keep, int nbatch, int seq_len, hipStream_t s) {
    const size_t total = (size_t)nbatch * seq_len;
    hipLaunchKernelGGL(rowkeep_kernel, dim3(f5_cdiv((long)total, 256)), dim3(256), 0, s, dur, keep, seq_len, total);
    F5_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// LayerNorm (affine), one wave per row
// =================================================================================================
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ out_f32,
                                                        op16_t* __restrict__ out_hi, op16_t* __restrict__ out_lo, int rows,
                                                        float eps) {
    constexpr int DIM = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 v[NV];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(x + (size_t)row * DIM + i * 256 + lane * 4);
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = f5_wave_sum(sum) * (1.0f / DIM);
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[i][e] - mean;
            sq += d * d;
        }
    const float rstd = rsqrtf(f5_wave_sum(sq) * (1.0f / DIM) + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + c);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * wv[e] + bv[e];
        if (out_f32) *reinterpret_cast<f32x4*>(out_f32 + (size_t)row * DIM + c) = y;
        if (out_hi) *reinterpret_cast<u32x2*>(out_hi + (size_t)row * DIM + c) = u32x2{f5_pack2(y[0], y[1]), f5_pack2(y[2], y[3])};
        if (out_lo)
            *reinterpret_cast<u32x2*>(out_lo + (size_t)row * DIM + c) = u32x2{f5_pack2_lo(y[0], y[1]), f5_pack2_lo(y[2], y[3])};
    }
}
int f5_launch_layernorm(const float* x, const float* w, const float* b, float* out_f32, op16_t* out_hi, op16_t* out_lo,
                        int rows, int dim, float eps, hipStream_t s) {
    F5_REQUIRE(dim % 256 == 0 && dim >= 256 && dim <= 1024, "layernorm: dim must be 256/512/768/1024 (got %d)", dim);
    const dim3 grid(f5_cdiv(rows, 4)), block(256);
#define LN_ARGS x, w, b, out_f32, out_hi, out_lo, rows, eps
    switch (dim / 256) {
        case 1: hipLaunchKernelGGL((layernorm_kernel<1>), grid, block, 0, s, LN_ARGS); break;
        case 2: hipLaunchKernelGGL((layernorm_kernel<2>), grid, block, 0, s, LN_ARGS); break;
        case 3: hipLaunchKernelGGL((layernorm_kernel<3>), grid, block, 0, s, LN_ARGS); break;
        default: hipLaunchKernelGGL((layernorm_kernel<4>), grid, block, 0, s, LN_ARGS); break;
    }
#undef LN_ARGS
    F5_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void im2col7_kernel(const float* __restrict__ x, op16_t* __restrict__ out_hi,
                                                      op16_t* __restrict__ out_lo, int seq_len, int channels, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over rows * 7 * 128
    if (i >= total) return;
    const size_t row = i / (7 * 128);
    const int rem = (int)(i - row * (7 * 128));
    const int t = rem >> 7, c = rem & 127;
    const int b = (int)(row / seq_len);
    const int n = (int)(row - (size_t)b * seq_len) + t - 3;
    float v = 0.0f;
    if (c < channels && n >= 0 && n < seq_len) v = x[((size_t)b * seq_len + n) * channels + c];
    op16_t h, l;
    f5_split(v, h, l);
    out_hi[i] = h;
    if (out_lo) out_lo[i] = l;
}
int f5_launch_im2col7(const float* x, op16_t* out_hi, op16_t* out_lo, int nbatch, int seq_len, int channels, hipStream_t s) {
    F5_REQUIRE(channels <= 128, "im2col7: channels must be <= 128");
    const size_t total = (size_t)nbatch * seq_len * 7 * 128;
    hipLaunchKernelGGL(im2col7_kernel, dim3(f5_cdiv((long)total, 256)), dim3(256), 0, s, x, out_hi, out_lo, seq_len, channels,
                       total);
    F5_LAUNCH_CHECK();
    return 0;
}

// =================================================================================================
// duration predictor helpers (duration.py)
// =================================================================================================
// fp32 [rows][cols] (optionally row-masked) -> bf16 (hi, lo) at column offset col0 of a [rows][ld] matrix
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ src, const uint8_t* __restrict__ rowkeep,
                                                        op16_t* __restrict__ out_hi, op16_t* __restrict__ out_lo, int cols, int ld,
                                                        int col0, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t row = i / cols;
    const int c = (int)(i - row * cols);
    float v = src[i];
    if (rowkeep != nullptr && rowkeep[row] == 0) v = 0.0f;
    op16_t h, l;
    f5_split(v, h, l);
    out_hi[row * ld + col0 + c] = h;
    if (out_lo) out_lo[row * ld + col0 + c] = l;
}
int f5_launch_pack_bf16(const float* src, const uint8_t* rowkeep, op16_t* out_hi, op16_t* out_lo, int rows, int cols, int ld,
                        int col0, hipStream_t s) {
    F5_REQUIRE(col0 >= 0 && col0 + cols <= ld, "pack_bf16: column range out of bounds");
    const size_t total = (size_t)rows * cols;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(f5_cdiv((long)total, 256)), dim3(256), 0, s, src, rowkeep, out_hi, out_lo, cols, ld,
                       col0, total);
    F5_LAUNCH_CHECK();
    return 0;
}

// nn.RMSNorm (duration.py:137) + masked mean over the sequence (utils.py:82-90) + Linear(dim -> 1, no bias) + Softplus
// (duration.py:188-190): one workgroup per batch element.  out[b] = softplus(sum_d w[d] * mean_n(mask * rms(x)[n,d] * g[d]))
__global__ __launch_bounds__(256) void duration_head_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ w, const uint8_t* __restrict__ mask,
                                                            float* __restrict__ out, int seq_len, int dim, float eps) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.0f;     // this wave's partial of sum_n mask[n] * sum_d (x[n,d] * rstd[n]) * g[d] * w[d]
    int cnt = 0;
    for (int n = wave; n < seq_len; n += 4) {
        const float* xr = x + ((size_t)b * seq_len + n) * dim;
        float ss = 0.0f, dot = 0.0f;
        for (int d = lane; d < dim; d += 64) {
            const float v = xr[d];
            ss += v * v;
            dot += v * g[d] * w[d];
        }
        ss = f5_wave_sum(ss);
        dot = f5_wave_sum(dot);
        if (mask[(size_t)b * seq_len + n]) {
            acc += dot * rsqrtf(ss / (float)dim + eps);
            cnt += 1;
        }
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int n = 0; n < seq_len; ++n) total += mask[(size_t)b * seq_len + n] ? 1 : 0;
        const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)(total > 0 ? total : 1);
        out[b] = mean > 20.0f ? mean : log1pf(expf(mean));
    }
}
int f5_launch_duration_head(const float* x, const float* g, const float* w, const uint8_t* mask, float* out, int B, int seq_len,
                            int dim, float eps, hipStream_t s) {
    hipLaunchKernelGGL(duration_head_kernel, dim3(B), dim3(256), 0, s, x, g, w, mask, out, seq_len, dim, eps);
    F5_LAUNCH_CHECK();
    return 0;
}
}  // namespace F5_NS
