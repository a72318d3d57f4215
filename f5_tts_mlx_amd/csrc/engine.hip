// Engine: weights arena, workspace plan, F5TTS.sample orchestration (eager or hipGraph) and the C ABI.
// Reference path: F5TTS.sample (cfm.py:312-397) -> fn/CFG (cfm.py:340-365) -> DiT.__call__ (dit.py:374-401).
//
// What is hoisted out of the ODE loop (numerically identical up to summation order):
//   * time MLP + all 22 adaLN linears + final adaLN for every function evaluation time (t only),
//   * the text path (TextEmbedding + ConvNeXtV2 blocks) for the cond and null branches (text only),
//   * the cond/text part of the input projection  W_c*cond + W_t*text + b  (dit.py:250).
// Per function evaluation the cond and null branches run as one batch of 2*B*N rows.
#include <stdarg.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/f5tts_hip.h"
#include "host_common.hpp"

// operand type of the per-op entry points (f5_op_*), set by f5_op_set_operand_type; engines carry their own
static Ops g_ops;

// ------------------------------------------------------------------------------------------------
// error string
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void f5_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* f5_last_error(void) { return g_err; }
extern "C" int f5_version(void) { return 1; }

struct MatF8 {          // MX-fp8 matrix in the arena: e4m3 [rows_pad][ld] + E8M0 [rows_pad][ld/32]
    size_t q = 0, s = 0;
    int rows = 0, ld = 0;
};
struct BlockW {
    MatBF qkv, o, ff1, ff2;
    MatF8 qkv8, o8, ff1_8, ff2_8;  // precision mxfp8 only (quantised from the bf16 copies at finalize)
    size_t bqkv, bo, bff1, bff2;  // fp32
};
struct TextBlockW {
    size_t dw_w, dw_b, ln_w, ln_b, b1, gamma, beta, b2;
    MatBF pw1, pw2;
};

struct Workspace {
    size_t total = 0;
    size_t scal, scal_words;   // per-call host scalars, contiguous: [status 1 | lens B | dur2 2B | tgrid nfe | dt steps | cfg 1 | lncnt] (32-bit words)
    size_t status;             // THE FIRST WORD OF THE WORKSPACE for every shape / solver (a C caller may read it without re-planning), staged
                               // with the scalars: bit 1 = the LN fold ran in this call (set by the host), bit 0 = a folded operand left the
                               // fp16 range (set by the producing epilogue); f5_sample_status / f5_sample_status_async
    size_t lncnt, lncnt_words; // row-block arrival counters of the fused LN tail (zeroed with the scalars; 0 words = no fusion)
    size_t lens, dur2, text, ids, keep, rowkeep;
    size_t tgrid, dt, cfgv, sinus, th, temb, mod;
    size_t rope_cos, rope_sin;
    size_t rope_t;                 // group-major q / k rotation tables (gemm.hpp rope_g4*): 2 x [16][N][4] floats
    size_t cond, traj, ytmp, kst, vel;
    size_t xin[2];
    size_t te[2], tg, grn_partial, grn_nx;
    size_t tln[2], tg2[2], ct[2];
    size_t hc, x;
    size_t xb[2], c1[2], h[2], qk[2], vt[2], ao[2], ffh[2];
    size_t h8, h8s, ao8, ao8s, ffh8, ffh8s;   // precision mxfp8: block-GEMM A operands as e4m3 + E8M0
    // LN fold (0 bytes each where the fold cannot run: plan_workspace)
    size_t lnstats;                // per 64-column slice and row (sum d, centred sum of squares) of d = x - lnmean, [D / 64][M2][2] floats
    size_t lnrowf;                 // row factors (rstd, rstd * (mean - lnmean)), [M2][2] floats
    size_t lnmean;                 // the row's mean at its previous LayerNorm = the shift of the next folded operand, 2 x [M2] floats: the
                                   // consumers that merge the statistics themselves (option fold_stats) read one and write the other
    size_t foldc[2];               // c1, c2 tables, [nfe][L][3 D + FF] floats each
    bool fold_planned;
    size_t vt_bytes;
};

struct GraphEntry {
    std::string key;
    hipGraphExec_t exec;     // segment 0 (the whole call when the graph is not split)
    std::vector<hipGraphExec_t> more;   // engine option "graph_split": the further segments (one per ODE step), launched in order
    const void* workspace;   // the captured nodes reference this buffer
    uint64_t stamp;          // last use (LRU)
    hipEvent_t done;         // recorded after every launch: an exec is only destroyed once its last replay has finished
};

// Per-engine launch options: everything that reaches a kernel as a by-value argument or changes the launch sequence of sample()
// lives in the engine handle, so two engines of one process keep their own configuration (f5_engine_set_option).  The process-wide
// f5_debug_set_* hooks only set the DEFAULTS a new engine starts from (and what the per-op entry points f5_op_* use).
struct F5Options {
    int q_premul = 1;     // q leaves the QKV epilogue multiplied by softmax_scale * log2(e) (single-segment operand modes)
    int qkv_tr = 1;       // 256x256 QKV kernel: q / k tiles accumulated transposed (pair-major rotation tables)
    int fuse_ln = 0;      // LN-modulate fused behind the small-tile residual GEMMs (measured slower, profiles/r02/ln_fusion_ab.txt)
    int gemm_flags = 0;   // F5GemmArgs::debug_flags of this engine's GEMM launches
    int attn_pipe = -1;   // large-grid attention: -1 = process default (f5_debug_set_attn_pipe), 0 = v2f, 1 = v2p (in-wave software pipeline)
    int ln_fold = -1;     // LN-modulate folded into the GEMMs around it (gemm.hpp fold_*): -1 = where it is measured faster (>= LN_FOLD_AUTO_ROWS
                          // rows and the four block GEMMs on the staged kernels, one-pass operand modes), 0 = never, 1 = wherever it can run
                          // (batch >= 4 at the 335M shape; fails loudly elsewhere)
    int fold_stats = -1;  // LN fold: 1 = the consumer GEMMs merge the producer's slice statistics into their row factors themselves (gemm.hpp fold_stats;
                          // width 1024 only), 0 = a f5_fold_rows launch between producer and consumer (round 4-5; rules the batch-1-sized route
                          // out), -1 = the statistics form on the batch-1-sized route only
    int graph_split = 0;  // 0: a call is ONE hipGraphExec (~5 000 kernel nodes at 32 points); 1: one exec per ODE step (prep rides in the first), launched
                          // back to back -- the same kernels with the same arguments, 31 replays of ~165 nodes (round-6 probe: does a long exec cost more per node?)
    int sat_check = 1;    // precision f16: every 16-bit operand producer reports values beyond +-65 504 in the status word (bit 2); 0 = A/B
    int null_keeps_cond = 0;   // the second (null) branch keeps the audio conditioning: DiT.__call__(drop_audio_cond=False, drop_text=True), dit.py:374-401
};
static F5Options g_default_options;
static int g_live_engines = 0;   // f5_debug_set_{ln_fusion,qkv_transposed,q_premul} only change the DEFAULTS: with engines alive they say so
// bumped by every process-wide launch knob change (f5_debug_set_*): part of the graph key, so a cached hipGraph captured under other
// knob values is never replayed
static int g_knob_epoch = 0;

static void destroy_graph_entry(GraphEntry& g) {
    if (g.done) {
        (void)hipEventSynchronize(g.done);        // the exec may still be running on the stream of its last launch
        (void)hipEventDestroy(g.done);
    }
    (void)hipGraphExecDestroy(g.exec);
    for (hipGraphExec_t x : g.more) (void)hipGraphExecDestroy(x);
}

struct f5_engine {
    f5_config cfg;
    int prec;
    int np;  // precision parts (1 or 2)
    char* arena = nullptr;
    size_t arena_bytes = 0;
    size_t arena_need = 0;
    bool finalized = false;
    std::unordered_map<std::string, std::vector<TensorDst>> tmap;
    // arena offsets
    size_t time_w0, time_b0, time_w2, time_b2;
    size_t ada_w, ada_b;  // [(6*depth+2)*D][D], [(6*depth+2)*D]
    size_t text_table, text_pos;
    std::vector<TextBlockW> tblocks;
    MatBF wx, wct;
    size_t bproj;
    MatBF conv_w[2];
    size_t conv_b[2];
    std::vector<BlockW> blocks;
    MatBF wout;
    size_t bout;
    Ops ops;                          // kernels of this engine's operand type
    F5Options opt = g_default_options;
    std::vector<GraphEntry> graphs;   // cached hipGraphExecs, at most graph_cap (least recently used is destroyed)
    std::vector<std::string> seen;    // signatures sampled once in "auto" graph mode (captured on the second sighting)
    int graph_cap = 8;
    uint64_t clock = 0;
};

static int nfe_per_step(int method) { return method == F5_EULER ? 1 : (method == F5_MIDPOINT ? 2 : 4); }

static MatF8 alloc_f8(Bump& b, int rows, int ld) {
    MatF8 m;
    m.rows = rows;
    m.ld = ld;
    const int rows_pad = (rows + 255) / 256 * 256;  // the MX GEMM reads whole 256-row weight tiles
    m.q = b.take((size_t)rows_pad * ld);
    m.s = b.take((size_t)rows_pad * (ld / 32));
    return m;
}

static void add_f32(f5_engine* e, const std::string& name, size_t off, std::vector<int64_t> shape) {
    TensorDst d;
    d.kind = 0;
    d.off = off;
    d.shape = shape;
    e->tmap[name].push_back(d);
}
static void add_mat(f5_engine* e, const std::string& name, const MatBF& m, int row0, int src_rows, int src_cols, int c0, int c1,
                    int dst_c0, std::vector<int64_t> shape) {
    TensorDst d;
    d.kind = 1;
    d.mat = m;
    d.row0 = row0;
    d.src_rows = src_rows;
    d.src_cols = src_cols;
    d.c0 = c0;
    d.c1 = c1;
    d.dst_c0 = dst_c0;
    d.shape = shape;
    e->tmap[name].push_back(d);
}

static int build_arena_plan(f5_engine* e) {
    const f5_config& c = e->cfg;
    const int D = c.dim, Dt = c.text_dim, FF = c.ff_dim, TF = c.text_ff_dim, L = c.depth, np = e->np;
    Bump b;
    const std::string p = "transformer.";
    // time MLP (fp32)
    e->time_w0 = b.take((size_t)D * c.freq_embed_dim * 4);
    e->time_b0 = b.take((size_t)D * 4);
    e->time_w2 = b.take((size_t)D * D * 4);
    e->time_b2 = b.take((size_t)D * 4);
    add_f32(e, p + "time_embed.time_mlp.layers.0.weight", e->time_w0, {D, c.freq_embed_dim});
    add_f32(e, p + "time_embed.time_mlp.layers.0.bias", e->time_b0, {D});
    add_f32(e, p + "time_embed.time_mlp.layers.2.weight", e->time_w2, {D, D});
    add_f32(e, p + "time_embed.time_mlp.layers.2.bias", e->time_b2, {D});
    // adaLN (fp32): all blocks + final stacked so one skinny GEMM builds the whole modulation table
    const size_t ada_rows = (size_t)(6 * L + 2) * D;
    e->ada_w = b.take(ada_rows * D * 4);
    e->ada_b = b.take(ada_rows * 4);
    // text embedding
    e->text_table = b.take((size_t)(c.text_num_embeds + 1) * Dt * 4);
    e->text_pos = b.take((size_t)c.text_max_pos * Dt * 4);
    add_f32(e, p + "text_embed.text_embed.weight", e->text_table, {c.text_num_embeds + 1, Dt});
    e->tblocks.resize(c.conv_layers);
    for (int i = 0; i < c.conv_layers; ++i) {
        TextBlockW& t = e->tblocks[i];
        const std::string q = p + "text_embed.text_blocks.layers." + std::to_string(i) + ".";
        t.dw_w = b.take((size_t)Dt * 7 * 4);
        t.dw_b = b.take((size_t)Dt * 4);
        t.ln_w = b.take((size_t)Dt * 4);
        t.ln_b = b.take((size_t)Dt * 4);
        t.b1 = b.take((size_t)TF * 4);
        t.gamma = b.take((size_t)TF * 4);
        t.beta = b.take((size_t)TF * 4);
        t.b2 = b.take((size_t)Dt * 4);
        t.pw1 = alloc_mat(b, TF, Dt, np);
        t.pw2 = alloc_mat(b, Dt, TF, np);
        add_f32(e, q + "dwconv.weight", t.dw_w, {Dt, 7, 1});
        add_f32(e, q + "dwconv.bias", t.dw_b, {Dt});
        add_f32(e, q + "norm.weight", t.ln_w, {Dt});
        add_f32(e, q + "norm.bias", t.ln_b, {Dt});
        add_mat(e, q + "pwconv1.weight", t.pw1, 0, TF, Dt, 0, Dt, 0, {TF, Dt});
        add_f32(e, q + "pwconv1.bias", t.b1, {TF});
        add_f32(e, q + "grn.gamma", t.gamma, {1, 1, TF});
        add_f32(e, q + "grn.beta", t.beta, {1, 1, TF});
        add_mat(e, q + "pwconv2.weight", t.pw2, 0, Dt, TF, 0, TF, 0, {Dt, TF});
        add_f32(e, q + "pwconv2.bias", t.b2, {Dt});
    }
    // input projection, split by input segment (x | cond | text), dit.py:250
    const int M = c.mel_dim, IN = 2 * M + Dt;
    e->wx = alloc_mat(b, D, 128, np);
    e->wct = alloc_mat(b, D, 128 + Dt, np);
    e->bproj = b.take((size_t)D * 4);
    add_mat(e, p + "input_embed.proj.weight", e->wx, 0, D, IN, 0, M, 0, {D, IN});
    add_mat(e, p + "input_embed.proj.weight", e->wct, 0, D, IN, M, 2 * M, 0, {D, IN});
    add_mat(e, p + "input_embed.proj.weight", e->wct, 0, D, IN, 2 * M, IN, 128, {D, IN});
    add_f32(e, p + "input_embed.proj.bias", e->bproj, {D});
    const int kc = c.conv_pos_kernel, gin = D / c.conv_pos_groups;
    for (int j = 0; j < 2; ++j) {
        e->conv_w[j] = alloc_mat(b, D, kc * gin, np);
        e->conv_b[j] = b.take((size_t)D * 4);
        const std::string q = p + "input_embed.conv_pos_embed.conv1d.layers." + std::to_string(j * 2) + ".";
        add_mat(e, q + "weight", e->conv_w[j], 0, D, kc * gin, 0, kc * gin, 0, {D, kc, gin});
        add_f32(e, q + "bias", e->conv_b[j], {D});
    }
    e->blocks.resize(L);
    for (int i = 0; i < L; ++i) {
        BlockW& w = e->blocks[i];
        const std::string q = p + "transformer_blocks." + std::to_string(i) + ".";
        w.qkv = alloc_mat(b, 3 * D, D, np);
        w.o = alloc_mat(b, D, D, np);
        w.ff1 = alloc_mat(b, FF, D, np);
        w.ff2 = alloc_mat(b, D, FF, np);
        if (e->prec == F5_PREC_MXFP8) {
            w.qkv8 = alloc_f8(b, 3 * D, D);
            w.o8 = alloc_f8(b, D, D);
            w.ff1_8 = alloc_f8(b, FF, D);
            w.ff2_8 = alloc_f8(b, D, FF);
        }
        w.bqkv = b.take((size_t)3 * D * 4);
        w.bo = b.take((size_t)D * 4);
        w.bff1 = b.take((size_t)FF * 4);
        w.bff2 = b.take((size_t)D * 4);
        add_f32(e, q + "attn_norm.linear.weight", e->ada_w + (size_t)i * 6 * D * D * 4, {6 * D, D});
        add_f32(e, q + "attn_norm.linear.bias", e->ada_b + (size_t)i * 6 * D * 4, {6 * D});
        const char* nm[3] = {"to_q", "to_k", "to_v"};
        for (int k = 0; k < 3; ++k) {
            add_mat(e, q + "attn." + nm[k] + ".weight", w.qkv, k * D, D, D, 0, D, 0, {D, D});
            add_f32(e, q + "attn." + nm[k] + ".bias", w.bqkv + (size_t)k * D * 4, {D});
        }
        add_mat(e, q + "attn.to_out.layers.0.weight", w.o, 0, D, D, 0, D, 0, {D, D});
        add_f32(e, q + "attn.to_out.layers.0.bias", w.bo, {D});
        add_mat(e, q + "ff.ff.layers.0.layers.0.weight", w.ff1, 0, FF, D, 0, D, 0, {FF, D});
        add_f32(e, q + "ff.ff.layers.0.layers.0.bias", w.bff1, {FF});
        add_mat(e, q + "ff.ff.layers.2.weight", w.ff2, 0, D, FF, 0, FF, 0, {D, FF});
        add_f32(e, q + "ff.ff.layers.2.bias", w.bff2, {D});
    }
    add_f32(e, p + "norm_out.linear.weight", e->ada_w + (size_t)L * 6 * D * D * 4, {2 * D, D});
    add_f32(e, p + "norm_out.linear.bias", e->ada_b + (size_t)L * 6 * D * 4, {2 * D});
    e->wout = alloc_mat(b, M, D, np);
    e->bout = b.take((size_t)M * 4);
    add_mat(e, p + "proj_out.weight", e->wout, 0, M, D, 0, D, 0, {M, D});
    add_f32(e, p + "proj_out.bias", e->bout, {M});
    e->arena_need = b.off;
    return 0;
}

extern "C" int f5_engine_create(const f5_config* cfg, int precision, f5_engine** out) {
    F5_REQUIRE(cfg && out, "f5_engine_create: null argument");
    const f5_config& c = *cfg;
    F5_REQUIRE(precision == F5_PREC_BF16 || precision == F5_PREC_BF16X3 || precision == F5_PREC_MXFP8 || precision == F5_PREC_F16,
               "unknown precision %d", precision);
    F5_REQUIRE(precision != F5_PREC_MXFP8 || (cfg->ff_dim % 256 == 0 && cfg->dim % 256 == 0), "mxfp8 needs dim and ff_dim to be multiples of 256");
    F5_REQUIRE(c.dim_head == 64, "dim_head must be 64 (got %d)", c.dim_head);
    F5_REQUIRE(c.heads * c.dim_head == c.dim, "heads * dim_head must equal dim (%d * %d != %d)", c.heads, c.dim_head, c.dim);
    F5_REQUIRE(c.dim % 256 == 0 && c.dim <= 1024, "dim must be a multiple of 256 and <= 1024 (got %d)", c.dim);
    F5_REQUIRE(c.dim / c.conv_pos_groups == 64, "dim / conv_pos_groups must be 64");
    F5_REQUIRE(c.conv_pos_kernel % 2 == 1 && c.conv_pos_kernel <= 31, "conv_pos_kernel must be odd and <= 31");
    F5_REQUIRE(c.text_dim % 256 == 0 && c.text_dim <= 1024, "text_dim must be a multiple of 256 and <= 1024 (got %d)", c.text_dim);
    F5_REQUIRE(c.ff_dim % 128 == 0 && c.text_ff_dim % 128 == 0, "ff dims must be multiples of 128");
    F5_REQUIRE(c.mel_dim >= 1 && c.mel_dim <= 128, "mel_dim must be in [1,128]");
    F5_REQUIRE(c.freq_embed_dim % 256 == 0 && c.freq_embed_dim <= 1024, "freq_embed_dim must be a multiple of 256");
    F5_REQUIRE(c.depth >= 1 && c.conv_layers >= 0, "depth must be >= 1 and conv_layers >= 0");
    f5_engine* e = new f5_engine();
    e->cfg = c;
    e->prec = precision;
    e->np = precision == F5_PREC_BF16X3 ? 2 : 1;
    e->ops.h = precision == F5_PREC_F16;
    build_arena_plan(e);
    ++g_live_engines;
    *out = e;
    return 0;
}

extern "C" void f5_engine_destroy(f5_engine* e) {
    if (!e) return;
    for (auto& g : e->graphs) destroy_graph_entry(g);
    --g_live_engines;
    delete e;
}

extern "C" int f5_engine_set_graph_cache(f5_engine* e, int max_graphs) {
    F5_REQUIRE(e && max_graphs >= 1 && max_graphs <= 1024, "graph cache size must be in [1, 1024]");
    e->graph_cap = max_graphs;
    while ((int)e->graphs.size() > e->graph_cap) {
        size_t lru = 0;
        for (size_t i = 1; i < e->graphs.size(); ++i)
            if (e->graphs[i].stamp < e->graphs[lru].stamp) lru = i;
        destroy_graph_entry(e->graphs[lru]);
        e->graphs.erase(e->graphs.begin() + lru);
    }
    return 0;
}
extern "C" int f5_engine_set_option(f5_engine* e, const char* name, int value) {
    F5_REQUIRE(e && name, "null argument");
    const std::string n(name);
    if (n == "q_premul") e->opt.q_premul = value ? 1 : 0;
    else if (n == "qkv_transposed") e->opt.qkv_tr = value ? 1 : 0;
    else if (n == "ln_fusion") e->opt.fuse_ln = value ? 1 : 0;
    else if (n == "gemm_flags") e->opt.gemm_flags = value;
    else if (n == "attn_pipe") e->opt.attn_pipe = value < 0 ? -1 : (value ? 1 : 0);
    else if (n == "null_keeps_cond") e->opt.null_keeps_cond = value ? 1 : 0;
    else if (n == "ln_fold") e->opt.ln_fold = value < 0 ? -1 : (value ? 1 : 0);
    else if (n == "sat_check") e->opt.sat_check = value ? 1 : 0;
    else if (n == "graph_split") e->opt.graph_split = value ? 1 : 0;
    else if (n == "fold_stats") e->opt.fold_stats = value < 0 ? -1 : (value ? 1 : 0);
    else {
        f5_set_error("unknown engine option %s (q_premul, qkv_transposed, ln_fusion, gemm_flags, attn_pipe, null_keeps_cond, ln_fold, sat_check)", name);
        return 2;
    }
    return 0;
}
extern "C" int f5_engine_get_option(f5_engine* e, const char* name, int* value) {
    F5_REQUIRE(e && name && value, "null argument");
    const std::string n(name);
    if (n == "q_premul") *value = e->opt.q_premul;
    else if (n == "qkv_transposed") *value = e->opt.qkv_tr;
    else if (n == "ln_fusion") *value = e->opt.fuse_ln;
    else if (n == "gemm_flags") *value = e->opt.gemm_flags;
    else if (n == "attn_pipe") *value = e->opt.attn_pipe;
    else if (n == "null_keeps_cond") *value = e->opt.null_keeps_cond;
    else if (n == "ln_fold") *value = e->opt.ln_fold;
    else if (n == "sat_check") *value = e->opt.sat_check;
    else if (n == "graph_split") *value = e->opt.graph_split;
    else if (n == "fold_stats") *value = e->opt.fold_stats;
    else {
        f5_set_error("unknown engine option %s", name);
        return 2;
    }
    return 0;
}
extern "C" int f5_engine_graph_count(f5_engine* e) { return e ? (int)e->graphs.size() : -1; }

extern "C" int f5_weights_bytes(f5_engine* e, size_t* bytes) {
    F5_REQUIRE(e && bytes, "null argument");
    *bytes = e->arena_need;
    return 0;
}

extern "C" int f5_set_weights_arena(f5_engine* e, void* dev_arena, size_t bytes, void* stream) {
    F5_REQUIRE(e && dev_arena, "null argument");
    F5_REQUIRE(bytes >= e->arena_need, "weights arena too small: %zu < %zu", bytes, e->arena_need);
    F5_REQUIRE(((uintptr_t)dev_arena & 255) == 0, "weights arena must be 256-byte aligned");
    e->arena = (char*)dev_arena;
    e->arena_bytes = bytes;
    F5_HIP_CHECK(hipMemsetAsync(dev_arena, 0, e->arena_need, (hipStream_t)stream));  // zero pads
    F5_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

extern "C" int f5_load_tensor(f5_engine* e, const char* name, const float* host, int ndim, const int64_t* shape) {
    F5_REQUIRE(e && name && host && shape, "null argument");
    F5_REQUIRE(e->arena, "f5_set_weights_arena must be called first");
    auto it = e->tmap.find(name);
    F5_REQUIRE(it != e->tmap.end(), "unknown tensor name '%s'", name);
    for (TensorDst& d : it->second) {
        F5_REQUIRE((int)d.shape.size() == ndim, "tensor '%s': expected %zu dims, got %d", name, d.shape.size(), ndim);
        size_t count = 1;
        for (int i = 0; i < ndim; ++i) {
            F5_REQUIRE(d.shape[i] == shape[i], "tensor '%s': dim %d is %lld, expected %lld", name, i, (long long)shape[i],
                       (long long)d.shape[i]);
            count *= (size_t)shape[i];
        }
        RC(f5_upload_tensor(e->arena, d, host, count, e->np, e->ops.h));
        d.loaded = true;
    }
    return 0;
}

extern "C" int f5_mark_weights_loaded(f5_engine* e) {
    F5_REQUIRE(e, "null argument");
    for (auto& kv : e->tmap)
        for (auto& d : kv.second) d.loaded = true;
    return 0;
}

// A second handle on the arena of `owner` (same configuration and precision, weights finalised): nothing is written, the caller keeps the
// arena alive for both.  What it is for: two handles driven by two host threads on two streams (two half batches of one sample() call fill
// the partly empty last rounds of each other's launches: engine.py Engine._sample_split); every handle has its own
// workspace, hipGraphs and status word, the weights are read-only on the hot path.
extern "C" int f5_share_weights(f5_engine* e, const f5_engine* owner) {
    F5_REQUIRE(e && owner && e != owner, "f5_share_weights: null argument / the same handle twice");
    F5_REQUIRE(owner->arena && owner->finalized, "f5_share_weights: the owner's weights are not finalised");
    F5_REQUIRE(e->prec == owner->prec && e->arena_need == owner->arena_need && memcmp(&e->cfg, &owner->cfg, sizeof(f5_config)) == 0,
               "f5_share_weights: the two handles differ in configuration or precision");
    e->arena = owner->arena;
    e->arena_bytes = owner->arena_bytes;
    for (auto& kv : e->tmap)
        for (auto& d : kv.second) d.loaded = true;
    e->finalized = true;
    return 0;
}

// One collective replicates the model: the arena is contiguous, so a host that owns an RCCL communicator (one process per GPU,
// xGMI) broadcasts it from `root` with a single ncclBroadcast.  RCCL is resolved at run time (dlopen) so that the library has
// no link-time dependency on it; the torch.distributed path of the Python wrapper (dist.py) does the same with dist.broadcast.
#include <dlfcn.h>
extern "C" int f5_broadcast_weights(f5_engine* e, void* nccl_comm, int root, int rank, void* stream) {
    F5_REQUIRE(e && e->arena && nccl_comm, "f5_broadcast_weights: null engine / arena / communicator");
    typedef int (*bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static bcast_fn fn = nullptr;
    if (!fn) {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        F5_REQUIRE(h != nullptr, "f5_broadcast_weights: librccl.so not found (%s)", dlerror());
        fn = (bcast_fn)dlsym(h, "ncclBroadcast");
        F5_REQUIRE(fn != nullptr, "f5_broadcast_weights: ncclBroadcast not found in librccl.so");
    }
    const int ncclUint8 = 1;
    const int rc = fn(e->arena, e->arena, e->arena_need, ncclUint8, root, nccl_comm, (hipStream_t)stream);
    F5_REQUIRE(rc == 0, "ncclBroadcast failed with ncclResult_t %d", rc);
    if (rank != root)
        for (auto& kv : e->tmap)
            for (auto& d : kv.second) d.loaded = true;
    return 0;
}

extern "C" int f5_finalize_weights(f5_engine* e, void* stream) {
    F5_REQUIRE(e && e->arena, "arena not set");
    for (auto& kv : e->tmap)
        for (auto& d : kv.second) F5_REQUIRE(d.loaded, "tensor '%s' was never loaded", kv.first.c_str());
    int rc = f5_launch_text_pos_table((float*)(e->arena + e->text_pos), e->cfg.text_max_pos, e->cfg.text_dim, (hipStream_t)stream);
    if (rc) return rc;
    if (e->prec == F5_PREC_MXFP8) {
        // MX-fp8 copies of the four block matrices, from the bf16 copies (pad rows of the arena are zero -> scale 0, bytes 0)
        for (const BlockW& w : e->blocks) {
            const MatBF* src[4] = {&w.qkv, &w.o, &w.ff1, &w.ff2};
            const MatF8* dst[4] = {&w.qkv8, &w.o8, &w.ff1_8, &w.ff2_8};
            for (int k = 0; k < 4; ++k) {
                rc = f5_launch_quantize_mx_bf16((const op16_t*)(e->arena + src[k]->hi), src[k]->ld, (uint8_t*)(e->arena + dst[k]->q),
                                                dst[k]->ld, (uint8_t*)(e->arena + dst[k]->s), src[k]->rows, src[k]->ld,
                                                (hipStream_t)stream);
                if (rc) return rc;
            }
        }
    }
    F5_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    e->finalized = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// workspace plan
// ------------------------------------------------------------------------------------------------
// LN fold, automatic mode: measured on sample(), 335M shape, f16 (profiles/r04/ln_fold_ab.jsonl): batch 4 -1.2 %, 8 -1.1 %, 12 +0.3 %,
// 16 +0.8 %, 32 +2.5 %: the LN launches it removes cost ~2 us per thousand rows, what it adds (x16 write in the residual epilogues, the
// row-factor kernel) has a floor
constexpr long LN_FOLD_AUTO_ROWS = 22000;
// The batch-1-sized route (round 6): all four block GEMMs are ONE round of workgroups, the consumers merge the slice statistics themselves
// (no row-factor launch), so the fold removes 44 of the 46 LN-modulate launches of a forward for a few hundred VALU instructions in the
// residual epilogues.  LN_FOLD_SMALL_AUTO: chosen by the automatic mode (-1) where the route applies.
constexpr long LN_FOLD_SMALL_ROWS = 2048;
constexpr bool LN_FOLD_SMALL_AUTO = false;     // measured -1.5 ... +0.7 % at batch 1 (profiles/r06): below the 4 % bar, so opt-in (ln_fold = 1)

static Workspace plan_workspace(const f5_engine* e, int B, int N, int nt, int steps, int method) {
    const f5_config& c = e->cfg;
    const int D = c.dim, Dt = c.text_dim, FF = c.ff_dim, TF = c.text_ff_dim, L = c.depth, np = e->np, mel = c.mel_dim;
    const size_t M1 = (size_t)B * N, M2 = 2 * M1;
    const int nfe = (steps - 1) * nfe_per_step(method);
    const int npad = (N + 63) / 64 * 64;
    Bump b;
    Workspace w;
    const size_t nfe1 = (size_t)(nfe > 0 ? nfe : 1);
    // one counter per 64 rows of the residual stream; only small-M launches fuse (gemm.hpp ln_counter), large ones get none
    w.lncnt_words = (M2 + 63) / 64 <= 512 ? (M2 + 63) / 64 : 0;
    w.scal_words = (size_t)1 + 3 * B + nfe1 + steps + 1 + w.lncnt_words;
    w.scal = b.take(w.scal_words * 4);
    w.status = w.scal;                      // offset 0, whatever the sizes
    w.lens = w.scal + 4;
    w.dur2 = w.lens + (size_t)B * 4;
    w.tgrid = w.dur2 + (size_t)2 * B * 4;
    w.dt = w.tgrid + nfe1 * 4;
    w.cfgv = w.dt + (size_t)steps * 4;
    w.lncnt = w.cfgv + 4;
    w.text = b.take((size_t)B * (nt > 0 ? nt : 1) * 4);
    w.ids = b.take(M2 * 4);
    w.keep = b.take(M2);
    w.rowkeep = b.take(M2);
    w.sinus = b.take((size_t)(nfe + 1) * c.freq_embed_dim * 4);
    w.th = b.take((size_t)(nfe + 1) * D * 4);
    w.temb = b.take((size_t)(nfe + 1) * D * 4);
    w.mod = b.take((size_t)(nfe + 1) * (6 * L + 2) * D * 4);
    w.rope_cos = b.take((size_t)N * 32 * 4);
    w.rope_sin = b.take((size_t)N * 32 * 4);
    w.rope_t = b.take((size_t)4 * 32 * N * 4);
    w.cond = b.take(M1 * mel * 4);
    w.traj = b.take((size_t)steps * M1 * mel * 4);
    w.ytmp = b.take(M1 * mel * 4);
    w.kst = b.take(3 * M1 * mel * 4);
    w.vel = b.take(M2 * mel * 4);
    for (int p = 0; p < 2; ++p) w.xin[p] = p < np ? b.take(M1 * 128 * 2) : 0;
    w.te[0] = b.take(M2 * Dt * 4);
    w.te[1] = b.take(M2 * Dt * 4);
    w.tg = b.take(M2 * TF * 4);
    w.grn_partial = b.take(f5_grn_partial_floats(2 * B, N, TF) * 4);
    w.grn_nx = b.take((size_t)2 * B * TF * 4);
    for (int p = 0; p < 2; ++p) {
        w.tln[p] = p < np ? b.take(M2 * Dt * 2) : 0;
        w.tg2[p] = p < np ? b.take(M2 * TF * 2) : 0;
        w.ct[p] = p < np ? b.take(M2 * (128 + Dt) * 2) : 0;
    }
    w.hc = b.take(M2 * D * 4);
    w.x = b.take(M2 * D * 4);
    w.vt_bytes = (size_t)2 * B * c.heads * 64 * npad * 2;
    for (int p = 0; p < 2; ++p) {
        w.xb[p] = p < np ? b.take(M2 * D * 2) : 0;
        w.c1[p] = p < np ? b.take(M2 * D * 2) : 0;
        w.h[p] = p < np ? b.take(M2 * D * 2) : 0;
        w.qk[p] = p < np ? b.take(M2 * 2 * D * 2) : 0;
        w.vt[p] = p < np ? b.take(w.vt_bytes) : 0;
        w.ao[p] = p < np ? b.take(M2 * D * 2) : 0;
        w.ffh[p] = p < np ? b.take(M2 * FF * 2) : 0;
    }
    // LN fold buffers only where the fold can run (a superset of ln_fold_state(): that one also knows the branch count of the call):
    // one-pass 16-bit operand modes, the option not 0, and -- in the automatic mode -- enough rows for it to be chosen with both branches
    // (... or few enough for the single-round kernels of the batch-1-sized route, gemm.hpp f5_gemm_fold_small: <= 2 048 rows at width 1024)
    w.fold_planned = np == 1 && e->prec != F5_PREC_MXFP8 && e->opt.ln_fold != 0 &&
                     (e->opt.ln_fold == 1 || (long)M2 >= LN_FOLD_AUTO_ROWS || ((long)M2 <= LN_FOLD_SMALL_ROWS && LN_FOLD_SMALL_AUTO));
    const size_t fp = w.fold_planned ? 1 : 0;
    w.lnstats = b.take(fp * M2 * (size_t)((D + 63) / 64) * 2 * 4);
    w.lnrowf = b.take(fp * M2 * 2 * 4);
    w.lnmean = b.take(fp * 2 * M2 * 4);
    for (int p = 0; p < 2; ++p) w.foldc[p] = b.take(fp * nfe1 * L * (size_t)(3 * D + FF) * 4);
    w.h8 = w.h8s = w.ao8 = w.ao8s = w.ffh8 = w.ffh8s = 0;
    if (e->prec == F5_PREC_MXFP8) {
        w.h8 = b.take(M2 * D);
        w.h8s = b.take(M2 * (D / 32));
        w.ao8 = b.take(M2 * D);
        w.ao8s = b.take(M2 * (D / 32));
        w.ffh8 = b.take(M2 * FF);
        w.ffh8s = b.take(M2 * (FF / 32));
    }
    w.total = b.off;
    return w;
}

extern "C" int f5_workspace_bytes(f5_engine* e, int B, int N, int nt, int steps, int method, size_t* bytes) {
    F5_REQUIRE(e && bytes, "null argument");
    F5_REQUIRE(B >= 1 && N >= 1 && steps >= 1, "bad sizes B=%d N=%d steps=%d", B, N, steps);
    F5_REQUIRE(method >= F5_EULER && method <= F5_RK4, "Unknown method: %d", method);
    *bytes = plan_workspace(e, B, N, nt, steps, method).total;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// launch context
// ------------------------------------------------------------------------------------------------
struct Ctx {
    f5_engine* e;
    Workspace w;
    char* ws;
    hipStream_t s;
    int B, N, nt, nb;   // nb = branches evaluated per function evaluation (1 or 2)
    int npad;
    bool use_mask;
    Ops ops;
    template <typename T>
    T* p(size_t off) const { return reinterpret_cast<T*>(ws + off); }
    template <typename T>
    T* a(size_t off) const { return reinterpret_cast<T*>(e->arena + off); }
    op16_t* pb(const size_t (&offs)[2], int part) const {
        return (part < e->np) ? reinterpret_cast<op16_t*>(ws + offs[part]) : nullptr;
    }
    const op16_t* wm(const MatBF& m, int part) const {
        if (part == 0) return reinterpret_cast<const op16_t*>(e->arena + m.hi);
        return e->np == 2 ? reinterpret_cast<const op16_t*>(e->arena + m.lo) : nullptr;
    }
    int nseg() const { return e->np == 2 ? 3 : 1; }
};

// factor folded into q by the QKV epilogue (0 = none): single-segment operand modes only, see run_dit
static float q_premul_factor(const f5_engine* e) {
    return (e->opt.q_premul && e->np == 1) ? (1.0f / sqrtf((float)e->cfg.dim_head)) * 1.4426950408889634f : 0.0f;
}

static F5GemmArgs gemm_base(const Ctx& c, const op16_t* a_hi, const op16_t* a_lo, int lda, const MatBF& w, int M, int N, int K,
                            const float* bias) {
    F5GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = a_hi;
    g.A[1] = a_lo;
    g.W[0] = c.wm(w, 0);
    g.W[1] = c.wm(w, 1);
    g.lda = lda;
    g.ldw = w.ld;
    g.M = M;
    g.N = N;
    g.K = K;
    g.nseg = c.nseg();
    g.bias = bias;
    g.debug_flags = c.e->opt.gemm_flags;      // this engine's own flags (f5_engine_set_option)
    return g;
}

// LN fold (gemm.hpp fold_*): 44 of the 46 LN-modulate launches of a forward disappear into the epilogues of the GEMMs around them
// (the first LN of block 0 follows conv-pos and the final one feeds a small-tile GEMM: both keep the LN kernel).  Only where all
// four block GEMMs of this shape run on the staged kernels, in the one-pass operand modes, with the transposed q / k tiles.
// -> 0 = off, 1 = on, -1 = required by the option but impossible here (error set)
static int ln_fold_state(const Ctx& c) {
    const f5_engine* e = c.e;
    const f5_config& cf = e->cfg;
    if (e->opt.ln_fold == 0) return 0;
    const int D = cf.dim, FF = cf.ff_dim, M = c.nb * c.B * c.N;
    bool ok = e->np == 1 && e->prec != F5_PREC_MXFP8 && e->opt.qkv_tr && !e->opt.fuse_ln && D % 256 == 0 && D <= 2048 && FF % 256 == 0 &&
              (e->opt.gemm_flags & 16384) == 0 && cf.depth >= 1;
    bool small = ok && D == 1024 && e->opt.fold_stats != 0 && (e->opt.gemm_flags & (8 | 256)) == 0;   // the single-round kernels: statistics form only
    if (ok) {
        F5GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.M = M;
        g.seq_len = c.N;
        g.nseg = 1;
        const int shapes[4][3] = {{3 * D, D, EPI_QKV_ROPE}, {D, D, EPI_RESID_GATE}, {FF, D, EPI_GELU_TANH}, {D, FF, EPI_RESID_GATE}};
        for (const auto& sh : shapes) {
            g.N = sh[0];
            g.K = sh[1];
            ok = ok && c.ops.gemm_runs_staged(g, sh[2]);
            small = small && c.ops.gemm_fold_small(g, sh[2], true);
        }
    }
    if (e->opt.ln_fold < 0 && !(M >= LN_FOLD_AUTO_ROWS && ok) && !(small && LN_FOLD_SMALL_AUTO)) return 0;
    if (!c.w.fold_planned) ok = small = false;      // (cannot happen: the plan's predicate is a superset of this one)
    if (!ok && !small && e->opt.ln_fold == 1) {
        f5_set_error("ln_fold = 1: this shape / precision cannot run the folded LN (needs f16 or bf16, qkv_transposed, no ln_fusion, dim %% 256 "
                     "== 0, and all four block GEMMs on the 256x256 / role-split 128x256 kernels -- batch >= 4 at the 335M shape -- or on the "
                     "single-round kernels of the batch-1-sized route: width 1024, fold_stats != 0)");
        return -1;
    }
    return ok ? 1 : (small ? 2 : 0);
}

// loop-invariant preparation: time/adaLN tables, text path, hoisted input projection, masks, rope
static int run_prep(const Ctx& c, int nfe) {
    const f5_engine* e = c.e;
    const f5_config& cf = e->cfg;
    const Workspace& w = c.w;
    const int D = cf.dim, Dt = cf.text_dim, TF = cf.text_ff_dim, L = cf.depth;
    const int M1 = c.B * c.N, M2 = 2 * M1;
    hipStream_t s = c.s;
    const Ops& K = c.ops;

    // --- t-only tables (dit.py:61-82, 267, 286)
    if (nfe > 0) {
        RC(f5_launch_time_sinus(c.p<float>(w.tgrid), c.p<float>(w.sinus), nfe, cf.freq_embed_dim, s));
        RC(f5_launch_skinny_gemm(c.p<float>(w.sinus), c.a<float>(e->time_w0), c.a<float>(e->time_b0), c.p<float>(w.th), nfe, D,
                                 cf.freq_embed_dim, 0, 1, s));
        RC(f5_launch_skinny_gemm(c.p<float>(w.th), c.a<float>(e->time_w2), c.a<float>(e->time_b2), c.p<float>(w.temb), nfe, D, D,
                                 0, 0, s));
        RC(f5_launch_skinny_gemm(c.p<float>(w.temb), c.a<float>(e->ada_w), c.a<float>(e->ada_b), c.p<float>(w.mod), nfe,
                                 (6 * L + 2) * D, D, 1, 0, s));
    }
    const int fold = ln_fold_state(c);
    if (fold < 0) return 2;
    if (fold && nfe > 0) {
        // c1 = W (1 + scale), c2 = W shift + bias for every evaluation and block: QKV against (scale_msa, shift_msa), FF1 against
        // (scale_mlp, shift_mlp); mod rows are [shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp] (dit.py:53-58)
        const int FF = cf.ff_dim;
        const size_t vstride = (size_t)(6 * L + 2) * D, ostride = (size_t)L * (3 * D + FF);
        // the 22 blocks of a projection in ONE launch when their weights sit at a uniform stride in the arena (they do: every block
        // allocates the same tensors in the same order), else block by block
        const BlockW& b0 = e->blocks[0];
        bool uniform = L >= 2;
        const long sq = L >= 2 ? (long)e->blocks[1].qkv.hi - (long)b0.qkv.hi : 0, sf = L >= 2 ? (long)e->blocks[1].ff1.hi - (long)b0.ff1.hi : 0;
        const long sbq = L >= 2 ? (long)e->blocks[1].bqkv - (long)b0.bqkv : 0, sbf = L >= 2 ? (long)e->blocks[1].bff1 - (long)b0.bff1 : 0;
        for (int i = 1; i < L && uniform; ++i) {
            const BlockW& bw = e->blocks[i];
            uniform = (long)bw.qkv.hi - (long)b0.qkv.hi == i * sq && (long)bw.ff1.hi - (long)b0.ff1.hi == i * sf &&
                      (long)bw.bqkv - (long)b0.bqkv == i * sbq && (long)bw.bff1 - (long)b0.bff1 == i * sbf && bw.qkv.ld == b0.qkv.ld &&
                      bw.ff1.ld == b0.ff1.ld;
        }
        uniform = uniform && sq > 0 && sf > 0 && sbq > 0 && sbf > 0 && sq % 8 == 0 && sf % 8 == 0 && sbq % 4 == 0 && sbf % 4 == 0;
        const int nlaunch = uniform ? 1 : L, count = uniform ? L : 1;
        for (int i = 0; i < nlaunch; ++i) {
            const BlockW& bw = e->blocks[i];
            const float* m6 = c.p<float>(w.mod) + (size_t)i * 6 * D;
            float* c1 = c.p<float>(w.foldc[0]) + (size_t)i * (3 * D + FF);
            float* c2 = c.p<float>(w.foldc[1]) + (size_t)i * (3 * D + FF);
            RC(K.fold_consts(c.wm(bw.qkv, 0), bw.qkv.ld, c.a<float>(bw.bqkv), m6 + D, m6, vstride, nfe, c1, c2, ostride, 3 * D, D, s, count,
                             (size_t)sq / 2, (size_t)sbq / 4, (size_t)6 * D, (size_t)(3 * D + FF)));
            RC(K.fold_consts(c.wm(bw.ff1, 0), bw.ff1.ld, c.a<float>(bw.bff1), m6 + 4 * D, m6 + 3 * D, vstride, nfe, c1 + 3 * D, c2 + 3 * D,
                             ostride, FF, D, s, count, (size_t)sf / 2, (size_t)sbf / 4, (size_t)6 * D, (size_t)(3 * D + FF)));
        }
    }
    RC(f5_launch_rope_table(c.p<float>(w.rope_cos), c.p<float>(w.rope_sin), c.N, cf.dim_head, s));
    {
        float* t = c.p<float>(w.rope_t);
        const size_t tn = (size_t)32 * c.N;
        const float qf = q_premul_factor(e);
        RC(f5_launch_rope_table_g4(t, t + 2 * tn, c.N, cf.dim_head, qf != 0.0f ? qf : 1.0f, s));      // q table | k table, 64 N floats each
    }
    RC(f5_launch_rowkeep(c.p<int>(w.dur2), c.p<uint8_t>(w.rowkeep), 2 * c.B, c.N, s));
    for (int p = 0; p < e->np; ++p) RC(K.zero_vt_pad(c.pb(w.vt, p), (size_t)2 * c.B * cf.heads * 64, c.N, c.npad, s));

    // --- text path for both branches (dit.py:196-229, convnext_v2.py:46-54)
    // conv_layers == 0 (dit.py:193-194): the plain embedding, no positional table and no masking; text_mask_padding == 0: no masking
    RC(f5_launch_text_embed(c.p<int>(w.text), c.nt, c.a<float>(e->text_table), cf.conv_layers > 0 ? c.a<float>(e->text_pos) : nullptr,
                            cf.text_max_pos, c.p<float>(w.te[0]), c.p<int>(w.ids), c.p<uint8_t>(w.keep), c.B, c.N, Dt,
                            (cf.conv_layers > 0 && cf.text_mask_padding) ? 1 : 0, s));
    int cur = 0;
    for (int i = 0; i < cf.conv_layers; ++i) {
        const TextBlockW& t = e->tblocks[i];
        RC(K.dwconv_ln(c.p<float>(w.te[cur]), c.a<float>(t.dw_w), c.a<float>(t.dw_b), c.a<float>(t.ln_w),
                               c.a<float>(t.ln_b), c.pb(w.tln, 0), c.pb(w.tln, 1), 2 * c.B, c.N, Dt, 1e-6f, s));
        F5GemmArgs g1 = gemm_base(c, c.pb(w.tln, 0), c.pb(w.tln, 1), Dt, t.pw1, M2, TF, Dt, c.a<float>(t.b1));
        g1.out_f32 = c.p<float>(w.tg);
        g1.ldo = TF;
        RC(K.gemm(g1, EPI_GELU_ERF, s));
        RC(K.grn(c.p<float>(w.tg), c.a<float>(t.gamma), c.a<float>(t.beta), c.p<float>(w.grn_partial),
                         c.p<float>(w.grn_nx), c.pb(w.tg2, 0), c.pb(w.tg2, 1), 2 * c.B, c.N, TF, s));
        F5GemmArgs g2 = gemm_base(c, c.pb(w.tg2, 0), c.pb(w.tg2, 1), TF, t.pw2, M2, Dt, TF, c.a<float>(t.b2));
        g2.out_f32 = c.p<float>(w.te[cur ^ 1]);
        g2.ldo = Dt;
        g2.resid = c.p<float>(w.te[cur]);
        g2.ldres = Dt;
        g2.rowkeep = c.p<uint8_t>(w.keep);
        RC(K.gemm(g2, EPI_RESID_KEEP, s));
        cur ^= 1;
    }
    // --- hoisted part of the input projection: Hc = [cond | text] * Wct^T + b   (dit.py:249-250)
    RC(K.pack_cond_text(c.p<float>(w.cond), c.p<int>(w.lens), c.p<float>(w.te[cur]), c.pb(w.ct, 0), c.pb(w.ct, 1), c.B,
                                c.N, cf.mel_dim, Dt, e->opt.null_keeps_cond, s));
    F5GemmArgs gh = gemm_base(c, c.pb(w.ct, 0), c.pb(w.ct, 1), 128 + Dt, e->wct, M2, D, 128 + Dt, c.a<float>(e->bproj));
    gh.out_f32 = c.p<float>(w.hc);
    gh.ldo = D;
    RC(K.gemm(gh, EPI_F32, s));
    return 0;
}

// one batched (cond + null) DiT forward; input = xin (bf16 padded state), modulation row `j`
// f5_debug_set_ln_fusion: LN-modulate fused behind the residual GEMMs of small-tile launches (gemm.hpp ln_counter).  OFF by
// default: bit-identical, but measured SLOWER at batch 1 (88.5 / 90.3 ms per sample against 75.1 / 78.6 on the same boxes,
// profiles/r02/ln_fusion_ab.txt): a cross-XCD hand-over inside a kernel is three dependent trips to the memory side (write-through
// ack, counter atomic, agent-scope re-read: ~10 us per GEMM) where the separate LN launch costs 5.2 us.
static int run_dit(const Ctx& c, int j) {
    const f5_engine* e = c.e;
    const f5_config& cf = e->cfg;
    const Workspace& w = c.w;
    const int D = cf.dim, FF = cf.ff_dim, L = cf.depth, H = cf.heads;
    const int M1 = c.B * c.N, M = c.nb * M1;
    hipStream_t s = c.s;
    const Ops& K = c.ops;
    const float* mod = c.p<float>(w.mod) + (size_t)j * (6 * L + 2) * D;
    const uint8_t* rowkeep = c.use_mask ? c.p<uint8_t>(w.rowkeep) : nullptr;
    const int* kvlen = c.use_mask ? c.p<int>(w.dur2) : nullptr;

    // x = Wx * x_t + Hc  (dit.py:250), both branches share x_t
    F5GemmArgs g0 = gemm_base(c, c.pb(w.xin, 0), c.pb(w.xin, 1), 128, e->wx, M, D, 128, nullptr);
    g0.a_row_mod = M1;
    g0.addrows = c.p<float>(w.hc);
    g0.ldadd = D;
    g0.out_f32 = c.p<float>(w.x);
    g0.ldo = D;
    g0.out_bf[0] = c.pb(w.xb, 0);
    g0.out_bf[1] = c.pb(w.xb, 1);
    g0.ldob = D;
    RC(K.gemm(g0, EPI_ADDROWS, s));

    // x += conv_pos_embed(x)  (dit.py:251)
    F5ConvPosArgs cp;
    memset(&cp, 0, sizeof(cp));
    cp.B = c.nb * c.B;
    cp.seq_len = c.N;
    cp.C = D;
    cp.groups = cf.conv_pos_groups;
    cp.taps = cf.conv_pos_kernel;
    cp.ld = D;
    cp.ldo = D;
    cp.nseg = c.nseg();
    cp.in[0] = c.pb(w.xb, 0);
    cp.in[1] = c.pb(w.xb, 1);
    cp.W[0] = c.wm(e->conv_w[0], 0);
    cp.W[1] = c.wm(e->conv_w[0], 1);
    cp.bias = c.a<float>(e->conv_b[0]);
    cp.mode = 0;
    cp.out_bf[0] = c.pb(w.c1, 0);
    cp.out_bf[1] = c.pb(w.c1, 1);
    RC(K.convpos(cp, s));
    cp.in[0] = c.pb(w.c1, 0);
    cp.in[1] = c.pb(w.c1, 1);
    cp.W[0] = c.wm(e->conv_w[1], 0);
    cp.W[1] = c.wm(e->conv_w[1], 1);
    cp.bias = c.a<float>(e->conv_b[1]);
    cp.mode = 1;
    cp.out_bf[0] = cp.out_bf[1] = nullptr;
    cp.out_f32 = c.p<float>(w.x);
    RC(K.convpos(cp, s));

    // q leaves the QKV epilogue multiplied by softmax_scale * log2(e) (one rounding, like the unscaled q): the attention kernels
    // then get their scores in exp2 units straight from the matrix cores (attention.hip v2f).  Single-segment operand modes only;
    // bf16x3 keeps the unscaled q (hi / lo split of the reference-exact value).  f5_debug_set_q_premul(0) = A/B.
    const float qpre = q_premul_factor(e);
    for (int i = 0; i < L && e->prec == F5_PREC_MXFP8; ++i) {
        // MX-fp8 block: the four GEMMs run on e4m3 operands with E8M0 block scales (v_mfma_scale_f32_32x32x64_f8f6f4); their A
        // operands are produced directly in that format by the LN kernel, the attention epilogue and the GELU epilogue.
        // q / k / V^T and the attention itself stay bf16, the residual stream fp32.
        const BlockW& bw = e->blocks[i];
        const float* m6 = mod + (size_t)i * 6 * D;
        auto f8args = [&](const uint8_t* a8, const uint8_t* as, int lda, const MatF8& wm, int N_, int K_, const float* bias) {
            F5GemmArgs g;
            memset(&g, 0, sizeof(g));
            g.A8 = a8;
            g.As = as;
            g.lda8 = lda;
            g.W8 = (const uint8_t*)(e->arena + wm.q);
            g.Ws = (const uint8_t*)(e->arena + wm.s);
            g.ldw8 = wm.ld;
            g.M = M;
            g.N = N_;
            g.K = K_;
            g.nseg = 1;
            g.bias = bias;
            g.debug_flags = e->opt.gemm_flags;
            return g;
        };
        RC(f5_launch_ln_modulate_f8(c.p<float>(w.x), m6 + D, m6, c.p<uint8_t>(w.h8), c.p<uint8_t>(w.h8s), M, D, 1e-6f, s));
        F5GemmArgs gq = f8args(c.p<uint8_t>(w.h8), c.p<uint8_t>(w.h8s), D, bw.qkv8, 3 * D, D, c.a<float>(bw.bqkv));
        gq.out_bf[0] = c.pb(w.qk, 0);
        gq.ldob = 2 * D;
        gq.rope_cos = c.p<float>(w.rope_cos);
        gq.rope_sin = c.p<float>(w.rope_sin);
        gq.seq_len = c.N;
        gq.npad = c.npad;
        gq.heads = H;
        gq.dmodel = D;
        gq.vt[0] = c.pb(w.vt, 0);
        gq.q_premul = qpre;
        RC(f5_launch_gemm_f8(gq, EPI_QKV_ROPE, s));

        F5AttnArgs at;
        memset(&at, 0, sizeof(at));
        at.qk[0] = c.pb(w.qk, 0);
        at.vt[0] = c.pb(w.vt, 0);
        at.out8 = c.p<uint8_t>(w.ao8);
        at.out8s = c.p<uint8_t>(w.ao8s);
        at.ldo8 = D;
        at.kv_len = kvlen;
        at.B = c.nb * c.B;
        at.H = H;
        at.seq_len = c.N;
        at.npad = c.npad;
        at.ldqk = 2 * D;
        at.ldo = D;
        at.dmodel = D;
        at.hp = 0;
        at.scale = 1.0f / sqrtf((float)cf.dim_head);
        at.q_prescaled = qpre != 0.0f;
        at.pipe = e->opt.attn_pipe;
        RC(K.attention(at, s));

        F5GemmArgs go = f8args(c.p<uint8_t>(w.ao8), c.p<uint8_t>(w.ao8s), D, bw.o8, D, D, c.a<float>(bw.bo));
        go.out_f32 = c.p<float>(w.x);
        go.ldo = D;
        go.gate = m6 + 2 * D;
        go.rowkeep = rowkeep;
        RC(f5_launch_gemm_f8(go, EPI_RESID_GATE, s));

        RC(f5_launch_ln_modulate_f8(c.p<float>(w.x), m6 + 4 * D, m6 + 3 * D, c.p<uint8_t>(w.h8), c.p<uint8_t>(w.h8s), M, D, 1e-6f, s));
        F5GemmArgs g1 = f8args(c.p<uint8_t>(w.h8), c.p<uint8_t>(w.h8s), D, bw.ff1_8, FF, D, c.a<float>(bw.bff1));
        g1.out8 = c.p<uint8_t>(w.ffh8);
        g1.out8s = c.p<uint8_t>(w.ffh8s);
        g1.ldo8 = FF;
        RC(f5_launch_gemm_f8(g1, EPI_GELU_TANH, s));
        F5GemmArgs g2 = f8args(c.p<uint8_t>(w.ffh8), c.p<uint8_t>(w.ffh8s), FF, bw.ff2_8, D, FF, c.a<float>(bw.bff2));
        g2.out_f32 = c.p<float>(w.x);
        g2.ldo = D;
        g2.gate = m6 + 5 * D;
        RC(f5_launch_gemm_f8(g2, EPI_RESID_GATE, s));
    }
    // LN-modulate fused behind the residual GEMMs where the launch is small-tile (gemm.hpp ln_counter): at batch 1 that removes
    // 2 of the 7 launches of a block; h_ready = the h operand of the next QKV / final projection already exists
    const float* mf = mod + (size_t)L * 6 * D;  // final adaLN: (scale, shift) order, dit.py:287
    bool h_ready = false;
    auto fuse_ln = [&](F5GemmArgs& g, const float* scale, const float* shift) -> bool {
        if (!e->opt.fuse_ln || w.lncnt_words == 0 || !K.gemm_resid_ln_fusable(g)) return false;
        g.ln_counter = c.p<int>(w.lncnt);
        g.ln_scale = scale;
        g.ln_shift = shift;
        g.ln_out[0] = c.pb(w.h, 0);
        g.ln_out[1] = c.pb(w.h, 1);
        g.ln_eps = 1e-6f;
        return true;
    };
    // LN fold: the residual GEMMs leave x (1 + scale) in the operand type in `h` and the row sums in `lnstats`; QKV / FF1 finish the
    // LN in their epilogues with the constants of this evaluation (run_prep)
    const int fold_state = ln_fold_state(c);
    if (fold_state < 0) return 2;
    const bool fold = fold_state >= 1;
    const float* fc1 = c.p<float>(w.foldc[0]) + (size_t)j * L * (3 * D + FF);
    const float* fc2 = c.p<float>(w.foldc[1]) + (size_t)j * L * (3 * D + FF);
    // the rows' shift m lives in lnmean[cur_mean]; with fold_stats a consumer reads it, writes the rows' new means into the other half and
    // the halves swap; with a f5_fold_rows launch the kernel updates lnmean[0] in place
    // statistics form: always on the batch-1-sized route (state 2: its kernels have no other), by option on the staged kernels (measured
    // slower there: every tile of a multi-round launch repeats the merge -- +1.9 % at batch 32, +1.6 % at 8, profiles/r06)
    const bool stats_form = fold_state == 2 || (fold && e->opt.fold_stats == 1 && D == 1024);
    int cur_mean = 0;
    auto lnmean = [&](int which) { return c.p<float>(w.lnmean) + (size_t)which * M; };
    auto fold_consumer = [&](F5GemmArgs& g, int block, int col0) {
        if (stats_form) {
            g.fold_stats = c.p<float>(w.lnstats);
            g.fold_stats_ld = M;
            g.fold_shift = lnmean(cur_mean);
            g.fold_mean_out = lnmean(cur_mean ^ 1);
            g.fold_eps = 1e-6f;
            cur_mean ^= 1;
        } else {
            g.fold_rowf = c.p<float>(w.lnrowf);
        }
        g.fold_c1 = fc1 + (size_t)block * (3 * D + FF) + col0;
        g.fold_c2 = fc2 + (size_t)block * (3 * D + FF) + col0;
    };
    auto fold_producer = [&](F5GemmArgs& g, const float* next_scale) {
        g.x16_out = c.pb(w.h, 0);
        g.ldx16 = D;
        g.x16_scale = next_scale;
        g.x16_shift = lnmean(cur_mean);            // the row's mean at the previous LayerNorm (LN kernel of block 0 / the last consumer / fold_rows)
        g.stats_out = c.p<float>(w.lnstats);
        g.stats_ld = M;
        g.x16_overflow = c.p<int>(w.status);
    };
    auto fold_rows = [&]() {
        if (stats_form) return 0;
        return K.fold_rows(c.p<float>(w.lnstats), M, D / 64, M, 1e-6f, c.p<float>(w.lnrowf), c.p<float>(w.lnmean), s);
    };
    bool h_folded = false;                       // `h` holds x (1 + scale) + row sums (fold) instead of the finished LN-modulate
    for (int i = 0; i < L && e->prec != F5_PREC_MXFP8; ++i) {
        const BlockW& bw = e->blocks[i];
        const float* m6 = mod + (size_t)i * 6 * D;  // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
        if (!h_ready) RC(K.ln_modulate(c.p<float>(w.x), m6 + D, m6, c.pb(w.h, 0), c.pb(w.h, 1), M, D, 1e-6f, s, fold ? c.p<float>(w.lnmean) : nullptr));
        F5GemmArgs gq = gemm_base(c, c.pb(w.h, 0), c.pb(w.h, 1), D, bw.qkv, M, 3 * D, D, c.a<float>(bw.bqkv));
        if (h_folded) fold_consumer(gq, i, 0);
        gq.out_bf[0] = c.pb(w.qk, 0);
        gq.out_bf[1] = c.pb(w.qk, 1);
        gq.ldob = 2 * D;
        gq.rope_cos = c.p<float>(w.rope_cos);
        gq.rope_sin = c.p<float>(w.rope_sin);
        gq.seq_len = c.N;
        gq.npad = c.npad;
        gq.heads = H;
        gq.dmodel = D;
        gq.vt[0] = c.pb(w.vt, 0);
        gq.vt[1] = c.pb(w.vt, 1);
        gq.q_premul = qpre;
        if (e->opt.qkv_tr) {                             // pair-major tables (the q pair carries qpre, or 1): used by the 256x256 kernel
            const float* t = c.p<float>(w.rope_t);
            const size_t tn = (size_t)32 * c.N;
            gq.rope_g4q = t;
            gq.rope_g4k = t + 2 * tn;
        }
        RC(K.gemm(gq, EPI_QKV_ROPE, s));

        F5AttnArgs at;
        memset(&at, 0, sizeof(at));
        for (int p = 0; p < 2; ++p) {
            at.qk[p] = c.pb(w.qk, p);
            at.vt[p] = c.pb(w.vt, p);
            at.out[p] = c.pb(w.ao, p);
        }
        at.kv_len = kvlen;
        at.B = c.nb * c.B;
        at.H = H;
        at.seq_len = c.N;
        at.npad = c.npad;
        at.ldqk = 2 * D;
        at.ldo = D;
        at.dmodel = D;
        at.hp = e->np == 2;
        at.scale = 1.0f / sqrtf((float)cf.dim_head);
        at.q_prescaled = qpre != 0.0f;
        at.pipe = e->opt.attn_pipe;
        RC(K.attention(at, s));

        F5GemmArgs go = gemm_base(c, c.pb(w.ao, 0), c.pb(w.ao, 1), D, bw.o, M, D, D, c.a<float>(bw.bo));
        go.out_f32 = c.p<float>(w.x);
        go.ldo = D;
        go.gate = m6 + 2 * D;
        go.rowkeep = rowkeep;
        const bool fused_mlp_ln = fuse_ln(go, m6 + 4 * D, m6 + 3 * D);     // h = LN(x) (1 + scale_mlp) + shift_mlp in the same launch
        if (fold) fold_producer(go, m6 + 4 * D);
        RC(K.gemm(go, EPI_RESID_GATE, s));
        if (fold) RC(fold_rows());

        if (!fused_mlp_ln && !fold) RC(K.ln_modulate(c.p<float>(w.x), m6 + 4 * D, m6 + 3 * D, c.pb(w.h, 0), c.pb(w.h, 1), M, D, 1e-6f, s));
        F5GemmArgs g1 = gemm_base(c, c.pb(w.h, 0), c.pb(w.h, 1), D, bw.ff1, M, FF, D, c.a<float>(bw.bff1));
        if (fold) fold_consumer(g1, i, 3 * D);
        g1.out_bf[0] = c.pb(w.ffh, 0);
        g1.out_bf[1] = c.pb(w.ffh, 1);
        g1.ldob = FF;
        RC(K.gemm(g1, EPI_GELU_TANH, s));
        F5GemmArgs g2 = gemm_base(c, c.pb(w.ffh, 0), c.pb(w.ffh, 1), FF, bw.ff2, M, D, FF, c.a<float>(bw.bff2));
        g2.out_f32 = c.p<float>(w.x);
        g2.ldo = D;
        g2.gate = m6 + 5 * D;
        // the LN that follows FF2: the next block's attention LN (scale_msa, shift_msa), or the final one (dit.py:287: scale, shift)
        const float* m6n = m6 + 6 * D;
        h_ready = (i + 1 < L) ? fuse_ln(g2, m6n + D, m6n) : fuse_ln(g2, mf, mf + D);
        h_folded = fold && i + 1 < L;            // (the final LN feeds the small mel projection: it keeps the LN kernel)
        if (h_folded) {
            fold_producer(g2, m6n + D);
            h_ready = true;
        }
        RC(K.gemm(g2, EPI_RESID_GATE, s));
        if (h_folded) RC(fold_rows());
    }
    if (!h_ready) RC(K.ln_modulate(c.p<float>(w.x), mf, mf + D, c.pb(w.h, 0), c.pb(w.h, 1), M, D, 1e-6f, s));
    F5GemmArgs gf = gemm_base(c, c.pb(w.h, 0), c.pb(w.h, 1), D, e->wout, M, cf.mel_dim, D, c.a<float>(e->bout));
    gf.out_f32 = c.p<float>(w.vel);
    gf.ldo = cf.mel_dim;
    RC(K.gemm(gf, EPI_F32, s));
    return 0;
}

// fp16 range detector (op16.hpp): while one of these is alive the 16-bit packers of every launch of the fp16 build report
// saturation into the call's status word (bit F5_STATUS_SATURATED).  Host-side: the pointer travels as a kernel argument, a captured
// graph keeps the status word of the workspace it was captured against (part of the graph key).
struct SatScope {
    int* saved;
    SatScope(const f5_engine* e, int* status_word) : saved(f5hf::f5_sat_flag_host) {
        if (e->prec == F5_PREC_F16 && e->opt.sat_check) f5hf::f5_sat_flag_host = status_word;
    }
    ~SatScope() { f5hf::f5_sat_flag_host = saved; }
};

// steps [i0, i1) of the solve; i0 == 0 also runs the per-call preparation (hoisted tables, text path, first operand pack)
static int run_sample_body(const Ctx& c, const f5_sample_args* a, int i0 = 0, int i1 = 1 << 30) {
    const f5_config& cf = c.e->cfg;
    const Workspace& w = c.w;
    const int mel = cf.mel_dim;
    const size_t M1 = (size_t)c.B * c.N;
    const int per = nfe_per_step(a->method);
    const int nfe = (a->steps - 1) * per;
    hipStream_t s = c.s;
    const Ops& K = c.ops;
    float* traj = c.p<float>(w.traj);
    if (i0 == 0) {
        RC(run_prep(c, nfe));
        RC(K.pack_x(traj, c.pb(w.xin, 0), c.pb(w.xin, 1), (int)M1, mel, s));
    }
    const float* pred = c.p<float>(w.vel);
    const float* nullp = c.nb == 2 ? pred + M1 * mel : nullptr;
    float* kst = c.p<float>(w.kst);
    float* ytmp = c.p<float>(w.ytmp);
    for (int i = i0; i + 1 < a->steps && i < i1; ++i) {
        float* y = traj + (size_t)i * M1 * mel;
        float* ynext = traj + (size_t)(i + 1) * M1 * mel;
        F5OdeArgs o;
        memset(&o, 0, sizeof(o));
        o.pred = pred;
        o.null_pred = nullp;
        o.cfg_ptr = c.p<float>(w.cfgv);   // staged per call: a by-value scalar would be frozen into the cached hipGraph
        o.base = y;
        o.dt_ptr = c.p<float>(w.dt) + i;
        o.divisor = 1.0f;
        o.rows = (int)M1;
        o.mel_dim = mel;
        o.xin_hi = c.pb(w.xin, 0);
        o.xin_lo = c.pb(w.xin, 1);
        if (a->method == F5_EULER) {
            RC(run_dit(c, i));
            o.coef = 1.0f;
            o.out = ynext;
            RC(K.ode_stage(o, s));
        } else if (a->method == F5_MIDPOINT) {
            RC(run_dit(c, 2 * i));
            o.coef = 0.5f;
            o.out = ytmp;
            RC(K.ode_stage(o, s));
            RC(run_dit(c, 2 * i + 1));
            o.coef = 1.0f;
            o.out = ynext;
            RC(K.ode_stage(o, s));
        } else {
            RC(run_dit(c, 4 * i));
            o.coef = 0.5f;
            o.out = ytmp;
            o.kstore = kst;
            RC(K.ode_stage(o, s));
            RC(run_dit(c, 4 * i + 1));
            o.kstore = kst + M1 * mel;
            RC(K.ode_stage(o, s));
            RC(run_dit(c, 4 * i + 2));
            o.coef = 1.0f;
            o.kstore = kst + 2 * M1 * mel;
            RC(K.ode_stage(o, s));
            RC(run_dit(c, 4 * i + 3));
            o.kstore = nullptr;
            o.mode = 1;
            o.divisor = 6.0f;
            o.k1 = kst;
            o.k2 = kst + M1 * mel;
            o.k3 = kst + 2 * M1 * mel;
            o.out = ynext;
            RC(K.ode_stage(o, s));
        }
    }
    return 0;
}

static int check_args(const f5_engine* e, const f5_sample_args* a) {
    F5_REQUIRE(e && a, "null argument");
    F5_REQUIRE(e->finalized, "weights are not finalized");
    F5_REQUIRE(a->B >= 1 && a->N >= 1 && a->nt >= 1 && a->steps >= 1, "bad sizes B=%d N=%d nt=%d steps=%d", a->B, a->N, a->nt,
               a->steps);
    // the RoPE / head-split epilogues step through a 4-row group with a single wrap at the sequence boundary
    F5_REQUIRE(a->N >= 4, "sequences shorter than 4 mel frames are not supported (N=%d)", a->N);
    F5_REQUIRE(a->method >= F5_EULER && a->method <= F5_RK4, "Unknown method: %d", a->method);
    F5_REQUIRE(a->text && a->cond && a->lens && a->durations && a->workspace, "null pointer in f5_sample_args");
    F5_REQUIRE(((uintptr_t)a->workspace & 255) == 0, "workspace must be 256-byte aligned");
    for (int b = 0; b < a->B; ++b) {
        F5_REQUIRE(a->durations[b] >= 1 && a->durations[b] <= a->N, "durations[%d]=%d out of range [1,%d]", b, a->durations[b],
                   a->N);
        F5_REQUIRE(a->lens[b] >= 0 && a->lens[b] <= a->N, "lens[%d]=%d out of range", b, a->lens[b]);
    }
    return 0;
}

// stage the call's inputs into the workspace (outside any graph: source pointers are caller owned)
static int stage_inputs(Ctx& c, const f5_sample_args* a, const float* x_override, const std::vector<float>& tnfe,
                        const std::vector<float>& dts) {
    const Workspace& w = c.w;
    const size_t M1 = (size_t)c.B * c.N;
    const int mel = c.e->cfg.mel_dim;
    hipStream_t s = c.s;
    // host scalars -> workspace as kernel arguments (no host buffer outlives this call, no host synchronisation)
    const size_t nfe1 = tnfe.empty() ? 1 : tnfe.size();
    std::vector<uint32_t> words(w.scal_words, 0u);
    F5_REQUIRE(w.scal_words == (size_t)1 + 3 * c.B + nfe1 + a->steps + 1 + w.lncnt_words && dts.size() <= (size_t)a->steps,
               "internal: scalar staging layout");
    const int fold_state = ln_fold_state(c);
    if (fold_state < 0) return 2;
    words[0] = fold_state >= 1 ? 2u : 0u;            // status word: bit 1 = the LN fold runs in this call, bit 0 is the kernels'
    uint32_t* wd = words.data() + 1;
    memcpy(wd, a->lens, (size_t)c.B * 4);
    for (int b = 0; b < c.B; ++b) wd[c.B + b] = wd[2 * c.B + b] = (uint32_t)a->durations[b];
    if (!tnfe.empty()) memcpy(wd + 3 * c.B, tnfe.data(), tnfe.size() * 4);
    if (!dts.empty()) memcpy(wd + 3 * c.B + nfe1, dts.data(), dts.size() * 4);
    memcpy(wd + 3 * c.B + nfe1 + a->steps, &a->cfg_strength, 4);
    RC(f5_launch_stage_words(words.data(), words.size(), c.p<uint32_t>(w.scal), s));
    // caller-owned device inputs -> workspace (the captured graph only references the workspace and the arena)
    RC(f5_launch_copy_words(a->text, c.ws + w.text, (size_t)c.B * c.nt, s));
    RC(f5_launch_copy_words(a->cond, c.ws + w.cond, M1 * mel, s));
    const float* x0 = x_override ? x_override : a->y0;
    F5_REQUIRE(x0 != nullptr, "initial state (y0) is null");
    RC(f5_launch_copy_words(x0, c.ws + w.traj, M1 * mel, s));
    return 0;
}

static Ctx make_ctx(f5_engine* e, const f5_sample_args* a, hipStream_t s) {
    Ctx c;
    c.e = e;
    c.w = plan_workspace(e, a->B, a->N, a->nt, a->steps, a->method);
    c.ws = (char*)a->workspace;
    c.s = s;
    c.B = a->B;
    c.N = a->N;
    c.nt = a->nt;
    c.nb = a->cfg_strength < 1e-5f ? 1 : 2;
    c.npad = (a->N + 63) / 64 * 64;
    c.use_mask = a->use_mask != 0;
    c.ops = e->ops;
    return c;
}

extern "C" int f5_sample(f5_engine* e, const f5_sample_args* a, void* stream) {
    RC(check_args(e, a));
    F5_REQUIRE(a->y0 && a->t && a->out, "null pointer in f5_sample_args (y0/t/out)");
    hipStream_t s = (hipStream_t)stream;
    Ctx c = make_ctx(e, a, s);
    F5_REQUIRE(a->workspace_bytes >= c.w.total, "workspace too small: %zu < %zu", a->workspace_bytes, c.w.total);
    const int mel = e->cfg.mel_dim;
    const size_t M1 = (size_t)c.B * c.N;
    SatScope sat(e, c.p<int>(c.w.status));

    // function-evaluation times and step sizes in fp32, as the solvers compute them (cfm.py:50-56,76-86,106-114)
    std::vector<float> tnfe, dts;
    for (int i = 0; i + 1 < a->steps; ++i) {
        const float t0 = a->t[i];
        const float dt = a->t[i + 1] - t0;
        dts.push_back(dt);
        if (a->method == F5_EULER) {
            tnfe.push_back(t0);
        } else if (a->method == F5_MIDPOINT) {
            tnfe.push_back(t0);
            tnfe.push_back(t0 + 0.5f * dt);
        } else {
            tnfe.push_back(t0);
            tnfe.push_back(t0 + 0.5f * dt);
            tnfe.push_back(t0 + 0.5f * dt);
            tnfe.push_back(t0 + dt);
        }
    }
    RC(stage_inputs(c, a, nullptr, tnfe, dts));

    // hipGraph cache.  The key is everything a captured node depends on BY VALUE: shapes, solver, branch count, masking and the
    // workspace address.  Per-call scalars (cfg strength, time grid, dt) are read from workspace memory staged above.
    char key[256];
    snprintf(key, sizeof(key), "B%d N%d nt%d st%d m%d nb%d mask%d ln%d qp%d qt%d gf%d ap%d nk%d lf%d sc%d gs%d fs%d ke%d ws%p", c.B, c.N, c.nt, a->steps, a->method, c.nb,
             (int)c.use_mask, e->opt.fuse_ln, e->opt.q_premul, e->opt.qkv_tr, e->opt.gemm_flags, e->opt.attn_pipe, e->opt.null_keeps_cond, e->opt.ln_fold, e->opt.sat_check, e->opt.graph_split, e->opt.fold_stats, g_knob_epoch,
             a->workspace);
    bool graph = a->use_graph == 1;
    if (a->use_graph == F5_GRAPH_AUTO) {
        // a text-to-speech service sees a new (N, nt) on almost every call and capture + instantiate of ~5000 nodes costs more
        // than one eager pass: run a signature eagerly the first time, capture it when it comes back
        bool cached = false;
        for (auto& g : e->graphs) cached |= g.key == key;
        bool seen = false;
        for (auto& k : e->seen) seen |= k == key;
        graph = cached || seen;
        if (!graph) {
            if (e->seen.size() >= 64) e->seen.erase(e->seen.begin());
            e->seen.push_back(key);
        }
    }
    if (graph) {
        GraphEntry* entry = nullptr;
        for (auto& g : e->graphs)
            if (g.key == key) {
                entry = &g;
                g.stamp = ++e->clock;
            }
        if (!entry) {
            // entries captured against another workspace are dead (the caller re-allocated it); then make room (LRU)
            for (size_t i = 0; i < e->graphs.size();) {
                if (e->graphs[i].workspace != a->workspace) {
                    destroy_graph_entry(e->graphs[i]);
                    e->graphs.erase(e->graphs.begin() + i);
                } else {
                    ++i;
                }
            }
            while ((int)e->graphs.size() >= e->graph_cap && !e->graphs.empty()) {
                size_t lru = 0;
                for (size_t i = 1; i < e->graphs.size(); ++i)
                    if (e->graphs[i].stamp < e->graphs[lru].stamp) lru = i;
                destroy_graph_entry(e->graphs[lru]);
                e->graphs.erase(e->graphs.begin() + lru);
            }
            // segments: the whole call, or (graph_split) one per ODE step
            const int nseg = (e->opt.graph_split && a->steps > 2) ? a->steps - 1 : 1;
            std::vector<hipGraphExec_t> execs;
            auto drop = [&]() {
                for (hipGraphExec_t x : execs) (void)hipGraphExecDestroy(x);
            };
            for (int sg = 0; sg < nseg; ++sg) {
                hipGraph_t graph_h = nullptr;
                F5_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                const int rc = nseg == 1 ? run_sample_body(c, a) : run_sample_body(c, a, sg, sg + 1);
                const hipError_t ec = hipStreamEndCapture(s, &graph_h);
                if (rc || ec != hipSuccess) {
                    if (graph_h) (void)hipGraphDestroy(graph_h);
                    drop();
                    if (rc) return rc;
                    F5_HIP_CHECK(ec);
                }
                hipGraphExec_t exec = nullptr;
                const hipError_t ei = hipGraphInstantiate(&exec, graph_h, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph_h);
                if (ei != hipSuccess) {
                    drop();
                    F5_HIP_CHECK(ei);
                }
                execs.push_back(exec);
            }
            hipEvent_t done = nullptr;
            if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) {
                drop();                                       // do not leak the instantiated graphs
                f5_set_error("f5_sample: hipEventCreateWithFlags failed");
                return 3;
            }
            GraphEntry ge{key, execs[0], {}, a->workspace, ++e->clock, done};
            ge.more.assign(execs.begin() + 1, execs.end());
            e->graphs.push_back(std::move(ge));
            entry = &e->graphs.back();
        }
        F5_HIP_CHECK(hipGraphLaunch(entry->exec, s));
        for (hipGraphExec_t x : entry->more) F5_HIP_CHECK(hipGraphLaunch(x, s));
        F5_HIP_CHECK(hipEventRecord(entry->done, s));
    } else {
        RC(run_sample_body(c, a));
    }
    const float* ylast = c.p<float>(c.w.traj) + (size_t)(a->steps - 1) * M1 * mel;
    RC(f5_launch_splice(c.p<float>(c.w.cond), ylast, c.p<int>(c.w.lens), a->out, c.B, c.N, mel, s));
    if (a->trajectory) RC(f5_launch_copy_words(c.ws + c.w.traj, a->trajectory, (size_t)a->steps * M1 * mel, s));
    return 0;
}

// Status word of the last f5_sample / f5_dit_forward that used this workspace (its FIRST 32-bit word, for every shape and solver):
// bit 1 = the LN fold ran in that call; bit 0 = a value of the folded operand (x - m)(1 + scale) did not fit fp16 (precision f16): the
// result is saturated there -- rerun with the engine option ln_fold = 0 (or in bf16).  f5_sample_status synchronises the stream;
// f5_sample_status_async only enqueues the 4-byte copy (flags should be pinned host memory; read it after the caller's own
// synchronisation / event): no host block per call.
extern "C" int f5_sample_status_async(f5_engine* e, const f5_sample_args* a, int* flags, void* stream) {
    F5_REQUIRE(e && a && flags && a->workspace && a->workspace_bytes >= 4, "null argument");
    F5_HIP_CHECK(hipMemcpyAsync(flags, (const char*)a->workspace, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}
// 1 when f5_sample with these arguments runs the LN fold (engine option "ln_fold", sizes, precision, branch count of cfg_strength) --
// the only configuration that can set bit 0 of the status word; host-side, no GPU work.  f5_dit_forward: pass steps = 2, method = F5_EULER.
extern "C" int f5_engine_ln_fold_active(f5_engine* e, const f5_sample_args* a, int* active) {
    F5_REQUIRE(e && a && active, "null argument");
    F5_REQUIRE(a->B >= 1 && a->N >= 1 && a->nt >= 1 && a->steps >= 1 && a->method >= F5_EULER && a->method <= F5_RK4, "bad sizes");
    const Ctx c = make_ctx(e, a, nullptr);
    const int st = ln_fold_state(c);
    if (st < 0) return 2;
    *active = st;
    return 0;
}
extern "C" int f5_sample_status(f5_engine* e, const f5_sample_args* a, int* flags, void* stream) {
    RC(f5_sample_status_async(e, a, flags, stream));
    F5_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

extern "C" int f5_dit_forward(f5_engine* e, const f5_sample_args* a, const float* x, float t, float* pred, float* null_pred,
                              void* stream) {
    RC(check_args(e, a));
    F5_REQUIRE(x && pred, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    f5_sample_args a2 = *a;
    a2.steps = 2;
    a2.method = F5_EULER;
    Ctx c = make_ctx(e, &a2, s);
    F5_REQUIRE(a->workspace_bytes >= c.w.total, "workspace too small: %zu < %zu", a->workspace_bytes, c.w.total);
    const int mel = e->cfg.mel_dim;
    const size_t M1 = (size_t)c.B * c.N;
    SatScope sat(e, c.p<int>(c.w.status));
    std::vector<float> tnfe(1, t), dts(1, 0.0f);
    RC(stage_inputs(c, &a2, x, tnfe, dts));
    const Ops& K = c.ops;
    RC(run_prep(c, 1));
    RC(K.pack_x(c.p<float>(c.w.traj), c.pb(c.w.xin, 0), c.pb(c.w.xin, 1), (int)M1, mel, s));
    RC(run_dit(c, 0));
    RC(f5_launch_copy_words(c.ws + c.w.vel, pred, M1 * mel, s));
    if (null_pred && c.nb == 2) RC(f5_launch_copy_words(c.ws + c.w.vel + M1 * mel * 4, null_pred, M1 * mel, s));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// per-op entry points
// ------------------------------------------------------------------------------------------------
// Test hooks.  They live in both kernel builds (bf16 / fp16 operands) and are process-wide: meant for the op-level tests and the
// A/B tools, which set them once; a change bumps g_knob_epoch, so no cached hipGraph captured under other values is replayed.
// The PRODUCT library keeps only the hooks that choose among the kernels sample() itself can reach (tile shapes, attention
// workgroup shapes, conv-pos pipeline step, tile numbering); everything that selects an experiment exists in the lab build only
// (F5_LAB=1 bash build.sh: superseded / rejected kernels, ablations, stream-K, priorities).
#define F5_DECL_KNOB(v) namespace f5bf { extern int v; } namespace f5hf { extern int v; }
#define F5_SET_BOTH(v, x) do { f5bf::v = (x); f5hf::v = (x); ++g_knob_epoch; } while (0)
#ifndef F5_LAB
#define F5_LAB 0
#endif
F5_DECL_KNOB(f5_convpos_tps)
F5_DECL_KNOB(f5_convpos_xcd_map)
F5_DECL_KNOB(f5_attn_wide)
F5_DECL_KNOB(f5_attn_kvsplit)
F5_DECL_KNOB(f5_attn_pipe)
F5_DECL_KNOB(f5_gemm_ring_default)
F5_DECL_KNOB(f5_gemm_order)
F5_DECL_KNOB(f5_gemm_nband)
F5_DECL_KNOB(f5_gemm_qkv_small_tile)
F5_DECL_KNOB(f5_gemm_debug_flags)
F5_DECL_KNOB(f5_gemm_tile_override)
extern "C" int f5_lab_build(void) { return F5_LAB; }      // 1 = this library also carries the experiments
extern "C" int f5_op_set_operand_type(int fp16) {
    F5_REQUIRE(fp16 == 0 || fp16 == 1, "operand type must be 0 (bf16) or 1 (fp16)");
    g_ops.h = fp16 != 0;
    return 0;
}
extern "C" int f5_op_get_operand_type(void) { return g_ops.h ? 1 : 0; }
// host-side conversions used by f5_load_tensor, exported for the CPU tests (no GPU needed)
extern "C" uint16_t f5_debug_f2h_bits(float f) { return f5_f2h_bits(f); }
extern "C" float f5_debug_h_bits2f(uint16_t h) { return f5_h_bits2f(h); }
extern "C" uint16_t f5_debug_f2bf_bits(float f) { return f5_f2bf_bits(f); }
extern "C" int f5_debug_set_convpos_tps(int v) {
    F5_REQUIRE(v == 0 || v == 1 || v == 2 || v == 4, "conv-pos taps per pipeline step must be 0 (auto), 1, 2 or 4");
    F5_SET_BOTH(f5_convpos_tps, v);
    return 0;
}
extern "C" int f5_debug_set_convpos_xcd_map(int on) {
    F5_SET_BOTH(f5_convpos_xcd_map, on ? 1 : 0);
    return 0;
}
// process DEFAULTS of the per-engine options (f5_engine_set_option changes one engine; engines that already exist keep theirs)
static void warn_defaults_only(const char* hook, const char* option) {
    if (g_live_engines > 0)
        fprintf(stderr, "[f5tts_hip] %s changes the default of NEW engines only: %d engine(s) already exist and keep their value "
                        "(use f5_engine_set_option(e, \"%s\", v) for those)\n", hook, g_live_engines, option);
}
extern "C" int f5_debug_set_ln_fusion(int on) {
    warn_defaults_only("f5_debug_set_ln_fusion", "ln_fusion");
    g_default_options.fuse_ln = on ? 1 : 0;
    return 0;
}
extern "C" int f5_debug_set_qkv_transposed(int on) {
    warn_defaults_only("f5_debug_set_qkv_transposed", "qkv_transposed");
    g_default_options.qkv_tr = on ? 1 : 0;
    return 0;
}
extern "C" int f5_debug_set_q_premul(int on) {
    warn_defaults_only("f5_debug_set_q_premul", "q_premul");
    g_default_options.q_premul = on ? 1 : 0;
    return 0;
}
extern "C" int f5_debug_set_attn_wide(int v) {
#if F5_LAB
    F5_REQUIRE(v >= -1 && v <= 2, "attention wide-workgroup switch must be -1 (auto), 0, 1 (256-query workgroups) or 2 (lab: role-split 512-query workgroups)");
#else
    F5_REQUIRE(v >= -1 && v <= 1, "attention wide-workgroup switch must be -1 (auto), 0 or 1 (256-query workgroups)");
#endif
    F5_SET_BOTH(f5_attn_wide, v);
    return 0;
}
extern "C" int f5_debug_set_attn_pipe(int v) {
    F5_REQUIRE(v == 0 || v == 1, "attention pipelining switch must be 0 (v2f) or 1 (v2p: in-wave software pipeline, one wave per SIMD)");
    F5_SET_BOTH(f5_attn_pipe, v);
    return 0;
}
extern "C" int f5_debug_set_attn_kvsplit(int v) {
    F5_REQUIRE(v == -1 || v == 1 || v == 2 || v == 4, "attention KV split must be -1 (auto), 1, 2 or 4");
    F5_SET_BOTH(f5_attn_kvsplit, v);
    return 0;
}
extern "C" int f5_debug_set_gemm_qkv_tile(int v) {
    F5_REQUIRE(v == 0 || v == 1 || v == 12 || v == 13 || v == 14, "small-M QKV tile must be 0 (auto), 1 (small tiles), 12 / 13 (8-wave 128x256 ring) or 14 (role-split 128x256)");
    F5_SET_BOTH(f5_gemm_qkv_small_tile, v);
    return 0;
}
extern "C" int f5_debug_set_gemm_nband(int v) {
    F5_REQUIRE(v >= 0 && v <= 16, "column-tile band width of the 256x256 tile numbering must be 0 (n fastest) .. 16");
    F5_SET_BOTH(f5_gemm_nband, v);
    return 0;
}
extern "C" int f5_debug_set_gemm_ring(int v) {
    F5_SET_BOTH(f5_gemm_ring_default, v ? 1 : 0);
    return 0;
}
extern "C" int f5_debug_set_gemm_order(int v) {
    F5_REQUIRE(v >= 0 && v <= 3, "gemm order must be 0 (auto), 1 (n fastest), 2 (m fastest) or 3 (band-major where it applies)");
    F5_SET_BOTH(f5_gemm_order, v);
    return 0;
}
// process-wide GEMM flags, OR-ed into every GEMM launch of the process (op-level tests, A/B tools); an engine's own flags are
// f5_engine_set_option(e, "gemm_flags", v)
extern "C" int f5_debug_set_gemm_flags(int v) {
    F5_SET_BOTH(f5_gemm_debug_flags, v);
    return 0;
}
extern "C" int f5_debug_set_gemm_tile(int sel) {
    F5_REQUIRE(sel >= 0 && sel <= 14, "gemm tile override must be 0 (auto) .. 14");
    F5_REQUIRE(F5_LAB || sel != 7, "gemm tile 7 (128x256, two workgroups per CU) exists in the lab build only");
    F5_SET_BOTH(f5_gemm_tile_override, sel);
    return 0;
}
#if F5_LAB
F5_DECL_KNOB(f5_attn_version)
F5_DECL_KNOB(f5_attn_ablation)
F5_DECL_KNOB(f5_attn_variant)
F5_DECL_KNOB(f5_attn_prio)
F5_DECL_KNOB(f5_gemm_big_kernel)
F5_DECL_KNOB(f5_gemm128_pad_lds)
F5_DECL_KNOB(f5_gemm_v3_stagger)
F5_DECL_KNOB(f5_gemm_v3_prio)
F5_DECL_KNOB(f5_gemm_streamk)
extern "C" int f5_debug_set_attn_version(int v) {
    F5_REQUIRE(v >= 1 && v <= 6, "attention version must be 1..6");
    F5_SET_BOTH(f5_attn_version, v);
    return 0;
}
extern "C" int f5_debug_set_attn_variant(int v) {
    F5_REQUIRE(v >= 0 && v <= 63, "attention variant bits: 1 = single-issue softmax VALU, 2 = one workgroup per CU, 4 = 2-D block numbering, 8 = eager rescale, 16 = per-tile maximum (v2w) in the large-grid kernel, 32 = role-split kernel keeps Q in LDS");
    F5_SET_BOTH(f5_attn_variant, v);
    return 0;
}
extern "C" int f5_debug_set_attn_ablation(int v) {
    F5_SET_BOTH(f5_attn_ablation, v);
    return 0;
}
extern "C" int f5_debug_set_attn_prio(int v) {
    F5_REQUIRE(v >= 0 && v <= 2, "attention priority scheme must be 0 (MFMA clusters), 1 (none) or 2 (softmax section)");
    F5_SET_BOTH(f5_attn_prio, v);
    return 0;
}
extern "C" int f5_debug_set_gemm_streamk(int v) {
    F5_REQUIRE(v >= 0 && v <= 2, "stream-K switch must be 0 (off), 1 (full) or 2 (hybrid)");
    if (v != 0) {
        RC(f5bf::f5_gemm_streamk_init());   // scratch on the CURRENT device; call outside of any stream capture
        RC(f5hf::f5_gemm_streamk_init());
    }
    F5_SET_BOTH(f5_gemm_streamk, v);
    return 0;
}
extern "C" int f5_debug_gemm_streamk_error() { return f5bf::f5_gemm_streamk_error() | f5hf::f5_gemm_streamk_error(); }
extern "C" int f5_debug_set_gemm_big_kernel(int v, int stagger_cycles) {
    F5_REQUIRE(v >= 2 && v <= 5, "big GEMM kernel must be 2 (256x256 role-split), 3 (128x256 v3), 4 (256x256 lock-step) or 5 (128x256 prefetching, two per CU)");
    F5_SET_BOTH(f5_gemm_big_kernel, v);
    F5_SET_BOTH(f5_gemm_v3_stagger, stagger_cycles);
    return 0;
}
extern "C" int f5_debug_set_gemm128_pad(int bytes) {
    F5_SET_BOTH(f5_gemm128_pad_lds, bytes);
    return 0;
}
extern "C" int f5_debug_set_gemm_v3_prio(int v) {
    F5_REQUIRE(v >= 0 && v <= 2, "128x256 GEMM priority scheme must be 0 (MFMA clusters), 1 (none) or 2 (epilogue)");
    F5_SET_BOTH(f5_gemm_v3_prio, v);
    return 0;
}
#endif  // F5_LAB

// op-level twins of the LN fold (gemm.hpp fold_*): what run_dit sets on its block GEMMs, for the per-op entry points
static struct {
    const float* next_scale = nullptr;       // producer: f5_op_gemm_resid_gate
    op16_t* x16 = nullptr;
    float* stats_out = nullptr;
    const float* row_shift = nullptr;
    float* ln_mean_out = nullptr;            // f5_op_ln_modulate: the row means (the first folded operand's shift)
    const float* rowf = nullptr;             // consumer: f5_op_gemm (epi 2), f5_op_qkv_rope
    const float* c1 = nullptr;
    const float* c2 = nullptr;
} g_op_fold;
extern "C" int f5_debug_set_op_fold_producer(const float* next_scale, void* x16_out, float* stats_out, const float* row_shift) {
    g_op_fold.next_scale = next_scale;
    g_op_fold.x16 = (op16_t*)x16_out;
    g_op_fold.stats_out = stats_out;
    g_op_fold.row_shift = row_shift;
    return 0;
}
extern "C" int f5_debug_set_op_ln_mean_out(float* mean_out) {
    g_op_fold.ln_mean_out = mean_out;
    return 0;
}
// op-level tests: where the 16-bit packers of the f5_op_* launches that follow report saturation (fp16 operand build; null = nowhere)
extern "C" int f5_debug_set_op_sat_flag(int* flag) {
    f5hf::f5_sat_flag_host = flag;
    return 0;
}
static int* g_op_fold_overflow = nullptr;
extern "C" int f5_debug_set_op_fold_overflow_flag(int* flag) {
    g_op_fold_overflow = flag;
    return 0;
}
extern "C" int f5_debug_set_op_fold_consumer(const float* rowf, const float* c1, const float* c2) {
    g_op_fold.rowf = rowf;
    g_op_fold.c1 = c1;
    g_op_fold.c2 = c2;
    return 0;
}
// the statistics form (gemm.hpp fold_stats): the consumers of the f5_op_* launches that follow merge the producer's slice statistics
// themselves; stats = the producer's stats_out ([16][ld][2]), shift = what it subtracted (or NULL), mean_out = where column tile 0
// writes the rows' means (or NULL).  Takes the place of `rowf` of f5_debug_set_op_fold_consumer (set that one with rowf = NULL).
static struct { const float* stats; int ld; const float* shift; float* mean_out; } g_op_fold_stats = {nullptr, 0, nullptr, nullptr};
extern "C" int f5_debug_set_op_fold_stats(const float* stats, int ld, const float* shift, float* mean_out) {
    g_op_fold_stats = {stats, ld, shift, mean_out};
    return 0;
}
static void op_fold_consumer(F5GemmArgs& g) {
    g.fold_rowf = g_op_fold.rowf;
    g.fold_c1 = g_op_fold.c1;
    g.fold_c2 = g_op_fold.c2;
    if (g_op_fold.rowf == nullptr) {
        g.fold_stats = g_op_fold_stats.stats;
        g.fold_stats_ld = g_op_fold_stats.ld;
        g.fold_shift = g_op_fold_stats.shift;
        g.fold_mean_out = g_op_fold_stats.mean_out;
        g.fold_eps = 1e-6f;
    }
}
extern "C" int f5_op_fold_rows(const float* stats, int nslice, int M, float* rowf, float* row_shift, void* stream) {
    return g_ops.fold_rows(stats, M, nslice, M, 1e-6f, rowf, row_shift, (hipStream_t)stream);
}
extern "C" int f5_op_fold_consts(const void* w_hi, int ldw, const float* bias, const float* scale, const float* shift, size_t vec_stride,
                                 int nvec, float* c1, float* c2, size_t out_stride, int N, int K, void* stream) {
    return g_ops.fold_consts((const op16_t*)w_hi, ldw, bias, scale, shift, vec_stride, nvec, c1, c2, out_stride, N, K, (hipStream_t)stream);
}

extern "C" int f5_op_gemm(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                          float* out_f32, void* out_bf_hi, void* out_bf_lo, int M, int N, int K, int lda, int ldw, int ldo,
                          int nseg, int epi, void* stream) {
    F5_REQUIRE(epi == EPI_F32 || epi == EPI_BF16 || epi == EPI_GELU_TANH || epi == EPI_GELU_ERF || epi == EPI_GELU_ERF_BF16,
               "f5_op_gemm supports epilogues 0-3 and 8 only");
    F5GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = (const op16_t*)a_hi;
    g.A[1] = (const op16_t*)a_lo;
    g.W[0] = (const op16_t*)w_hi;
    g.W[1] = (const op16_t*)w_lo;
    g.lda = lda;
    g.ldw = ldw;
    g.M = M;
    g.N = N;
    g.K = K;
    g.nseg = nseg;
    g.bias = bias;
    g.out_f32 = out_f32;
    g.ldo = ldo;
    g.out_bf[0] = (op16_t*)out_bf_hi;
    g.out_bf[1] = (op16_t*)out_bf_lo;
    g.ldob = ldo;
    if (epi == EPI_GELU_TANH && (g_op_fold.rowf != nullptr || g_op_fold_stats.stats != nullptr)) op_fold_consumer(g);
    return g_ops.gemm(g, epi, (hipStream_t)stream);
}

// MX-fp8 GEMM: A8/W8 e4m3 [rows][ld] + E8M0 scales [rows][K/32].  epi 0: out_f32 = acc + bias; 1: out_bf = bf16(acc + bias);
// 2: (out8, out8s) = MX-fp8(gelu_tanh(acc + bias)); 4: out_f32 += gate * ((acc + bias) * keep[row])
extern "C" int f5_op_gemm_f8(const void* a8, const void* a_scales, const void* w8, const void* w_scales, const float* bias,
                             const float* gate, const uint8_t* rowkeep, float* out_f32, void* out_bf, void* out8, void* out8_scales,
                             int M, int N, int K, int lda8, int ldw8, int ldo, int epi, void* stream) {
    F5_REQUIRE(epi == EPI_F32 || epi == EPI_BF16 || epi == EPI_GELU_TANH || epi == EPI_RESID_GATE,
               "f5_op_gemm_f8 supports epilogues 0, 1, 2 and 4");
    F5GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A8 = (const uint8_t*)a8;
    g.As = (const uint8_t*)a_scales;
    g.W8 = (const uint8_t*)w8;
    g.Ws = (const uint8_t*)w_scales;
    g.lda8 = lda8;
    g.ldw8 = ldw8;
    g.M = M;
    g.N = N;
    g.K = K;
    g.nseg = 1;
    g.bias = bias;
    g.gate = gate;
    g.rowkeep = rowkeep;
    g.out_f32 = out_f32;
    g.ldo = ldo;
    g.out_bf[0] = (op16_t*)out_bf;
    g.ldob = ldo;
    g.out8 = (uint8_t*)out8;
    g.out8s = (uint8_t*)out8_scales;
    g.ldo8 = ldo;
    return f5_launch_gemm_f8(g, epi, (hipStream_t)stream);
}
extern "C" int f5_op_quantize_mx(const float* x, int ldx, void* q, int ldq, void* scales, int rows, int cols, void* stream) {
    return f5_launch_quantize_mx(x, ldx, (uint8_t*)q, ldq, (uint8_t*)scales, rows, cols, (hipStream_t)stream);
}

// group-major twins of the rotation tables ([dim_head/4][seq_len][4] = (cos, cos, sin, sin) of two neighbouring pairs; the q table
// multiplied by qscale; 16 seq_len dim_head/4 bytes each, 16-byte aligned) and an op-level hook that hands them to f5_op_qkv_rope: with
// both set (and the q factor of f5_debug_set_op_q_premul folded into the q table by the caller) the staged kernels accumulate the
// q / k tiles transposed, as sample() does.  Null = off.
static const float* g_op_rope_t[2] = {nullptr, nullptr};
extern "C" int f5_debug_set_op_rope_tables_g4(const float* tq, const float* tk) {
    g_op_rope_t[0] = tq;
    g_op_rope_t[1] = tk;
    return 0;
}
// op-level twin of the engine's q pre-multiplication (run_dit): when set (single-segment operands only), f5_op_qkv_rope scales
// the q columns by this factor and f5_op_attention treats q as already carrying scale * log2(e).  0 (default) = plain q.
static float g_op_q_premul = 0.0f;
extern "C" int f5_debug_set_op_q_premul(float v) {
    g_op_q_premul = v;
    return 0;
}
extern "C" int f5_op_attention(const void* qk_hi, const void* qk_lo, const void* vt_hi, const void* vt_lo, void* out_hi,
                               void* out_lo, const int32_t* kv_len, int B, int H, int seq_len, int npad, int dmodel, float scale,
                               int hp, void* stream) {
    F5AttnArgs at;
    memset(&at, 0, sizeof(at));
    at.qk[0] = (const op16_t*)qk_hi;
    at.qk[1] = (const op16_t*)qk_lo;
    at.vt[0] = (const op16_t*)vt_hi;
    at.vt[1] = (const op16_t*)vt_lo;
    at.out[0] = (op16_t*)out_hi;
    at.out[1] = (op16_t*)out_lo;
    at.kv_len = kv_len;
    at.B = B;
    at.H = H;
    at.seq_len = seq_len;
    at.npad = npad;
    at.ldqk = 2 * dmodel;
    at.ldo = dmodel;
    at.dmodel = dmodel;
    at.hp = hp;
    at.scale = scale;
    at.q_prescaled = g_op_q_premul != 0.0f && hp == 0;
    at.pipe = -1;                                     // the process default (f5_debug_set_attn_pipe)
    return g_ops.attention(at, (hipStream_t)stream);
}

extern "C" int f5_op_qkv_rope(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                              const float* rope_cos, const float* rope_sin, void* qk_hi, void* qk_lo, void* vt_hi, void* vt_lo,
                              int B, int seq_len, int npad, int heads, int dmodel, int nseg, void* stream) {
    F5GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = (const op16_t*)a_hi;
    g.A[1] = (const op16_t*)a_lo;
    g.W[0] = (const op16_t*)w_hi;
    g.W[1] = (const op16_t*)w_lo;
    g.lda = dmodel;
    g.ldw = dmodel;
    g.M = B * seq_len;
    g.N = 3 * dmodel;
    g.K = dmodel;
    g.nseg = nseg;
    g.bias = bias;
    g.out_bf[0] = (op16_t*)qk_hi;
    g.out_bf[1] = (op16_t*)qk_lo;
    g.ldob = 2 * dmodel;
    g.rope_cos = rope_cos;
    g.rope_sin = rope_sin;
    g.seq_len = seq_len;
    g.npad = npad;
    g.heads = heads;
    g.dmodel = dmodel;
    g.q_premul = nseg == 1 ? g_op_q_premul : 0.0f;
    g.rope_g4q = g_op_rope_t[0];
    g.rope_g4k = g_op_rope_t[1];
    g.vt[0] = (op16_t*)vt_hi;
    g.vt[1] = (op16_t*)vt_lo;
    if (g_op_fold.rowf != nullptr || g_op_fold_stats.stats != nullptr) op_fold_consumer(g);
    return g_ops.gemm(g, EPI_QKV_ROPE, (hipStream_t)stream);
}

extern "C" int f5_op_rope_table_g4(float* tq, float* tk, int seq_len, int dim_head, float qscale, void* stream) {
    return f5_launch_rope_table_g4(tq, tk, seq_len, dim_head, qscale, (hipStream_t)stream);
}
extern "C" int f5_op_rope_table(float* cos_t, float* sin_t, int seq_len, int dim_head, void* stream) {
    return f5_launch_rope_table(cos_t, sin_t, seq_len, dim_head, (hipStream_t)stream);
}

extern "C" int f5_op_convpos(const void* in_hi, const void* in_lo, const void* w_hi, const void* w_lo, const float* bias,
                             void* out_hi, void* out_lo, float* out_f32, int B, int seq_len, int C, int groups, int taps,
                             int nseg, int mode, void* stream) {
    F5ConvPosArgs cp;
    memset(&cp, 0, sizeof(cp));
    cp.in[0] = (const op16_t*)in_hi;
    cp.in[1] = (const op16_t*)in_lo;
    cp.W[0] = (const op16_t*)w_hi;
    cp.W[1] = (const op16_t*)w_lo;
    cp.bias = bias;
    cp.B = B;
    cp.seq_len = seq_len;
    cp.C = C;
    cp.groups = groups;
    cp.taps = taps;
    cp.ld = C;
    cp.ldo = C;
    cp.nseg = nseg;
    cp.mode = mode;
    cp.out_bf[0] = (op16_t*)out_hi;
    cp.out_bf[1] = (op16_t*)out_lo;
    cp.out_f32 = out_f32;
    return g_ops.convpos(cp, (hipStream_t)stream);
}

// MFMA rate yardstick of the current operand type (f5_op_set_operand_type): see rowops.hip mfma_peak_kernel
extern "C" int f5_op_mfma_peak(const void* operands, int blocks, int iters, float* sink, double* flops, void* stream) {
    return g_ops.h ? f5hf::f5_launch_mfma_peak((const f5hf::op16_t*)operands, blocks, iters, sink, flops, (hipStream_t)stream)
                   : f5bf::f5_launch_mfma_peak((const f5bf::op16_t*)operands, blocks, iters, sink, flops, (hipStream_t)stream);
}

extern "C" int f5_op_ln_modulate(const float* x, const float* scale, const float* shift, void* out_hi, void* out_lo, int rows,
                                 int dim, void* stream) {
    return g_ops.ln_modulate(x, scale, shift, (op16_t*)out_hi, (op16_t*)out_lo, rows, dim, 1e-6f, (hipStream_t)stream, g_op_fold.ln_mean_out);
}

extern "C" int f5_op_dwconv_ln(const float* x, const float* dw_w, const float* dw_b, const float* ln_w, const float* ln_b,
                               void* out_hi, void* out_lo, int nbatch, int seq_len, int dim, void* stream) {
    return g_ops.dwconv_ln(x, dw_w, dw_b, ln_w, ln_b, (op16_t*)out_hi, (op16_t*)out_lo, nbatch, seq_len, dim, 1e-6f,
                               (hipStream_t)stream);
}

extern "C" size_t f5_op_grn_scratch_floats(int nbatch, int seq_len, int dim) {
    return f5_grn_partial_floats(nbatch, seq_len, dim) + (size_t)nbatch * dim;
}
extern "C" int f5_op_grn(const float* g, const float* gamma, const float* beta, float* scratch, void* out_hi, void* out_lo,
                         int nbatch, int seq_len, int dim, void* stream) {
    float* nx = scratch + f5_grn_partial_floats(nbatch, seq_len, dim);
    return g_ops.grn(g, gamma, beta, scratch, nx, (op16_t*)out_hi, (op16_t*)out_lo, nbatch, seq_len, dim, (hipStream_t)stream);
}

extern "C" int f5_op_text_embed(const int32_t* text, int nt, const float* table, const float* pos_table, int max_pos, float* out,
                                int32_t* ids_out, uint8_t* keep_out, int B, int seq_len, int dim, void* stream) {
    return f5_launch_text_embed(text, nt, table, pos_table, max_pos, out, ids_out, keep_out, B, seq_len, dim, 1, (hipStream_t)stream);
}
extern "C" int f5_op_text_embed_nomask(const int32_t* text, int nt, const float* table, const float* pos_table, int max_pos,
                                       float* out, int32_t* ids_out, uint8_t* keep_out, int B, int seq_len, int dim, void* stream) {
    return f5_launch_text_embed(text, nt, table, pos_table, max_pos, out, ids_out, keep_out, B, seq_len, dim, 0, (hipStream_t)stream);
}
// out = (resid + A W^T + bias) * keep[row]   (convnext_v2.py:53-54 + dit.py:225; keep may be NULL)
extern "C" int f5_op_gemm_resid_keep(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                                     const float* resid, const uint8_t* rowkeep, float* out, int M, int N, int K, int lda, int ldw,
                                     int ldo, int nseg, void* stream) {
    F5GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = (const op16_t*)a_hi;
    g.A[1] = (const op16_t*)a_lo;
    g.W[0] = (const op16_t*)w_hi;
    g.W[1] = (const op16_t*)w_lo;
    g.lda = lda;
    g.ldw = ldw;
    g.M = M;
    g.N = N;
    g.K = K;
    g.nseg = nseg;
    g.bias = bias;
    g.resid = resid;
    g.ldres = ldo;
    g.rowkeep = rowkeep;
    g.out_f32 = out;
    g.ldo = ldo;
    F5_REQUIRE(resid != nullptr && out != nullptr, "gemm_resid_keep: null resid / out");
    return g_ops.gemm(g, EPI_RESID_KEEP, (hipStream_t)stream);
}
extern "C" int f5_op_pack_bf16(const float* src, const uint8_t* rowkeep, void* out_hi, void* out_lo, int rows, int cols, int ld,
                               int col0, void* stream) {
    return g_ops.pack_bf16(src, rowkeep, (op16_t*)out_hi, (op16_t*)out_lo, rows, cols, ld, col0, (hipStream_t)stream);
}
extern "C" int f5_op_duration_head(const float* x, const float* g, const float* w, const uint8_t* mask, float* out, int B,
                                   int seq_len, int dim, float eps, void* stream) {
    return f5_launch_duration_head(x, g, w, mask, out, B, seq_len, dim, eps, (hipStream_t)stream);
}
extern "C" int f5_op_text_pos_table(float* table, int max_pos, int dim, void* stream) {
    return f5_launch_text_pos_table(table, max_pos, dim, (hipStream_t)stream);
}
extern "C" int f5_op_time_sinus(const float* t, float* out, int n, int dim, void* stream) {
    return f5_launch_time_sinus(t, out, n, dim, (hipStream_t)stream);
}
extern "C" int f5_op_skinny_gemm(const float* a, const float* w, const float* b, float* out, int M, int N, int K, int silu_in,
                                 int silu_out, void* stream) {
    return f5_launch_skinny_gemm(a, w, b, out, M, N, K, silu_in, silu_out, (hipStream_t)stream);
}
extern "C" int f5_op_cfg_axpy(const float* pred, const float* null_pred, float cfg, const float* base, const float* dt_dev,
                              float coef, float divisor, float* out, void* xin_hi, void* xin_lo, int rows, int mel_dim,
                              void* stream) {
    F5OdeArgs o;
    memset(&o, 0, sizeof(o));
    o.pred = pred;
    o.null_pred = null_pred;
    o.cfg = cfg;
    o.base = base;
    o.dt_ptr = dt_dev;
    o.coef = coef;
    o.divisor = divisor;
    o.out = out;
    o.xin_hi = (op16_t*)xin_hi;
    o.xin_lo = (op16_t*)xin_lo;
    o.rows = rows;
    o.mel_dim = mel_dim;
    return g_ops.ode_stage(o, (hipStream_t)stream);
}

// x[row][col] += gate[col] * ((A W^T + bias)[row][col] * keep[row])   (dit.py:319,323; Vocos layer scale + residual)
extern "C" int f5_op_gemm_resid_gate(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                                     const float* gate, const uint8_t* rowkeep, float* x, int M, int N, int K, int lda, int ldw,
                                     int ldx, int nseg, void* stream) {
    F5GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = (const op16_t*)a_hi;
    g.A[1] = (const op16_t*)a_lo;
    g.W[0] = (const op16_t*)w_hi;
    g.W[1] = (const op16_t*)w_lo;
    g.lda = lda;
    g.ldw = ldw;
    g.M = M;
    g.N = N;
    g.K = K;
    g.nseg = nseg;
    g.bias = bias;
    g.gate = gate;
    g.rowkeep = rowkeep;
    g.out_f32 = x;
    g.ldo = ldx;
    F5_REQUIRE(gate != nullptr && x != nullptr, "gemm_resid_gate: null gate / x");
    if (g_op_fold.x16 != nullptr) {
        g.x16_out = g_op_fold.x16;
        g.ldx16 = N;
        g.x16_scale = g_op_fold.next_scale;
        g.x16_shift = g_op_fold.row_shift;
        g.stats_out = g_op_fold.stats_out;
        g.stats_ld = M;
        g.x16_overflow = g_op_fold_overflow;
    }
    return g_ops.gemm(g, EPI_RESID_GATE, (hipStream_t)stream);
}
// the same with the LN-modulate of the next sub-layer fused behind it (small-tile shapes only, see gemm.hpp ln_counter):
// h = LN(x_new) * (1 + ln_scale) + ln_shift, bit-identical to f5_op_gemm_resid_gate followed by f5_op_ln_modulate.
// counters: >= ceil(M / 64) ints, zero on entry (left zero on exit).  Returns an error for shapes that run the large-tile kernels.
extern "C" int f5_op_gemm_resid_gate_ln(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                                        const float* gate, const uint8_t* rowkeep, float* x, const float* ln_scale,
                                        const float* ln_shift, void* h_hi, void* h_lo, int* counters, int M, int N, int K, int lda,
                                        int ldw, int nseg, void* stream) {
    F5GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = (const op16_t*)a_hi;
    g.A[1] = (const op16_t*)a_lo;
    g.W[0] = (const op16_t*)w_hi;
    g.W[1] = (const op16_t*)w_lo;
    g.lda = lda;
    g.ldw = ldw;
    g.M = M;
    g.N = N;
    g.K = K;
    g.nseg = nseg;
    g.bias = bias;
    g.gate = gate;
    g.rowkeep = rowkeep;
    g.out_f32 = x;
    g.ldo = N;
    g.ln_counter = counters;
    g.ln_scale = ln_scale;
    g.ln_shift = ln_shift;
    g.ln_out[0] = (op16_t*)h_hi;
    g.ln_out[1] = (op16_t*)h_lo;
    g.ln_eps = 1e-6f;
    F5_REQUIRE(gate && x && ln_scale && ln_shift && h_hi && counters, "gemm_resid_gate_ln: null pointer");
    return g_ops.gemm(g, EPI_RESID_GATE, (hipStream_t)stream);
}
// out_f32 = A[row % a_row_mod] W^T + addrows[row], out_bf = the same rounded to the operand type (dit.py:250: the per-step
// x projection added to the hoisted cond / text part; both CFG branches share the x rows through a_row_mod)
extern "C" int f5_op_gemm_addrows(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* addrows,
                                  int a_row_mod, float* out_f32, void* out_hi, void* out_lo, int M, int N, int K, int lda, int ldw,
                                  int ldo, int nseg, void* stream) {
    F5GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A[0] = (const op16_t*)a_hi;
    g.A[1] = (const op16_t*)a_lo;
    g.W[0] = (const op16_t*)w_hi;
    g.W[1] = (const op16_t*)w_lo;
    g.lda = lda;
    g.ldw = ldw;
    g.M = M;
    g.N = N;
    g.K = K;
    g.nseg = nseg;
    g.a_row_mod = a_row_mod;
    g.addrows = addrows;
    g.ldadd = ldo;
    g.out_f32 = out_f32;
    g.ldo = ldo;
    g.out_bf[0] = (op16_t*)out_hi;
    g.out_bf[1] = (op16_t*)out_lo;
    g.ldob = ldo;
    F5_REQUIRE(addrows != nullptr && out_f32 != nullptr && out_hi != nullptr, "gemm_addrows: null addrows / out");
    return g_ops.gemm(g, EPI_ADDROWS, (hipStream_t)stream);
}
extern "C" int f5_op_layernorm(const float* x, const float* w, const float* b, float* out_f32, void* out_hi, void* out_lo,
                               int rows, int dim, void* stream) {
    return g_ops.layernorm(x, w, b, out_f32, (op16_t*)out_hi, (op16_t*)out_lo, rows, dim, 1e-6f, (hipStream_t)stream);
}
extern "C" int f5_op_im2col7(const float* x, void* out_hi, void* out_lo, int nbatch, int seq_len, int channels, void* stream) {
    return g_ops.im2col7(x, (op16_t*)out_hi, (op16_t*)out_lo, nbatch, seq_len, channels, (hipStream_t)stream);
}
