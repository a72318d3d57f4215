// Vocos vocoder behind the C ABI: mel (b, n, 100) -> waveform, one call (`f5_vocode`), hipGraph captured.
//
// Reference call sites: cfm.py:19 (import), :399-400 (`self._vocoder(out)` at the end of F5TTS.sample), :446 / :471
// (`Vocos.from_pretrained("lucasnewman/vocos-mel-24khz")` wired into F5TTS).  The implementation of `vocos_mlx` is a third-party
// dependency that is NOT in /root/reference (pyproject.toml:42, unpinned): what runs here is restated from the published Vocos
// architecture (gemelo-ai/vocos `VocosBackbone` + `ISTFTHead`, mel-24khz configuration):
//   embed   = Conv1d(n_mels -> dim, k = 7, pad 3)            -> im2col (tap major, channels padded to 128) + MFMA GEMM
//   norm    = LayerNorm(dim, eps 1e-6)
//   8 x ConvNeXtBlock: depthwise Conv1d(k = 7) -> LayerNorm -> Linear(dim -> 3 dim) -> GELU(erf) -> Linear(3 dim -> dim)
//                      -> x + gamma * (.)                     -> dwconv_ln kernel, two MFMA GEMMs with fused epilogues
//   final_layer_norm, head.out = Linear(dim -> n_fft + 2)     -> (log-magnitude | phase), exp clipped at 1e2
//   ISTFT(n_fft 1024, hop 256, hann, center)                  -> per-frame inverse FFT in LDS + overlap-add, whole batch in 2 launches
// PARITY UNPINNED: no reference-side vector exists for this stage (no vocos_mlx, no checkpoint, no network); tests compare with
// oracle/vocos_oracle.py (a CPU restatement of the same published architecture) and torch.istft.
#include <stdarg.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/f5tts_hip.h"
#include "host_common.hpp"

int f5_launch_istft(const float* x, int ldx, const float* window, float* frames_scratch, float* wave, int B, int nframes, int hop,
                    hipStream_t s);   // audio.hip

struct VBlock {
    size_t dw_w, dw_b, ln_w, ln_b, b1, b2, gamma;
    MatBF pw1, pw2;
};
struct VGraph {
    int B, N;
    const void* workspace;
    hipGraphExec_t exec;
    uint64_t stamp;
    hipEvent_t done;         // recorded after every launch: an exec is only destroyed once its last replay has finished
};
static void destroy_vgraph(VGraph& g) {
    if (g.done) {
        (void)hipEventSynchronize(g.done);
        (void)hipEventDestroy(g.done);
    }
    (void)hipGraphExecDestroy(g.exec);
}
struct VWorkspace {
    size_t total = 0;
    size_t mel, x0, x, y, frames, wave;
    size_t a0[2], h[2], g[2];
};

struct f5_vocoder {
    f5_vocos_config cfg;
    int np = 1;
    Ops ops;
    char* arena = nullptr;
    size_t arena_need = 0, arena_bytes = 0;
    bool finalized = false;
    std::unordered_map<std::string, std::vector<TensorDst>> tmap;
    MatBF w_embed, w_head;
    size_t b_embed, norm_w, norm_b, fnorm_w, fnorm_b, b_head, window;
    bool embed_loaded = false;
    std::vector<VBlock> blocks;
    std::vector<VGraph> graphs;
    uint64_t clock = 0;
};

static void vadd_f32(f5_vocoder* v, const std::string& name, size_t off, std::vector<int64_t> shape) {
    TensorDst d;
    d.kind = 0;
    d.off = off;
    d.shape = shape;
    v->tmap[name].push_back(d);
}
static void vadd_mat(f5_vocoder* v, const std::string& name, const MatBF& m, int rows, int cols) {
    TensorDst d;
    d.kind = 1;
    d.mat = m;
    d.src_rows = rows;
    d.src_cols = cols;
    d.c0 = 0;
    d.c1 = cols;
    d.shape = {rows, cols};
    v->tmap[name].push_back(d);
}

extern "C" int f5_vocoder_create(const f5_vocos_config* cfg, int precision, f5_vocoder** out) {
    F5_REQUIRE(cfg && out, "f5_vocoder_create: null argument");
    const f5_vocos_config& c = *cfg;
    F5_REQUIRE(precision == F5_PREC_BF16 || precision == F5_PREC_BF16X3 || precision == F5_PREC_F16,
               "vocoder precision must be bf16, bf16x3 or f16 (got %d)", precision);
    F5_REQUIRE(c.n_fft == 1024, "vocoder: only n_fft = 1024 is supported (got %d)", c.n_fft);
    F5_REQUIRE(c.hop_length > 0 && c.n_fft % c.hop_length == 0, "vocoder: hop_length must divide n_fft");
    F5_REQUIRE(c.n_mels >= 1 && c.n_mels <= 128, "vocoder: n_mels must be in [1, 128]");
    F5_REQUIRE(c.dim % 256 == 0 && c.dim >= 256 && c.dim <= 1024, "vocoder: dim must be 256 / 512 / 768 / 1024");
    F5_REQUIRE(c.intermediate_dim % 128 == 0 && c.intermediate_dim >= 128, "vocoder: intermediate_dim must be a multiple of 128");
    F5_REQUIRE(c.num_layers >= 1 && c.num_layers <= 64, "vocoder: num_layers out of range");
    f5_vocoder* v = new f5_vocoder();
    v->cfg = c;
    v->np = precision == F5_PREC_BF16X3 ? 2 : 1;
    v->ops.h = precision == F5_PREC_F16;
    const int D = c.dim, I = c.intermediate_dim, NF = c.n_fft + 2;
    Bump b;
    v->w_embed = alloc_mat(b, D, 7 * 128, v->np);      // [co][tap][c padded to 128]; loaded through the layout-aware path
    v->b_embed = b.take((size_t)D * 4);
    v->norm_w = b.take((size_t)D * 4);
    v->norm_b = b.take((size_t)D * 4);
    vadd_f32(v, "backbone.embed.bias", v->b_embed, {D});
    vadd_f32(v, "backbone.norm.weight", v->norm_w, {D});
    vadd_f32(v, "backbone.norm.bias", v->norm_b, {D});
    v->blocks.resize(c.num_layers);
    for (int i = 0; i < c.num_layers; ++i) {
        VBlock& k = v->blocks[i];
        const std::string p = "backbone.convnext." + std::to_string(i) + ".";
        k.dw_w = b.take((size_t)D * 7 * 4);
        k.dw_b = b.take((size_t)D * 4);
        k.ln_w = b.take((size_t)D * 4);
        k.ln_b = b.take((size_t)D * 4);
        k.b1 = b.take((size_t)I * 4);
        k.b2 = b.take((size_t)D * 4);
        k.gamma = b.take((size_t)D * 4);
        k.pw1 = alloc_mat(b, I, D, v->np);
        k.pw2 = alloc_mat(b, D, I, v->np);
        // depthwise weight: upstream (dim, 1, 7) and MLX (dim, 7, 1) are the same bytes, [dim][7]; both shapes are accepted
        vadd_f32(v, p + "dwconv.weight", k.dw_w, {D, 1, 7});
        vadd_f32(v, p + "dwconv.bias", k.dw_b, {D});
        vadd_f32(v, p + "norm.weight", k.ln_w, {D});
        vadd_f32(v, p + "norm.bias", k.ln_b, {D});
        vadd_mat(v, p + "pwconv1.weight", k.pw1, I, D);
        vadd_f32(v, p + "pwconv1.bias", k.b1, {I});
        vadd_mat(v, p + "pwconv2.weight", k.pw2, D, I);
        vadd_f32(v, p + "pwconv2.bias", k.b2, {D});
        vadd_f32(v, p + "gamma", k.gamma, {D});
    }
    v->fnorm_w = b.take((size_t)D * 4);
    v->fnorm_b = b.take((size_t)D * 4);
    vadd_f32(v, "backbone.final_layer_norm.weight", v->fnorm_w, {D});
    vadd_f32(v, "backbone.final_layer_norm.bias", v->fnorm_b, {D});
    v->w_head = alloc_mat(b, NF, D, v->np);
    v->b_head = b.take((size_t)NF * 4);
    vadd_mat(v, "head.out.weight", v->w_head, NF, D);
    vadd_f32(v, "head.out.bias", v->b_head, {NF});
    v->window = b.take((size_t)c.n_fft * 4);
    v->arena_need = b.off;
    *out = v;
    return 0;
}

extern "C" void f5_vocoder_destroy(f5_vocoder* v) {
    if (!v) return;
    for (auto& g : v->graphs) destroy_vgraph(g);
    delete v;
}

extern "C" int f5_vocoder_weights_bytes(f5_vocoder* v, size_t* bytes) {
    F5_REQUIRE(v && bytes, "null argument");
    *bytes = v->arena_need;
    return 0;
}

extern "C" int f5_vocoder_set_weights_arena(f5_vocoder* v, void* dev_arena, size_t bytes, void* stream) {
    F5_REQUIRE(v && dev_arena, "null argument");
    F5_REQUIRE(bytes >= v->arena_need, "vocoder weights arena too small: %zu < %zu", bytes, v->arena_need);
    F5_REQUIRE(((uintptr_t)dev_arena & 255) == 0, "vocoder weights arena must be 256-byte aligned");
    v->arena = (char*)dev_arena;
    v->arena_bytes = bytes;
    F5_HIP_CHECK(hipMemsetAsync(dev_arena, 0, v->arena_need, (hipStream_t)stream));   // zero pads
    F5_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

// Tensor names are the upstream Vocos state-dict names (`backbone.embed.weight`, `backbone.convnext.3.pwconv1.weight`,
// `head.out.bias` ...), which are also what an MLX module tree of the same structure produces; conv weights are accepted in the
// PyTorch layout (out, in, k) and in the MLX layout (out, k, in) -- told apart by their shape.
extern "C" int f5_vocoder_load_tensor(f5_vocoder* v, const char* name, const float* host, int ndim, const int64_t* shape) {
    F5_REQUIRE(v && name && host && shape, "null argument");
    F5_REQUIRE(v->arena, "f5_vocoder_set_weights_arena must be called first");
    const f5_vocos_config& c = v->cfg;
    if (std::string(name) == "backbone.embed.weight") {
        F5_REQUIRE(ndim == 3 && shape[0] == c.dim, "tensor '%s': expected (dim, n_mels, 7) or (dim, 7, n_mels)", name);
        const bool torch_layout = shape[1] == c.n_mels && shape[2] == 7;
        const bool mlx_layout = shape[1] == 7 && shape[2] == c.n_mels;
        F5_REQUIRE(torch_layout || mlx_layout, "tensor '%s': expected (dim, n_mels, 7) or (dim, 7, n_mels)", name);
        std::vector<float> m((size_t)c.dim * 7 * 128, 0.0f);
        for (int co = 0; co < c.dim; ++co)
            for (int tap = 0; tap < 7; ++tap)
                for (int ci = 0; ci < c.n_mels; ++ci)
                    m[((size_t)co * 7 + tap) * 128 + ci] = torch_layout ? host[((size_t)co * c.n_mels + ci) * 7 + tap]
                                                                         : host[((size_t)co * 7 + tap) * c.n_mels + ci];
        TensorDst d;
        d.kind = 1;
        d.mat = v->w_embed;
        d.src_rows = c.dim;
        d.src_cols = 7 * 128;
        d.c0 = 0;
        d.c1 = 7 * 128;
        RC(f5_upload_tensor(v->arena, d, m.data(), m.size(), v->np, v->ops.h));
        v->embed_loaded = true;
        return 0;
    }
    auto it = v->tmap.find(name);
    F5_REQUIRE(it != v->tmap.end(), "unknown vocoder tensor name '%s'", name);
    for (TensorDst& d : it->second) {
        size_t count = 1, want = 1;
        for (int i = 0; i < ndim; ++i) count *= (size_t)shape[i];
        for (int64_t s : d.shape) want *= (size_t)s;
        bool ok = (int)d.shape.size() == ndim;
        for (int i = 0; ok && i < ndim; ++i) ok = d.shape[i] == shape[i];
        // depthwise conv weight in the MLX layout (dim, 7, 1): same bytes as (dim, 1, 7)
        if (!ok && d.kind == 0 && ndim == 3 && d.shape.size() == 3 && shape[0] == d.shape[0] && shape[1] == 7 && shape[2] == 1) ok = true;
        // LayerScale gamma is sometimes stored (1, 1, dim)
        if (!ok && d.kind == 0 && count == want && d.shape.size() == 1) ok = true;
        F5_REQUIRE(ok && count == want, "tensor '%s': unexpected shape", name);
        RC(f5_upload_tensor(v->arena, d, host, count, v->np, v->ops.h));
        d.loaded = true;
    }
    return 0;
}

extern "C" int f5_vocoder_mark_weights_loaded(f5_vocoder* v) {
    F5_REQUIRE(v, "null argument");
    for (auto& kv : v->tmap)
        for (auto& d : kv.second) d.loaded = true;
    v->embed_loaded = true;
    return 0;
}

extern "C" int f5_vocoder_finalize(f5_vocoder* v, void* stream) {
    F5_REQUIRE(v && v->arena, "vocoder arena not set");
    F5_REQUIRE(v->embed_loaded, "vocoder tensor 'backbone.embed.weight' was never loaded");
    for (auto& kv : v->tmap)
        for (auto& d : kv.second) F5_REQUIRE(d.loaded, "vocoder tensor '%s' was never loaded", kv.first.c_str());
    // periodic Hann window of the ISTFT head (torch.hann_window(n_fft)), computed in double like numpy's np.hanning(n+1)[:-1]
    std::vector<float> win(v->cfg.n_fft);
    for (int i = 0; i < v->cfg.n_fft; ++i) win[i] = (float)(0.5 - 0.5 * cos(2.0 * 3.14159265358979323846 * i / v->cfg.n_fft));
    F5_HIP_CHECK(hipMemcpy(v->arena + v->window, win.data(), win.size() * 4, hipMemcpyHostToDevice));
    (void)stream;
    v->finalized = true;
    return 0;
}

static VWorkspace vplan(const f5_vocoder* v, int B, int N) {
    const f5_vocos_config& c = v->cfg;
    const size_t rows = (size_t)B * N;
    Bump b;
    VWorkspace w;
    w.mel = b.take(rows * c.n_mels * 4);
    w.x0 = b.take(rows * c.dim * 4);
    w.x = b.take(rows * c.dim * 4);
    w.y = b.take(rows * (c.n_fft + 2) * 4);
    w.frames = b.take(rows * c.n_fft * 4);
    w.wave = b.take((size_t)B * c.hop_length * (N > 1 ? N - 1 : 1) * 4);
    for (int p = 0; p < 2; ++p) {
        w.a0[p] = p < v->np ? b.take(rows * 7 * 128 * 2) : 0;
        w.h[p] = p < v->np ? b.take(rows * c.dim * 2) : 0;
        w.g[p] = p < v->np ? b.take(rows * c.intermediate_dim * 2) : 0;
    }
    w.total = b.off;
    return w;
}

extern "C" int f5_vocoder_workspace_bytes(f5_vocoder* v, int B, int N, size_t* bytes) {
    F5_REQUIRE(v && bytes, "null argument");
    F5_REQUIRE(B >= 1 && N >= 2, "vocoder: need B >= 1 and N >= 2 frames (got B=%d N=%d)", B, N);
    *bytes = vplan(v, B, N).total;
    return 0;
}

static int vocode_body(const f5_vocoder* v, const VWorkspace& w, char* ws, int B, int N, hipStream_t s) {
    const f5_vocos_config& c = v->cfg;
    const Ops& K = v->ops;
    const int D = c.dim, I = c.intermediate_dim, NF = c.n_fft + 2, rows = B * N, nseg = v->np == 2 ? 3 : 1;
    auto P = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    auto PB = [&](const size_t (&offs)[2], int part) { return part < v->np ? reinterpret_cast<op16_t*>(ws + offs[part]) : (op16_t*)nullptr; };
    auto A = [&](size_t off) { return reinterpret_cast<const float*>(v->arena + off); };
    auto WM = [&](const MatBF& m, int part) {
        return part == 0 ? reinterpret_cast<const op16_t*>(v->arena + m.hi) : (v->np == 2 ? reinterpret_cast<const op16_t*>(v->arena + m.lo) : nullptr);
    };
    auto gemm = [&](const op16_t* ah, const op16_t* al, int lda, const MatBF& wm, int Nn, int Kk, const float* bias) {
        F5GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A[0] = ah;
        g.A[1] = al;
        g.W[0] = WM(wm, 0);
        g.W[1] = WM(wm, 1);
        g.lda = lda;
        g.ldw = wm.ld;
        g.M = rows;
        g.N = Nn;
        g.K = Kk;
        g.nseg = nseg;
        g.bias = bias;
        return g;
    };
    // embed conv as a GEMM over im2col rows
    RC(K.im2col7(P(w.mel), PB(w.a0, 0), PB(w.a0, 1), B, N, c.n_mels, s));
    F5GemmArgs ge = gemm(PB(w.a0, 0), PB(w.a0, 1), 7 * 128, v->w_embed, D, 7 * 128, A(v->b_embed));
    ge.out_f32 = P(w.x0);
    ge.ldo = D;
    RC(K.gemm(ge, EPI_F32, s));
    RC(K.layernorm(P(w.x0), A(v->norm_w), A(v->norm_b), P(w.x), nullptr, nullptr, rows, D, 1e-6f, s));
    for (const VBlock& k : v->blocks) {
        RC(K.dwconv_ln(P(w.x), A(k.dw_w), A(k.dw_b), A(k.ln_w), A(k.ln_b), PB(w.h, 0), PB(w.h, 1), B, N, D, 1e-6f, s));
        F5GemmArgs g1 = gemm(PB(w.h, 0), PB(w.h, 1), D, k.pw1, I, D, A(k.b1));
        g1.out_bf[0] = PB(w.g, 0);
        g1.out_bf[1] = PB(w.g, 1);
        g1.ldob = I;
        RC(K.gemm(g1, EPI_GELU_ERF_BF16, s));
        F5GemmArgs g2 = gemm(PB(w.g, 0), PB(w.g, 1), I, k.pw2, D, I, A(k.b2));
        g2.out_f32 = P(w.x);
        g2.ldo = D;
        g2.gate = A(k.gamma);                     // LayerScale gamma = the per-column gate of the residual epilogue
        RC(K.gemm(g2, EPI_RESID_GATE, s));
    }
    RC(K.layernorm(P(w.x), A(v->fnorm_w), A(v->fnorm_b), nullptr, PB(w.h, 0), PB(w.h, 1), rows, D, 1e-6f, s));
    F5GemmArgs gh = gemm(PB(w.h, 0), PB(w.h, 1), D, v->w_head, NF, D, A(v->b_head));
    gh.out_f32 = P(w.y);
    gh.ldo = NF;
    RC(K.gemm(gh, EPI_F32, s));
    RC(f5_launch_istft(P(w.y), NF, A(v->window), P(w.frames), P(w.wave), B, N, c.hop_length, s));
    return 0;
}

// mel dev [B][N][n_mels] fp32 -> wave dev [B][hop * (N - 1)] fp32.  Replaces `self._vocoder(out)` (cfm.py:399-400).
extern "C" int f5_vocode(f5_vocoder* v, const float* mel, int B, int N, float* wave, void* workspace, size_t workspace_bytes,
                         int use_graph, void* stream) {
    F5_REQUIRE(v && mel && wave && workspace, "f5_vocode: null argument");
    F5_REQUIRE(v->finalized, "vocoder weights are not finalized");
    F5_REQUIRE(B >= 1 && N >= 2, "vocoder: need B >= 1 and N >= 2 frames (got B=%d N=%d)", B, N);
    F5_REQUIRE(((uintptr_t)workspace & 255) == 0, "vocoder workspace must be 256-byte aligned");
    const VWorkspace w = vplan(v, B, N);
    F5_REQUIRE(workspace_bytes >= w.total, "vocoder workspace too small: %zu < %zu", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const size_t rows = (size_t)B * N;
    // caller buffers are staged outside the graph: the captured nodes only reference the workspace and the arena
    RC(f5_launch_copy_words(mel, ws + w.mel, rows * v->cfg.n_mels, s));      // copy KERNELS, not memcpy nodes (rowops.hip)
    if (use_graph) {
        hipGraphExec_t exec = nullptr;
        for (auto& g : v->graphs)
            if (g.B == B && g.N == N && g.workspace == workspace) {
                exec = g.exec;
                g.stamp = ++v->clock;
            }
        if (!exec) {
            for (size_t i = 0; i < v->graphs.size();) {
                if (v->graphs[i].workspace != workspace) {
                    destroy_vgraph(v->graphs[i]);
                    v->graphs.erase(v->graphs.begin() + i);
                } else {
                    ++i;
                }
            }
            while (v->graphs.size() >= 8) {
                size_t lru = 0;
                for (size_t i = 1; i < v->graphs.size(); ++i)
                    if (v->graphs[i].stamp < v->graphs[lru].stamp) lru = i;
                destroy_vgraph(v->graphs[lru]);
                v->graphs.erase(v->graphs.begin() + lru);
            }
            hipGraph_t graph = nullptr;
            F5_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            const int rc = vocode_body(v, w, ws, B, N, s);
            const hipError_t ec = hipStreamEndCapture(s, &graph);
            if (rc) {
                if (graph) (void)hipGraphDestroy(graph);
                return rc;
            }
            F5_HIP_CHECK(ec);
            F5_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
            hipEvent_t done = nullptr;
            F5_HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
            v->graphs.push_back({B, N, workspace, exec, ++v->clock, done});
        }
        F5_HIP_CHECK(hipGraphLaunch(exec, s));
        for (auto& g : v->graphs)
            if (g.exec == exec) F5_HIP_CHECK(hipEventRecord(g.done, s));
    } else {
        RC(vocode_body(v, w, ws, B, N, s));
    }
    RC(f5_launch_copy_words(ws + w.wave, wave, (size_t)B * v->cfg.hop_length * (N - 1), s));
    return 0;
}
