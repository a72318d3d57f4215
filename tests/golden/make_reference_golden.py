"""Golden vectors produced by the REFERENCE's own Python source.

    python tests/golden/make_reference_golden.py          (build container only: needs /root/reference)

MLX is not installable here, so the reference modules (`f5_tts_mlx/{dit,cfm,rope,convnext_v2,audio,utils}.py`) are imported
unmodified from /root/reference on top of `oracle/mlx_shim.py`, a numpy emulation of the mlx primitives they call.  What the
vectors therefore pin is the reference's code path (layer order, masks, CFG, solvers, index arithmetic, quirks) with numpy
arithmetic underneath; the per-primitive semantics are the shim's reading of the MLX documentation (see its header).

Random draws are injected (the MLX PRNG is not emulated): `mx.random.normal` / `uniform` are replaced by queues for the
duration of a call, in the order the reference draws them.  Weights are `synthetic_weights(REF_CFG, seed)` — regenerated, not
stored.  The DiT config is the smallest one the reference's hard-coded `ConvPositionEmbedding(groups=16)` shares with the engine
(64 channels per group => dim 1024).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REFERENCE)

from oracle import mlx_shim  # noqa: E402

mx, nn = mlx_shim.install()

import f5_tts_mlx.audio as ref_audio  # noqa: E402
import f5_tts_mlx.cfm as ref_cfm  # noqa: E402
import f5_tts_mlx.dit as ref_dit  # noqa: E402
import f5_tts_mlx.duration as ref_duration  # noqa: E402
import f5_tts_mlx.rope as ref_rope  # noqa: E402
import f5_tts_mlx.utils as ref_utils  # noqa: E402

from f5_tts_mlx_amd.duration import synthetic_duration_weights  # noqa: E402
from f5_tts_mlx_amd.weights import DiTConfig, synthetic_weights  # noqa: E402

REF_CFG = dict(dim=1024, depth=2, heads=16, dim_head=64, ff_mult=2, mel_dim=100, text_num_embeds=64, text_dim=512,
               conv_layers=2, conv_pos_groups=16)
WEIGHTS_SEED = 7
DUR_CFG = dict(dim=512, depth=3, text_num_embeds=70, text_dim=512, conv_layers=2, ff_mult=2)     # from_pretrained's, 3 of 8 blocks
DUR_SEED = 5
GEN_VOCAB = " abcdefghijklmnopqrstuvwxyz,.!?'-ABCDEFGHIJKLMNOPQRSTUVWXYZ0123"        # 64 symbols; anything else -> 0 (utils.py:129)
GEN_CASES = dict(
    single_duration=dict(generation_text="The quick brown fox.", duration=6.4, steps=3, method="euler", seed=3),
    single_estimated=dict(generation_text="Hello there; general Kenobi", estimate_duration=True, steps=2, method="rk4", seed=11, speed=1.3),
    sentences_predicted=dict(generation_text="One fish. Two fish! Red fish", steps=2, method="midpoint", seed=5, speed=0.1, cfg_strength=1.5),
)


class injected_random:
    """Replace mx.random.{normal,uniform} by queues of pre-drawn arrays (checked for shape) while active."""

    def __init__(self, normal=(), uniform=()):
        self.normal, self.uniform = list(normal), list(uniform)

    def __enter__(self):
        self.saved = (mx.random.normal, mx.random.uniform, mx.random.seed)

        def normal(shape=(), dtype=np.float32):
            a = self.normal.pop(0)
            assert tuple(int(s) for s in shape) == a.shape, (shape, a.shape)
            return mx.array(a)

        def uniform(low=0.0, high=1.0, shape=(), dtype=np.float32):
            a = self.uniform.pop(0)
            assert tuple(int(s) for s in shape) == a.shape, (shape, a.shape)
            assert np.all(a >= low) and np.all(a <= high)
            return mx.array(a)

        mx.random.normal, mx.random.uniform, mx.random.seed = normal, uniform, lambda s: None
        return self

    def __exit__(self, *exc):
        mx.random.normal, mx.random.uniform, mx.random.seed = self.saved
        assert exc[0] is not None or (not self.normal and not self.uniform), "unused injected draws"


def build_reference_model(cfg: DiTConfig, weights):
    dit = ref_dit.DiT(dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, dim_head=cfg.dim_head, ff_mult=cfg.ff_mult, mel_dim=cfg.mel_dim,
                      text_num_embeds=cfg.text_num_embeds, text_dim=cfg.text_dim,
                      text_mask_padding=bool(getattr(cfg, "text_mask_padding", True)), conv_layers=cfg.conv_layers)
    model = ref_cfm.F5TTS(transformer=dit)
    model.load_weights(list(weights.items()))
    return model


def f32(x):
    return np.asarray(x, dtype=np.float32)


def main():
    cfg = DiTConfig(**REF_CFG)
    weights = synthetic_weights(cfg, seed=WEIGHTS_SEED)
    model = build_reference_model(cfg, weights)
    r = np.random.default_rng(2024)
    meta = dict(cfg=json.dumps(REF_CFG), weights_seed=WEIGHTS_SEED)

    # ---- 1. DiT forward (dit.py:362-401): cond / null branch, with and without a key mask -------------------
    B, N, NT = 2, 40, 12
    x = f32(r.standard_normal((B, N, cfg.mel_dim)))
    cond = f32(r.standard_normal((B, N, cfg.mel_dim)))
    cond[:, 14:] = 0
    text = r.integers(0, cfg.text_num_embeds, (B, NT)).astype(np.int32)
    text[1, 9:] = -1
    time = f32([0.3, 0.3])
    mask = np.arange(N)[None, :] < np.array([N, 33])[:, None]
    fwd = {}
    for tag, (da, dt, m) in dict(cond=(False, False, None), null=(True, True, None), cond_masked=(False, False, mask),
                                 null_masked=(True, True, mask)).items():
        fwd[tag] = f32(model.transformer(x=mx.array(x), cond=mx.array(cond), text=mx.array(text), time=mx.array(time),
                                         drop_audio_cond=da, drop_text=dt, mask=None if m is None else mx.array(m)))
    scalar_t = f32(model.transformer(x=mx.array(x[:1]), cond=mx.array(cond[:1]), text=mx.array(text[:1]), time=mx.array(np.float32(0.7)),
                                     drop_audio_cond=False, drop_text=False))
    np.savez_compressed(os.path.join(HERE, "ref_dit_forward.npz"), x=x, cond=cond, text=text, time=time, mask=mask,
                        out_scalar_time=scalar_t, **{"out_" + k: v for k, v in fwd.items()}, **meta)

    # ---- 2. F5TTS.sample (cfm.py:264-402): ragged batch of 2 and batch 1, three solvers ----------------------
    n_ref = 16
    durations = np.array([44, 37], np.int32)
    cond_mel = f32(r.standard_normal((2, n_ref, cfg.mel_dim)))
    ref_lens = np.array([16, 13], np.int32)
    stext = r.integers(0, cfg.text_num_embeds, (2, 14)).astype(np.int32)
    stext[1, 11:] = -1
    z = [f32(r.standard_normal((cfg.mel_dim, int(d)))) for d in durations]
    samples = {}
    for method, steps in (("euler", 5), ("midpoint", 4), ("rk4", 3)):
        with injected_random(normal=z):
            out, traj = model.sample(mx.array(cond_mel), mx.array(stext), mx.array(durations), lens=mx.array(ref_lens), steps=steps,
                                     method=method, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=1)
        samples[f"out_{method}"] = f32(out)
        samples[f"traj_{method}"] = f32(traj)
        samples[f"steps_{method}"] = steps
    with injected_random(normal=z[:1]):        # batch 1: no attention mask (cfm.py:333-336), int duration, no sway, cfg 0
        out1, traj1 = model.sample(mx.array(cond_mel[:1]), mx.array(stext[:1]), int(durations[0]), steps=4, method="euler",
                                   cfg_strength=0.0, sway_sampling_coef=None)
    np.savez_compressed(os.path.join(HERE, "ref_sample.npz"), cond=cond_mel, lens=ref_lens, text=stext, durations=durations,
                        z0=z[0], z1=z[1], out_b1_nocfg=f32(out1), traj_b1_nocfg=f32(traj1), **samples, **meta)

    # ---- 2b. duration / lens clamps (cfm.py:301-303,317-319): more text tokens than reference frames
    # (conditioning region extended over zero padding), duration below lens+1 raised, duration above max_duration clipped
    r2 = np.random.default_rng(77)
    c_cond = f32(r2.standard_normal((2, 8, cfg.mel_dim)))
    c_lens = np.array([8, 6], np.int32)
    c_text = r2.integers(0, cfg.text_num_embeds, (2, 14)).astype(np.int32)
    c_text[1, 11:] = -1
    c_req = np.array([10, 50], np.int32)
    c_final = [15, 45]                                   # max(max(#tokens, lens) + 1, requested) clipped to max_duration = 45
    cz = [f32(r2.standard_normal((cfg.mel_dim, d))) for d in c_final]
    with injected_random(normal=cz):
        c_out, c_traj = model.sample(mx.array(c_cond), mx.array(c_text), mx.array(c_req), lens=mx.array(c_lens), steps=3, method="euler",
                                     cfg_strength=2.0, sway_sampling_coef=-1.0, seed=1, max_duration=45)
    assert c_out.shape == (2, 45, cfg.mel_dim)
    np.savez_compressed(os.path.join(HERE, "ref_sample_clamps.npz"), cond=c_cond, lens=c_lens, text=c_text, durations=c_req,
                        final_durations=np.array(c_final, np.int32), max_duration=45, z0=cz[0], z1=cz[1], out=f32(c_out), traj=f32(c_traj), **meta)

    # ---- 3. F5TTS.__call__ loss forward (cfm.py:169-251), every draw recorded in reference order --------------
    Bl, Nl = 2, 48
    mel_in = f32(r.standard_normal((Bl, Nl, cfg.mel_dim)))
    ltext = r.integers(0, cfg.text_num_embeds, (Bl, 12)).astype(np.int32)
    llens = np.array([48, 41], np.int32)
    draws = dict(frac_lengths=f32([0.75, 0.9]), span_rand=f32([0.35, 0.6]), x0=f32(r.standard_normal((Bl, Nl, cfg.mel_dim))),
                 time=f32([0.21, 0.83]))
    losses = {}
    for name, (ra, rc) in dict(keep=(0.9, 0.9), drop_audio=(0.1, 0.9), drop_both=(0.9, 0.1)).items():
        with injected_random(normal=[draws["x0"]], uniform=[draws["frac_lengths"], draws["span_rand"], draws["time"], f32([ra]), f32([rc])]):
            losses["loss_" + name] = np.float32(model(mx.array(mel_in), mx.array(ltext), lens=mx.array(llens)))
    np.savez_compressed(os.path.join(HERE, "ref_cfm_loss.npz"), mel=mel_in, text=ltext, lens=llens, **draws, **losses, **meta)

    # ---- 4. host-side pieces: masks, tokenisers, RoPE tables, time embedding, mel front-end --------------------
    lens_a = np.array([5, 9, 1], np.int32)
    vocab = {c: i for i, c in enumerate([" ", "a", "b", "c", "ni3", "hao3", "!"])}
    tok_in = [["a", "b", " ", "ni3", "hao3", "?"], ["c", "!"]]
    with injected_random(uniform=[f32([0.5, 0.1, 0.99])]):
        frac_mask = ref_utils.mask_from_frac_lengths(mx.array(np.array([20, 31, 8], np.int32)), mx.array(f32([0.7, 0.85, 1.0])), max_length=32)
    rot = ref_rope.RotaryEmbedding(64)
    freqs, xpos = rot.forward_from_seq_len(24)
    assert xpos == 1.0 or xpos is None or float(xpos) == 1.0
    qh = f32(r.standard_normal((1, 2, 24, 64)))
    sway = {}
    for coef in (None, -1.0, 0.5):
        t = mx.linspace(0, 1, 9)
        if coef is not None:
            t = t + coef * (mx.cos(mx.pi / 2 * t) - 1 + t)
        sway[str(coef)] = f32(t)
    audio = f32(0.1 * r.standard_normal(24000 // 2))
    np.savez_compressed(
        os.path.join(HERE, "ref_host.npz"),
        lens=lens_a, lens_mask=np.asarray(ref_utils.lens_to_mask(mx.array(lens_a))),
        lens_mask_len12=np.asarray(ref_utils.lens_to_mask(mx.array(lens_a), length=12)),
        frac_mask=np.asarray(frac_mask),
        utf8=np.asarray(ref_utils.list_str_to_tensor(["hello", "héllo wörld", ""])),
        vocab=json.dumps(vocab), tok_in=json.dumps(tok_in), tok_idx=np.asarray(ref_utils.list_str_to_idx(tok_in, vocab)),
        freqs_cis=f32(ref_rope.precompute_freqs_cis(512, 64)),
        pos_idx=np.asarray(ref_rope.get_pos_embed_indices(mx.array(np.array([0, 3], np.int32)), 10, max_pos=12)),
        rope_freqs=f32(freqs), rope_in=qh, rope_out=f32(ref_rope.apply_rotary_pos_emb(mx.array(qh), freqs)),
        time_in=f32([0.0, 0.25, 1.0]), time_sinus=f32(ref_dit.SinusPositionEmbedding(256)(mx.array(f32([0.0, 0.25, 1.0])))),
        sway_none=sway["None"], sway_m1=sway["-1.0"], sway_p05=sway["0.5"],
        audio=audio, mel=f32(ref_audio.log_mel_spectrogram(mx.array(audio))),
        mel_filters=f32(ref_audio.mel_filters(24000, 1024, 100)), hanning=f32(ref_audio.hanning(1024)))
    # ---- 5. DurationPredictor.__call__ (duration.py:192-251), return_loss=False ----------------------------------
    dp = ref_duration.DurationPredictor(transformer=ref_duration.DurationTransformer(heads=8, **DUR_CFG))
    dp.load_weights(list(synthetic_duration_weights(seed=DUR_SEED, **DUR_CFG).items()))
    dur = {}
    for tag, (Bd, Nd, ntd) in dict(b1=(1, 50, 20), b2_long_text=(2, 36, 44)).items():
        dmel = f32(r.standard_normal((Bd, Nd, 100)) * 1.5 - 1.0)
        dtext = r.integers(0, DUR_CFG["text_num_embeds"], (Bd, ntd)).astype(np.int32)
        dtext[-1, ntd - 5:] = -1
        dlens = np.array([Nd] * Bd if Bd == 1 else [Nd, Nd - 9], np.int32)
        dur.update({f"{tag}_mel": dmel, f"{tag}_text": dtext, f"{tag}_lens": dlens,
                    f"{tag}_seconds": f32(dp(mx.array(dmel), mx.array(dtext), lens=mx.array(dlens)))})
    np.savez_compressed(os.path.join(HERE, "ref_duration.npz"), cfg=json.dumps(DUR_CFG), weights_seed=DUR_SEED, **dur)

    # ---- 6. generate() (generate.py:113-245): the API wrapper around sample --------------------------------------
    # Stand-ins (test infrastructure, identical on the engine side of the comparison): `soundfile` -> scipy WAV reader
    # / capture of the written array; `sounddevice` unused (output_path given); `jieba.cut` -> the package's emulation
    # of jieba's segmentation of single-byte text; `F5TTS.from_pretrained` -> the synthetic-weight model above plus a
    # character vocabulary, the duration predictor of section 5 and a FAKE vocoder (frame n -> 256 samples
    # mel[n, j % 100]); `mx.random.seed/normal` -> the package's threefry emulation of MLX's generator (rng.py).
    import scipy.io.wavfile as wavfile
    from f5_tts_mlx_amd.rng import mlx_like_normal
    from f5_tts_mlx_amd.utils import _ascii_segments
    written = {}

    def sf_read(path):
        sr, a = wavfile.read(path)
        return a.astype(np.float64) / 32768.0, sr

    sys.modules["soundfile"].read = sf_read
    sys.modules["soundfile"].write = lambda path, data, sr: written.__setitem__(path, (np.asarray(data, dtype=np.float32), sr))
    sys.modules["jieba"].cut = _ascii_segments
    import f5_tts_mlx.generate as ref_generate  # noqa: E402
    ref_utils.jieba.cut = _ascii_segments

    gen_model = build_reference_model(cfg, weights)
    gen_model._vocab_char_map = {c: i for i, c in enumerate(GEN_VOCAB)}
    gen_model._vocoder = lambda mel: mx.array(np.asarray(mel)[0][:, np.arange(256) % 100].reshape(-1))
    gen_model._duration_predictor = dp
    dp._vocab_char_map = gen_model._vocab_char_map
    ref_generate.F5TTS.from_pretrained = classmethod(lambda cls, name, quantization_bits=None: gen_model)
    seed_state = {}
    saved = (mx.random.seed, mx.random.normal)
    mx.random.seed = lambda s: seed_state.__setitem__("seed", int(s))
    mx.random.normal = lambda shape=(), dtype=np.float32: mx.array(mlx_like_normal(seed_state["seed"], tuple(int(v) for v in shape)))
    wav = os.path.join(ROOT, "f5_tts_mlx_amd", "assets", "test_en_1_ref_short.wav")
    caption = "Some call me nature, others call me mother nature."
    gen = {}
    for tag, kw in GEN_CASES.items():
        ref_generate.generate(ref_audio_path=wav, ref_audio_text=caption, output_path=tag, **kw)
        wave, sr = written[tag]
        assert sr == 24000 and wave.ndim == 1 and wave.shape[0] > 0
        gen["wave_" + tag] = wave
    mx.random.seed, mx.random.normal = saved
    np.savez_compressed(os.path.join(HERE, "ref_generate.npz"), cases=json.dumps(GEN_CASES), vocab=GEN_VOCAB, caption=caption,
                        dur_cfg=json.dumps(DUR_CFG), dur_seed=DUR_SEED, **gen, **meta)
    print("generate():", {k: v.shape for k, v in gen.items()})

    print("reference goldens written:", {k: v.shape for k, v in fwd.items()}, {k: float(v) for k, v in losses.items()})


if __name__ == "__main__":
    main()
