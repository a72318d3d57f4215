"""The HIP engine against golden vectors produced by the REFERENCE's own Python source (tests/golden/ref_*.npz, made by
tests/golden/make_reference_golden.py over oracle/mlx_shim.py — see that script's header for what the vectors do and do not pin).

Gate (BASELINE.json north_star): mean |mel_engine - mel_reference| <= 1e-3, in the bf16x3 precision mode; plain bf16 drift is
reported and loosely bounded.  Nothing here reads /root/reference: only the committed vectors travel to the GPU box.
"""
import json
import os

import numpy as np
import pytest
import torch

from f5test import DEV, DiTConfig, report, synthetic_weights
from f5_tts_mlx_amd.cfm import F5TTS
from f5_tts_mlx_amd.dit import DiT
from f5_tts_mlx_amd.utils import lens_to_mask, list_str_to_idx, list_str_to_tensor, mask_from_frac_lengths

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MEL_L1_TOL = 1e-3


def load(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="module")
def models():
    g = load("ref_dit_forward.npz")
    cfg = DiTConfig(**json.loads(str(g["cfg"])))
    w = synthetic_weights(cfg, seed=int(g["weights_seed"]))
    out = {}
    for prec in ("bf16x3", "bf16", "f16"):
        m = DiT.from_config(cfg, precision=prec, device=DEV)
        m.load_weights(w)
        out[prec] = m
    return cfg, out


def test_dit_forward_vs_reference_code(models):
    cfg, ms = models
    g = load("ref_dit_forward.npz")
    x, cond, text, time, mask = (torch.from_numpy(g[k]) for k in ("x", "cond", "text", "time", "mask"))
    for prec, tol in (("bf16x3", 2e-4), ("bf16", 3e-2)):
        for tag, (drop, m) in dict(cond=(False, None), null=(True, None), cond_masked=(False, mask), null_masked=(True, mask)).items():
            got = ms[prec](x=x, cond=cond, text=text, time=time, drop_audio_cond=drop, drop_text=drop, mask=m)
            _, mean, refm = report(f"dit[{prec}] {tag} vs reference code", got.cpu(), torch.from_numpy(g["out_" + tag]))
            assert mean <= tol * max(1.0, refm), (prec, tag)
    got = ms["bf16x3"](x=x[:1], cond=cond[:1], text=text[:1], time=torch.tensor(0.7), drop_audio_cond=False, drop_text=False)
    _, mean, _ = report("dit[bf16x3] scalar time vs reference code", got.cpu(), torch.from_numpy(g["out_scalar_time"]))
    assert mean <= 2e-4


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_sample_vs_reference_code(models, method):
    cfg, ms = models
    g = load("ref_sample.npz")
    durations = torch.from_numpy(g["durations"])
    y0 = torch.zeros((2, int(durations.max()), cfg.mel_dim))
    for i, z in enumerate((g["z0"], g["z1"])):
        y0[i, :z.shape[1]] = torch.from_numpy(z.T)
    kw = dict(duration=durations, lens=torch.from_numpy(g["lens"]), steps=int(g[f"steps_{method}"]), method=method, cfg_strength=2.0,
              sway_sampling_coef=-1.0, y0=y0)
    want_out, want_traj = torch.from_numpy(g[f"out_{method}"]), torch.from_numpy(g[f"traj_{method}"])
    out, traj = F5TTS(transformer=ms["bf16x3"]).sample(torch.from_numpy(g["cond"]), torch.from_numpy(g["text"]), **kw)
    assert traj.shape == want_traj.shape and torch.equal(traj[0].cpu(), want_traj[0])
    _, l1, _ = report(f"sample[bf16x3] {method} final mel vs reference code", out.cpu(), want_out)
    _, l1t, _ = report(f"sample[bf16x3] {method} trajectory vs reference code", traj.cpu(), want_traj)
    assert l1 <= MEL_L1_TOL and l1t <= MEL_L1_TOL
    # conditioning frames are spliced back bit-exactly (cfm.py:395-397)
    cm = lens_to_mask(torch.maximum((torch.from_numpy(g["text"]) != -1).sum(-1), torch.from_numpy(g["lens"])), out.shape[1])
    assert torch.equal(out.cpu()[cm], want_out[cm])
    outb, _ = F5TTS(transformer=ms["bf16"]).sample(torch.from_numpy(g["cond"]), torch.from_numpy(g["text"]), **kw)
    _, l1b, _ = report(f"sample[bf16] {method} drift vs reference code", outb.cpu(), want_out)
    assert l1b <= 5e-2


def test_sample_batch1_no_cfg_vs_reference_code(models):
    cfg, ms = models
    g = load("ref_sample.npz")
    y0 = torch.from_numpy(g["z0"].T.copy())[None]
    out, traj = F5TTS(transformer=ms["bf16x3"]).sample(torch.from_numpy(g["cond"][:1]), torch.from_numpy(g["text"][:1]),
                                                       duration=int(g["durations"][0]), steps=4, method="euler", cfg_strength=0.0,
                                                       sway_sampling_coef=None, y0=y0)
    _, l1, _ = report("sample[bf16x3] B1 no-cfg vs reference code", out.cpu(), torch.from_numpy(g["out_b1_nocfg"]))
    _, l1t, _ = report("sample[bf16x3] B1 no-cfg trajectory vs reference code", traj.cpu(), torch.from_numpy(g["traj_b1_nocfg"]))
    assert l1 <= MEL_L1_TOL and l1t <= MEL_L1_TOL


def test_cfm_loss_vs_reference_code(models):
    cfg, ms = models
    g = load("ref_cfm_loss.npz")
    tts = F5TTS(ms["bf16x3"])
    for name, (ra, rc) in dict(keep=(0.9, 0.9), drop_audio=(0.1, 0.9), drop_both=(0.9, 0.1)).items():
        rand = dict(x0=torch.from_numpy(g["x0"]), time=torch.from_numpy(g["time"]), frac_lengths=torch.from_numpy(g["frac_lengths"]),
                    span_rand=torch.from_numpy(g["span_rand"]), rand_audio_drop=ra, rand_cond_drop=rc)
        got = float(tts(torch.from_numpy(g["mel"]), torch.from_numpy(g["text"]), lens=torch.from_numpy(g["lens"]), rand=rand))
        want = float(g["loss_" + name])
        print(f"cfm loss [{name}]: engine {got:.6f} reference code {want:.6f}")
        assert abs(got - want) <= 2e-4 * want


def test_host_side_and_mel_front_end_vs_reference_code():
    """index paths bit-exact; the HIP mel front-end within the mel-L1 gate of audio.py's output"""
    from f5_tts_mlx_amd.audio import MelSpec
    g = load("ref_host.npz")
    lens = torch.from_numpy(g["lens"])
    assert np.array_equal(lens_to_mask(lens).numpy(), g["lens_mask"])
    assert np.array_equal(lens_to_mask(lens, 12).numpy(), g["lens_mask_len12"])
    fm = mask_from_frac_lengths(torch.tensor([20, 31, 8], dtype=torch.int32), torch.tensor([0.7, 0.85, 1.0]), max_length=32,
                                rand=torch.tensor([0.5, 0.1, 0.99]))
    assert np.array_equal(fm.numpy(), g["frac_mask"])
    assert np.array_equal(list_str_to_tensor(["hello", "héllo wörld", ""]).numpy(), g["utf8"])
    assert np.array_equal(list_str_to_idx(json.loads(str(g["tok_in"])), json.loads(str(g["vocab"]))).numpy(), g["tok_idx"])
    mel = MelSpec()(torch.from_numpy(g["audio"]).to(DEV)).cpu()
    ref = torch.from_numpy(g["mel"])
    _, l1, _ = report("mel front-end vs reference code", mel, ref)
    assert mel.shape == ref.shape and l1 <= 1e-4


def test_duration_predictor_vs_reference_code():
    """DurationPredictor on the HIP ops vs duration.py's own output (tests/golden/ref_duration.npz)"""
    from f5_tts_mlx_amd.duration import DurationPredictor, DurationTransformer, synthetic_duration_weights
    g = load("ref_duration.npz")
    kw = json.loads(str(g["cfg"]))
    w = synthetic_duration_weights(seed=int(g["weights_seed"]), **kw)
    for prec, tol in (("bf16x3", 2e-4), ("bf16", 2e-2)):
        dp = DurationPredictor(DurationTransformer(heads=8, precision=prec, device=DEV, **kw))
        dp.load_weights(w)
        for tag in ("b1", "b2_long_text"):
            got = dp(torch.from_numpy(g[f"{tag}_mel"]), torch.from_numpy(g[f"{tag}_text"]), lens=torch.from_numpy(g[f"{tag}_lens"]))
            mx, _, refm = report(f"duration predictor [{prec}] {tag} vs reference code", got.cpu(), torch.from_numpy(g[f"{tag}_seconds"]))
            assert mx <= tol * max(1.0, refm), (prec, tag)


def test_generate_vs_reference_code(models, tmp_path):
    """generate() (generate.py:113-245) end to end vs the reference's own function run over the shim: RMS rule, seconds -> frames,
    estimated / predicted durations, sentence loop, text conversion, reference trim by samples, WAV output.  Stand-ins shared by
    both sides (see make_reference_golden.py §6): fake vocoder, character vocabulary, rng.py noise for `seed`."""
    import os as _os
    from f5_tts_mlx_amd import generate as G
    from f5_tts_mlx_amd.duration import DurationPredictor, DurationTransformer, synthetic_duration_weights
    cfg, ms = models
    g = load("ref_generate.npz")
    vocab = {c: i for i, c in enumerate(str(g["vocab"]))}
    dkw = json.loads(str(g["dur_cfg"]))
    dp = DurationPredictor(DurationTransformer(heads=8, precision="bf16x3", device=DEV, **dkw))
    dp.load_weights(synthetic_duration_weights(seed=int(g["dur_seed"]), **dkw))
    idx = torch.arange(256, device=DEV) % 100
    f5 = F5TTS(transformer=ms["bf16x3"], vocab_char_map=vocab, vocoder=lambda mel: mel[0][:, idx].reshape(-1), duration_predictor=dp)
    wav = _os.path.join(_os.path.dirname(G.__file__), "assets", "test_en_1_ref_short.wav")
    for tag, kw in json.loads(str(g["cases"])).items():
        out = str(tmp_path / f"{tag}.wav")
        wave = G.generate(ref_audio_path=wav, ref_audio_text=str(g["caption"]), output_path=out, f5tts=f5, **kw)
        want = torch.from_numpy(g["wave_" + tag])
        assert wave.shape == want.shape, (tag, wave.shape, want.shape)          # identical frame counts on every sentence
        _, l1, _ = report(f"generate[{tag}] wave (fake vocoder = mel samples) vs reference code", wave.cpu(), want)
        assert l1 <= MEL_L1_TOL, tag
        sr, written = __import__("scipy.io.wavfile", fromlist=["read"]).read(out)
        assert sr == 24000 and np.array_equal(written, wave.cpu().numpy())


def test_sample_duration_clamps_vs_reference_code(models):
    """more text tokens than reference frames, duration raised to lens + 1 and clipped to max_duration (cfm.py:301-303,317-319)"""
    cfg, ms = models
    g = load("ref_sample_clamps.npz")
    nmax = int(g["max_duration"])
    y0 = torch.zeros((2, nmax, cfg.mel_dim))
    for i, z in enumerate((g["z0"], g["z1"])):
        y0[i, :z.shape[1]] = torch.from_numpy(z.T)
    out, traj = F5TTS(transformer=ms["bf16x3"]).sample(torch.from_numpy(g["cond"]), torch.from_numpy(g["text"]),
                                                       duration=torch.from_numpy(g["durations"]), lens=torch.from_numpy(g["lens"]),
                                                       steps=3, method="euler", y0=y0, max_duration=nmax)
    assert tuple(out.shape) == g["out"].shape
    _, l1, _ = report("sample[bf16x3] clamps final mel vs reference code", out.cpu(), torch.from_numpy(g["out"]))
    _, l1t, _ = report("sample[bf16x3] clamps trajectory vs reference code", traj.cpu(), torch.from_numpy(g["traj"]))
    assert l1 <= MEL_L1_TOL and l1t <= MEL_L1_TOL


def test_benched_f16_mode_vs_reference_code(models):
    """the mode bench.py times (`f16`: IEEE-half MFMA operands) against the vectors the REFERENCE's own code
    produced -- one hop, not engine -> oracle -> reference: DiT forwards (cond / null, masked / unmasked), the ragged batch-2
    sample() for the three solvers, batch 1 without CFG, and the duration clamps.  Gate: mel L1 <= 1e-3 on final mels and
    trajectories (the forward gets the relative bound the bf16 mode has, tightened by the three extra significand bits)."""
    cfg, ms = models
    m = ms["f16"]
    g = load("ref_dit_forward.npz")
    x, cond, text, time, mask = (torch.from_numpy(g[k]) for k in ("x", "cond", "text", "time", "mask"))
    for tag, (drop, mk) in dict(cond=(False, None), null=(True, None), cond_masked=(False, mask), null_masked=(True, mask)).items():
        got = m(x=x, cond=cond, text=text, time=time, drop_audio_cond=drop, drop_text=drop, mask=mk)
        _, mean, refm = report(f"dit[f16] {tag} vs reference code", got.cpu(), torch.from_numpy(g["out_" + tag]))
        assert mean <= 4e-3 * max(1.0, refm), tag
    g = load("ref_sample.npz")
    durations = torch.from_numpy(g["durations"])
    y0 = torch.zeros((2, int(durations.max()), cfg.mel_dim))
    for i, z in enumerate((g["z0"], g["z1"])):
        y0[i, :z.shape[1]] = torch.from_numpy(z.T)
    f5 = F5TTS(transformer=m)
    for method in ("euler", "midpoint", "rk4"):
        kw = dict(duration=durations, lens=torch.from_numpy(g["lens"]), steps=int(g[f"steps_{method}"]), method=method, cfg_strength=2.0,
                  sway_sampling_coef=-1.0, y0=y0)
        out, traj = f5.sample(torch.from_numpy(g["cond"]), torch.from_numpy(g["text"]), **kw)
        _, l1, _ = report(f"sample[f16] {method} final mel vs reference code", out.cpu(), torch.from_numpy(g[f"out_{method}"]))
        _, l1t, _ = report(f"sample[f16] {method} trajectory vs reference code", traj.cpu(), torch.from_numpy(g[f"traj_{method}"]))
        assert l1 <= MEL_L1_TOL and l1t <= MEL_L1_TOL, method
    out, traj = f5.sample(torch.from_numpy(g["cond"][:1]), torch.from_numpy(g["text"][:1]), duration=int(g["durations"][0]), steps=4,
                          method="euler", cfg_strength=0.0, sway_sampling_coef=None, y0=torch.from_numpy(g["z0"].T.copy())[None])
    _, l1, _ = report("sample[f16] B1 no-cfg vs reference code", out.cpu(), torch.from_numpy(g["out_b1_nocfg"]))
    assert l1 <= MEL_L1_TOL
    g = load("ref_sample_clamps.npz")
    nmax = int(g["max_duration"])
    y0 = torch.zeros((2, nmax, cfg.mel_dim))
    for i, z in enumerate((g["z0"], g["z1"])):
        y0[i, :z.shape[1]] = torch.from_numpy(z.T)
    out, _ = f5.sample(torch.from_numpy(g["cond"]), torch.from_numpy(g["text"]), duration=torch.from_numpy(g["durations"]),
                       lens=torch.from_numpy(g["lens"]), steps=3, method="euler", y0=y0, max_duration=nmax)
    _, l1, _ = report("sample[f16] clamps final mel vs reference code", out.cpu(), torch.from_numpy(g["out"]))
    assert l1 <= MEL_L1_TOL
