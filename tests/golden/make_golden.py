"""Generate the committed golden fixtures from the oracle (fp64).  The reference (MLX) cannot be imported in
the build container, so these vectors pin OUR restatement; rerun with
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from f5test import O, TINY, synth_inputs, synthetic_weights  # noqa: E402


def main():
    cfg = TINY
    w = synthetic_weights(cfg, seed=42)
    cond, text, durations, y0 = synth_inputs(cfg, 2, 72, nt=14, n_ref=20, seed=123, ragged=True)
    steps = 5
    out, traj = O.sample(O.DiTOracle(cfg, w, dtype=torch.float64), cond, text, torch.tensor(durations), y0=y0, steps=steps,
                         method="euler")
    np.savez_compressed(os.path.join(HERE, "tiny_sample_euler.npz"), weights_seed=42, cond=cond.numpy(), text=text.numpy(),
                        durations=np.asarray(durations), y0=y0.numpy(), steps=steps, out=out.numpy().astype(np.float32),
                        traj_last=traj[-1].numpy().astype(np.float32))
    import scipy.io.wavfile as wf
    sr, a = wf.read(os.path.join(ROOT, "f5_tts_mlx_amd", "assets", "test_en_1_ref_short.wav"))
    audio = (a.astype(np.float64) / 32768.0).astype(np.float32)
    mel = O.log_mel_spectrogram(audio, dtype=np.float64)[0].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "fixture_mel.npz"), mel=mel)
    # flow-matching loss forward (cfm.py:169-251) with every random draw recorded
    g = torch.Generator().manual_seed(77)
    B, N = 2, 64
    mel_in = torch.randn((B, N, cfg.mel_dim), generator=g)
    rand = dict(x0=torch.randn((B, N, cfg.mel_dim), generator=g), time=torch.tensor([0.21, 0.83]),
                frac_lengths=torch.tensor([0.75, 0.9]), span_rand=torch.tensor([0.35, 0.6]))
    ltext = torch.randint(0, cfg.text_num_embeds, (B, 16), generator=g, dtype=torch.int32)
    lens = torch.tensor([64, 51], dtype=torch.int32)
    losses = {}
    for name, (ra, rc) in dict(keep=(0.9, 0.9), drop_audio=(0.1, 0.9), drop_both=(0.9, 0.1)).items():
        losses[name] = float(O.cfm_loss(O.DiTOracle(cfg, w, dtype=torch.float64), mel_in, ltext, lens=lens, rand_audio_drop=ra,
                                        rand_cond_drop=rc, **rand))
    np.savez_compressed(os.path.join(HERE, "tiny_cfm_loss.npz"), weights_seed=42, mel=mel_in.numpy(), text=ltext.numpy(), lens=lens.numpy(),
                        x0=rand["x0"].numpy(), time=rand["time"].numpy(), frac_lengths=rand["frac_lengths"].numpy(),
                        span_rand=rand["span_rand"].numpy(), loss_keep=losses["keep"], loss_drop_audio=losses["drop_audio"],
                        loss_drop_both=losses["drop_both"])
    # MX-fp8 quantisation (oracle/mx_oracle.py): bytes and scales of a fixed tensor
    from oracle import mx_oracle as MX
    xq = (torch.randn((8, 128), generator=g) * torch.logspace(-3, 2, 8)[:, None]).float()
    q, e8 = MX.mx_quantize(xq)
    np.savez_compressed(os.path.join(HERE, "mx_quantize.npz"), x=xq.numpy(), q=q.numpy(), e8=e8.numpy())
    print("wrote golden fixtures", out.shape, mel.shape, losses)


if __name__ == "__main__":
    main()
