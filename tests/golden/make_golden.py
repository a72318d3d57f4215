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
    print("wrote golden fixtures", out.shape, mel.shape)


if __name__ == "__main__":
    main()
