#!/usr/bin/env python
"""Produce the two known-answer files that would pin the last unpinned rows of the hot path -- to be run
by anyone with the reference's own runtime (Apple silicon or any machine where `pip install mlx vocos-mlx` works); nothing in
this repository can run it (no mlx here, no network).

    python tools/make_mlx_goldens.py [--out tests/golden] [--vocos lucasnewman/vocos-mel-24khz]

Writes
  tests/golden/mlx_rng.npz     `mx.random.seed(s); mx.random.normal((100, d))` -- exactly what f5_tts_mlx/cfm.py:369-375 draws --
                               for a few (seed, duration) pairs, incl. odd element counts and a 64-bit seed, plus the raw key
                               material (`mx.random.key(s)`, `mx.random.split`) so a mismatch can be localised
  tests/golden/mlx_vocos.npz   `vocos_mlx.Vocos.from_pretrained(name).decode(mel)` for a seeded mel (1, 64, 100) and (2, 33, 100),
                               plus the flattened parameter names and shapes of the checkpoint (cfm.py:446)

tests/test_mlx_goldens.py activates by itself once the files exist: the RNG vectors must match rng.mlx_like_normal (host) and
f5_noise_normal (device) BIT FOR BIT; the vocoder names / shapes must map onto vocos.vocos_param_specs, and -- when the checkpoint
itself is reachable through $F5_VOCOS_PATH -- the HIP vocoder must reproduce the waves to 1e-4.
Only numpy is used for the file format, so the outputs load anywhere."""
import argparse
import json
import os

import numpy as np

RNG_CASES = [(0, 1), (0, 937), (3, 50), (1234, 333), (2 ** 40 + 7, 2), (2 ** 63 - 1, 937), (42, 4096)]
MEL_CASES = [(1, 64, 11), (2, 33, 12)]          # (batch, frames, numpy seed)


def rng_goldens(mx):
    out = {}
    for seed, dur in RNG_CASES:
        mx.random.seed(seed)
        x = mx.random.normal((100, dur))
        out[f"normal_s{seed}_d{dur}"] = np.array(x, dtype=np.float32)
        # consecutive draws advance the global key: the second draw pins the split rule
        y = mx.random.normal((100, 3))
        out[f"second_s{seed}_d{dur}"] = np.array(y, dtype=np.float32)
        k = mx.random.key(seed)
        out[f"key_s{seed}"] = np.array(k, dtype=np.uint32)
        out[f"split_s{seed}"] = np.array(mx.random.split(k), dtype=np.uint32)
        out[f"bits_s{seed}"] = np.array(mx.random.bits((7,), key=k), dtype=np.uint32)
        out[f"uniform_s{seed}"] = np.array(mx.random.uniform(-1.0, 1.0, (7,), key=k), dtype=np.float32)
    # a keyed normal without the global state, the element-count-odd case
    out["normal_key5_shape3x5"] = np.array(mx.random.normal((3, 5), key=mx.random.key(5)), dtype=np.float32)
    return out


def vocos_goldens(mx, name):
    from mlx.utils import tree_flatten
    from vocos_mlx import Vocos
    voc = Vocos.from_pretrained(name)
    out = {}
    params = tree_flatten(voc.parameters())
    out["param_names_json"] = np.frombuffer(json.dumps([[k, list(v.shape), str(v.dtype)] for k, v in params]).encode(), dtype=np.uint8)
    for b, n, seed in MEL_CASES:
        mel = (np.random.default_rng(seed).standard_normal((b, n, 100)) * 2.0 - 1.0).astype(np.float32)
        wave = voc.decode(mx.array(mel))
        out[f"mel_b{b}_n{n}"] = mel
        out[f"wave_b{b}_n{n}"] = np.array(wave, dtype=np.float32)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    ap.add_argument("--vocos", default="lucasnewman/vocos-mel-24khz", help="checkpoint name or local path for vocos_mlx.Vocos.from_pretrained")
    ap.add_argument("--skip-vocos", action="store_true")
    args = ap.parse_args()
    import mlx
    import mlx.core as mx
    os.makedirs(args.out, exist_ok=True)
    meta = np.frombuffer(json.dumps({"mlx_version": getattr(mlx, "__version__", "unknown"), "default_device": str(mx.default_device())}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(args.out, "mlx_rng.npz"), meta_json=meta, **rng_goldens(mx))
    print("wrote", os.path.join(args.out, "mlx_rng.npz"))
    if not args.skip_vocos:
        np.savez_compressed(os.path.join(args.out, "mlx_vocos.npz"), meta_json=meta, **vocos_goldens(mx, args.vocos))
        print("wrote", os.path.join(args.out, "mlx_vocos.npz"))


if __name__ == "__main__":
    main()
