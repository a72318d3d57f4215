"""Flow-matching sampler (reference: f5_tts_mlx/cfm.py).

ein notation: b - batch, n - sequence, nt - text sequence, nw - raw wave length, d - dimension.

`F5TTS.sample` keeps the reference's signature, defaults, return tuple and error behaviour; the host
side reproduces the reference's scalar logic (text -> ids, lens/duration clamp, time grid), and the
GPU side — masks, noise splice, the ODE loop with classifier-free guidance, 2 x NFE DiT forwards —
runs inside one C-ABI call (`f5_sample`, hipGraph captured).  Tensors are torch (ROCm) tensors
instead of `mx.array`.
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Callable, List, Literal, Optional, Union

import numpy as np
import torch

from .audio import MelSpec
from .dit import DiT
from .engine import noise_normal
from .utils import (default, exists, fetch_from_hub, lens_to_mask, list_str_to_idx, list_str_to_tensor,
                    mask_from_frac_lengths)
from .weights import convert_upstream_weights, dequantize_mlx_checkpoint

# The three fixed-grid solvers (cfm.py:38-122) live in the engine, fused with the CFG combine (csrc/rowops.hip: ode_stage_kernel,
# sequenced by csrc/engine.hip: run_sample_body); `sample()` has no host-side ODE loop.  The module-level names the reference
# exports stay importable for user code that integrates its own vector field: one explicit Runge-Kutta driver parameterised by a
# tableau (stage offsets c, stage weights a of the previous stage only -- all three schemes are "chain" schemes -- and output
# weights b), over torch tensors instead of mx.arrays.


def _explicit_rk(func: Callable, y0: torch.Tensor, t, c, a, b) -> torch.Tensor:
    """Fixed-grid explicit RK: returns the states at every grid point, y0 first, shape (len(t), *y0.shape)."""
    states = [y0]
    y = y0
    for i in range(len(t) - 1):
        h = t[i + 1] - t[i]
        ks = []
        for cj, aj in zip(c, a):
            yj = y if not ks else y + (aj * h) * ks[-1]
            ks.append(func(t[i] + cj * h, yj))
        incr = None
        for bj, kj in zip(b, ks):
            if bj != 0.0:
                incr = bj * kj if incr is None else incr + bj * kj
        y = y + h * incr
        states.append(y)
    return torch.stack(states)


def odeint_euler(func, y0, t):
    """cfm.py:38-61: forward Euler on the grid t."""
    return _explicit_rk(func, y0, t, c=(0.0,), a=(0.0,), b=(1.0,))


def odeint_midpoint(func, y0, t):
    """cfm.py:64-91: explicit midpoint rule."""
    return _explicit_rk(func, y0, t, c=(0.0, 0.5), a=(0.0, 0.5), b=(0.0, 1.0))


def odeint_rk4(func, y0, t):
    """cfm.py:94-122: classic fourth-order Runge-Kutta."""
    return _explicit_rk(func, y0, t, c=(0.0, 0.5, 0.5, 1.0), a=(0.0, 0.5, 0.5, 1.0), b=(1 / 6, 2 / 6, 2 / 6, 1 / 6))


def time_grid(steps: int, sway_sampling_coef: Optional[float]) -> np.ndarray:
    """cfm.py:377-381 in float32: `steps` grid POINTS on [0, 1], optionally sway-warped."""
    t = np.linspace(0, 1, steps, dtype=np.float32)
    if exists(sway_sampling_coef):
        t = (t + np.float32(sway_sampling_coef) * (np.cos(np.float32(math.pi / 2) * t) - np.float32(1) + t)).astype(np.float32)
    return t


def prepare_lengths(text: torch.Tensor, cond_seq_len: int, batch: int, duration, lens, max_duration: int, method: str):
    """Host-side scalar logic of `F5TTS.sample` (cfm.py:288-319, :383-390) on CPU tensors.

    Returns (text int32 (b, nt), lens int64 (b,), duration int64 (b,), max_duration int).  Raises the
    reference's ValueErrors before anything reaches the GPU."""
    if not exists(lens):
        lens = torch.full((batch,), cond_seq_len, dtype=torch.int64)
    lens = torch.as_tensor(lens).to("cpu", torch.int64)
    text = torch.as_tensor(text).to("cpu", torch.int32)
    text_lens = (text != -1).sum(dim=-1)
    lens = torch.maximum(text_lens, lens)                                   # cfm.py:301-303
    if duration is None:
        raise ValueError("Duration must be provided or a duration predictor must be set.")   # cfm.py:310
    if isinstance(duration, int):
        duration = torch.full((batch,), duration, dtype=torch.int64)
    duration = torch.as_tensor(duration).to("cpu", torch.int64)
    duration = torch.maximum(lens + 1, duration)                            # cfm.py:317
    duration = torch.clip(duration, 0, max_duration)                        # cfm.py:318
    max_duration = int(duration.max().item())                               # cfm.py:319
    if method not in ("midpoint", "euler", "rk4"):
        raise ValueError(f"Unknown method: {method}")                       # cfm.py:390
    return text, lens, duration, max_duration


class F5TTS:
    """Conditional flow matching wrapper (cfm.py:128-167).  `__call__` evaluates the training loss forward-only
    (no autograd through the engine); the optimiser/trainer is out of scope."""

    def __init__(
        self,
        transformer: DiT,
        audio_drop_prob=0.3,
        cond_drop_prob=0.2,
        num_channels=None,
        mel_spec_module=None,
        mel_spec_kwargs: dict = dict(),
        frac_lengths_mask: tuple[float, float] = (0.7, 1.0),
        vocab_char_map: dict[str, int] | None = None,
        vocoder: Callable | None = None,
        duration_predictor=None,
    ):
        self.frac_lengths_mask = frac_lengths_mask
        # host audio goes to the MODEL's GPU, not to whichever device happens to be current
        self._mel_spec = default(mel_spec_module, MelSpec(**dict(dict(device=getattr(transformer, "device", None)), **mel_spec_kwargs)))
        num_channels = default(num_channels, self._mel_spec.n_mels)
        self.num_channels = num_channels
        self.audio_drop_prob = audio_drop_prob
        self.cond_drop_prob = cond_drop_prob
        self.transformer = transformer
        self.dim = transformer.dim
        self._vocab_char_map = vocab_char_map
        self._vocoder = vocoder
        self._duration_predictor = duration_predictor

    def eval(self):
        return self

    def __call__(self, inp, text, *, lens=None, rand: Optional[dict] = None, generator: Optional[torch.Generator] = None):
        """cfm.py:169-251 — flow-matching loss (value only).  Extension: `rand` may inject any of the reference's random
        draws {"frac_lengths" (b,), "span_rand" (b,), "x0" (b,n,d), "time" (b,), "rand_audio_drop", "rand_cond_drop"};
        the rest come from torch's CPU generator (MLX's stream is not reproduced)."""
        rand = dict(rand or {})
        dev = self.transformer.device
        inp = torch.as_tensor(inp)
        if inp.ndim == 2:                                                     # :177-180
            inp = self._mel_spec(inp)
            inp = inp.transpose(1, 2)     # the reference's "b d n -> b n d" on an already (b, n, d) array: kept as is
            assert inp.shape[-1] == self.num_channels
        inp = inp.to(torch.float32)
        batch, seq_len = inp.shape[:2]
        if isinstance(text, list):                                            # :185-190
            text = list_str_to_idx(text, self._vocab_char_map) if exists(self._vocab_char_map) else list_str_to_tensor(text)
            assert text.shape[0] == batch
        text = torch.as_tensor(text)
        if not exists(lens):
            lens = torch.full((batch,), seq_len, dtype=torch.int32)           # :193-194
        lens = torch.as_tensor(lens).cpu()
        mask = lens_to_mask(lens, length=seq_len)                             # :196

        def draw(name, shape, lo=0.0, hi=1.0):
            if name in rand:
                return torch.as_tensor(rand[name], dtype=torch.float32).reshape(shape)
            return torch.rand(shape, generator=generator) * (hi - lo) + lo

        frac_lengths = draw("frac_lengths", (batch,), *self.frac_lengths_mask)                      # :199
        rand_span_mask = mask_from_frac_lengths(lens, frac_lengths, max_length=seq_len,
                                                rand=draw("span_rand", (batch,)))                   # :200
        rand_span_mask = rand_span_mask & mask                                                      # :202-203
        x1 = inp.to(dev)
        x0 = (torch.as_tensor(rand["x0"], dtype=torch.float32) if "x0" in rand
              else torch.randn(tuple(x1.shape), generator=generator)).to(dev)                       # :209
        time = draw("time", (batch,))                                                               # :212
        t = time.to(dev)[:, None, None]
        phi = (1 - t) * x0 + t * x1                                                                 # :216
        flow = x1 - x0
        span = rand_span_mask.to(dev)
        cond = torch.where(span[..., None], torch.zeros_like(x1), x1)                               # :220-224
        rand_audio_drop = float(draw("rand_audio_drop", (1,)))                                      # :228-229
        rand_cond_drop = float(draw("rand_cond_drop", (1,)))
        drop_text = rand_cond_drop < self.cond_drop_prob
        drop_audio_cond = (rand_audio_drop < self.audio_drop_prob) or drop_text                     # :232
        pred = self.transformer(x=phi, cond=cond, text=text, time=time, drop_audio_cond=drop_audio_cond,
                                drop_text=drop_text)                                                # :234-241
        loss = torch.square(pred - flow)                                                            # :245
        m = span[..., None].expand(-1, -1, self.num_channels)
        masked = torch.where(m, loss, torch.zeros_like(loss))
        loss = masked.sum() / torch.clamp(m.sum().to(torch.float32), min=1e-6)                      # :249
        return loss.mean()

    def predict_duration(self, cond, text, speed: float = 1.0):
        """cfm.py:253-262."""
        duration_in_sec = self._duration_predictor(cond, text)
        frame_rate = self._mel_spec.sample_rate // self._mel_spec.hop_length
        return (duration_in_sec * frame_rate / speed).to(torch.int32)

    def sample(
        self,
        cond: torch.Tensor,                      # b n d  |  b nw
        text: Union[torch.Tensor, List[str], List[List[str]]],
        duration: Union[int, torch.Tensor, None] = None,
        *,
        lens: Optional[torch.Tensor] = None,
        steps=8,
        method: Literal["euler", "midpoint", "rk4"] = "rk4",
        cfg_strength=2.0,
        speed=1.0,
        sway_sampling_coef=-1.0,
        seed: Optional[int] = None,
        max_duration=4096,
        y0: Optional[torch.Tensor] = None,       # extension: inject the initial noise (b, n, d)
        use_graph="auto",                        # extension: True / False / "auto" (graph from the 2nd call of a shape on)
        pad_to: Optional[int] = None,            # extension: padded length of a SHARD of a larger batch = that batch's max duration
                                                 # (the key-padding mask of cfm.py:333-336 is then built even for a one-utterance shard)
    ) -> tuple[torch.Tensor, torch.Tensor]:
        self.eval()
        device = self.transformer.device
        cond = torch.as_tensor(cond)

        # raw wave (cfm.py:283-286): batch 1 only, like the reference's `rearrange("1 n -> n")`
        if cond.ndim == 2:
            assert cond.shape[0] == 1, "raw-wave conditioning supports batch 1 (cfm.py:284)"
            cond = self._mel_spec(cond[0])
            assert cond.shape[-1] == self.num_channels

        batch, cond_seq_len = cond.shape[:2]
        if isinstance(text, list):
            if exists(self._vocab_char_map):
                text = list_str_to_idx(text, self._vocab_char_map)
            else:
                text = list_str_to_tensor(text)
            assert text.shape[0] == batch
        if duration is None and self._duration_predictor is not None:
            duration = self.predict_duration(cond, text, speed)
        max_duration_cap = int(max_duration)
        text, lens, duration, max_duration = prepare_lengths(text, cond_seq_len, batch, duration, lens, max_duration, method)
        self.last_durations = [int(d) for d in duration.tolist()]     # extension: frames per element after the clamps (cfm.py:317-318)
        if pad_to is not None:
            # GRN (convnext_v2.py:16) and the unmasked conv-pos-embed (dit.py:251) see the padding, so a shard of a batch only
            # reproduces its rows of the unsharded call when it is padded to the WHOLE batch's maximum (dist.shard_batch)
            if int(pad_to) < max_duration:
                raise ValueError(f"pad_to={pad_to} is shorter than this shard's longest duration {max_duration}")
            if int(pad_to) > max_duration_cap:       # the cap prepare_lengths enforces on durations (cfm.py:318) holds for the padding too
                raise ValueError(f"pad_to={pad_to} exceeds max_duration={max_duration_cap}")
            max_duration = int(pad_to)
        cond = cond.to(device, torch.float32)

        # pad the conditioning mel to max_duration (cfm.py:321); masks are built on the GPU from lens/durations
        if max_duration >= cond_seq_len:
            cond_p = torch.zeros((batch, max_duration, self.num_channels), dtype=torch.float32, device=device)
            cond_p[:, :cond_seq_len] = cond
        else:  # mx.pad with a negative width raises in the reference
            raise ValueError("duration shorter than the conditioning audio")

        # noise input (cfm.py:369-375): channel-major draw per element, zero padded, "b d n -> b n d" -- drawn on the GPU
        # (csrc/noise.hip; rng.mlx_like_normal is the host restatement the tests compare it with).  The reference re-seeds with
        # the same `seed` for every element; without a seed every element gets fresh entropy.
        if y0 is None:
            durs = [int(d) for d in duration.tolist()]
            seeds = [seed if exists(seed) else int(np.random.SeedSequence().entropy % (1 << 63)) for _ in durs]
            y0 = noise_normal(seeds, durs, max_duration, self.num_channels, device)
        y0 = torch.as_tensor(y0).to(device, torch.float32).contiguous()
        assert y0.shape == (batch, max_duration, self.num_channels)

        t = time_grid(steps, sway_sampling_coef)

        out, trajectory = self.transformer.engine.sample(
            text.to(device).contiguous(), cond_p, lens.tolist(), duration.tolist(), y0, t, method=method,
            cfg_strength=float(cfg_strength), use_mask=(batch > 1 or pad_to is not None), use_graph=use_graph, return_trajectory=True)

        if exists(self._vocoder):
            out = self._vocoder(out)
        return out, trajectory

    # ------------------------------------------------------------------------------------------
    def load_weights(self, weights) -> None:
        """`weights`: dict or list of (name, array) with reference (MLX-layout) names (cfm.py:517)."""
        self.transformer.load_weights(dict(weights))

    @classmethod
    def from_pretrained(cls, hf_model_name_or_path: str, convert_weights=None, quantization_bits: int | None = None,
                        precision: str = "f16", device: str = "cuda:0",
                        vocoder_name_or_path: str | None = "lucasnewman/vocos-mel-24khz",
                        allow_missing_vocoder: bool = False, verify_precision: bool = True) -> "F5TTS":
        """cfm.py:404-520.  Loads `model_v1.safetensors` (or `model_v1_{4,8}b.safetensors`) + `vocab.txt` (+ the duration
        predictor `duration_v2.safetensors` when present) from a local directory or the HF hub (network required), and the
        Vocos vocoder (cfm.py:446) from `vocoder_name_or_path`, `$F5_VOCOS_PATH`, or the hub.  MLX int4/int8 checkpoints are
        expanded to fp32 on load (weights.dequantize_mlx_checkpoint) and run like a full-precision checkpoint.  If the vocoder
        cannot be found this raises, like the reference's `Vocos.from_pretrained` (cfm.py:446); pass
        `allow_missing_vocoder=True` (or `vocoder_name_or_path=None`) to get a model whose sample() returns mel frames.
        verify_precision (precision "f16" only; no reference counterpart): the FIRST sample() call of the loaded checkpoint is also run
        in precision "bf16x3" and compared (engine.Engine `verify_calls`); if fp16's 11 significand bits lose more than the 1e-3 parity
        gate on it the model warns and stays on bf16x3.  Synthetic weights pass with a margin of 2.5-4; a real checkpoint has never been
        run here, so it is checked instead of trusted."""
        path = fetch_from_hub(hf_model_name_or_path, quantization_bits=quantization_bits)
        if path is None:
            raise ValueError(f"Could not find model {hf_model_name_or_path}")
        vocab_path = Path(path) / "vocab.txt"
        vocab = {v: i for i, v in enumerate(Path(vocab_path).read_text().split("\n"))}
        if len(vocab) == 0:
            raise ValueError(f"Could not load vocab from {vocab_path}")
        from safetensors.numpy import load_file

        # duration predictor (cfm.py:425-442)
        duration_predictor = None
        duration_model_path = Path(path) / "duration_v2.safetensors"
        if duration_model_path.exists():
            from .duration import DurationPredictor, DurationTransformer
            duration_predictor = DurationPredictor(
                transformer=DurationTransformer(dim=512, depth=8, heads=8, text_dim=512, ff_mult=2, conv_layers=2,
                                                text_num_embeds=len(vocab) - 1, precision=precision, device=device),
                vocab_char_map=vocab)
            duration_predictor.load_weights(load_file(str(duration_model_path)))

        # vocoder (cfm.py:446, 471)
        vocoder = None
        import os
        from .vocos import Vocos
        vprec = precision if precision in ("bf16", "bf16x3", "f16") else "bf16"     # the MX-fp8 mode covers the DiT block GEMMs only
        for cand in (vocoder_name_or_path, os.environ.get("F5_VOCOS_PATH")):
            if cand and Path(cand).exists():
                vocoder = Vocos.from_pretrained(cand, precision=vprec, device=device).decode
                break
        if vocoder is None and vocoder_name_or_path:
            try:
                from huggingface_hub import snapshot_download
                vdir = snapshot_download(repo_id=vocoder_name_or_path, allow_patterns=["*.safetensors", "*.yaml", "*.json"])
                vocoder = Vocos.from_pretrained(vdir, precision=vprec, device=device).decode
            except Exception as exc:
                if not allow_missing_vocoder:
                    raise RuntimeError(
                        f"vocoder {vocoder_name_or_path!r} unavailable ({type(exc).__name__}: {exc}).  Point "
                        "vocoder_name_or_path or $F5_VOCOS_PATH at a local vocos-mel-24khz directory, or pass "
                        "allow_missing_vocoder=True to get mel frames out of sample().") from exc
                print(f"[f5_tts_mlx_amd] vocoder {vocoder_name_or_path!r} unavailable ({type(exc).__name__}); "
                      f"sample() will return mel frames.")

        model_filename = "model_v1.safetensors"                            # cfm.py:448-453
        if exists(quantization_bits):
            model_filename = f"model_v1_{quantization_bits}b.safetensors"
            convert_weights = False
        else:
            convert_weights = default(convert_weights, True)
        weights = load_file(str(Path(path) / model_filename))
        if convert_weights:
            weights = convert_upstream_weights(weights)
        if exists(quantization_bits):
            weights = dequantize_mlx_checkpoint(weights, quantization_bits)
        weights = {k: np.asarray(v, dtype=np.float32) for k, v in weights.items()}
        f5tts = cls(
            transformer=DiT(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4,
                            text_num_embeds=len(vocab) - 1, text_mask_padding=True, precision=precision, device=device),
            vocab_char_map=vocab,
            vocoder=vocoder,
            duration_predictor=duration_predictor,
        )
        f5tts.load_weights(weights)
        if verify_precision and precision == "f16":
            f5tts.transformer.engine.verify_calls = 1
        return f5tts
