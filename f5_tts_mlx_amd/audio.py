"""Mel front-end (reference: f5_tts_mlx/audio.py) on the HIP engine.

`mel_filters` / `hanning` build small constant tables on the host exactly as the reference does
(float32 arithmetic, HTK scale, no norm); the STFT + filterbank + log run in one HIP kernel
(`f5_mel_spectrogram`, csrc/audio.hip).
"""
from __future__ import annotations

import ctypes as C
import math
from functools import lru_cache
from typing import Optional

import numpy as np
import torch

from . import engine as _eng


@lru_cache(maxsize=None)
def mel_filters(sample_rate: int, n_fft: int, n_mels: int, f_min: float = 0, f_max: Optional[float] = None,
                norm: Optional[str] = None, mel_scale: str = "htk") -> np.ndarray:
    """audio.py:12-98.  Returns (n_mels, n_fft // 2 + 1) float32 (torch-compatible filterbank)."""
    if mel_scale != "htk":
        raise NotImplementedError("only the HTK mel scale is used on the sampling path (audio.py:188)")

    def hz_to_mel(freq):
        return 2595.0 * math.log10(1.0 + freq / 700.0)

    f_max = f_max or sample_rate / 2
    n_freqs = n_fft // 2 + 1
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs, dtype=np.float32)
    m_pts = np.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2, dtype=np.float32)
    f_pts = (np.float32(700.0) * (np.float32(10.0) ** (m_pts / np.float32(2595.0)) - np.float32(1.0))).astype(np.float32)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down_slopes = (-slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    filterbank = np.maximum(np.float32(0), np.minimum(down_slopes, up_slopes))
    if norm == "slaney":
        enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
        filterbank = filterbank * enorm[None, :]
    return np.ascontiguousarray(filterbank.T.astype(np.float32))


@lru_cache(maxsize=None)
def hanning(size: int) -> np.ndarray:
    """audio.py:101-112 — periodic Hann window."""
    return np.hanning(size + 1)[:-1].astype(np.float32)


_dev_tables = {}


def _tables(device: torch.device, sample_rate: int, n_fft: int, n_mels: int):
    key = (str(device), sample_rate, n_fft, n_mels)
    if key not in _dev_tables:
        _dev_tables[key] = (torch.from_numpy(hanning(n_fft)).to(device),
                            torch.from_numpy(mel_filters(sample_rate, n_fft, n_mels)).to(device))
    return _dev_tables[key]


def log_mel_spectrogram(audio: torch.Tensor, sample_rate: int = 24_000, n_mels: int = 100, n_fft: int = 1024,
                        hop_length: int = 256, padding: int = 0, device=None) -> torch.Tensor:
    """audio.py:162-210.  audio: [t] or [b, t] (device tensor or anything torch.as_tensor accepts).
    Returns (b, frames, n_mels) float32 on the GPU — the layout the reference code produces."""
    lib = _eng.load_library()
    audio = torch.as_tensor(audio)
    if device is not None:
        audio = audio.to(device)
    elif not audio.is_cuda:
        audio = audio.to("cuda")                                       # host input and no device given: current GPU
    audio = audio.to(torch.float32)
    if audio.ndim == 1:
        audio = audio[None]
    if padding > 0:
        audio = torch.nn.functional.pad(audio, (0, padding))
    audio = audio.contiguous()
    window, fb = _tables(audio.device, sample_rate, n_fft, n_mels)
    b, L = audio.shape
    frames = L // hop_length
    out = torch.empty((b, frames, n_mels), dtype=torch.float32, device=audio.device)
    # one launch for the whole batch (the reference loops over it in Python, audio.py:195), on the current stream of the device
    # that holds the audio, with that device current (a launch on another device's stream would be a device mismatch)
    with torch.cuda.device(audio.device):
        stream = _eng.stream_ptr(audio.device)
        _eng.check(lib.f5_mel_spectrogram_batch(_eng.ptr(audio), b, C.c_int64(L), _eng.ptr(window), _eng.ptr(fb), n_fft, hop_length,
                                                n_mels, _eng.ptr(out), stream), "f5_mel_spectrogram_batch")
    return out


class MelSpec:
    """audio.py:213-230."""

    def __init__(self, sample_rate=24_000, n_fft=1024, hop_length=256, n_mels=100, device=None):
        self.device = device                # where host audio is moved to (the owning model's device); None = current GPU
        self.sample_rate = sample_rate
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.n_mels = n_mels

    def __call__(self, audio, **kwargs) -> torch.Tensor:
        return log_mel_spectrogram(audio, sample_rate=self.sample_rate, n_mels=self.n_mels, n_fft=self.n_fft,
                                   hop_length=self.hop_length, device=self.device)
