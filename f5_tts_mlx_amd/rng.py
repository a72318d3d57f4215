"""Initial-noise generation for F5TTS.sample (cfm.py:369-375).

The reference draws `mx.random.normal((num_channels, dur))` after `mx.random.seed(seed)`.  MLX's PRNG
is third-party arithmetic (threefry2x32 counter RNG); MLX is not installable here, so the emulation
below is written from the published algorithm and is UNVERIFIED against MLX itself.  Parity tests
therefore inject `y0` explicitly; this module only makes `seed=` deterministic and cheap.

Published algorithm (mlx/random.cpp):  key(seed) = (seed >> 32, seed & 0xffffffff);  the global key
sequence does  key, subkey = split(key)  per draw;  bits(n) = threefry2x32(key, counters) laid out as
[first outputs..., second outputs...];  uniform = bits / (2^32 - 1) scaled to [nextafter(-1,0), 1);
normal = sqrt(2) * erfinv(uniform).
"""
from __future__ import annotations

import numpy as np
from scipy.special import erfinv

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_M = np.uint64(0xFFFFFFFF)


def _rotl(x, r):
    return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & _M


def threefry2x32(key, c0, c1):
    """Threefry-2x32, 20 rounds.  key: (k0, k1) uint32; c0, c1: uint32 arrays."""
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    ks = (k0, k1, (k0 ^ k1 ^ np.uint64(0x1BD11BDA)) & _M)
    x0 = (c0.astype(np.uint64) + ks[0]) & _M
    x1 = (c1.astype(np.uint64) + ks[1]) & _M
    for i in range(5):
        for r in _ROT[i % 2]:
            x0 = (x0 + x1) & _M
            x1 = _rotl(x1, r) ^ x0
        x0 = (x0 + ks[(i + 1) % 3]) & _M
        x1 = (x1 + ks[(i + 2) % 3] + np.uint64(i + 1)) & _M
    return x0.astype(np.uint32), x1.astype(np.uint32)


def _bits(key, n):
    half = (n + 1) // 2
    c0 = np.arange(half, dtype=np.uint32)
    c1 = c0 + np.uint32(half)
    a, b = threefry2x32(key, c0, c1)
    return np.concatenate([a, b])[:n]


def _split(key):
    b = _bits(key, 4)
    return (b[0], b[1]), (b[2], b[3])


def mlx_like_normal(seed: int, shape) -> np.ndarray:
    """float32 normal draw emulating `mx.random.seed(seed); mx.random.normal(shape)` (unverified)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    key = (np.uint32(seed >> 32), np.uint32(seed & 0xFFFFFFFF))
    _, sub = _split(key)
    n = int(np.prod(shape))
    u = _bits(sub, n).astype(np.float32) / np.float32(0xFFFFFFFF)
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = np.minimum(u, np.nextafter(np.float32(1.0), np.float32(0.0)))
    u = (np.float32(1.0) - lo) * u + lo
    return (np.float32(np.sqrt(2.0)) * erfinv(u.astype(np.float64))).astype(np.float32).reshape(shape)
