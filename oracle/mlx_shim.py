"""A numpy emulation of the slice of the `mlx` API that the reference (`/root/reference/f5_tts_mlx`) uses.

TEST INFRASTRUCTURE ONLY.  Purpose: MLX cannot be installed in the build container, so the reference's own Python source
cannot run as is.  With this shim registered under the module names `mlx`, `mlx.core`, `mlx.nn`, `einx` (plus empty stand-ins
for `vocos_mlx`, `jieba`, `pypinyin`, `soundfile`, `sounddevice`), the reference's *code* — DiT, CFM sampler, solvers, RoPE, ConvNeXt, mel front-end, masks,
tokenisers — executes unmodified on numpy arrays, and `tests/golden/make_reference_golden.py` records its outputs as golden
vectors that pin `oracle/f5_oracle.py` (and, through the GPU tests, the engine).

What this does and does not establish: control flow, layer order, index arithmetic, masking, quirks and every formula written in
the reference's Python are the reference's.  The meaning of each mlx primitive (Linear = x Wᵀ + b, Conv1d channels-last with
weight (out, k, in/groups), LayerNorm biased variance, GELU forms, SDPA boolean mask True = keep, default float32 / int32
dtypes, `as_strided` in elements, …) is this file's reading of the MLX documentation — SURVEY.md §8(c) lists the same
assumptions — and the arithmetic is numpy's, not MLX's kernels.

`install()` registers the modules; nothing else in the repository imports this file except the golden generator and tests.
"""
from __future__ import annotations

import math
import sys
import types

import numpy as np


# ------------------------------------------------------------------------------------------------
# array type: float32 / int32 discipline like MLX, plus the array methods the reference calls
# ------------------------------------------------------------------------------------------------
def _down(a: np.ndarray) -> np.ndarray:
    if a.dtype == np.float64:
        return a.astype(np.float32)
    if a.dtype == np.int64:
        return a.astype(np.int32)
    return a


class array(np.ndarray):
    def __new__(cls, obj, dtype=None):
        a = _down(np.asarray(obj))
        if dtype is not None:
            a = a.astype(dtype)
        return a.view(cls)

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        ins = [np.asarray(i) if isinstance(i, np.ndarray) else i for i in inputs]
        res = getattr(ufunc, method)(*ins, **kwargs)
        if isinstance(res, tuple):
            return tuple(_wrap(r) for r in res)
        return _wrap(res)

    # methods mlx arrays have and numpy arrays do not
    def sin(self): return _wrap(np.sin(np.asarray(self)))
    def cos(self): return _wrap(np.cos(np.asarray(self)))
    def exp(self): return _wrap(np.exp(np.asarray(self)))
    def log(self): return _wrap(np.log(np.asarray(self)))
    def sqrt(self): return _wrap(np.sqrt(np.asarray(self)))
    def square(self): return _wrap(np.square(np.asarray(self)))
    def abs(self): return _wrap(np.abs(np.asarray(self)))
    def split(self, n, axis=0): return [_wrap(p) for p in np.split(np.asarray(self), n, axis=axis)]
    def moveaxis(self, a, b): return _wrap(np.moveaxis(np.asarray(self), a, b))
    def expand(self, *shape):        # not an MLX method either (dit.py:162 uses it); broadcast, as the call intends
        return _wrap(np.broadcast_to(np.asarray(self), shape))
    def astype(self, dtype, *a, **k): return _wrap(np.asarray(self).astype(dtype))
    def max(self, *a, **k): return _wrap(np.asarray(self).max(*a, **k))
    def min(self, *a, **k): return _wrap(np.asarray(self).min(*a, **k))
    def sum(self, *a, **k): return _wrap(np.asarray(self).sum(*a, **k))
    def mean(self, *a, **k): return _wrap(np.asarray(self).mean(*a, **k))


def _wrap(x):
    if isinstance(x, (bool, int, float)):
        return x
    return _down(np.asarray(x)).view(array)


def _raw(x):
    return np.asarray(x) if isinstance(x, np.ndarray) else x


# ------------------------------------------------------------------------------------------------
# mlx.core
# ------------------------------------------------------------------------------------------------
def _build_core() -> types.ModuleType:
    mx = types.ModuleType("mlx.core")
    mx.array = array
    mx.float32, mx.float16, mx.int32, mx.int64, mx.uint32, mx.bool_ = np.float32, np.float16, np.int32, np.int64, np.uint32, np.bool_
    mx.bfloat16 = np.float32
    mx.pi = math.pi

    def arange(*args, dtype=None):
        a = np.arange(*args)
        return _wrap(a.astype(dtype) if dtype is not None else a)

    mx.arange = arange
    mx.linspace = lambda a, b, num=50, dtype=np.float32: _wrap(np.linspace(a, b, num).astype(dtype))
    mx.zeros = lambda shape, dtype=np.float32: _wrap(np.zeros(shape, dtype))
    mx.ones = lambda shape, dtype=np.float32: _wrap(np.ones(shape, dtype))
    mx.full = lambda shape, v, dtype=None: _wrap(np.full(shape, v) if dtype is None else np.full(shape, v, dtype))
    mx.zeros_like = lambda a: _wrap(np.zeros_like(_raw(a)))
    mx.ones_like = lambda a: _wrap(np.ones_like(_raw(a)))
    mx.expand_dims = lambda a, axis: _wrap(np.expand_dims(_raw(a), axis))
    mx.squeeze = lambda a, axis=None: _wrap(np.squeeze(_raw(a), axis=axis))
    mx.stack = lambda arrs, axis=0: _wrap(np.stack([_raw(a) for a in arrs], axis=axis))
    mx.concatenate = lambda arrs, axis=0: _wrap(np.concatenate([_raw(a) for a in arrs], axis=axis))
    mx.split = lambda a, n, axis=0: [_wrap(p) for p in np.split(_raw(a), n, axis=axis)]
    mx.where = lambda c, a, b: _wrap(np.where(_raw(c), _raw(a), _raw(b)))
    mx.maximum = lambda a, b: _wrap(np.maximum(_raw(a), _raw(b)))
    mx.minimum = lambda a, b: _wrap(np.minimum(_raw(a), _raw(b)))
    mx.clip = lambda a, lo, hi: _wrap(np.clip(_raw(a), lo, hi))
    mx.abs = lambda a: _wrap(np.abs(_raw(a)))
    mx.exp = lambda a: _wrap(np.exp(_raw(a)))
    mx.log = lambda a: _wrap(np.log(_raw(a)))
    mx.cos = lambda a: _wrap(np.cos(_raw(a)))
    mx.sin = lambda a: _wrap(np.sin(_raw(a)))
    mx.sqrt = lambda a: _wrap(np.sqrt(np.asarray(a, dtype=np.float32)))
    mx.sum = lambda a, axis=None, keepdims=False: _wrap(np.sum(_raw(a), axis=axis, keepdims=keepdims))
    mx.mean = lambda a, axis=None, keepdims=False: _wrap(np.mean(_raw(a), axis=axis, keepdims=keepdims))
    mx.square = lambda a: _wrap(np.square(_raw(a)))
    mx.outer = lambda a, b: _wrap(np.outer(_raw(a), _raw(b)))
    mx.matmul = lambda a, b: _wrap(np.matmul(_raw(a), _raw(b)))
    mx.einsum = lambda spec, *ops: _wrap(np.einsum(spec, *[_raw(o) for o in ops]))
    mx.eval = lambda *a, **k: None

    def load(path, format=None):
        """mx.load(path, format="safetensors") -> dict of arrays (cfm.py:441,474)"""
        from safetensors.numpy import load_file
        return {k: _wrap(v) for k, v in load_file(str(path)).items()}

    mx.load = load
    mx.compile = lambda fn, *a, **k: fn

    def pad(a, pad_width, mode="constant", constant_values=0):
        return _wrap(np.pad(_raw(a), pad_width, mode=mode, constant_values=constant_values))

    mx.pad = pad

    def as_strided(a, shape, strides, offset=0):        # MLX strides are in ELEMENTS
        a = np.ascontiguousarray(_raw(a))
        return _wrap(np.lib.stride_tricks.as_strided(a.reshape(-1)[offset:], shape=shape,
                                                     strides=[s * a.itemsize for s in strides]).copy())

    mx.as_strided = as_strided

    fft = types.ModuleType("mlx.core.fft")
    fft.rfft = lambda a, n=None, axis=-1: np.fft.rfft(_raw(a), n=n, axis=axis).astype(np.complex64).view(array)
    mx.fft = fft
    linalg = types.ModuleType("mlx.core.linalg")
    linalg.norm = lambda a, ord=None, axis=None, keepdims=False: _wrap(np.linalg.norm(_raw(a), ord=ord, axis=axis, keepdims=keepdims))
    mx.linalg = linalg

    # random: the MLX generator is not reproduced; golden scripts replace these functions by injected draws
    rnd = types.ModuleType("mlx.core.random")
    state = {"rng": np.random.default_rng(0)}
    rnd.seed = lambda s: state.__setitem__("rng", np.random.default_rng(int(s)))
    rnd.normal = lambda shape=(), dtype=np.float32: _wrap(state["rng"].standard_normal(shape).astype(dtype))
    rnd.uniform = lambda low=0.0, high=1.0, shape=(), dtype=np.float32: _wrap(state["rng"].uniform(low, high, shape).astype(dtype))
    mx.random = rnd

    fast = types.ModuleType("mlx.core.fast")

    def sdpa(q, k, v, *, scale, mask=None):
        q, k, v = _raw(q).astype(np.float32), _raw(k).astype(np.float32), _raw(v).astype(np.float32)
        s = np.matmul(q * np.float32(scale), np.swapaxes(k, -1, -2))
        if mask is not None:
            m = _raw(mask)
            s = np.where(m, s, -np.inf) if m.dtype == np.bool_ else s + m
        s = s - s.max(axis=-1, keepdims=True)
        e = np.exp(s)
        return _wrap(np.matmul(e / e.sum(axis=-1, keepdims=True), v))

    fast.scaled_dot_product_attention = sdpa
    mx.fast = fast
    return mx


# ------------------------------------------------------------------------------------------------
# mlx.nn
# ------------------------------------------------------------------------------------------------
def _build_nn(mx) -> types.ModuleType:
    nn = types.ModuleType("mlx.nn")

    class Module:
        def __init__(self):
            pass

        def eval(self):
            return self

        def train(self, mode=True):
            return self

        def named_modules(self, prefix=""):
            """(dotted path, module) for this module and everything below it: attributes that are Modules and the elements
            of list / tuple attributes (MLX names list elements `attr.N`)."""
            out = [(prefix, self)]
            for name, val in vars(self).items():
                if name.startswith("_"):
                    continue                      # MLX: attributes starting with "_" are not part of the parameter tree
                path = f"{prefix}.{name}" if prefix else name
                if isinstance(val, Module):
                    out += val.named_modules(path)
                elif isinstance(val, (list, tuple)):
                    for i, v in enumerate(val):
                        if isinstance(v, Module):
                            out += v.named_modules(f"{path}.{i}")
            return out

        def parameters(self):
            """flat {dotted name: array} of every array attribute in the module tree (tree_flatten of MLX's nested dict)"""
            out = {}
            for path, mod in self.named_modules():
                for name, val in vars(mod).items():
                    if not name.startswith("_") and isinstance(val, np.ndarray):
                        out[f"{path}.{name}" if path else name] = val
            return out

        def load_weights(self, weights, strict=True):
            items = weights.items() if isinstance(weights, dict) else weights
            for name, value in items:
                obj = self
                parts = name.split(".")
                for part in parts[:-1]:
                    obj = obj[int(part)] if isinstance(obj, (list, tuple)) else getattr(obj, part)
                leaf = parts[-1]
                if strict and not hasattr(obj, leaf):
                    raise ValueError(f"load_weights: no parameter {name}")
                new = _wrap(np.asarray(value))
                if strict and hasattr(obj, leaf) and tuple(np.shape(getattr(obj, leaf))) != tuple(new.shape):
                    raise ValueError(f"load_weights: {name} has shape {tuple(new.shape)}, the module expects {tuple(np.shape(getattr(obj, leaf)))}")
                setattr(obj, leaf, new)
            return self

    class Linear(Module):
        def __init__(self, input_dims, output_dims, bias=True):
            super().__init__()
            self.weight = mx.zeros((output_dims, input_dims))
            if bias:
                self.bias = mx.zeros((output_dims,))

        def __call__(self, x):
            y = np.matmul(_raw(x), _raw(self.weight).T)
            if hasattr(self, "bias"):
                y = y + _raw(self.bias)
            return _wrap(y)

    class Embedding(Module):
        def __init__(self, num_embeddings, dims):
            super().__init__()
            self.weight = mx.zeros((num_embeddings, dims))

        def __call__(self, idx):
            return _wrap(_raw(self.weight)[_raw(idx)])

    class LayerNorm(Module):
        def __init__(self, dims, eps=1e-5, affine=True, bias=True):
            super().__init__()
            self.eps = eps
            if affine:
                self.weight = mx.ones((dims,))
                if bias:
                    self.bias = mx.zeros((dims,))

        def __call__(self, x):
            x = _raw(x).astype(np.float32)
            mu = x.mean(axis=-1, keepdims=True)
            var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
            y = (x - mu) / np.sqrt(var + np.float32(self.eps))
            if hasattr(self, "weight"):
                y = y * _raw(self.weight)
                if hasattr(self, "bias"):
                    y = y + _raw(self.bias)
            return _wrap(y)

    class RMSNorm(Module):
        def __init__(self, dims, eps=1e-5):
            super().__init__()
            self.eps = eps
            self.weight = mx.ones((dims,))

        def __call__(self, x):
            x = _raw(x).astype(np.float32)
            return _wrap(x / np.sqrt((x * x).mean(axis=-1, keepdims=True) + np.float32(self.eps)) * _raw(self.weight))

    class Conv1d(Module):
        """channels-last input (b, l, c_in); weight (c_out, k, c_in / groups); cross-correlation; zero padding"""

        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
            super().__init__()
            assert stride == 1 and dilation == 1
            self.padding, self.groups = padding, groups
            self.weight = mx.zeros((out_channels, kernel_size, in_channels // groups))
            if bias:
                self.bias = mx.zeros((out_channels,))

        def __call__(self, x):
            x = _raw(x).astype(np.float32)
            w = _raw(self.weight).astype(np.float32)
            b, l, cin = x.shape
            cout, k, cg = w.shape
            g = self.groups
            xp = np.pad(x, ((0, 0), (self.padding, self.padding), (0, 0)))
            lout = xp.shape[1] - k + 1
            win = np.lib.stride_tricks.sliding_window_view(xp, k, axis=1)          # (b, lout, cin, k)
            win = win.reshape(b, lout, g, cg, k)
            wg = w.reshape(g, cout // g, k, cg)
            y = np.einsum("blgck,gokc->blgo", win, wg).reshape(b, lout, cout)
            if hasattr(self, "bias"):
                y = y + _raw(self.bias)
            return _wrap(y)

    class Sequential(Module):
        def __init__(self, *layers):
            super().__init__()
            self.layers = list(layers)

        def __call__(self, x):
            for layer in self.layers:
                x = layer(x)
            return x

    def _erf(x):
        from scipy.special import erf
        return erf(x)

    class GELU(Module):
        def __init__(self, approx="none"):
            super().__init__()
            assert approx in ("none", "tanh", "precise", "fast")
            self.approx = approx

        def __call__(self, x):
            x = _raw(x).astype(np.float32)
            if self.approx == "tanh" or self.approx == "precise":
                return _wrap(0.5 * x * (1.0 + np.tanh(np.float32(math.sqrt(2.0 / math.pi)) * (x + np.float32(0.044715) * x ** 3))))
            if self.approx == "fast":
                return _wrap(x / (1.0 + np.exp(-1.702 * x)))
            return _wrap(0.5 * x * (1.0 + _erf(x / np.float32(math.sqrt(2.0)))).astype(np.float32))

    class SiLU(Module):
        def __call__(self, x):
            x = _raw(x).astype(np.float32)
            return _wrap(x / (1.0 + np.exp(-x)))

    class Mish(Module):
        def __call__(self, x):
            x = _raw(x).astype(np.float32)
            return _wrap(x * np.tanh(np.logaddexp(0.0, x)))

    class Softplus(Module):
        def __call__(self, x):
            return _wrap(np.logaddexp(0.0, _raw(x).astype(np.float32)))

    class Dropout(Module):
        def __init__(self, p=0.5):
            super().__init__()
            self.p = p

        def __call__(self, x):
            return x

    losses = types.ModuleType("mlx.nn.losses")

    def mse_loss(pred, target, reduction="mean"):
        d = (_raw(pred) - _raw(target)) ** 2
        return _wrap(d if reduction == "none" else (d.mean() if reduction == "mean" else d.sum()))

    losses.mse_loss = mse_loss
    nn.losses = losses
    for cls in (Module, Linear, Embedding, LayerNorm, RMSNorm, Conv1d, Sequential, GELU, SiLU, Mish, Softplus, Dropout):
        setattr(nn, cls.__name__, cls)

    class QuantizedLinear(Module):
        """Parameter SHAPES of mlx.nn.QuantizedLinear (MLX documentation of `mx.quantize`: uint32 words holding 32 / bits
        elements each along the input axis, one scale and one bias per group of `group_size` input elements); the packed matmul
        itself is not emulated -- the loader tests only need the tree that `load_weights(strict)` validates against."""

        def __init__(self, input_dims, output_dims, bias=True, group_size=64, bits=4):
            super().__init__()
            self.group_size, self.bits = group_size, bits
            self.weight = _wrap(np.zeros((output_dims, input_dims * bits // 32), np.uint32))
            self.scales = mx.zeros((output_dims, input_dims // group_size))
            self.biases = mx.zeros((output_dims, input_dims // group_size))
            if bias:
                self.bias = mx.zeros((output_dims,))

        def __call__(self, x):
            raise NotImplementedError("QuantizedLinear matmul is not emulated")

    def quantize(model, group_size=64, bits=4, class_predicate=None):
        """nn.quantize: replace, in place, every leaf module for which class_predicate(path, module) holds (default: modules
        with a `to_quantized` method, i.e. Linear and Embedding) by its quantised counterpart (cfm.py:510-515)."""
        pred = class_predicate or (lambda p, m: isinstance(m, (Linear, Embedding)))
        replaced = []
        for path, mod in model.named_modules():
            for name, val in list(vars(mod).items()):
                if name.startswith("_"):
                    continue
                sub = f"{path}.{name}" if path else name
                def swap(m, where):
                    if isinstance(m, Linear) and pred(where, m):
                        out_d, in_d = m.weight.shape
                        replaced.append(where)
                        return QuantizedLinear(in_d, out_d, bias=hasattr(m, "bias"), group_size=group_size, bits=bits)
                    return m
                if isinstance(val, Module):
                    setattr(mod, name, swap(val, sub))
                elif isinstance(val, list):
                    for i, v in enumerate(val):
                        if isinstance(v, Module):
                            val[i] = swap(v, f"{sub}.{i}")
        model._quantized_paths = replaced
        return model

    nn.QuantizedLinear = QuantizedLinear
    nn.quantize = quantize
    return nn


# ------------------------------------------------------------------------------------------------
# einx (only the elementwise named-axis calls the reference makes)
# ------------------------------------------------------------------------------------------------
def _build_einx() -> types.ModuleType:
    einx = types.ModuleType("einx")

    def _elementwise(fn):
        def call(pattern, *ops):
            lhs, out = pattern.split("->")
            ins = [s.split() for s in lhs.split(",")]
            out_axes = out.split()
            assert len(ins) == len(ops), pattern
            arrs = []
            for axes, op in zip(ins, ops):
                a = np.asarray(op)
                if not axes:                              # scalar operand ("b n, b n d, -> b n d")
                    arrs.append(a)
                    continue
                assert a.ndim == len(axes), (pattern, a.shape)
                a = np.transpose(a, [axes.index(ax) for ax in out_axes if ax in axes])
                dims = iter(a.shape)
                shape = [next(dims) if ax in axes else 1 for ax in out_axes]
                arrs.append(a.reshape(shape))
            return _wrap(fn(*arrs))
        return call

    einx.less = _elementwise(np.less)
    einx.greater_equal = _elementwise(np.greater_equal)
    einx.divide = _elementwise(np.divide)
    einx.multiply = _elementwise(np.multiply)
    einx.add = _elementwise(np.add)
    einx.where = _elementwise(np.where)
    return einx


_INSTALLED = False


def install():
    """Register the emulation under the module names the reference imports.  Idempotent."""
    global _INSTALLED
    if _INSTALLED:
        return sys.modules["mlx.core"], sys.modules["mlx.nn"]
    mx = _build_core()
    nn = _build_nn(mx)
    mlx = types.ModuleType("mlx")
    mlx.core, mlx.nn = mx, nn
    sys.modules.update({"mlx": mlx, "mlx.core": mx, "mlx.nn": nn, "mlx.nn.losses": nn.losses, "einx": _build_einx()})
    for name, attrs in (("vocos_mlx", {"Vocos": type("Vocos", (), {})}), ("jieba", {"setLogLevel": lambda *_a: None}),
                        ("pypinyin", {"lazy_pinyin": None, "Style": type("Style", (), {"TONE3": 0})}),
                        ("soundfile", {}), ("sounddevice", {})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    _INSTALLED = True
    return mx, nn
