"""ctypes binding of the C-ABI library (include/f5tts_hip.h) + a thin Engine object.

PyTorch is used for device memory, streams and torch.distributed only; every FLOP of the sampling
path runs in the hand-written HIP kernels of `csrc/`.  There is NO fallback: if the shared library
is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from .weights import DiTConfig, check_weights

# F5TTS_HIP_LIB: another build of the SAME library (A/B runs of two builds, tools/sample_ab.py); never a fallback -- a path that does
# not exist fails like a missing default build does
_LIB_PATH = Path(os.environ["F5TTS_HIP_LIB"]).resolve() if os.environ.get("F5TTS_HIP_LIB") else Path(__file__).resolve().parent / "csrc" / "libf5tts_hip.so"
_lib = None

METHODS = {"euler": 0, "midpoint": 1, "rk4": 2}
# MFMA operand encodings (include/f5tts_hip.h).  "f16": IEEE half operands, the one-pass mode that meets the 1e-3 mel-L1 gate;
# "bf16": bfloat16 operands (outside the gate by 2-4x); "bf16x3": hi/lo split, 3 passes, fp32-class; "mxfp8": MX-fp8 block GEMMs
# (gfx950 scaled MFMA), everything else bf16
PRECISIONS = {"bf16": 0, "bf16x3": 1, "mxfp8": 2, "f16": 3}
GRAPH_MODES = {False: 0, True: 1, "off": 0, "on": 1, "auto": 2}


def operand_dtype(precision: str) -> torch.dtype:
    """torch dtype of the 16-bit MFMA operand buffers of a precision mode."""
    return torch.float16 if precision == "f16" else torch.bfloat16


class operand_type:
    """Context manager: the per-op entry points (f5_op_*) take bf16 operands by default; inside this block they take the
    operand type of `precision` (process-wide switch, not re-entrant -- same contract as an engine handle)."""

    def __init__(self, precision: str):
        self.fp16 = 1 if precision == "f16" else 0
        self._saved = []

    def __enter__(self):
        lib = load_library()
        self._saved.append(int(lib.f5_op_get_operand_type()))      # nested blocks restore what they found, not bf16
        check(lib.f5_op_set_operand_type(self.fp16), "f5_op_set_operand_type")
        return self

    def __exit__(self, *exc):
        check(load_library().f5_op_set_operand_type(self._saved.pop()), "f5_op_set_operand_type")
        return False


class F5Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dim", "depth", "heads", "dim_head", "ff_dim", "mel_dim", "text_num_embeds", "text_dim", "text_ff_dim",
        "conv_layers", "conv_pos_kernel", "conv_pos_groups", "freq_embed_dim", "text_max_pos", "text_mask_padding")]


class F5SampleArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("N", C.c_int32), ("nt", C.c_int32),
        ("text", C.c_void_p), ("cond", C.c_void_p), ("lens", C.c_void_p), ("durations", C.c_void_p),
        ("y0", C.c_void_p), ("t", C.c_void_p),
        ("steps", C.c_int32), ("method", C.c_int32), ("cfg_strength", C.c_float),
        ("use_mask", C.c_int32), ("use_graph", C.c_int32),
        ("out", C.c_void_p), ("trajectory", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
    ]


def library_path() -> Path:
    return _LIB_PATH


def load_library() -> C.CDLL:
    """Load libf5tts_hip.so (built by `__graft_entry__.build()` / csrc/build.sh). Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise RuntimeError(
            f"HIP extension not built: {_LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or f5_tts_mlx_amd/csrc/build.sh). There is no CPU fallback.")
    lib = C.CDLL(str(_LIB_PATH))
    lib.f5_last_error.restype = C.c_char_p
    lib.f5_version.restype = C.c_int
    lib.f5_op_grn_scratch_floats.restype = C.c_size_t
    lib.f5_engine_destroy.restype = None
    lib.f5_debug_f2h_bits.restype = C.c_uint16
    lib.f5_debug_f2h_bits.argtypes = [C.c_float]
    lib.f5_debug_f2bf_bits.restype = C.c_uint16
    lib.f5_debug_f2bf_bits.argtypes = [C.c_float]
    lib.f5_debug_h_bits2f.restype = C.c_float
    lib.f5_debug_h_bits2f.argtypes = [C.c_uint16]
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load_library().f5_last_error().decode("utf-8", "replace")
        if msg.startswith("Unknown method"):
            raise ValueError(msg)
        raise RuntimeError(f"{what or 'f5 call'} failed (rc={rc}): {msg}")


def ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def stream_ptr(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def noise_normal(seeds: Sequence[int], durations: Sequence[int], N: int, mel: int, device: torch.device) -> torch.Tensor:
    """Initial noise of F5TTS.sample (cfm.py:369-375) drawn on the GPU: element b is the channel-major draw
    `mx.random.seed(seeds[b]); mx.random.normal((mel, durations[b]))`, zero padded to N frames -> (B, N, mel) fp32.
    Same numbers as rng.mlx_like_normal (the host restatement of MLX's published generator; unverifiable against MLX here)."""
    lib = load_library()
    B = len(durations)
    if len(seeds) != B:
        raise ValueError("one seed per batch element")
    device = torch.device(device)
    y0 = torch.empty((B, int(N), int(mel)), dtype=torch.float32, device=device)
    CH = 256                                                     # f5_noise_normal takes at most 256 elements per call
    scratch = torch.empty(3 * min(B, CH), dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        for b0 in range(0, B, CH):
            nb = min(CH, B - b0)
            c_seeds = (C.c_uint64 * nb)(*[int(s) & 0xFFFFFFFFFFFFFFFF for s in seeds[b0:b0 + nb]])
            c_durs = (C.c_int32 * nb)(*[int(d) for d in durations[b0:b0 + nb]])
            check(lib.f5_noise_normal(c_seeds, nb, c_durs, int(N), int(mel), ptr(y0[b0:b0 + nb]), ptr(scratch), stream_ptr(device)),
                  "f5_noise_normal")
    return y0


def to_c_config(cfg: DiTConfig) -> F5Config:
    return F5Config(cfg.dim, cfg.depth, cfg.heads, cfg.dim_head, cfg.ff_dim, cfg.mel_dim, cfg.text_num_embeds, cfg.text_dim,
                    cfg.text_ff_dim, cfg.conv_layers, cfg.conv_pos_kernel, cfg.conv_pos_groups, cfg.freq_embed_dim,
                    cfg.text_max_pos, int(bool(getattr(cfg, "text_mask_padding", True))))


def _aligned_bytes(nbytes: int, device: torch.device) -> torch.Tensor:
    """uint8 device buffer, 256-byte aligned (torch's caching allocator aligns to 512 B)."""
    t = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
    off = (-t.data_ptr()) % 256
    return t[off:off + int(nbytes)]


class OperandRangeWarning(RuntimeWarning):
    """A 16-bit MFMA operand left the fp16 range in a call: the folded LayerNorm operand (the engine fell back to the unfolded path) or
    any other producer (the engine fell back to precision bf16x3)."""


RANGE_CHECKS = ("sync", "auto", "async", "off")
OPTION_NAMES = ("q_premul", "qkv_transposed", "ln_fusion", "gemm_flags", "attn_pipe", "null_keeps_cond", "ln_fold", "sat_check",
                "graph_split", "fold_stats")
# Engine.split_batch: sample() of at least this many utterances runs as TWO half batches on two HIP streams (Engine._sample_split);
# 0 = never, the default.  The launches of a DiT block are a strict chain and each is 3.7 ... 11 rounds of one-workgroup-per-CU tiles, so
# its last, partly filled round idles CUs (profiles/r06/m_sweep_tile_rounds.jsonl: +22 ... 29 us at every round boundary); a second,
# independent chain gives the dispatcher workgroups to fill them with.  Measured at the 335M shape, N = 937, bit-identical results
# (profiles/r06/split_sample_ab.jsonl, two_streams_probe.jsonl): with every MFMA operand ZERO the split is -5.4 % at 32 utterances and
# -6.0 % at 64 -- the tails are real; with real operand values it is -0.1 / -0.4 / -3.0 % at 32 on three boxes, -2.0 % at 64, -3.5 % at 48:
# the chip runs these launches under its power cap, and what fills idle CUs comes back as clock.  Hence opt-in: worth it for >= 48.
SPLIT_MIN_BATCH = 0
STATUS_FOLD_OVERFLOW, STATUS_FOLD_RAN, STATUS_SATURATED = 1, 2, 4      # include/f5tts_hip.h F5_STATUS_*


class Engine:
    """One engine handle per device.  Not re-entrant (same contract as the C ABI).

    range_check -- what happens to the status word of a call (include/f5tts_hip.h f5_sample_status).  Only precision "f16" can set it:
    bit 0 = the folded LayerNorm operand (x - m)(1 + scale) left the fp16 range (LN fold active, batch >= 12 at the 335M shape), bit 2 =
    some other 16-bit operand producer (LN-modulate, q / k / v, the GELU output, conv-pos, ...) clamped a value beyond +-65 504 -- at
    every batch size, every block of every evaluation.  Whether that happens depends on the INPUT (reference audio, text, length) as well
    as on the checkpoint, so the default looks at every call:
      "sync"   (default) read the word after every call (one stream synchronisation).  Bit 0: switch this engine's ln_fold to 0, warn
               (OperandRangeWarning) and RE-RUN the call unfolded.  Bit 2: warn, build a precision-"bf16x3" engine from the host weights
               this engine was loaded from (fp32-class arithmetic, bf16's range: 3x the matrix work) and RE-RUN the call there; later
               calls go straight to it.  The caller always gets a result that is not saturated.
      "async"  never block: a 4-byte copy into pinned memory rides behind the call and is looked at when the NEXT call starts (or in
               synchronize() / check_status()); the flagged call's own output was saturated (finite, clamped) and the warning says so;
               the calls that follow run unfolded / on the bf16x3 engine
      "auto"   "sync" until `range_probation` (3) calls of the SAME shape in a row came back clean, "async" from then on; a new shape
               starts a new probation
      "off"    ignore the word
    keep_host_weights -- keep a reference to the dict given to load_weights (no copy) so that the bf16x3 fall-back can be built; without
    it (or when the arena arrived by broadcast) a saturated call can only be reported.

    verify_calls -- the range detector sees RANGE, not PRECISION: fp16's 11-bit significand can lose a result without any value
    reaching +-65 504 (large attention logits: outlier q / k rows scaled 300x give a finite, un-clamped mel that is 3.4e-2 off,
    tests/test_model_gpu.py::test_f16_range_stress_outlier_weights).  The next `verify_calls` sample() calls are therefore ALSO run on
    the bf16x3 engine and compared (mean |delta| of the final mel against `verify_tol` = the 1e-3 parity gate); beyond it the engine
    warns (OperandRangeWarning), hands back the bf16x3 result and stays on bf16x3.  0 by default (a cross-check costs a bf16x3 call);
    F5TTS.from_pretrained sets 1: a real checkpoint is cross-checked on its first call.
    """

    def __init__(self, cfg: DiTConfig, precision: str = "bf16", device: str | torch.device = "cuda:0", range_check: str = "sync",
                 keep_host_weights: bool = True, share_weights_with: Optional["Engine"] = None):
        if range_check not in RANGE_CHECKS:
            raise ValueError(f"range_check must be one of {RANGE_CHECKS}")
        if precision not in PRECISIONS:
            raise ValueError(f"unknown precision {precision!r}; expected one of {sorted(PRECISIONS)}")
        self.lib = load_library()
        self.cfg = cfg
        self.precision = precision
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP engine needs a GPU device (cuda:N); there is no CPU path")
        torch.cuda.set_device(self.device)
        self._h = C.c_void_p()
        ccfg = to_c_config(cfg)
        check(self.lib.f5_engine_create(C.byref(ccfg), PRECISIONS[precision], C.byref(self._h)), "f5_engine_create")
        nbytes = C.c_size_t()
        check(self.lib.f5_weights_bytes(self._h, C.byref(nbytes)), "f5_weights_bytes")
        # share_weights_with: a second handle on the finalised arena of another engine of the same configuration and precision
        # (f5_share_weights: nothing is copied; own workspace, graphs and status word) -- the sibling that runs the second half batch of
        # a split sample()
        if share_weights_with is None:
            self.arena = _aligned_bytes(nbytes.value, self.device)
            check(self.lib.f5_set_weights_arena(self._h, ptr(self.arena), C.c_size_t(self.arena.numel()), stream_ptr(self.device)),
                  "f5_set_weights_arena")
        else:
            check(self.lib.f5_share_weights(self._h, share_weights_with._h), "f5_share_weights")
            self.arena = share_weights_with.arena
        self._workspace: Optional[torch.Tensor] = None
        self.weights_ready = share_weights_with is not None
        self.split_batch = SPLIT_MIN_BATCH     # sample() of >= this many utterances: two half batches on two streams (0 = never: default)
        self.split_events = 0                  # calls that ran split
        self._sibling: Optional["Engine"] = None
        self._pool: Optional[ThreadPoolExecutor] = None
        self._split_seen: set = set()
        # hipGraph capture is illegal on the legacy default stream: the engine owns a side stream and
        # orders it against the caller's current stream with events (wait_stream)
        self._stream = torch.cuda.Stream(device=self.device)
        self.range_check = range_check
        self.range_probation = 3
        self.range_events = 0                  # calls whose folded operand overflowed (the engine fell back to ln_fold = 0)
        self.saturation_events = 0             # calls in which another 16-bit producer saturated (the engine fell back to bf16x3)
        self._clean_calls = 0                  # clean calls in a row of the shape `_clean_shape` ("auto")
        self._clean_shape = None
        # pinned ring of status words: one slot per call whose check is still pending ("async"); slots are handed out from a free list,
        # so a synchronous read can never be given the slot of a pending asynchronous one; a caller that runs 16 calls ahead of the GPU
        # waits for the oldest one
        self._status_host = torch.zeros(16, dtype=torch.int32).pin_memory() if precision == "f16" else None
        self._status_pending: list = []        # [(event, slot, what)] oldest first
        self._status_free = list(range(16))
        self.keep_host_weights = bool(keep_host_weights)
        self._host_weights: Optional[Dict[str, np.ndarray]] = None
        self._fallback: Optional["Engine"] = None      # the bf16x3 engine (built on demand from the host weights) ...
        self._use_fallback = False                     # ... and whether calls go to it (a call saturated, or a cross-check failed)
        self.verify_calls = 0
        self.verify_tol = 1e-3
        self.verify_events = 0                         # cross-checks that failed (the engine switched to bf16x3)
        self.last_verify_l1: Optional[float] = None

    def _run_on_side_stream(self, fn, cur=None):
        cur = torch.cuda.current_stream(self.device) if cur is None else cur      # (torch's current stream is per host thread)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            fn(C.c_void_p(self._stream.cuda_stream))
        cur.wait_stream(self._stream)

    def __del__(self):
        try:
            if getattr(self, "_pool", None) is not None:
                self._pool.shutdown(wait=True)
                self._pool = None
            if getattr(self, "_h", None) is not None and self._h.value:
                self.lib.f5_engine_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # ---- weights ---------------------------------------------------------------------------
    def load_weights(self, weights: Dict[str, np.ndarray]) -> None:
        """Upload reference-named fp32 tensors (weights.param_specs) into the arena."""
        check_weights(self.cfg, weights)
        for name, arr in weights.items():
            if name == "transformer.rotary_embed.inv_freq":
                continue
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            check(self.lib.f5_load_tensor(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.ndim, shape),
                  f"f5_load_tensor({name})")
        self.finalize()
        if self.precision == "f16" and self.keep_host_weights:
            self._host_weights = weights           # a reference, not a copy: what the bf16x3 fall-back is loaded from

    def finalize(self) -> None:
        check(self.lib.f5_finalize_weights(self._h, stream_ptr(self.device)), "f5_finalize_weights")
        self.weights_ready = True

    def set_graph_cache(self, max_graphs: int) -> None:
        """Bound the number of cached hipGraphExecs (default 8); the least recently used one is destroyed."""
        check(self.lib.f5_engine_set_graph_cache(self._h, int(max_graphs)), "f5_engine_set_graph_cache")
        if self._sibling is not None:
            self._sibling.set_graph_cache(max_graphs)

    def graph_count(self) -> int:
        return int(self.lib.f5_engine_graph_count(self._h))

    def set_option(self, name: str, value: int) -> None:
        """Per-engine launch option ("q_premul", "qkv_transposed", "ln_fusion", "gemm_flags", "attn_pipe", "null_keeps_cond", "ln_fold",
        "sat_check"; include/f5tts_hip.h): other engines of the process keep their own values, cached hipGraphs are keyed on them."""
        check(self.lib.f5_engine_set_option(self._h, name.encode(), int(value)), f"f5_engine_set_option({name})")
        if self._fallback is not None and name == "null_keeps_cond":
            self._fallback.set_option(name, value)
        if self._sibling is not None:
            self._sibling.set_option(name, value)

    def _ensure_sibling(self) -> "Engine":
        """A second handle on THIS engine's weights arena (own workspace, own hipGraphs, own status word) and the host thread that drives
        it: the second half batch of a split sample()."""
        if self._sibling is None:
            sib = Engine(self.cfg, precision=self.precision, device=self.device, range_check="off", keep_host_weights=False,
                         share_weights_with=self)
            for name in OPTION_NAMES:
                sib.set_option(name, self.get_option(name))
            sib.split_batch = 0
            self._sibling = sib
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="f5-half-batch")
        return self._sibling

    def _ensure_fallback(self) -> Optional["Engine"]:
        """The precision-"bf16x3" twin of this engine (fp32-class arithmetic, bf16's range), built once from the host weights."""
        if self._fallback is None and self._host_weights is not None:
            fb = Engine(self.cfg, precision="bf16x3", device=self.device, range_check="off")
            fb.load_weights(self._host_weights)
            fb.set_option("null_keeps_cond", self.get_option("null_keeps_cond"))
            self._fallback = fb
        return self._fallback

    def get_option(self, name: str) -> int:
        v = C.c_int()
        check(self.lib.f5_engine_get_option(self._h, name.encode(), C.byref(v)), f"f5_engine_get_option({name})")
        return int(v.value)

    def mark_loaded_from_broadcast(self) -> None:
        """Arena content arrived by a collective (dist.broadcast_weights) instead of load_weights."""
        check(self.lib.f5_mark_weights_loaded(self._h), "f5_mark_weights_loaded")
        self.finalize()

    # ---- workspace -------------------------------------------------------------------------
    def workspace(self, B: int, N: int, nt: int, steps: int, method: str) -> torch.Tensor:
        if method not in METHODS:
            raise ValueError(f"Unknown method: {method}")
        nbytes = C.c_size_t()
        check(self.lib.f5_workspace_bytes(self._h, B, N, nt, steps, METHODS[method], C.byref(nbytes)), "f5_workspace_bytes")
        if self._workspace is None or self._workspace.numel() < nbytes.value:
            self._workspace = None
            self._workspace = _aligned_bytes(nbytes.value, self.device)
        return self._workspace

    # ---- hot path --------------------------------------------------------------------------
    def _args(self, text, cond, lens, durations, y0, t, steps, method, cfg_strength, use_mask, use_graph, out, traj, ws):
        B, N, _ = cond.shape
        keep = dict(lens=np.ascontiguousarray(lens, dtype=np.int32), dur=np.ascontiguousarray(durations, dtype=np.int32),
                    t=np.ascontiguousarray(t, dtype=np.float32))
        a = F5SampleArgs()
        a.B, a.N, a.nt = B, N, text.shape[1]
        a.text, a.cond = text.data_ptr(), cond.data_ptr()
        a.lens, a.durations = keep["lens"].ctypes.data, keep["dur"].ctypes.data
        a.y0 = 0 if y0 is None else y0.data_ptr()
        a.t = keep["t"].ctypes.data
        a.steps, a.method, a.cfg_strength = steps, METHODS[method], float(cfg_strength)
        if use_graph not in GRAPH_MODES:
            raise ValueError(f"use_graph must be one of {list(GRAPH_MODES)}")
        a.use_mask, a.use_graph = int(use_mask), GRAPH_MODES[use_graph]
        a.out = 0 if out is None else out.data_ptr()
        a.trajectory = 0 if traj is None else traj.data_ptr()
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        return a, keep

    def _check_inputs(self, text, cond, lens, durations):
        assert cond.is_cuda and cond.dtype == torch.float32 and cond.is_contiguous() and cond.ndim == 3
        assert cond.shape[2] == self.cfg.mel_dim
        assert text.is_cuda and text.dtype == torch.int32 and text.is_contiguous() and text.ndim == 2
        assert text.shape[0] == cond.shape[0]
        assert len(lens) == cond.shape[0] and len(durations) == cond.shape[0]
        if not self.weights_ready:
            raise RuntimeError("weights not loaded")

    def sample(self, text: torch.Tensor, cond: torch.Tensor, lens: Sequence[int], durations: Sequence[int], y0: torch.Tensor,
               t: np.ndarray, method: str = "euler", cfg_strength: float = 2.0, use_mask: Optional[bool] = None,
               use_graph=True, return_trajectory: bool = True, out: Optional[torch.Tensor] = None,
               trajectory: Optional[torch.Tensor] = None):
        """ODE solve on the GPU. cond (B,N,mel) fp32 zero-padded to N=max(durations); text (B,nt) int32
        (-1 padded); y0 (B,N,mel).  use_graph: True / False / "auto" (eager on the first sighting of a shape
        signature, captured hipGraph from the second on).  Returns (out (B,N,mel), trajectory (steps,B,N,mel) or None)."""
        if method not in METHODS:
            raise ValueError(f"Unknown method: {method}")
        self._check_inputs(text, cond, lens, durations)
        self.check_status(block=False)                 # a pending check of the previous call: resolved here if it has arrived
        if self._use_fallback:                         # an earlier call saturated fp16 / failed its cross-check: bf16x3 from now on
            return self._fallback.sample(text, cond, lens, durations, y0, t, method=method, cfg_strength=cfg_strength, use_mask=use_mask,
                                         use_graph=use_graph, return_trajectory=return_trajectory, out=out, trajectory=trajectory)
        B, N, mel = cond.shape
        steps = int(len(t))
        assert y0.shape == cond.shape and y0.dtype == torch.float32 and y0.is_contiguous() and y0.is_cuda
        if use_mask is None:
            use_mask = B > 1                       # cfm.py:333-336
        if 0 < self.split_batch <= B and self.verify_calls == 0:
            done = self._sample_split(text, cond, lens, durations, y0, t, steps, method, cfg_strength, use_mask, use_graph,
                                      return_trajectory, out, trajectory)
            if done is not None:
                return done                        # (None: a half reported its status word -- the whole call is repeated below, unsplit)
        ws = self.workspace(B, N, text.shape[1], steps, method)
        if out is None:
            out = torch.empty_like(cond)
        if return_trajectory and trajectory is None:
            trajectory = torch.empty((steps, B, N, mel), dtype=torch.float32, device=self.device)
        a, keep = self._args(text, cond, lens, durations, y0, t, steps, method, cfg_strength, use_mask, use_graph, out,
                             trajectory if return_trajectory else None, ws)
        launch = lambda: self._run_on_side_stream(lambda st: check(self.lib.f5_sample(self._h, C.byref(a), st), "f5_sample"))  # noqa: E731
        launch()
        saturated = self._after_call(a, launch, f"sample(B={B}, N={N}, {method}, {steps} points)", (B, N, steps, method))
        del keep
        if saturated and self._use_fallback:           # ("sync": re-run THIS call in bf16x3, into the same output buffers)
            return self._fallback.sample(text, cond, lens, durations, y0, t, method=method, cfg_strength=cfg_strength, use_mask=use_mask,
                                         use_graph=use_graph, return_trajectory=return_trajectory, out=out, trajectory=trajectory)
        if self.verify_calls > 0 and self.precision == "f16" and self._ensure_fallback() is not None:
            self.verify_calls -= 1
            ref, ref_traj = self._fallback.sample(text, cond, lens, durations, y0, t, method=method, cfg_strength=cfg_strength,
                                                  use_mask=use_mask, use_graph=False, return_trajectory=return_trajectory)
            self.last_verify_l1 = l1 = float((out - ref).abs().mean())
            if not (l1 <= self.verify_tol):
                self.verify_events += 1
                self._use_fallback = True
                warnings.warn(f"fp16 precision cross-check failed in sample(B={B}, N={N}, {method}, {steps} points): mean |f16 - bf16x3| of the final "
                              f"mel is {l1:.3e} (tolerance {self.verify_tol:g}) although no operand left the fp16 range -- 11 significand bits "
                              "are not enough for this checkpoint / input (large attention logits, massive activations); this engine now runs "
                              "every call in precision 'bf16x3' (3x the matrix work) and returns the bf16x3 result of this call.",
                              OperandRangeWarning, stacklevel=2)
                out.copy_(ref)
                if return_trajectory:
                    trajectory.copy_(ref_traj)
        return out, (trajectory if return_trajectory else None)

    # ---- two half batches on two streams ---------------------------------------------------------
    def _sample_split(self, text, cond, lens, durations, y0, t, steps, method, cfg_strength, use_mask, use_graph, return_trajectory,
                      out, trajectory):
        """sample() of a large batch as two independent half batches: this engine runs utterances [0, h) on its stream, a sibling handle
        on the same weights arena runs [h, B) on its own, each enqueued by its own host thread (the HIP runtime lets a thread run only a
        few dozen launches ahead of the GPU, so one thread cannot keep two streams fed).  Utterances do not interact (both halves keep
        the padded length N and the batch's key-padding mask rule), so the result is bit-identical to the unsplit call
        (tests/test_model_gpu.py::test_split_sample_is_bit_identical).  What it gains and why it is opt-in: SPLIT_MIN_BATCH above.  The first call of a
        shape (graph capture) runs the halves one after the other.  The status words of both halves are read synchronously whatever
        `range_check` says (the host is blocked for the length of the call anyway); if either is set the result is dropped and None is
        returned: the caller repeats the whole call unsplit, with the warnings and fall-backs of the ordinary path."""
        B, N, mel = cond.shape
        h = (B + 1) // 2
        sib = self._ensure_sibling()
        cur = torch.cuda.current_stream(self.device)
        if out is None:
            out = torch.empty_like(cond)
        nt = text.shape[1]
        calls = []
        for eng, (b0, b1) in ((self, (0, h)), (sib, (h, B))):
            tr = torch.empty((steps, b1 - b0, N, mel), dtype=torch.float32, device=self.device) if return_trajectory else None
            ws = eng.workspace(b1 - b0, N, nt, steps, method)
            a, keep = eng._args(text[b0:b1], cond[b0:b1], lens[b0:b1], durations[b0:b1], y0[b0:b1], t, steps, method, cfg_strength,
                                use_mask, use_graph, out[b0:b1], tr, ws)
            calls.append((eng, a, keep, tr))

        def run(eng, a):
            torch.cuda.set_device(self.device)         # (the worker thread's current device)
            eng._run_on_side_stream(lambda st: check(eng.lib.f5_sample(eng._h, C.byref(a), st), "f5_sample"), cur)

        key = (B, N, nt, steps, method, use_graph, bool(use_mask), return_trajectory, cfg_strength >= 1e-5)
        if key in self._split_seen:
            fut = self._pool.submit(run, calls[1][0], calls[1][1])
            try:
                run(calls[0][0], calls[0][1])
            finally:
                fut.result()
        else:
            for eng, a, _, _ in calls:
                run(eng, a)
            self._split_seen.add(key)
        flags = 0
        if self._status_host is not None and self.range_check != "off":
            for eng, a, _, _ in calls:
                flags |= eng._status_word_blocking(a, cur)
        if flags & (STATUS_FOLD_OVERFLOW | STATUS_SATURATED):
            return None
        self.split_events += 1
        if return_trajectory:
            if trajectory is None:
                trajectory = torch.cat([calls[0][-1], calls[1][-1]], dim=1)
            else:
                trajectory[:, :h].copy_(calls[0][-1])
                trajectory[:, h:].copy_(calls[1][-1])
        return out, (trajectory if return_trajectory else None)

    def _status_word_blocking(self, a, cur=None) -> int:
        """The status word of the call described by `a`, read behind it on this engine's stream; waits for it."""
        if not self._status_free:
            self._status_pending[0][0].synchronize()
            self.check_status(block=False)
        slot = self._status_free.pop(0)
        holder = {}

        def enqueue_read(st):
            check(self.lib.f5_sample_status_async(self._h, C.byref(a), C.c_void_p(self._status_host[slot:].data_ptr()), st), "f5_sample_status_async")
            holder["ev"] = torch.cuda.Event()
            holder["ev"].record(torch.cuda.current_stream(self.device))

        self._run_on_side_stream(enqueue_read, cur)
        holder["ev"].synchronize()
        flags = int(self._status_host[slot])
        self._status_free.append(slot)
        return flags

    # ---- status word (fp16 operand range) ------------------------------------------------------
    def _fall_back(self, what: str, rerun: bool) -> None:
        self.range_events += 1
        self._clean_calls = 0
        self.set_option("ln_fold", 0)
        tail = ("re-running it unfolded" if rerun else
                "THAT call's output is saturated (finite, clamped at +-65504) -- repeat it, or construct the engine with range_check='sync'")
        warnings.warn(f"ln_fold: the residual stream times (1 + scale) left the fp16 range (|v| > 65504) in {what}; this engine now runs "
                      f"with ln_fold = 0 (or use precision 'bf16' / 'bf16x3'); {tail}.", OperandRangeWarning, stacklevel=4)

    def _saturation_fall_back(self, what: str, rerun: bool) -> None:
        """Bit 2 of the status word: some 16-bit operand producer clamped a value at +-65 504.  fp16 cannot carry this model on this
        input: build the bf16x3 engine (fp32-class arithmetic, bf16's range) from the host weights and send the calls there."""
        self.saturation_events += 1
        self._clean_calls = 0
        if self._ensure_fallback() is not None:
            self._use_fallback = True
            tail = ("re-running it in bf16x3" if rerun else
                    "THAT call's output is saturated (finite, wrong) -- repeat it, or construct the engine with range_check='sync'")
            how = "this engine now runs every call on a precision-'bf16x3' engine built from the same weights (3x the matrix work)"
        else:
            tail = "the output of that call is saturated (finite, wrong)"
            how = ("no host weights are kept (keep_host_weights=False / arena received by broadcast): construct the model with precision "
                   "'bf16x3' (or 'bf16')")
        warnings.warn(f"fp16 operands saturated: a value beyond +-65504 reached a 16-bit MFMA operand (LN-modulate, q / k / v, the GELU output, "
                      f"conv-pos, ...) in {what}; {how}; {tail}.", OperandRangeWarning, stacklevel=4)

    def _resolve(self, flags: int, what: str, shape, rerun: bool) -> bool:
        """Act on one status word; returns True when the call saturated (bit 2) -- "sync": the caller re-runs it on the fall-back."""
        if flags & STATUS_SATURATED:
            if not self._use_fallback or rerun:
                self._saturation_fall_back(what, rerun)
            return True
        if flags & STATUS_FOLD_OVERFLOW:
            if self.get_option("ln_fold") != 0:            # (calls enqueued before the first warning arrived are flagged too: once)
                self._fall_back(what, rerun)
            return False
        if shape == self._clean_shape:
            self._clean_calls += 1
        else:
            self._clean_shape, self._clean_calls = shape, 1
        return False

    def check_status(self, block: bool = True) -> int:
        """Resolve the pending status checks ("async" mode), oldest first; with block=False only those whose copy has arrived.
        Returns the OR of the words looked at.  Called at the start of every call and by synchronize()."""
        seen = 0
        while self._status_pending:
            ev, slot, what, shape = self._status_pending[0]
            if not block and not ev.query():
                break
            ev.synchronize()
            self._status_pending.pop(0)
            flags = int(self._status_host[slot])
            self._status_free.append(slot)
            seen |= flags
            self._resolve(flags, what, shape, rerun=False)
        return seen

    def synchronize(self) -> None:
        """Wait for everything this engine enqueued and resolve a pending status check."""
        self._stream.synchronize()
        torch.cuda.current_stream(self.device).synchronize()
        self.check_status(block=True)
        if self._fallback is not None:
            self._fallback.synchronize()
        if self._sibling is not None:
            self._sibling.synchronize()

    def _after_call(self, a, launch, what: str, shape=None) -> bool:
        """Status handling of one f5_sample / f5_dit_forward call (class docstring).  Returns True when the call saturated fp16 and has
        to be repeated on the bf16x3 fall-back ("sync" only; the LN-fold overflow is repeated here, unfolded)."""
        if self._status_host is None or self.range_check == "off":
            return False
        mode = self.range_check
        if mode == "auto":
            mode = "async" if (shape == self._clean_shape and self._clean_calls >= self.range_probation) else "sync"
        if not self._status_free:                      # 16 calls ahead of the GPU: wait for the oldest check
            self._status_pending[0][0].synchronize()
            self.check_status(block=False)
        slot = self._status_free.pop(0)
        holder = {}

        def enqueue_read(st):
            check(self.lib.f5_sample_status_async(self._h, C.byref(a), C.c_void_p(self._status_host[slot:].data_ptr()), st), "f5_sample_status_async")
            holder["ev"] = torch.cuda.Event()
            holder["ev"].record(torch.cuda.current_stream(self.device))

        self._run_on_side_stream(enqueue_read)
        if mode != "sync":
            self._status_pending.append((holder["ev"], slot, what, shape))
            return False
        holder["ev"].synchronize()
        flags = int(self._status_host[slot])
        if (flags & STATUS_FOLD_OVERFLOW) and not (flags & STATUS_SATURATED):
            self._resolve(flags, what, shape, rerun=True)
            launch()                                   # ln_fold is 0 now: this run cannot carry bit 0; bit 2 is looked at below
            self._run_on_side_stream(enqueue_read)
            holder["ev"].synchronize()
            flags = int(self._status_host[slot]) & ~STATUS_FOLD_OVERFLOW
        self._status_free.append(slot)
        return self._resolve(flags, what, shape, rerun=True)

    def dit_forward(self, x: torch.Tensor, text: torch.Tensor, cond: torch.Tensor, lens, durations, t: float,
                    cfg_strength: float = 2.0, use_mask: Optional[bool] = None):
        """One DiT evaluation (cond branch and, when cfg_strength >= 1e-5, null branch)."""
        self._check_inputs(text, cond, lens, durations)
        self.check_status(block=False)
        if self._use_fallback:
            return self._fallback.dit_forward(x, text, cond, lens, durations, t, cfg_strength=cfg_strength, use_mask=use_mask)
        B, N, mel = cond.shape
        if use_mask is None:
            use_mask = B > 1
        ws = self.workspace(B, N, text.shape[1], 2, "euler")
        pred = torch.empty_like(cond)
        null = torch.empty_like(cond) if cfg_strength >= 1e-5 else None
        a, keep = self._args(text, cond, lens, durations, None, np.zeros(2, np.float32), 2, "euler", cfg_strength, use_mask,
                             False, None, None, ws)
        launch = lambda: self._run_on_side_stream(lambda st: check(   # noqa: E731
            self.lib.f5_dit_forward(self._h, C.byref(a), ptr(x), C.c_float(t), ptr(pred), ptr(null), st), "f5_dit_forward"))
        launch()
        saturated = self._after_call(a, launch, f"dit_forward(B={B}, N={N})", (B, N, 2, "forward"))
        del keep
        if saturated and self._use_fallback:
            return self._fallback.dit_forward(x, text, cond, lens, durations, t, cfg_strength=cfg_strength, use_mask=use_mask)
        return pred, null
