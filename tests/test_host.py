"""CPU tests: host-side logic of the drop-in API, C-ABI surface (no compute calls), weights inventory."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from f5test import O, ROOT, TINY, F5TTS_335M, synthetic_weights
from f5_tts_mlx_amd import engine as E
from f5_tts_mlx_amd import utils as U
from f5_tts_mlx_amd.cfm import (prepare_lengths, time_grid)
from f5_tts_mlx_amd.rng import mlx_like_normal, threefry2x32
from f5_tts_mlx_amd.weights import check_weights, convert_upstream_weights, num_params, param_specs

ASSETS = os.path.join(ROOT, "f5_tts_mlx_amd", "assets")


# ---- utils (bit-exact index paths) ---------------------------------------------------------------
def test_vocab_and_tokenizers_match_oracle():
    vocab = {v: i for i, v in enumerate(open(os.path.join(ASSETS, "vocab.txt")).read().split("\n"))}
    assert len(vocab) == 2546 and vocab[" "] == 0 and vocab["a"] == 62
    texts = [list("some call me nature"), list("héllo wörld!!"), list("x")]
    a = U.list_str_to_idx(texts, vocab)
    b = O.list_str_to_idx(texts, vocab)
    assert a.dtype == torch.int32 and torch.equal(a, b)
    assert (a[2, 1:] == -1).all() and a.shape == (3, 19)
    assert torch.equal(U.list_str_to_tensor(["ab", "héé"]), O.list_str_to_tensor(["ab", "héé"]))
    assert U.list_str_to_tensor(["ab", "héé"]).tolist() == [[97, 98, -1, -1, -1], [104, 195, 169, 195, 169]]


def test_masks_and_padding():
    t = torch.tensor([3, 0, 5])
    assert torch.equal(U.lens_to_mask(t), O.lens_to_mask(t))
    assert U.lens_to_mask(t).shape == (3, 5) and U.lens_to_mask(t, 7).shape == (3, 7)
    assert U.lens_to_mask(t).tolist()[0] == [True, True, True, False, False]
    x = torch.arange(4)
    assert U.pad_to_length(x, 6, value=-1).tolist() == [0, 1, 2, 3, -1, -1] and U.pad_to_length(x, 2).tolist() == [0, 1]
    with pytest.raises(ValueError):
        U.pad_to_length(torch.zeros(2, 2, 2), 5)
    assert U.pad_sequence([torch.ones(2), torch.ones(4)]).shape == (2, 4)


def test_convert_char_to_pinyin_ascii():
    out = U.convert_char_to_pinyin(["Some call me nature, others call me mother nature. Hello;world"])
    assert "".join(out[0]) == "Some call me nature, others call me mother nature. Hello, world"
    assert U.convert_char_to_pinyin(["“q” ‘x’"]) == [list("\"q\" 'x'")]


# ---- sampler host logic ----------------------------------------------------------------------------
def test_time_grid_matches_oracle():
    for steps, s in ((32, -1.0), (8, -1.0), (16, None), (2, 0.5)):
        assert np.allclose(time_grid(steps, s), O.time_grid(steps, s).numpy(), atol=1.2e-7)
    assert time_grid(32, -1.0).dtype == np.float32 and len(time_grid(32, -1.0)) == 32


def test_prepare_lengths_matches_oracle_and_raises():
    cfg = TINY
    from f5test import synth_inputs
    cond, text, durations, y0 = synth_inputs(cfg, 3, 60, nt=30, n_ref=12, seed=4, ragged=True)
    _, _, aux = O.sample(O.DiTOracle(cfg, synthetic_weights(cfg)), cond, text, torch.tensor(durations), y0=y0, steps=2,
                         method="euler", return_aux=True)
    t2, lens, dur, md = prepare_lengths(text, 12, 3, torch.tensor(durations), None, 4096, "euler")
    assert torch.equal(lens, aux["lens"]) and torch.equal(dur, aux["duration"]) and md == 60
    assert lens.tolist() == [30, 27, 24]                       # text longer than the reference audio extends lens
    _, _, d2, md2 = prepare_lengths(text, 12, 3, 5, None, 4096, "rk4")
    assert d2.tolist() == [31, 28, 25] and md2 == 31           # duration lifted to lens + 1 (cfm.py:317)
    _, _, d3, md3 = prepare_lengths(text, 12, 3, 9000, None, 4096, "midpoint")
    assert d3.tolist() == [4096] * 3 and md3 == 4096           # clipped (cfm.py:318)
    with pytest.raises(ValueError, match="Duration must be provided or a duration predictor must be set."):
        prepare_lengths(text, 12, 3, None, None, 4096, "euler")
    with pytest.raises(ValueError, match="Unknown method: heun"):
        prepare_lengths(text, 12, 3, 50, None, 4096, "heun")


def test_rng_emulation_is_deterministic():
    a, b, c = mlx_like_normal(7, (100, 50)), mlx_like_normal(7, (100, 50)), mlx_like_normal(8, (100, 50))
    assert np.array_equal(a, b) and not np.array_equal(a, c) and a.dtype == np.float32
    z = mlx_like_normal(0, (100, 937))
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02 and np.isfinite(z).all()
    # Threefry-2x32 known-answer test (Random123 kat_vectors, 20 rounds)
    x0, x1 = threefry2x32((np.uint32(0), np.uint32(0)), np.array([0], np.uint32), np.array([0], np.uint32))
    assert (int(x0[0]), int(x1[0])) == (0x6B200159, 0x99BA4EFE)
    x0, x1 = threefry2x32((np.uint32(0xFFFFFFFF), np.uint32(0xFFFFFFFF)), np.array([0xFFFFFFFF], np.uint32),
                          np.array([0xFFFFFFFF], np.uint32))
    assert (int(x0[0]), int(x1[0])) == (0x1CB996FC, 0xBB002BE7)


# ---- weights ----------------------------------------------------------------------------------------
def test_param_inventory():
    assert num_params(F5TTS_335M) == 337_096_804
    names = [n for n, _, _ in param_specs(F5TTS_335M)]
    assert len(names) == len(set(names))
    assert "transformer.transformer_blocks.21.ff.ff.layers.2.weight" in names
    w = synthetic_weights(TINY, seed=1)
    check_weights(TINY, w)
    assert np.array_equal(w["transformer.proj_out.weight"], synthetic_weights(TINY, seed=1)["transformer.proj_out.weight"])
    bad = dict(w)
    bad.pop("transformer.proj_out.bias")
    with pytest.raises(ValueError, match="missing parameter"):
        check_weights(TINY, bad)


def test_convert_upstream_weights_roundtrip():
    """cfm.py:477-508: upstream (PyTorch F5-TTS) names/layouts -> reference names/layouts."""
    w = synthetic_weights(TINY, seed=3)
    up = {}
    for k, v in w.items():
        k2 = (k.replace(".to_out.layers", ".to_out").replace(".text_blocks.layers", ".text_blocks")
              .replace(".ff.ff.layers.0.layers.0", ".ff.ff.0.0").replace(".ff.ff.layers.2", ".ff.ff.2")
              .replace(".time_mlp.layers", ".time_mlp").replace(".conv1d.layers", ".conv1d"))
        if ".dwconv.weight" in k or ".conv1d.layers.0.weight" in k or ".conv1d.layers.2.weight" in k:
            v = np.swapaxes(v, 1, 2)                              # torch conv layout (out, in/g, k)
        up["ema_model." + k2] = v
    up["ema_model.mel_spec.mel_stft.window"] = np.zeros(4, np.float32)
    up["initted"] = np.zeros(1, np.float32)
    up["step"] = np.zeros(1, np.float32)
    back = convert_upstream_weights(up)
    assert set(back) == set(w)
    for k in w:
        assert np.array_equal(back[k], w[k]), k


# ---- C ABI surface (no GPU needed) ---------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "f5tts_hip.h")).read()
    names = sorted(set(re.findall(r"\b(f5_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    lib = E.load_library()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.f5_version() >= 1
    # the lab header's hooks exist exactly when the library says it is the lab build (product: none of them leaks in)
    lab_hdr = open(os.path.join(ROOT, "include", "f5tts_hip_lab.h")).read()
    lab_names = sorted(set(re.findall(r"\b(f5_[a-z0-9_]+)\s*\(", lab_hdr)) - set(names))
    assert len(lab_names) >= 8
    have = [hasattr(lib, n) for n in lab_names]
    assert all(have) if lib.f5_lab_build() else not any(have), dict(zip(lab_names, have))


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "f5tts_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",'
                   'sizeof(f5_config), sizeof(f5_sample_args), offsetof(f5_sample_args, text), offsetof(f5_sample_args, steps),'
                   'offsetof(f5_sample_args, workspace_bytes));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    A = E.F5SampleArgs
    assert got == [C.sizeof(E.F5Config), C.sizeof(A), A.text.offset, A.steps.offset, A.workspace_bytes.offset]


def test_engine_handle_errors_and_sizes_without_gpu():
    lib = E.load_library()
    h = C.c_void_p()
    cfg = E.to_c_config(F5TTS_335M)
    assert lib.f5_engine_create(C.byref(cfg), 0, C.byref(h)) == 0
    n1, n2, ws1, ws32 = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    assert lib.f5_weights_bytes(h, C.byref(n1)) == 0 and n1.value > 337_096_804 * 2
    assert lib.f5_workspace_bytes(h, 1, 937, 160, 32, 0, C.byref(ws1)) == 0
    assert lib.f5_workspace_bytes(h, 32, 937, 160, 32, 0, C.byref(ws32)) == 0 and ws32.value > 8 * ws1.value
    assert lib.f5_workspace_bytes(h, 1, 937, 160, 32, 7, C.byref(ws1)) != 0
    assert lib.f5_last_error().decode().startswith("Unknown method")
    lib.f5_engine_destroy(h)
    h3 = C.c_void_p()
    assert lib.f5_engine_create(C.byref(cfg), 1, C.byref(h3)) == 0
    assert lib.f5_weights_bytes(h3, C.byref(n2)) == 0 and n2.value > n1.value          # bf16x3 carries hi + lo
    # loading before an arena is set must fail loudly, unknown names too
    buf = (C.c_float * 4)()
    shp = (C.c_int64 * 1)(4)
    assert lib.f5_load_tensor(h3, b"transformer.proj_out.bias", buf, 1, shp) != 0
    lib.f5_engine_destroy(h3)
    bad = E.to_c_config(type(F5TTS_335M)(dim=1000))
    hb = C.c_void_p()
    assert lib.f5_engine_create(C.byref(bad), 0, C.byref(hb)) != 0 and b"dim" in lib.f5_last_error()


def test_f16_precision_and_host_half_conversion_without_gpu():
    """precision "f16": an engine can be created / sized without a GPU, its arena equals the bf16 one (one 16-bit copy per
    matrix), and the host-side float -> fp16 conversion used by f5_load_tensor is IEEE round-to-nearest-even with
    saturation (checked exhaustively over all fp16 values and their midpoints, and on random floats against numpy)."""
    lib = E.load_library()
    cfg = E.to_c_config(F5TTS_335M)
    sizes = {}
    for prec in ("bf16", "f16"):
        h, n = C.c_void_p(), C.c_size_t()
        assert lib.f5_engine_create(C.byref(cfg), E.PRECISIONS[prec], C.byref(h)) == 0
        assert lib.f5_weights_bytes(h, C.byref(n)) == 0
        assert lib.f5_engine_graph_count(h) == 0 and lib.f5_engine_set_graph_cache(h, 4) == 0
        assert lib.f5_engine_set_graph_cache(h, 0) != 0
        sizes[prec] = n.value
        lib.f5_engine_destroy(h)
    assert sizes["bf16"] == sizes["f16"]
    assert lib.f5_engine_create(C.byref(cfg), 9, C.byref(C.c_void_p())) != 0 and b"precision" in lib.f5_last_error()
    assert lib.f5_op_set_operand_type(2) != 0 and lib.f5_op_set_operand_type(1) == 0 and lib.f5_op_set_operand_type(0) == 0
    f2h = lambda x: int(lib.f5_debug_f2h_bits(C.c_float(float(x))))
    # every finite fp16 value converts to itself and back
    allh = np.arange(65536, dtype=np.uint16)
    vals = allh.view(np.float16).astype(np.float32)
    fin = np.isfinite(vals)
    for hb, v in zip(allh[fin][::7], vals[fin][::7]):
        assert f2h(v) == int(hb), (hb, v)
        assert lib.f5_debug_h_bits2f(int(hb)) == float(v)
    # random floats (normals, subnormal range, near-overflow) against numpy's RNE conversion
    r = np.random.default_rng(0)
    xs = np.concatenate([r.standard_normal(3000) * 3, r.standard_normal(2000) * 1e-6, r.standard_normal(2000) * 3e4,
                         np.array([0.0, -0.0, 65504.0, 65519.9, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0001, 6.1e-5])]).astype(np.float32)
    xs = xs[np.abs(xs) < 65520]
    want = xs.astype(np.float16).view(np.uint16)
    got = np.array([f2h(x) for x in xs], dtype=np.uint16)
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:5]
    # beyond the fp16 range: saturate to +-65504 (numpy would give inf), like the device producers (csrc/op16.hpp f5_sat)
    assert f2h(1e6) == 0x7BFF and f2h(-7e4) == 0xFBFF and f2h(float("inf")) == 0x7BFF
    assert (f2h(float("nan")) & 0x7C00) == 0x7C00 and (f2h(float("nan")) & 0x3FF) != 0
    # bf16 helper agrees with torch
    xb = torch.from_numpy(xs[:2000])
    wantb = xb.to(torch.bfloat16).view(torch.int16).numpy().astype(np.uint16)
    gotb = np.array([int(lib.f5_debug_f2bf_bits(C.c_float(float(x)))) for x in xs[:2000]], dtype=np.uint16)
    assert np.array_equal(gotb, wantb)


def test_ascii_segmentation_follows_jieba_two_stage_split():
    """utils._ascii_segments restates jieba.cut for single-byte text: runs of [a-zA-Z0-9+#&._%-] are blocks whose alphanumeric
    parts stay whole and whose symbol stretches stay ONE piece (so the reference inserts a space token before them)."""
    seg = U._ascii_segments
    assert seg("wait... ok") == ["wait", "...", " ", "ok"]
    assert seg("a--b") == ["a", "--", "b"]
    assert seg("3.5% of C++") == ["3.5%", " ", "of", " ", "C", "++"]
    assert seg("x, y!  z") == ["x", ",", " ", "y", "!", " ", " ", "z"]
    assert U.convert_char_to_pinyin(["wait... ok"]) == [list("wait ... ok")]
    assert U.convert_char_to_pinyin(["no--way"]) == [list("no --way")[:2] + [" ", "-", "-"] + [" ", "w", "a", "y"]]


def test_engine_refuses_cpu_device():
    with pytest.raises(RuntimeError, match="GPU"):
        E.Engine(TINY, device="cpu")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(E, "_lib", None)
    monkeypatch.setattr(E, "_LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        E.load_library()


# ---- generate() host pieces (generate.py:30-36, 104-111, 154-163) -------------------------------------
def test_generate_helpers(tmp_path):
    from f5_tts_mlx_amd import generate as G
    assert G.split_sentences("Hello there. How are you? Fine") == ["Hello there.", "How are you?"]   # trailing fragment dropped
    assert G.split_sentences("no punctuation") == []
    assert G.FRAMES_PER_SEC == 93.75 and int(10.0 * G.FRAMES_PER_SEC) == 937
    audio = torch.zeros(256 * 40)
    d = G.estimated_duration(audio, "abcd", "abcdefgh", speed=1.0)
    assert abs(d - (40 + int(40 / 4 * 8)) / 93.75) < 1e-9
    a, sr = G.read_wav(os.path.join(ASSETS, "test_en_1_ref_short.wav"))
    assert sr == 24000 and a.dtype == np.float64 and a.shape == (127987,) and abs(np.sqrt(np.mean(a ** 2)) - 0.12888) < 1e-4
    p = tmp_path / "o.wav"
    G.write_wav(str(p), np.linspace(-0.5, 0.5, 1000, dtype=np.float32))
    b, sr2 = G.read_wav(str(p))
    assert sr2 == 24000 and np.allclose(b, np.linspace(-0.5, 0.5, 1000), atol=1e-6)
    assert G.DEFAULT_REF_TEXT == "Some call me nature, others call me mother nature."


def test_mlx_affine_quantisation_roundtrip():
    """weights.{quantize,dequantize}_mlx_affine / dequantize_mlx_checkpoint: packing order, group layout, error bound."""
    from f5_tts_mlx_amd.weights import dequantize_mlx_affine, dequantize_mlx_checkpoint, quantize_mlx_affine
    rng = np.random.default_rng(0)
    w = rng.standard_normal((24, 192)).astype(np.float32)
    for bits in (4, 8):
        pk, sc, bi = quantize_mlx_affine(w, bits)
        assert pk.dtype == np.uint32 and pk.shape == (24, 192 * bits // 32) and sc.shape == (24, 3) == bi.shape
        d = dequantize_mlx_affine(pk, sc, bi, bits)
        step = (w.reshape(24, 3, 64).max(-1) - w.reshape(24, 3, 64).min(-1)) / (2 ** bits - 1)
        assert np.all(np.abs(d - w).reshape(24, 3, 64).max(-1) <= 0.5 * step * (1 + 1e-5) + 1e-7)
        # first element of a word sits in the least significant bits
        q0 = np.rint((w[0, 0] - bi[0, 0]) / sc[0, 0])
        assert (int(pk[0, 0]) & ((1 << bits) - 1)) == int(q0)
    ck = {"a.weight": pk, "a.scales": sc, "a.biases": bi, "a.bias": np.ones(24, np.float32), "n.weight": np.ones(5, np.float32)}
    out = dequantize_mlx_checkpoint(ck, 8)
    assert set(out) == {"a.weight", "a.bias", "n.weight"} and out["a.weight"].shape == (24, 192)
    with pytest.raises(ValueError):
        dequantize_mlx_affine(pk, sc[:, :2], bi[:, :2], 8)


def test_bench_flop_model_matches_survey():
    """bench.py's algorithmic flop count per utterance-forward is SURVEY.md §8(d)'s F(937) = 442 304 409 600; the hoisted
    (executed) figure drops the per-sample work and counts the x-part of the input projection at its padded K = 128"""
    import importlib
    import sys
    from f5test import ROOT
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    assert bench.flops_forward(937, hoisted=False) == 442_304_409_600
    executed = bench.flops_forward(937, hoisted=True)
    assert executed == 432_959_365_120 + 937 * 2 * (128 - 100) * 1024      # SURVEY's F_h + the K padding 100 -> 128
    assert executed < bench.flops_forward(937, hoisted=False)
    assert abs(62 * 442_304_409_600 / 1e12 - 27.423) < 1e-3                # reference-equivalent TFLOP per 32-point Euler utterance
