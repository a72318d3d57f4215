"""Known-answer tests against the REAL reference runtime (MLX, vocos_mlx) -- the two rows of SURVEY.md §8 that no code in this
repository can pin (a18: `mx.random.seed(s); mx.random.normal((100, d))`, cfm.py:369-375; a22: `vocos_mlx.Vocos.decode`,
cfm.py:399-400,446).  They activate by themselves once tests/golden/mlx_rng.npz / mlx_vocos.npz exist; the files are produced by
`tools/make_mlx_goldens.py` on a machine that has mlx.  Until then they skip and both rows stay "parity unpinned"."""
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TXT = os.path.join(ROOT, "tests", "golden", "ref_text_tokens.json")
RNG = os.path.join(ROOT, "tests", "golden", "mlx_rng.npz")
VOC = os.path.join(ROOT, "tests", "golden", "mlx_vocos.npz")

need_rng = pytest.mark.skipif(not os.path.exists(RNG), reason="tests/golden/mlx_rng.npz absent: run tools/make_mlx_goldens.py where mlx is installed")
need_voc = pytest.mark.skipif(not os.path.exists(VOC), reason="tests/golden/mlx_vocos.npz absent: run tools/make_mlx_goldens.py where mlx / vocos_mlx are installed")


need_txt = pytest.mark.skipif(not os.path.exists(TXT), reason="tests/golden/ref_text_tokens.json absent: run tools/make_mlx_goldens.py --only-text where jieba + pypinyin are installed")


@need_txt
def test_text_to_tokens_matches_the_reference_with_real_jieba():
    """the text -> token path (utils.py:139-173) is an index path: bit-exact against the reference's own function run
    with the REAL jieba / pypinyin.  Single-byte strings go through the package's `_ascii_segments` here (no jieba in this image) and
    must give the recorded tokens AND the recorded jieba segmentation; strings that need jieba / pypinyin are compared where those are
    importable and must raise (not guess) where they are not."""
    from f5_tts_mlx_amd import utils as U
    g = json.load(open(TXT, encoding="utf-8"))
    have_backends = U._text_backends()[0] is not None
    n_ascii = n_other = 0
    for text, toks, segs in zip(g["texts"], g["tokens"], g["segments"]):
        t2 = text.translate(U._QUOTES).translate(U._OOV)
        if len(t2.encode("utf-8")) == len(t2):
            assert U._ascii_segments(t2) == list(segs), text          # the emulation against jieba's own segmentation
            assert U.convert_char_to_pinyin([text]) == [toks], text
            n_ascii += 1
        elif have_backends:
            assert U.convert_char_to_pinyin([text]) == [toks], text
            n_other += 1
        else:
            with pytest.raises(RuntimeError, match="jieba"):
                U.convert_char_to_pinyin([text])
            n_other += 1
    assert n_ascii >= 15 and n_other >= 10


def _cases(g):
    for k in g.files:
        m = re.fullmatch(r"normal_s(\d+)_d(\d+)", k)
        if m:
            yield int(m.group(1)), int(m.group(2)), g[k]


def test_the_generator_script_is_self_consistent():
    """no mlx here: at least the script must parse, and its case list must cover odd element counts and a 64-bit seed"""
    import ast
    src = open(os.path.join(ROOT, "tools", "make_mlx_goldens.py")).read()
    tree = ast.parse(src)
    ns = {}
    exec(compile(ast.Module([n for n in tree.body if isinstance(n, ast.Assign)], []), "cases", "exec"), ns)
    assert any(s >= 2 ** 32 for s, _ in ns["RNG_CASES"]) and any(d % 2 for _, d in ns["RNG_CASES"])
    # the text section: ASCII / punctuation-heavy, CJK and mixed strings, and the package handles every single-byte one
    from f5_tts_mlx_amd import utils as U
    texts = ns["TEXT_CASES"]
    single = [t for t in texts if len(t.translate(U._QUOTES).translate(U._OOV).encode("utf-8")) == len(t.translate(U._QUOTES).translate(U._OOV))]
    assert len(texts) >= 40 and len(single) >= 15 and len(texts) - len(single) >= 15
    for t in single:
        assert "".join(U._ascii_segments(t.translate(U._QUOTES).translate(U._OOV))) == t.translate(U._QUOTES).translate(U._OOV)
    assert all(b * n * 100 > 0 for b, n, _ in ns["MEL_CASES"])


@need_rng
def test_host_generator_matches_mlx_bit_for_bit():
    from f5_tts_mlx_amd.rng import _bits, _split, mlx_like_normal
    g = np.load(RNG)
    n = 0
    for seed, dur, ref in _cases(g):
        got = mlx_like_normal(seed, (100, dur))
        assert got.dtype == np.float32 and got.shape == ref.shape
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (seed, dur, float(np.abs(got - ref).max()))
        key = (np.uint32(seed >> 32), np.uint32(seed & 0xFFFFFFFF))
        assert np.array_equal(np.asarray(key, np.uint32), g[f"key_s{seed}"].reshape(-1))
        (a, b) = _split(key)
        assert np.array_equal(np.asarray([a, b], np.uint32).reshape(-1), g[f"split_s{seed}"].reshape(-1))
        assert np.array_equal(_bits(key, 7), g[f"bits_s{seed}"])
        n += 1
    assert n >= 5


@need_rng
@pytest.mark.gpu
def test_device_generator_matches_mlx_bit_for_bit():
    import torch
    from f5_tts_mlx_amd.engine import noise_normal
    g = np.load(RNG)
    for seed, dur, ref in _cases(g):
        got = noise_normal([seed], [dur], dur, 100, "cuda:0")
        torch.cuda.synchronize()
        got = got[0].cpu().numpy().T                                   # (100, dur): the reference's channel-major draw
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (seed, dur, float(np.abs(got - ref).max()))


def _mlx_to_upstream(name: str, shape):
    """vocos_mlx parameter path -> upstream Vocos name (vocos.vocos_param_specs); conv weights are (out, k, in) in MLX"""
    n = name
    n = n.replace("backbone.convnext.layers.", "backbone.convnext.")
    return n


@need_voc
def test_vocos_checkpoint_names_and_shapes_map_onto_the_engine():
    from f5_tts_mlx_amd.vocos import vocos_param_specs
    g = np.load(VOC)
    params = json.loads(bytes(g["param_names_json"]).decode())
    want = dict(vocos_param_specs())
    got = {}
    for name, shape, _dtype in params:
        if name.startswith(("feature_extractor.", "head.istft.")):
            continue
        got[_mlx_to_upstream(name, shape)] = tuple(shape)
    missing = sorted(set(want) - set(got))
    extra = sorted(set(got) - set(want))
    assert not missing and not extra, (missing[:5], extra[:5])
    for k, shp in want.items():
        g_shape = got[k]
        ok = tuple(shp) == g_shape or (len(shp) == 3 and (shp[0], shp[2], shp[1]) == g_shape)      # MLX conv layout (out, k, in)
        assert ok, (k, shp, g_shape)


@need_voc
@pytest.mark.gpu
def test_hip_vocoder_reproduces_vocos_mlx_waves():
    path = os.environ.get("F5_VOCOS_PATH")
    if not path or not os.path.exists(path):
        pytest.skip("the vocos-mel-24khz checkpoint itself is needed next to its golden: set F5_VOCOS_PATH")
    import torch
    from f5_tts_mlx_amd.vocos import Vocos
    g = np.load(VOC)
    voc = Vocos.from_pretrained(path, precision="bf16x3", device="cuda:0")
    for k in g.files:
        m = re.fullmatch(r"mel_b(\d+)_n(\d+)", k)
        if not m:
            continue
        ref = g[f"wave_b{m.group(1)}_n{m.group(2)}"]
        got = voc.decode(torch.from_numpy(g[k])).cpu().numpy().reshape(ref.shape)
        err = float(np.abs(got - ref).max())
        assert err <= 1e-4, (k, err)
