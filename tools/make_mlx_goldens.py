#!/usr/bin/env python
"""Produce the two known-answer files that would pin the last unpinned rows of the hot path -- to be run
by anyone with the reference's own runtime (Apple silicon or any machine where `pip install mlx vocos-mlx` works); nothing in
this repository can run it (no mlx here, no network).

    python tools/make_mlx_goldens.py [--out tests/golden] [--vocos lucasnewman/vocos-mel-24khz]

Writes
  tests/golden/ref_text_tokens.json   `convert_char_to_pinyin(TEXT_CASES)` (f5_tts_mlx/utils.py:139-173) computed with the REAL jieba and
                               pypinyin (the text -> token path is an index path: bit-exact).  Needs only jieba + pypinyin, no mlx:
                               `python tools/make_mlx_goldens.py --only-text`.  Until the file exists the package's ASCII
                               segmentation (`f5_tts_mlx_amd.utils._ascii_segments`, a restatement of jieba's published algorithm)
                               is checked only against itself and the CJK branch is dormant (INTEGRATION.md says so)
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

# ASCII, punctuation-heavy, mixed and CJK strings: what generate() may be handed (f5_tts_mlx/generate.py:140-176)
TEXT_CASES = [
    "Some call me nature, others call me mother nature.",
    "Hello;world",
    "wait... ok",
    "no--way",
    "C++ & C# at 99.5% -- e.g. v2.0_beta+3",
    "a.b.c d_e-f  g",
    "It's 'quoted' \"twice\": done",
    "\u201cq\u201d \u2018x\u2019",
    "tabs\tand\nnewlines\r\nhere",
    "3.14 is pi; 2.718 is e",
    "email me@example.com, or call +1-555-0100!",
    "x",
    "",
    "   leading and trailing   ",
    "ALLCAPS and MiXeD 123abc",
    "semi;colons;everywhere;",
    "(parentheses) [brackets] {braces} <angles>",
    "dollars $5.00 & cents 50c #tag @user",
    "end.",
    "e.g., i.e., etc.",
    "\u4f60\u597d\uff0c\u4e16\u754c\u3002",
    "\u4eca\u5929\u5929\u6c14\u5f88\u597d\uff0c\u6211\u4eec\u53bb\u516c\u56ed\u6563\u6b65\u5427\uff01",
    "\u6211\u7231 Python \u548c C++\u3002",
    "Hello \u4e16\u754c, this is \u6df7\u5408 text.",
    "\u884c\u957f\u5728\u94f6\u884c\u91cc\u884c\u8d70",
    "\u300a\u7ea2\u697c\u68a6\u300b\u662f\u4e00\u90e8\u5c0f\u8bf4\u2014\u2014\u5f88\u6709\u540d\u2026",
    "\u4e00\u4e2a\u4e0d\u884c\uff1f\u4e0d\u8981\uff01",
    "2024\u5e749\u670825\u65e5 10:30",
    "\u30c6\u30b9\u30c8 test \ud55c\uad6d\uc5b4",
    "caf\u00e9 na\u00efve r\u00e9sum\u00e9",
    "\u4e2d\u6587mixed\u4e2d\u6587 with\u82f1\u6587inside",
    "\u3002\uff0c\u3001\uff1b\uff1a\uff1f\uff01",
    "A\u3002B\uff0cC",
    "\u8c22\u8c22 thanks \u8c22\u8c22\uff01OK?",
    "100% \u540c\u610f",
    "\u4e0d",
    "\u957f\u5927 \u957f\u5ea6 \u91cd\u8981 \u91cd\u590d",
    "\u6211\u4eec;\u4f60\u4eec",
    "The quick brown fox jumps over the lazy dog. \u654f\u6377\u7684\u68d5\u8272\u72d0\u72f8\u8df3\u8fc7\u4e86\u61d2\u72d7\u3002",
    "\U0001f600 emoji \u548c text",
]

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


def text_goldens(reference_root):
    """the reference's OWN convert_char_to_pinyin (utils.py:139-173) with the real jieba / pypinyin.  Its module imports mlx at the top:
    where mlx is absent only the one function is compiled out of the file (no stand-ins for jieba / pypinyin -- they are the point)."""
    import ast
    import jieba
    import pypinyin
    from pypinyin import Style, lazy_pinyin
    path = os.path.join(reference_root, "f5_tts_mlx", "utils.py")
    tree = ast.parse(open(path, encoding="utf-8").read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "convert_char_to_pinyin"]
    assert fn, f"convert_char_to_pinyin not found in {path}"
    ns = {"jieba": jieba, "lazy_pinyin": lazy_pinyin, "Style": Style}
    exec(compile(ast.Module(fn, []), path, "exec"), ns)
    tokens = ns["convert_char_to_pinyin"](TEXT_CASES)
    return {"jieba_version": getattr(jieba, "__version__", "unknown"), "pypinyin_version": getattr(pypinyin, "__version__", "unknown"),
            "source": "f5_tts_mlx/utils.py:139-173 convert_char_to_pinyin, real jieba + pypinyin", "texts": TEXT_CASES, "tokens": tokens,
            # what jieba.cut sees inside the function (after its two translation tables, utils.py:141-148): pins the ASCII segmentation
            "segments": [list(jieba.cut(t.translate(str.maketrans({"\u201c": '"', "\u201d": '"', "\u2018": "'", "\u2019": "'"})).translate(str.maketrans({";": ","}))))
                         for t in TEXT_CASES]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("F5_REFERENCE", "/root/reference"), help="checkout of lucasnewman/f5-tts-mlx (for --only-text / the text section)")
    ap.add_argument("--only-text", action="store_true", help="write ref_text_tokens.json only (needs jieba + pypinyin, not mlx)")
    ap.add_argument("--skip-text", action="store_true")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    ap.add_argument("--vocos", default="lucasnewman/vocos-mel-24khz", help="checkpoint name or local path for vocos_mlx.Vocos.from_pretrained")
    ap.add_argument("--skip-vocos", action="store_true")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    if not args.skip_text:
        with open(os.path.join(args.out, "ref_text_tokens.json"), "w", encoding="utf-8") as f:
            json.dump(text_goldens(args.reference), f, ensure_ascii=True, indent=0)
        print("wrote", os.path.join(args.out, "ref_text_tokens.json"))
    if args.only_text:
        return
    import mlx
    import mlx.core as mx
    meta = np.frombuffer(json.dumps({"mlx_version": getattr(mlx, "__version__", "unknown"), "default_device": str(mx.default_device())}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(args.out, "mlx_rng.npz"), meta_json=meta, **rng_goldens(mx))
    print("wrote", os.path.join(args.out, "mlx_rng.npz"))
    if not args.skip_vocos:
        np.savez_compressed(os.path.join(args.out, "mlx_vocos.npz"), meta_json=meta, **vocos_goldens(mx, args.vocos))
        print("wrote", os.path.join(args.out, "mlx_vocos.npz"))


if __name__ == "__main__":
    main()
