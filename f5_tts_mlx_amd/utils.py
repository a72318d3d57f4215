"""Host utilities on the sampling path (reference: f5_tts_mlx/utils.py).

ein notation: b - batch, n - sequence, nt - text sequence, nw - raw wave length, d - dimension.
Tensors are torch (CPU for these index paths); semantics follow the reference line by line.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import torch
import torch.nn.functional as F


def exists(v):
    return v is not None


def default(v, d):
    return v if exists(v) else d


def divisible_by(num, den):
    return (num % den) == 0


def lens_to_mask(t: torch.Tensor, length: Optional[int] = None) -> torch.Tensor:  # Bool['b n']
    """utils.py:39-47."""
    if not exists(length):
        length = int(t.max().item())
    seq = torch.arange(length, device=t.device)
    return seq[None, :] < t[:, None]


def pad_to_length(t: torch.Tensor, length: int, value=0) -> torch.Tensor:
    """utils.py:93-103."""
    ndim = t.ndim
    seq_len = t.shape[-1]
    if length > seq_len:
        if ndim in (1, 2):
            t = F.pad(t, (0, length - seq_len), value=value)
        else:
            raise ValueError(f"Unsupported padding dims: {ndim}")
    return t[..., :length]


def mask_from_start_end_indices(seq_len: torch.Tensor, start: torch.Tensor, end: torch.Tensor,
                                max_length: Optional[int] = None) -> torch.Tensor:
    """utils.py:50-58 (`max_length` is used as given: the reference's default to seq_len.max() is commented out)."""
    seq = torch.arange(max_length, dtype=torch.int32, device=start.device)
    return (seq[None, :] >= start[:, None]) & (seq[None, :] < end[:, None])


def mask_from_frac_lengths(seq_len: torch.Tensor, frac_lengths: torch.Tensor, max_length: Optional[int] = None,
                           rand: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """utils.py:61-79.  Extension: `rand` injects the uniform draw of :68 (parity tests); index math in float32 -> int32
    truncation exactly as `astype(mx.int32)`."""
    seq_len = torch.as_tensor(seq_len)
    lengths = (frac_lengths.to(torch.float32) * seq_len.to(torch.float32)).to(torch.int32)
    max_start = seq_len.to(torch.int32) - lengths
    if rand is None:
        rand = torch.rand(frac_lengths.shape, generator=generator)
    start = torch.clamp((max_start.to(torch.float32) * rand.to(torch.float32)).to(torch.int32), min=0)
    end = start + lengths
    out = mask_from_start_end_indices(seq_len, start, end, max_length)
    if exists(max_length):
        out = pad_to_length(out, max_length)
    return out


def pad_sequence(t: Sequence[torch.Tensor], padding_value=0) -> torch.Tensor:
    """utils.py:106-109."""
    max_len = max([i.shape[-1] for i in t])
    return torch.stack([pad_to_length(i, max_len, padding_value) for i in t])


# simple utf-8 tokenizer, since paper went character based

def list_str_to_tensor(text: List[str], padding_value=-1) -> torch.Tensor:  # Int['b nt']
    """utils.py:115-118."""
    list_tensors = [torch.tensor([*bytes(t, "UTF-8")], dtype=torch.int32) for t in text]
    return pad_sequence(list_tensors, padding_value=-1)


# char tokenizer, based on custom dataset's extracted .txt file

def list_str_to_idx(text: List[Union[str, List[str]]], vocab_char_map: Dict[str, int], padding_value=-1) -> torch.Tensor:
    """utils.py:124-133 — unknown characters map to 0, padding is -1."""
    list_idx_tensors = [torch.tensor([vocab_char_map.get(c, 0) for c in t], dtype=torch.int32) for t in text]
    return pad_sequence(list_idx_tensors, padding_value=padding_value)


# convert char to pinyin

_ZH_PUNCT = "。，、；：？！《》【】—…"
_QUOTES = str.maketrans({"“": '"', "”": '"', "‘": "'", "’": "'"})     # librispeech test-clean carries zh quotes
_OOV = str.maketrans({";": ","})


def _text_backends():
    """(segmenter, romaniser): jieba + pypinyin when importable (what the reference uses), else the single-byte emulation."""
    try:
        import jieba  # type: ignore
        from pypinyin import Style, lazy_pinyin  # type: ignore
        jieba.setLogLevel(20)
        return (lambda t: list(jieba.cut(t))), (lambda t: lazy_pinyin(t, style=Style.TONE3, tone_sandhi=True))
    except Exception:  # pragma: no cover - environment dependent
        return None, None


def convert_char_to_pinyin(text_list: List[str], polyphone: bool = True) -> List[List[str]]:
    """utils.py:139-173: token list per text — single-byte runs character by character (a space is inserted before a
    multi-character run unless the previous token is one of ` :'"`), pure-CJK runs as tone-numbered pinyin syllables each preceded
    by a space, mixed runs character by character.  jieba / pypinyin are not installable here: without them single-byte text
    goes through `_ascii_segments` (jieba's behaviour on such text) and anything else raises."""
    segment, romanise = _text_backends()
    result = []
    for text in text_list:
        text = text.translate(_QUOTES).translate(_OOV)
        if segment is None:
            if len(text.encode("utf-8")) != len(text):
                raise RuntimeError("convert_char_to_pinyin: non single-byte text needs jieba + pypinyin (not installed)")
            pieces = _ascii_segments(text)
        else:
            pieces = segment(text)
        toks: List[str] = []
        for piece in pieces:
            width = len(piece.encode("utf-8"))
            if width == len(piece):                                   # letters, digits, ascii symbols
                if toks and width > 1 and toks[-1] not in " :'\"":
                    toks.append(" ")
                toks += list(piece)
            elif polyphone and width == 3 * len(piece):               # CJK only
                for syllable in romanise(piece):
                    if syllable not in _ZH_PUNCT:
                        toks.append(" ")
                    toks.append(syllable)
            else:                                                     # mixed
                for ch in piece:
                    if ord(ch) < 256 or ch in _ZH_PUNCT:
                        toks.append(ch)
                    else:
                        toks.append(" ")
                        toks += romanise(ch)
        result.append(toks)
    return result


def _ascii_segments(text: str) -> List[str]:
    """Segmentation jieba.cut (default mode, HMM on) produces for single-byte text, restated from jieba's published algorithm
    (jieba/__init__.py `Tokenizer.cut`, jieba/finalseg `cut`; jieba itself is not installable here):
    1. the sentence is split into blocks by `re_han_default = [\u4E00-\u9FD5a-zA-Z0-9+#&._%-]+`;
    2. a matching block holds no dictionary word when it is single-byte, so it reaches the HMM fallback `finalseg.cut`, whose
       `re_skip = [a-zA-Z0-9]+(?:\.\d+)?%?` keeps alphanumeric runs whole and yields the stretches BETWEEN them ("...", "--",
       "+#") as ONE multi-character piece each (the reference then puts a space token in front of such a piece);
    3. a non-matching block (whitespace, other punctuation) is split on `(\r\n|\s)`: whitespace items as they are, everything
       else character by character."""
    import re
    out: List[str] = []
    for blk in re.split(r"([a-zA-Z0-9+#&._%\-]+)", text):
        if not blk:
            continue
        if re.fullmatch(r"[a-zA-Z0-9+#&._%\-]+", blk):
            if len(blk) == 1:
                out.append(blk)
            else:
                out += [x for x in re.split(r"([a-zA-Z0-9]+(?:\.\d+)?%?)", blk) if x]
        else:
            for x in re.split(r"(\r\n|\s)", blk):
                if not x:
                    continue
                if re.fullmatch(r"\r\n|\s", x):
                    out.append(x)
                else:
                    out += list(x)
    return out


def fetch_from_hub(hf_repo: str, quantization_bits: Optional[int] = None):
    """utils.py:179-192 — local directory or HF snapshot (needs network)."""
    from pathlib import Path
    p = Path(hf_repo)
    if p.exists():
        return p
    from huggingface_hub import snapshot_download
    model_filename = "model_v1.safetensors"
    if exists(quantization_bits):
        model_filename = f"model_v1_{quantization_bits}b.safetensors"
    return Path(snapshot_download(repo_id=hf_repo, allow_patterns=[model_filename, "duration_v2.safetensors", "*.txt"]))
