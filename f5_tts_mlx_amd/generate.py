"""Text-to-speech entry point (reference: f5_tts_mlx/generate.py).

`generate(...)` keeps the reference's keyword arguments, defaults and control flow (single vs per-sentence generation,
seconds -> frames with 93.75 frames/s, RMS normalisation that only scales UP, reference trimmed by SAMPLES).  Differences:
WAV I/O uses scipy (soundfile is not installed), playback (`AudioPlayer`, PortAudio) is out of scope, and the function
returns the waveform (the reference returns None).  Without `output_path` nothing is played; the wave is just returned.
"""
from __future__ import annotations

import argparse
import datetime
import pkgutil
import re
import sys
from typing import Literal, Optional

import numpy as np
import torch

from .cfm import F5TTS
from .utils import convert_char_to_pinyin

SAMPLE_RATE = 24_000
HOP_LENGTH = 256
FRAMES_PER_SEC = SAMPLE_RATE / HOP_LENGTH
TARGET_RMS = 0.1
DEFAULT_REF_TEXT = "Some call me nature, others call me mother nature."   # generate.py:143


def split_sentences(text):
    """generate.py:30-36."""
    sentence_endings = re.compile(r"([.!?;:])")
    sentences = sentence_endings.split(text)
    sentences = [sentences[i] + sentences[i + 1] for i in range(0, len(sentences) - 1, 2)]
    return [sentence.strip() for sentence in sentences if sentence.strip()]


def estimated_duration(ref_audio, ref_text: str, gen_text: str, speed: float = 1.0):
    """generate.py:104-111 (quirk kept: the 'pause punctuation' pattern is a literal sequence, not a class)."""
    ref_audio_len = ref_audio.shape[0] // HOP_LENGTH
    zh_pause_punc = r"。，、；：？！"
    ref_text_len = len(ref_text.encode("utf-8")) + 3 * len(re.findall(zh_pause_punc, ref_text))
    gen_text_len = len(gen_text.encode("utf-8")) + 3 * len(re.findall(zh_pause_punc, gen_text))
    duration_in_frames = ref_audio_len + int(ref_audio_len / ref_text_len * gen_text_len / speed)
    print(f"Got estimated duration: {duration_in_frames / FRAMES_PER_SEC}")
    return duration_in_frames / FRAMES_PER_SEC


def read_wav(path_or_bytes):
    """float64 samples in [-1, 1) + sample rate (what `soundfile.read` returns for PCM16/float WAVs)."""
    import io
    import scipy.io.wavfile as wf
    src = io.BytesIO(path_or_bytes) if isinstance(path_or_bytes, (bytes, bytearray)) else path_or_bytes
    sr, a = wf.read(src)
    if a.dtype == np.int16:
        a = a.astype(np.float64) / 32768.0
    elif a.dtype == np.int32:
        a = a.astype(np.float64) / 2147483648.0
    elif a.dtype == np.uint8:
        a = (a.astype(np.float64) - 128.0) / 128.0
    else:
        a = a.astype(np.float64)
    if a.ndim > 1:
        a = a[:, 0]
    return a, sr


def write_wav(path, wave: np.ndarray, sample_rate: int = SAMPLE_RATE) -> None:
    import scipy.io.wavfile as wf
    wf.write(path, sample_rate, np.asarray(wave, dtype=np.float32))


def generate(
    generation_text: str,
    duration: Optional[float] = None,
    estimate_duration: bool = False,
    model_name: str = "lucasnewman/f5-tts-mlx",
    ref_audio_path: Optional[str] = None,
    ref_audio_text: Optional[str] = None,
    steps: int = 8,
    method: Literal["euler", "midpoint"] = "rk4",
    cfg_strength: float = 2.0,
    sway_sampling_coef: float = -1.0,
    speed: float = 1.0,  # used when duration is None as part of the duration heuristic
    seed: Optional[int] = None,
    quantization_bits: Optional[int] = None,
    output_path: Optional[str] = None,
    f5tts: Optional[F5TTS] = None,          # extension: reuse an already loaded model
    batch_sentences: bool = False,          # extension: all sentences as ONE ragged sample() batch instead of the reference's loop
):
    """generate.py:113-245.  `batch_sentences=True` (opt-in, no reference counterpart; SURVEY.md section 8(f)2 names it as the point of
    owning the app loop): the sentences of a multi-sentence text are sampled as one ragged batch -- one `sample()` call, one set of
    launches at M = 2 x (sum of frames) instead of one set per sentence.  It is NOT bit-compatible with the loop: in a batch the
    key-padding mask exists (cfm.py:333-336), and GRN (convnext_v2.py:16) and the conv position embedding (dit.py:251) see the padding
    up to the longest sentence, exactly as the reference's own `sample()` behaves for a batch; what it equals is `sample()` of that
    batch (tests/test_model_gpu.py::test_generate_batch_sentences), each element vocoded on its own frames (the padded tail of a
    shorter sentence never reaches the vocoder)."""
    if f5tts is None:
        f5tts = F5TTS.from_pretrained(model_name, quantization_bits=quantization_bits)
    if getattr(f5tts, "_vocoder", None) is None:
        # sample() would hand back mel frames; trimming them by audio samples and writing a WAV would produce garbage
        raise RuntimeError("generate() needs a model with a vocoder (F5TTS(..., vocoder=...) / from_pretrained with Vocos)")

    if ref_audio_path is None:
        data = pkgutil.get_data("f5_tts_mlx_amd", "assets/test_en_1_ref_short.wav")   # generate.py:133-143
        audio, sr = read_wav(data)
        ref_audio_text = DEFAULT_REF_TEXT
    else:
        audio, sr = read_wav(ref_audio_path)
        if sr != SAMPLE_RATE:
            raise ValueError("Reference audio must have a sample rate of 24kHz")

    audio = torch.from_numpy(np.asarray(audio)).to(torch.float32)
    ref_audio_duration = audio.shape[0] / SAMPLE_RATE
    print(f"Got reference audio with duration: {ref_audio_duration:.2f} seconds")

    rms = torch.sqrt(torch.mean(torch.square(audio)))
    if rms < TARGET_RMS:
        audio = audio * TARGET_RMS / rms

    sentences = split_sentences(generation_text)
    is_single_generation = len(sentences) <= 1 or duration is not None

    def run(text_for_model, dur):
        wave, _ = f5tts.sample(audio[None], text=text_for_model, duration=dur, steps=steps, method=method, speed=speed,
                               cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, seed=seed)
        wave = wave.reshape(-1) if wave.ndim > 1 and wave.shape[0] == 1 else wave
        return wave[audio.shape[0]:]                                  # trim the reference by SAMPLES (generate.py:183)

    start_date = datetime.datetime.now()
    if is_single_generation:
        if duration is not None:
            duration = int(duration * FRAMES_PER_SEC)
        elif estimate_duration:
            duration = int(estimated_duration(audio, ref_audio_text, generation_text, speed) * FRAMES_PER_SEC)
        text = convert_char_to_pinyin([ref_audio_text + " " + generation_text])
        wave = run(text, duration)
    elif batch_sentences:
        # every sentence gets the reference mel as conditioning and its own duration; sample() pads to the longest (cfm.py:319-321)
        texts = convert_char_to_pinyin([ref_audio_text + " " + t for t in sentences])
        cond = f5tts._mel_spec(audio[None].to(f5tts.transformer.device)).repeat(len(sentences), 1, 1)
        # (the whole text for every sentence, like the loop's first sentence; the loop's carried-over `duration` -- frames x 93.75
        # again from the second sentence on, generate.py:203-208 -- is a quirk of the loop and is not reproduced here)
        if estimate_duration:
            durs = torch.full((len(sentences),), int(estimated_duration(audio, ref_audio_text, generation_text, speed) * FRAMES_PER_SEC))
        else:
            durs = None                                               # duration predictor, per element (cfm.py:307-308)
        # mel frames out of sample() (vocoder detached for the call), then every sentence is vocoded ON ITS OWN FRAMES: frames past a
        # shorter sentence's duration hold unconstrained ODE state, and a vocoder with a receptive field (Vocos: a k = 7 ConvNeXt stack)
        # would leak them into the last ~0.25 s of that sentence before the trim
        voc = f5tts._vocoder
        try:
            f5tts._vocoder = None
            mel, _ = f5tts.sample(cond, text=texts, duration=durs, steps=steps, method=method, speed=speed, cfg_strength=cfg_strength,
                                  sway_sampling_coef=sway_sampling_coef, seed=seed)
        finally:
            f5tts._vocoder = voc
        frames = f5tts.last_durations                                 # per-element frame counts after the clamps of cfm.py:317-318
        wave = torch.cat([voc(mel[i:i + 1, :int(frames[i])]).reshape(-1)[audio.shape[0]:] for i in range(len(sentences))], dim=0)
    else:
        output = []
        for sentence_text in sentences:
            if duration is not None:
                duration = int(duration * FRAMES_PER_SEC)
            elif estimate_duration:                                   # quirk kept: uses the WHOLE text (generate.py:208)
                duration = int(estimated_duration(audio, ref_audio_text, generation_text, speed) * FRAMES_PER_SEC)
            text = convert_char_to_pinyin([ref_audio_text + " " + sentence_text])
            output.append(run(text, duration))
        wave = torch.cat(output, dim=0)

    if wave.is_cuda:
        torch.cuda.synchronize()
    generated_duration = wave.shape[0] / SAMPLE_RATE
    print(f"Generated {generated_duration:.2f}s of audio in {datetime.datetime.now() - start_date}.")

    if output_path is not None:
        write_wav(output_path, wave.detach().cpu().numpy(), SAMPLE_RATE)
    return wave


# command line: the reference's flags and defaults (generate.py:248-362), table driven
_CLI = (
    ("--model", str, "lucasnewman/f5-tts-mlx", "checkpoint: hub name or local directory"),
    ("--text", str, None, "what to say; read from stdin when omitted"),
    ("--duration", float, None, "total length (reference + generated) in seconds"),
    ("--estimate-duration", bool, False, "derive the length from the reference audio and the text sizes"),
    ("--ref-audio", str, None, "24 kHz mono WAV to clone (default: the bundled sample)"),
    ("--ref-text", str, None, "transcript of --ref-audio"),
    ("--output", str, None, "WAV file to write"),
    ("--steps", int, 8, "ODE grid points (steps - 1 solver updates)"),
    ("--cfg", float, 2.0, "classifier-free guidance strength"),
    ("--sway-coef", float, -1.0, "sway-sampling coefficient of the time grid"),
    ("--speed", float, 1.0, "speaking-rate factor used by the duration heuristics"),
    ("--seed", int, None, "noise seed"),
)


def main(argv=None):
    ap = argparse.ArgumentParser(
        description="F5-TTS text to speech on MI355X (same flags as f5_tts_mlx.generate)",
        epilog="Text front-end: single-byte (ASCII / Latin) text is tokenised exactly like the reference; Chinese text needs "
               "jieba and pypinyin (utils.py:139-173), which this package uses when they are importable and otherwise refuses "
               "with an error rather than guessing.")
    for flag, kind, dflt, doc in _CLI:
        ap.add_argument(flag, type=kind, default=dflt, help=doc)
    ap.add_argument("--method", default="rk4", choices=("euler", "midpoint", "rk4"), help="ODE solver")
    ap.add_argument("--q", type=int, default=None, choices=(4, 8), help="load the MLX 4/8-bit checkpoint (model_v1_{4,8}b.safetensors).  The group-quantised weights are EXPANDED to fp32 on load and run "
                         "through the same 16-bit MFMA kernels as the full checkpoint: unlike in the reference this saves neither memory nor time here")
    ns = ap.parse_args(argv)

    text = ns.text
    if text is None:
        if sys.stdin.isatty():
            print("Please enter the text to generate:")
            text = input("> ").strip()
        else:
            text = sys.stdin.read().strip()                            # generate.py:330-335 strips both

    generate(text, duration=ns.duration, estimate_duration=ns.estimate_duration, model_name=ns.model, ref_audio_path=ns.ref_audio,
             ref_audio_text=ns.ref_text, steps=ns.steps, method=ns.method, cfg_strength=ns.cfg, sway_sampling_coef=ns.sway_coef,
             speed=ns.speed, seed=ns.seed, quantization_bits=ns.q, output_path=ns.output)


if __name__ == "__main__":
    main()
