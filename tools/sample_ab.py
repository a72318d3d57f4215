"""A/B of per-engine options at sample() level: ONE engine (335M synthetic weights), the bench workload, rounds that alternate the
settings (each setting has its own cached hipGraph, the option values are part of the graph key).

    python tools/sample_ab.py [--batch 1] [--rounds 6] [--iters 10] name=value[,name=value...] [name=value...] ...
e.g. python tools/sample_ab.py gemm_flags=0 gemm_flags=4096          (two settings of one engine option)
Prints the ms per sample() of every round and setting, and whether the final mels of all settings are bit-identical."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
# A/B of two LIBRARIES: run the tool once per library with F5TTS_HIP_LIB=<other build> (f5_tts_mlx_amd/engine.py reads it)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--precision", default="f16")
    ap.add_argument("settings", nargs="+")
    a = ap.parse_args()
    from f5_tts_mlx_amd.cfm import F5TTS
    from f5_tts_mlx_amd.dit import DiT
    from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights
    dev = torch.device("cuda:0")
    model = DiT.from_config(F5TTS_335M, precision=a.precision, device=dev)
    model.load_weights(synthetic_weights(F5TTS_335M, seed=42))
    f5 = F5TTS(transformer=model, vocoder=None)
    cond, text, y0, _ = bench.synth_batch(a.batch, first=0, device=dev)
    kw = dict(duration=bench.N_FRAMES, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0, use_graph=True)
    settings = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in s.split(",")) for s in a.settings]

    def apply(st):
        for k, v in st.items():
            model.engine.set_option(k, v)

    outs, ms = [], [[] for _ in settings]
    for st in settings:                                     # capture + warm every setting
        apply(st)
        for _ in range(3):
            out, _ = f5.sample(cond, text, **kw)
        torch.cuda.synchronize()
        outs.append(out.clone())
    for r in range(a.rounds):
        for i, st in enumerate(settings):
            apply(st)
            f5.sample(cond, text, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                f5.sample(cond, text, **kw)
            torch.cuda.synchronize()
            ms[i].append(round((time.perf_counter() - t0) / a.iters * 1e3, 2))
    print(json.dumps(dict(lib=os.path.basename(os.environ.get("F5TTS_HIP_LIB", "libf5tts_hip.so")), batch=a.batch, precision=a.precision, settings=a.settings, ms=ms,
                          median=[sorted(m)[len(m) // 2] for m in ms],
                          identical=[bool(torch.equal(outs[0], o)) for o in outs])))


if __name__ == "__main__":
    main()
