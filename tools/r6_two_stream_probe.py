"""Round 6 probe: does running two half batches on two HIP streams fill the tails of each other's launches?  (GPU box, measurement only.)

At batch 32 every block GEMM is 3.67 / 7.34 / 11 rounds of 256 one-per-CU workgroups; the block is a strict chain, so a launch's last,
partly filled round idles CUs.  Two independent half batches on two streams give the dispatcher a second kernel to fill them with.
Measured here with what exists: two engines (own weights, own workspace) of 16 utterances each, replaying their graphs from two host
threads, against one engine with all 32 -- same utterances, same solver.

usage: python tools/r6_two_stream_probe.py [rounds] > gpurun_out/TAG/two_streams.jsonl
"""
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from f5_tts_mlx_amd.cfm import F5TTS  # noqa: E402
from f5_tts_mlx_amd.dit import DiT  # noqa: E402
from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
N = bench.N_FRAMES
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
weights = synthetic_weights(F5TTS_335M, seed=42)


def model():
    m = DiT.from_config(F5TTS_335M, precision="f16", device=dev)
    m.load_weights(weights)
    return F5TTS(transformer=m)


def kw_for(y0):
    return dict(duration=N, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0, use_graph=True)


full = model()
c32, t32, y32, _ = bench.synth_batch(32, first=0, device=dev)
halves = [model(), model()]
hin = [bench.synth_batch(16, first=0, device=dev), bench.synth_batch(16, first=16, device=dev)]
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]


def run_full():
    out, _ = full.sample(c32, text=t32, **kw_for(y32))
    return out


def run_half(i):
    c, t, y, _ = hin[i]
    with torch.cuda.stream(streams[i]):
        out, _ = halves[i].sample(c, text=t, **kw_for(y))
        streams[i].synchronize()
    return out


# warm-up and graph capture, one at a time
ref = run_full()
outs = [run_half(0), run_half(1)]
torch.cuda.synchronize()
split_l1 = float((torch.cat(outs, 0).float() - ref.float()).abs().mean())


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def both_serial():
    run_half(0)
    run_half(1)


def both_concurrent():
    th = [threading.Thread(target=run_half, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()


rec = dict(kind="two_streams", utterances=32, frames=N, split_vs_full_mean_abs=split_l1, ms={})
for r in range(rounds):
    for key, fn in (("one_engine_b32", run_full), ("two_b16_serial", both_serial), ("two_b16_two_streams", both_concurrent)):
        rec["ms"].setdefault(key, []).append(round(wall(fn), 2))
rec["best"] = {k: min(v) for k, v in rec["ms"].items()}
rec["two_streams_vs_b32"] = round(rec["best"]["two_b16_two_streams"] / rec["best"]["one_engine_b32"], 4)
print(json.dumps(rec), flush=True)

# the clock follows the average power of the last seconds: sustained sequences, per-call times, in both orders
seq = dict(kind="two_streams_sustained", calls_per_leg=6, legs=[])
for key, fn in (("one_engine_b32", run_full), ("two_b16_two_streams", both_concurrent), ("one_engine_b32", run_full),
                ("two_b16_two_streams", both_concurrent), ("two_b16_serial", both_serial), ("one_engine_b32", run_full)):
    seq["legs"].append({key: [round(wall(fn), 1) for _ in range(6)]})
print(json.dumps(seq), flush=True)

# ---- variants: four quarter batches on four streams; two halves enqueued from ONE host thread (no status read inside sample())
quarters = [model() for _ in range(4)]
qin = [bench.synth_batch(8, first=8 * i, device=dev) for i in range(4)]
qstreams = [torch.cuda.Stream(device=dev) for _ in range(4)]


def run_quarter(i):
    c, t, y, _ = qin[i]
    with torch.cuda.stream(qstreams[i]):
        out, _ = quarters[i].sample(c, text=t, **kw_for(y))
        qstreams[i].synchronize()
    return out


def four_concurrent():
    th = [threading.Thread(target=run_quarter, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()


for i in range(4):
    run_quarter(i)
for h in halves:
    h.transformer.engine.range_check = "off"


def two_one_thread():
    for i in range(2):
        c, t, y, _ = hin[i]
        with torch.cuda.stream(streams[i]):
            halves[i].sample(c, text=t, **kw_for(y))
    for st_ in streams:
        st_.synchronize()


two_one_thread()
seq = dict(kind="two_streams_variants", calls_per_leg=5, legs=[])
for key, fn in (("one_engine_b32", run_full), ("four_b8_four_streams", four_concurrent), ("two_b16_one_host_thread", two_one_thread),
                ("one_engine_b32", run_full), ("two_b16_one_host_thread", two_one_thread), ("four_b8_four_streams", four_concurrent)):
    seq["legs"].append({key: [round(wall(fn), 1) for _ in range(5)]})
print(json.dumps(seq), flush=True)
