"""Round-3 probe 3 (GPU box): two half-batch sample() calls on two CU-MASKED streams (each stream owns half of the CUs of every
XCD, hipExtStreamCreateWithCUMask; tools/probes/cumask.hip established the bit layout and that captured graphs keep the mask)
against one batch-32 call on the whole chip.  The idea: the HBM-bound phases of one lane (residual epilogues, LN-modulate) run
under the matrix-bound phases of the other, and a lane that multiplies while the other streams gets the power budget."""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402
from bench import synth_batch, N_FRAMES  # noqa: E402

dev = torch.device("cuda:0")
hip = C.CDLL("libamdhip64.so")


def masked_stream(lo_half: bool, ncu=256, split=128):
    words = (ncu + 31) // 32
    m = (C.c_uint32 * words)()
    for i in range(ncu):
        if (i < split) == lo_half:
            m[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def main():
    from f5_tts_mlx_amd.cfm import F5TTS
    from f5_tts_mlx_amd.dit import DiT
    from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights
    weights = synthetic_weights(F5TTS_335M, seed=42)
    models = []
    for _ in range(2):
        m = DiT.from_config(F5TTS_335M, precision="f16", device=dev)
        m.load_weights(weights)
        models.append(F5TTS(transformer=m))
    kw = dict(duration=N_FRAMES, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, use_graph=True)
    c32, t32, y32, _ = synth_batch(32, 0, dev)
    halves = [(c32[:16].contiguous(), t32[:16].contiguous(), y32[:16].contiguous()), (c32[16:].contiguous(), t32[16:].contiguous(), y32[16:].contiguous())]

    def timed(fn, iters=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3, r

    ms32, o32 = timed(lambda: models[0].sample(c32, t32, y0=y32, **kw)[0])
    res = dict(b32_full_chip_ms=round(ms32, 1))
    print(json.dumps(res), flush=True)
    for split in (128,):
        lanes = [masked_stream(True, split=split), masked_stream(False, split=split)]
        # the engine's own side stream IS the masked stream (graphs are captured on and replayed into it)
        for i in range(2):
            models[i].transformer.engine._stream = lanes[i]
            models[i].transformer.engine.set_graph_cache(0) if False else None
        outer = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

        def run_half(i):
            with torch.cuda.stream(outer[i]):
                c, t, y = halves[i]
                return models[i].sample(c, t, y0=y, **kw)[0]

        # new streams -> new graph captures (the cached graphs were captured on another stream: fine to replay, but the mask comes
        # from the stream the graph is launched into, so nothing to invalidate)
        ms_a, _ = timed(lambda: run_half(0))
        ms_both, (oa, ob) = timed(lambda: (run_half(0), run_half(1)))
        for delay in (0.5, 2.0):
            def both_delayed():
                a = run_half(0)
                with torch.cuda.stream(outer[1]):
                    torch.cuda._sleep(int(delay * 1e-3 * 100e6))
                return a, run_half(1)
            ms_d, _ = timed(both_delayed)
            res[f"split{split}_two_lanes_delay{delay}ms"] = round(ms_d, 1)
        res[f"split{split}_lane0_alone_ms"] = round(ms_a, 1)
        res[f"split{split}_two_lanes_ms"] = round(ms_both, 1)
        res[f"split{split}_equal_b32"] = bool(torch.equal(torch.cat([oa, ob]), o32))
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
