"""The multi-GPU protocol with the REAL engine in more than one process: two ranks share the one GPU of the
test box and talk over gloo (RCCL refuses two ranks on one device; the collective is the only thing that differs from the 8-GPU
job).  Rank 0 owns the weights, the arena travels by ONE broadcast, rank 1 marks it loaded and finalises, every rank samples
its shard of a RAGGED batch padded to the GLOBAL maximum duration (`dist.shard_batch`), rank 0 gathers -- and the gathered
result must equal the single-process batch bit for bit."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _inputs(cfg, B, N, seed=3):
    r = np.random.default_rng(seed)
    durations = [N - 11 * i for i in range(B)]
    cond = torch.from_numpy(r.standard_normal((B, 24, cfg.mel_dim)).astype(np.float32))
    text = torch.from_numpy(r.integers(0, cfg.text_num_embeds, (B, 20)).astype(np.int32))
    for i in range(1, B):
        text[i, 20 - i:] = -1
    y0 = np.zeros((B, N, cfg.mel_dim), np.float32)
    for i, d in enumerate(durations):
        y0[i, :d] = r.standard_normal((cfg.mel_dim, d)).astype(np.float32).T
    return cond, text, durations, torch.from_numpy(y0)


def _worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        from f5_tts_mlx_amd.cfm import F5TTS
        from f5_tts_mlx_amd.dist import broadcast_weights, gather_outputs, shard_batch, shard_ranges
        from f5_tts_mlx_amd.dit import DiT
        from f5_tts_mlx_amd.weights import TINY, synthetic_weights
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = "cuda:0"
        cfg = TINY
        B, N = 6, 120
        cond, text, durations, y0 = _inputs(cfg, B, N)
        model = DiT.from_config(cfg, precision="f16", device=dev)
        if rank == 0:
            model.load_weights(synthetic_weights(cfg, seed=42))      # only rank 0 ever sees the checkpoint
        eng = model.engine
        assert eng.weights_ready == (rank == 0)
        ms = broadcast_weights(eng, src=0)                           # the CUDA arena over the process group
        assert eng.weights_ready and ms >= 0.0
        idx, npad = shard_batch(durations, world, rank)
        assert npad == N
        sl = slice(idx.start, idx.stop)
        kw = dict(steps=5, method="midpoint", cfg_strength=2.0, sway_sampling_coef=-1.0)
        # a one-utterance shard would drop the key-padding mask (cfm.py:333-336 builds it only when batch > 1): shards of >= 2 here
        out, _ = F5TTS(transformer=model).sample(cond[sl], text[sl], duration=torch.tensor(durations[sl.start:sl.stop]),
                                                 y0=y0[sl].contiguous(), pad_to=npad, **kw)
        assert out.shape[1] == N                                      # padded to the GLOBAL length, not the shard's own maximum
        counts = [e - s for s, e in shard_ranges(B, world)]
        g = gather_outputs(out, counts)
        ok, msg = True, ""
        if rank == 0:
            ref, _ = F5TTS(transformer=model).sample(cond, text, duration=torch.tensor(durations), y0=y0, **kw)
            torch.cuda.synchronize()
            ok = g is not None and tuple(g.shape) == tuple(ref.shape)
            if ok:
                eq = torch.equal(g.to(ref.device), ref)
                diff = float((g.to(ref.device) - ref).abs().max())
                ok, msg = eq, f"max diff {diff:.3e}"
        else:
            ok = g is None
        q.put((rank, bool(ok), msg))
        dist.destroy_process_group()
    except Exception as exc:  # noqa: BLE001
        import traceback
        q.put((rank, False, traceback.format_exc()[-1500:] + repr(exc)))


def test_two_ranks_real_engine_shards_equal_single_process_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 23000 + (os.getpid() % 3000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, ok, msg = q.get(timeout=600)
        res[rank] = (ok, msg)
    for p in procs:
        p.join(timeout=120)
    assert res[0][0] and res[1][0], res
