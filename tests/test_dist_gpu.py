"""The multi-GPU protocol with the REAL engine in more than one process: two ranks share the one GPU of the
test box and talk over gloo (RCCL refuses two ranks on one device; the collective is the only thing that differs from the 8-GPU
job); where two devices are visible the same protocol also runs over nccl (= RCCL), one rank per device.  Rank 0 owns the weights, the arena travels by ONE broadcast, rank 1 marks it loaded and finalises, every rank samples
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


def _worker(rank, world, port, q, backend="gloo"):
    try:
        import torch.distributed as dist
        from f5_tts_mlx_amd.cfm import F5TTS
        from f5_tts_mlx_amd.dist import broadcast_weights, gather_outputs, shard_batch, shard_ranges
        from f5_tts_mlx_amd.dit import DiT
        from f5_tts_mlx_amd.weights import TINY, synthetic_weights
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        # gloo: both ranks on the one GPU of the test box; nccl (= RCCL): one rank per device, as the 8-GPU job runs
        dev = "cuda:0" if backend == "gloo" else f"cuda:{rank}"
        if backend == "nccl":
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
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


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_ranks_real_engine_shards_equal_single_process_batch(backend):
    """gloo: two ranks share the test box's one GPU.  nccl: the same protocol over RCCL with one rank per device -- runs wherever two
    devices are visible (the driver's multi-GPU node), skips on a one-GPU box."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (RCCL refuses two ranks on one device)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 23000 + (os.getpid() % 3000) + (7 if backend == "nccl" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, ok, msg = q.get(timeout=600)
        res[rank] = (ok, msg)
    for p in procs:
        p.join(timeout=120)
    assert res[0][0] and res[1][0], res


def test_one_utterance_shards_keep_the_key_mask():
    """a shard of ONE utterance (batch 6 over 6 GPUs) is padded to the global maximum by `pad_to` and must still
    run with the key-padding mask of the unsharded call (cfm.py:333-336 builds it when batch > 1) -- the padded frames are keys
    otherwise.  Every utterance of a ragged batch sampled alone with pad_to equals its row of the batched call (same arithmetic,
    possibly other tile shapes: 1e-5), and NOT the unmasked single-utterance result."""
    from f5_tts_mlx_amd.cfm import F5TTS
    from f5_tts_mlx_amd.dit import DiT
    from f5_tts_mlx_amd.weights import TINY, synthetic_weights
    cfg, B, N = TINY, 6, 120
    cond, text, durations, y0 = _inputs(cfg, B, N)
    model = DiT.from_config(cfg, precision="bf16x3", device="cuda:0")
    model.load_weights(synthetic_weights(cfg, seed=42))
    f5 = F5TTS(transformer=model)
    kw = dict(steps=5, method="midpoint", cfg_strength=2.0, sway_sampling_coef=-1.0)
    ref, _ = f5.sample(cond, text, duration=torch.tensor(durations), y0=y0, **kw)
    torch.cuda.synchronize()
    worst, apart = 0.0, 0.0
    for i in range(B):
        sl = slice(i, i + 1)
        out, _ = f5.sample(cond[sl], text[sl], duration=torch.tensor(durations[sl]), y0=y0[sl].contiguous(), pad_to=N, **kw)
        d = durations[i]
        worst = max(worst, float((out[0, :d] - ref[i, :d]).abs().max()))
        if d < N:
            # the same shard WITHOUT the mask (what round 3 did): engine level, use_mask=False
            eng = model.engine
            lens = torch.maximum((text[sl] != -1).sum(-1), torch.tensor([cond.shape[1]]))
            condp = torch.zeros((1, N, cfg.mel_dim), device="cuda:0")
            condp[:, :cond.shape[1]] = cond[sl].to("cuda:0")
            from f5_tts_mlx_amd.cfm import time_grid
            o2, _ = eng.sample(text[sl].to("cuda:0").contiguous(), condp, lens.tolist(), [d], y0[sl].to("cuda:0").contiguous(),
                               time_grid(5, -1.0), method="midpoint", cfg_strength=2.0, use_mask=False, use_graph=False)
            apart = max(apart, float((o2[0, :d] - ref[i, :d]).abs().max()))
    print(f"[one-utterance shards] worst |shard - batch row| = {worst:.3e}; unmasked shard is {apart:.3e} away")
    assert worst <= 1e-5 and apart > 1e-3


def test_pad_to_respects_max_duration():
    from f5_tts_mlx_amd.cfm import F5TTS
    from f5_tts_mlx_amd.dit import DiT
    from f5_tts_mlx_amd.weights import TINY, synthetic_weights
    cond, text, durations, y0 = _inputs(TINY, 2, 60)
    model = DiT.from_config(TINY, precision="f16", device="cuda:0")
    model.load_weights(synthetic_weights(TINY, seed=42))
    with pytest.raises(ValueError, match="exceeds max_duration"):
        F5TTS(transformer=model).sample(cond, text, duration=torch.tensor(durations), y0=y0, steps=3, method="euler", pad_to=5000)
