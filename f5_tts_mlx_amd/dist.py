"""Multi-GPU support: utterance sharding + one RCCL broadcast of the weights arena.

The sampling path has no cross-utterance arithmetic, so scale-out is data parallel with NO per-step
communication: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), rank 0 holds
the checkpoint, the contiguous device arena (bf16: ~0.7 GB, bf16x3: ~1.4 GB) is replicated with a single
broadcast, every rank samples its shard.  Caveat carried over from the reference: a ragged batch must be
padded to the GLOBAL max duration on every rank (GRN and the unmasked conv-pos-embed make results depend
on the padded length, convnext_v2.py:16, dit.py:251) — `shard_batch` returns that length.
"""
from __future__ import annotations

import time
from typing import List, Sequence, Tuple

import torch


def shard_ranges(n_items: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced [start, end) ranges; the first (n_items % world) ranks get one extra item."""
    q, r = divmod(n_items, world)
    out, s = [], 0
    for k in range(world):
        e = s + q + (1 if k < r else 0)
        out.append((s, e))
        s = e
    return out


def shard_batch(durations: Sequence[int], world: int, rank: int) -> Tuple[range, int]:
    """Indices of this rank's utterances and the padded length every rank must use (global max)."""
    s, e = shard_ranges(len(durations), world)[rank]
    return range(s, e), int(max(durations)) if len(durations) else 0


def broadcast_weights(engine, src: int = 0, group=None) -> float:
    """Replicate rank `src`'s weights arena to every rank with one collective. Returns milliseconds."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0.0
    arena = engine.arena
    if arena.is_cuda:
        torch.cuda.synchronize(arena.device)
    t0 = time.perf_counter()
    dist.broadcast(arena, src=src, group=group)
    if arena.is_cuda:
        torch.cuda.synchronize(arena.device)
    ms = (time.perf_counter() - t0) * 1e3
    if dist.get_rank(group) != src:
        engine.mark_loaded_from_broadcast()
    return ms


def gather_outputs(local: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor | None:
    """Collect per-rank result slabs (b_r, N, d) on rank 0 (optional convenience; not on the hot path)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bmax = max(counts)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()                       # gloo gathers host tensors (the RCCL job gathers device tensors)
    pad = torch.zeros((bmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
