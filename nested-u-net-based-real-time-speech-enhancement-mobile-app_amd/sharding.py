"""Multi-GPU plumbing: the path shards by independent streams (utterances), one process per GPU,
weights replicated, no activation ever crosses xGMI.  The only collective is the reduction of
the throughput counters at the end of a measurement window (RCCL on GPUs, gloo in CPU tests).
The reference has no distributed code at all; this mirrors nothing."""
from __future__ import annotations

from typing import Optional, Tuple


def stream_range(total_streams: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the streams rank ``rank`` owns; remainders go to the lowest ranks."""
    if not (0 <= rank < world) or total_streams < 0:
        raise ValueError("bad rank/world")
    base, rem = divmod(total_streams, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_throughput(frames: int, elapsed_s: float, dist=None, device=None) -> Tuple[int, float]:
    """(sum of frames over ranks, max of elapsed seconds over ranks).  ``dist`` is
    ``torch.distributed`` (initialised) or None for a single process."""
    if dist is None:
        return int(frames), float(elapsed_s)
    import torch
    f = torch.tensor([float(frames)], dtype=torch.float64, device=device)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(round(f.item())), float(t.item())


def collective_proof(rank_rate: float, dist=None, device=None) -> dict:
    """Evidence that the collective really spanned the job: every rank contributes a 1 to an all-reduce(SUM) (``ranks_reduced``
    must equal the world size) and its own frames/s to an all-gather (``per_rank`` -- rank order).  ``backend`` is what
    ``torch.distributed`` runs on ("nccl" is RCCL on ROCm; "gloo" in the CPU tests; "none" for a single process)."""
    if dist is None:
        return {"ranks_reduced": 1, "backend": "none", "per_rank": [float(rank_rate)]}
    import torch
    one = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    mine = torch.tensor([float(rank_rate)], dtype=torch.float64, device=device)
    every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    return {"ranks_reduced": int(round(one.item())), "backend": str(dist.get_backend()), "per_rank": [float(x.item()) for x in every]}
