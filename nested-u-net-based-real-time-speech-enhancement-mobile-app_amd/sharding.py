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
