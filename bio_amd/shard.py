"""Sharding a batch of pairs over ranks (one process per GPU), SURVEY.md 8(e).

Pairs are independent, so the batch is cut into `world` CONTIGUOUS ranges balanced by
cell count (sum of len1*len2) and every rank aligns its own range on its own GPU.  There
is no data-path collective.  When the caller wants the whole result on one rank the
per-rank results are gathered as host objects over the process group (gloo or RCCL
alike); a pipeline that consumes results where they were produced passes gather=False.

The same cut is what bioalign_batch() applies inside one process across the devices of
its context (bio_amd/csrc/api.cpp: run_share).
"""
from __future__ import annotations

import numpy as np


def shard_ranges(len1, len2, world: int):
    """Cuts pairs [0, n) into `world` contiguous ranges of nearly equal cell count.

    Returns bounds[world+1] (int64), bounds[0] = 0, bounds[world] = n, non-decreasing.
    Range r is [bounds[r], bounds[r+1]).  Pairs with zero cells still cost a launch slot,
    so every pair counts at least one cell.
    """
    l1 = np.asarray(len1, dtype=np.int64)
    l2 = np.asarray(len2, dtype=np.int64)
    if l1.shape != l2.shape:
        raise ValueError("len1 and len2 differ in shape")
    if world < 1:
        raise ValueError("world must be >= 1")
    n = l1.size
    cells = np.maximum(l1 * l2, 1)
    csum = np.cumsum(cells)
    total = int(csum[-1]) if n else 0
    bounds = np.zeros(world + 1, dtype=np.int64)
    for r in range(1, world):
        # first index whose running total reaches r/world of the work
        target = (total * r + world - 1) // world
        bounds[r] = int(np.searchsorted(csum, target, side="left")) + 1 if n else 0
    bounds[world] = n
    np.minimum(bounds, n, out=bounds)
    np.maximum.accumulate(bounds, out=bounds)
    return bounds


def align_sharded(mode: str, s1s, s2s, a, gap_open: int, gap_extend: int | None = None, *,
                  group=None, gather: bool = True, dst: int = 0, align_fn=None):
    """Aligns the batch across the ranks of a torch.distributed process group.

    Every rank calls this with the SAME batch (or at least the same lengths); rank r
    aligns range r of shard_ranges() with `align_fn(mode, s1s, s2s, a, gap_open, gap_extend)`
    -> (scores, t1s, t2s).  The default align_fn is the GPU batch entry point; there is no
    CPU fallback here (tests inject the oracle explicitly).

    gather=True : rank `dst` returns (scores, t1s, t2s) for the whole batch in input
                  order, the other ranks return None.
    gather=False: every rank returns ((lo, hi), (scores, t1s, t2s)) for its own range.
    """
    import torch.distributed as dist

    if gap_extend is None:
        gap_extend = gap_open
    if align_fn is None:
        from .align import _run_batch

        def align_fn(mode, x, y, a, go, ge):
            return _run_batch(mode, x, y, a, go, ge) if x else ([], [], [])

    if len(s1s) != len(s2s):
        raise ValueError("s1s and s2s must have the same length")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    bounds = shard_ranges([len(s) for s in s1s], [len(s) for s in s2s], world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    mine = align_fn(mode, list(s1s[lo:hi]), list(s2s[lo:hi]), a, gap_open, gap_extend)
    if not gather:
        return (lo, hi), mine
    if world == 1:
        return mine
    parts = [None] * world if rank == dst else None
    dist.gather_object(mine, parts, dst=dst, group=group)
    if rank != dst:
        return None
    scores, t1s, t2s = [], [], []
    for p in parts:
        scores += list(p[0])
        t1s += list(p[1])
        t2s += list(p[2])
    return scores, t1s, t2s


def max_over_ranks(seconds: float, device=None, group=None) -> float:
    """The timing reduction bench.py's contract asks for: max of a per-rank duration."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
