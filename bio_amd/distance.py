"""All-pairs distance matrices over alignment scores.

The reference's DNA8List.DistanceMatrix (dna_list.go:951-969) and StrList.DistanceMatrix (str.go:450-469) call a
distance function f(l[i], l[j]) for every j < i and mirror the value into a full [][]float64 with a zero
diagonal.  When f is a function of an alignment score -- an AlignDist -- the n(n-1)/2 alignments are one
score-only batch on the GPU (bioalign_all_pairs): the list lies in HBM once and pairs address it by offset.
"""
from __future__ import annotations

import numpy as np

from . import _native
from .subst_matrix import score_table


class AlignDist:
    """A DistFunc8 / DistFuncStr (dna_list.go:951, str.go:450) of the form to_dist(score of AlignPair(mode, a, b,
    aligner, gap)), or of the Gotoh score when gapExtend differs from gapOpen.  to_dist maps an int64 array of
    scores to float64 distances (default: -score, so that a higher score is a smaller distance)."""

    def __init__(self, mode: str, a, gapOpen: int, gapExtend: int | None = None, to_dist=None):
        if mode not in _native.MODES:
            raise KeyError(mode)
        self.mode, self.a = mode, a
        self.gap_open = int(gapOpen)
        self.gap_extend = self.gap_open if gapExtend is None else int(gapExtend)
        self.to_dist = to_dist or (lambda s: -s.astype(np.float64))

    def __call__(self, s: bytes, t: bytes) -> float:
        """One pair, as the reference would call it."""
        from .align import _run_batch

        scores, _, _ = _run_batch(self.mode, [bytes(s)], [bytes(t)], self.a, self.gap_open, self.gap_extend, score_only=True)
        return float(self.to_dist(np.asarray(scores, dtype=np.int64))[0])


def AlignScores(l, f: AlignDist, engine=None, pair_range=None) -> np.ndarray:
    """scores[i*(i-1)/2 + j] = score of aligning l[i] (rows) with l[j] (columns), j < i.
    `l`: a list of bytes, or a FastaBatch (records already in HBM).  pair_range=(first, end): only those pairs."""
    from .align import _gap, engine as default_engine, pack_sequences
    from .fasta import FastaBatch

    eng = engine or default_engine()
    if isinstance(l, FastaBatch):
        syms = np.unique(l._fa.seq_bytes()).tolist()
        data, off = np.zeros(0, np.uint8), np.zeros(1, np.uint64)
    else:
        data, off = pack_sequences([bytes(s) for s in l])
        syms = np.unique(data).tolist()
    table, code1, code2 = score_table(f.a, syms, syms)
    pb = _native.PackedBatch(data, off, np.zeros(0, np.uint8), off, f.mode, table, code1, code2, f.gap_open, f.gap_extend,
                             gap_symbol=_gap(), flags=_native.SCORE_ONLY)
    return eng.all_pairs(pb, fasta=l._fa if isinstance(l, FastaBatch) else None, pair_range=pair_range)


def pair_ranges(lens, world: int):
    """Cuts the n(n-1)/2 pairs (numbered i(i-1)/2 + j, j < i) into `world` contiguous ranges of nearly equal cell
    count, as bioalign_all_pairs does across the devices of one context: row i costs (len_i + 1) times the sum of
    (len_j + 1) over j < i.  -> bounds[world + 1], bounds[0] = 0, bounds[world] = n(n-1)/2, non-decreasing."""
    if world < 1:
        raise ValueError("world must be >= 1")
    w = np.asarray(lens, dtype=np.float64) + 1.0
    n = w.size
    total = n * (n - 1) // 2
    bounds = np.zeros(world + 1, dtype=np.int64)
    bounds[world] = total
    if n >= 2:
        cum = np.concatenate(([0.0], np.cumsum(w)))          # cum[i] = sum over j < i
        rowpre = np.concatenate(([0.0], np.cumsum(w * cum[:-1])))
        for r in range(1, world):
            target = rowpre[n] * r / world
            i = min(max(int(np.searchsorted(rowpre, target, side="right")), 1), n) - 1
            j = 0
            if i >= 1:
                j = min(int(np.searchsorted(cum[:i + 1], (target - rowpre[i]) / w[i], side="left")), i)
            bounds[r] = min(i * (i - 1) // 2 + j if i > 0 else 0, total)
    np.maximum.accumulate(bounds, out=bounds)
    return bounds


def AlignScoresSharded(l, f: AlignDist, *, group=None, scores_fn=None) -> np.ndarray:
    """AlignScores across the ranks of a torch.distributed process group (one process per GPU):
    every rank calls this with the same list, aligns range `rank` of pair_ranges() on its own GPU
    (bioalign_all_pairs_range) and all ranks return the whole score vector.  Pairs are independent: the only
    traffic is the gather of the scores themselves (8 bytes per pair), after the kernels.

    scores_fn(l, f, first, end) -> int64 scores of pairs [first, end): the default is the GPU; there is no CPU
    fallback (the CPU test injects the oracle explicitly)."""
    import torch
    import torch.distributed as dist

    if scores_fn is None:
        def scores_fn(l, f, first, end):
            return AlignScores(l, f, pair_range=(first, end))

    l = [bytes(s) for s in l]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    bounds = pair_ranges([len(s) for s in l], world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    mine = np.asarray(scores_fn(l, f, lo, hi), dtype=np.int64) if hi > lo else np.zeros(0, np.int64)
    if world == 1:
        return mine
    # equal-sized all_gather (ranges differ in pair count: pad to the longest)
    xdev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    longest = int(np.max(np.diff(bounds)))
    buf = torch.zeros(max(longest, 1), dtype=torch.int64)
    buf[:hi - lo] = torch.from_numpy(mine)
    buf = buf.to(xdev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return np.concatenate([parts[r].cpu().numpy()[:int(bounds[r + 1] - bounds[r])] for r in range(world)])


def DistanceMatrix(l, f) -> list:
    """dna_list.go:955-969 / str.go:454-469: d[i][j] = d[j][i] = f(l[i], l[j]) for j < i, zero diagonal, as a list
    of lists of floats.  An empty list panics in the reference (d[0] of an empty slice): IndexError here.
    f: an AlignDist (batched on the GPU), or any callable (called pair by pair, as the reference does)."""
    n = l.count if hasattr(l, "count") and not isinstance(l, (list, tuple)) else len(l)
    if n == 0:
        raise IndexError("index out of range [0] with length 0")
    d = np.zeros((n, n), dtype=np.float64)
    if n > 1:
        ii, jj = np.tril_indices(n, -1)  # row by row: pair i(i-1)/2 + j
        if isinstance(f, AlignDist):
            v = np.asarray(f.to_dist(AlignScores(l, f)), dtype=np.float64)
        else:
            v = np.array([f(l[i], l[j]) for i, j in zip(ii.tolist(), jj.tolist())], dtype=np.float64)
        d[ii, jj] = v
        d[jj, ii] = v
    return d.tolist()
