"""Greedy fragment assembly on real overlap alignments.

The reference's AlignLong (align_long.go:17-90) finds overlaps by exact substring search -- the last half of s must
occur in t verbatim (align_long.go:35-45) -- so one read error breaks the assembly.  AlignLongOverlap keeps the
reference's greedy merge (steps 2-4, align_long.go:46-89) statement by statement and replaces step 1 by ONE call
(bioalign_all_ordered_pairs) that aligns every ordered pair of fragments in "overlap" mode (suffix of s against
prefix of t, align_pair.go:184-195) on the GPU, the fragments lying in HBM once.  Only scores and end cells come back (BIOALIGN_SCORE_ONLY | BIOALIGN_END_CELLS: the end cell's column is
the length of t's prefix inside the overlap); there is no traceback at all.

This changes behaviour on purpose (an overlap may contain mismatches and gaps), which is why SURVEY ranks it last.
On error-free reads WITHOUT repeated half-fragments it returns what AlignLong returns.  With repeats the two can
differ in either direction: the reference takes the first verbatim occurrence of s's last half anywhere in t
(strings.Index, align_long.go:41) without looking at t's prefix and ranks by that position, this code takes the
leftmost best bottom-row cell of an overlap alignment, with score and half-length thresholds
(tests/test_assembly.py::test_repeated_half_fragment_records_both_behaviours holds one fixture for each direction).
Parity is unpinned beyond that: the reference has no test of AlignLong.
"""
from __future__ import annotations

import numpy as np

from . import _native
from .subst_matrix import MatchAligner, score_table


def overlap_end_cells(ss, pairs, a, indelPenalty: int):
    """Overlap alignments of (ss[si], ss[ti]) for every (si, ti) in pairs -> (scores, j_end): int64 arrays; j_end =
    length of the prefix of ss[ti] that the alignment covers (column of the end cell, align_pair.go:169-182).

    One call of bioalign_all_ordered_pairs: the fragments lie in HBM once, all n(n-1) ordered pairs are aligned
    (score and end cell only, no traceback) and the pairs asked for are picked from the result."""
    from .align import _gap, engine, pack_sequences

    n = len(ss)
    data, off = pack_sequences(ss)
    syms = np.unique(data).tolist()
    table, code1, code2 = score_table(a, syms, syms)
    pb = _native.PackedBatch(data, off, np.zeros(0, np.uint8), off, "overlap", table, code1, code2, indelPenalty,
                             indelPenalty, gap_symbol=_gap(), flags=_native.SCORE_ONLY | _native.END_CELLS)
    scores, _, j_end = engine().all_ordered_pairs(pb)
    si = np.fromiter((p[0] for p in pairs), dtype=np.int64, count=len(pairs))
    ti = np.fromiter((p[1] for p in pairs), dtype=np.int64, count=len(pairs))
    idx = si * (n - 1) + np.where(ti < si, ti, ti - 1)
    return scores[idx].astype(np.int64), j_end[idx].astype(np.int64)


def AlignLongOverlap(ss, a=None, indelPenalty: int = 2, minScorePerColumn: float = 0.8, end_cells=None) -> bytes:
    """AlignLong (align_long.go:17) with alignment-based overlaps.  All fragments have the same length; an overlap
    counts when it covers at least half a fragment of t (the reference's requirement, align_long.go:10-13) and
    scores at least minScorePerColumn per covered column of t.  -> the assembled string, or b"" if the fragments
    do not assemble into one.

    end_cells(ss, pairs, a, indelPenalty) -> (scores, j_end): the default is the GPU batch; there is no CPU
    fallback (the CPU test of the merge logic injects the oracle explicitly)."""
    ss = [bytes(s) for s in ss]
    if len(ss) == 0:
        return b""
    if len(ss) == 1:
        return ss[0]
    a = a if a is not None else MatchAligner(1, -1)
    half = len(ss[0]) // 2
    # 1. pairwise overlap alignments of all fragments (the reference skips t == s, align_long.go:38-40)
    pairs = [(si, ti) for si in range(len(ss)) for ti in range(len(ss)) if ss[ti] != ss[si]]
    overlaps = []
    if pairs:
        scores, j_end = (end_cells or overlap_end_cells)(ss, pairs, a, indelPenalty)
        for (si, ti), sc, j in zip(pairs, scores.tolist(), j_end.tolist()):
            if j >= half and j > 0 and sc >= minScorePerColumn * j:
                overlaps.append([si, ti, j - half])  # x + half = overlap, as in the reference
    left = 0
    for _ in range(len(ss), 1, -1):
        # 2. the two fragments with the largest overlap (first one wins ties, align_long.go:51-61)
        x_max, o_max = -1, 0
        for i, (si, ti, x) in enumerate(overlaps):
            s, t = ss[si], ss[ti]
            if s == b"" or t == b"" or t == s:
                continue
            if x > x_max:
                x_max, o_max = x, i
        if x_max == -1:
            return b""
        # 3. merge them
        left, right = overlaps[o_max][0], overlaps[o_max][1]
        ss[left] += ss[right][x_max + half:]
        ss[right] = b""
        # overlaps from si are deleted (swap with the last), overlaps from ti become overlaps from si
        i = 0
        while i < len(overlaps):
            if overlaps[i][0] == left:
                last = len(overlaps) - 1
                if i < last:
                    overlaps[i] = overlaps[last]
                overlaps.pop()
                continue
            if overlaps[i][0] == right:
                overlaps[i][0] = left
            i += 1
    return ss[left]
