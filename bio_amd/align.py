"""Host mirror of the reference's alignment API (Go package `bio`), routed through
the C ABI to the gfx950 kernels.  Same names, argument meaning and error
behaviour as the reference (file:line relative to the reference tree):

  AlignGlobal, AlignLocal          align.go:76, 152
  AlignPair                        align_pair.go:3
  AlignGlobalLS                    align_linear.go:9   (returns what AlignGlobal returns,
                                   as the disabled test align_test.go:55-66 expects)
  AlignGlobalAffine                align_affine.go:12  (score as written; canonical trace)
  LinearGap, ConstantGap           align.go:18, 41     (stay on the CPU)
  GapSymbol                        align.go:3          (module variable, read at call time)

Sequences are `bytes` (the reference's `type Seq []byte`, seq.go:18).  Empty
traces come back as None (Go's nil slice).  The batch entry points
(AlignPairBatch, AlignAffineBatch) are additive.

No alignment is ever computed on the CPU here.
"""
from __future__ import annotations

import numpy as np

from . import _native
from .subst_matrix import cooccurrence, score_table

GapSymbol = ord("-")  # align.go:3

_engine = None


def engine() -> _native.Engine:
    global _engine
    if _engine is None:
        _engine = _native.Engine()
    return _engine


def set_engine(e) -> None:
    global _engine
    _engine = e


def _gap() -> int:
    import bio_amd

    return int(getattr(bio_amd, "GapSymbol", GapSymbol)) & 0xFF


# ---------------------------------------------------------------------------
# rescoring (CPU, O(trace length))
# ---------------------------------------------------------------------------
def LinearGap(s: bytes, t: bytes, a, gapPenalty: int) -> int:
    """align.go:18-31."""
    if len(s) != len(t):
        raise RuntimeError("Sequences have different lengths")  # the reference panics
    g = _gap()
    score = 0
    for sa, ta in zip(s, t):
        if sa == g or ta == g:
            score -= gapPenalty
        else:
            score += a.Score(sa, ta)
    return score


def ConstantGap(s: bytes, t: bytes, a, gapPenalty: int) -> int:
    """align.go:41-69."""
    if len(s) != len(t):
        raise RuntimeError("Sequences have different lengths")
    g = _gap()
    score, open_gap = 0, 0  # 0 none, 1 gap in s, 2 gap in t
    for sa, ta in zip(s, t):
        if sa == g:
            if open_gap != 1:
                score -= gapPenalty
                open_gap = 1
        elif ta == g:
            if open_gap != 2:
                score -= gapPenalty
                open_gap = 2
        else:
            score += a.Score(sa, ta)
            open_gap = 0
    return score


# ---------------------------------------------------------------------------
# batches
# ---------------------------------------------------------------------------
def pack_sequences(seqs):
    """list of bytes -> (concatenated uint8 array, uint64 offsets[n+1])."""
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        np.cumsum([len(s) for s in seqs], out=off[1:])
    data = np.frombuffer(b"".join(seqs), dtype=np.uint8) if seqs else np.zeros(0, dtype=np.uint8)
    return data, off


LONG_MIN_CELLS = 1 << 24  # pairs of at least this many cells go to the long-mode kernels (bioalign_long)


def _long_pair(s1, s2, max_symbols=100):
    """One pair per matrix instead of one pair per lane: same results, the whole chip on the one pair.  "auto": try the
    long kernels, stay on the batch kernels if they refuse the parameters."""
    return "auto" if len(s1) * len(s2) >= LONG_MIN_CELLS and len(set(bytes(s2))) <= max_symbols else False


def _run_batch(mode, s1s, s2s, a, gap_open, gap_extend, score_only=False, long=False, scale=1):
    if mode not in _native.MODES:
        raise KeyError(mode)
    d1, o1 = pack_sequences(s1s)
    d2, o2 = pack_sequences(s2s)
    # (a Score that raises fails the batch only if some pair puts those two bytes against each other, as in the reference)
    table, code1, code2 = score_table(a, np.unique(d1).tolist(), np.unique(d2).tolist(),
                                      needed=cooccurrence(d1, o1, d2, o2) if len(s1s) > 1 else None)
    if scale != 1:
        table = table * np.int32(scale)  # dyadic penalties: everything in units of 1 / scale
    pb = _native.PackedBatch(d1, o1, d2, o2, mode, table, code1, code2, gap_open, gap_extend,
                             gap_symbol=_gap(), flags=_native.SCORE_ONLY if score_only else 0)
    try:
        scores, t1, t2, t_off, t_len = engine().batch(pb, long=long)
    except _native.BioAlignError as e:
        # a pair routed to the long kernels by its size alone that they do not take stays on the batch kernels
        if not (long == "auto" and e.code == _native.EUNSUPPORTED):
            raise
        scores, t1, t2, t_off, t_len = engine().batch(pb, long=False)
    if score_only:
        return [int(x) for x in scores], None, None
    t1b, t2b = t1.tobytes(), t2.tobytes()
    out1, out2 = [], []
    for p in range(len(s1s)):
        o, l = int(t_off[p]), int(t_len[p])
        out1.append(t1b[o:o + l] if l else None)
        out2.append(t2b[o:o + l] if l else None)
    return [int(x) for x in scores], out1, out2


def _trace_quirk(t1, t2):
    """AlignPair's trace() (align_pair.go:197-216) appends a symbol only if it is not 0 -- a NUL byte of a sequence (or
    a GapSymbol of 0) is silently dropped from its own trace -- and then reverses both traces in one loop that runs
    over len(t1)/2 and indexes both with len(t1)-1: when the two traces have lost different numbers of bytes, t2 comes
    out partly reversed, or the loop indexes past its end and Go panics.  Reproduced here on the host, O(trace), only
    for traces that hold a 0 byte; (t1, t2) are the aligned traces the kernels return, front to back."""
    if (t1 is None or 0 not in t1) and (t2 is None or 0 not in t2):
        return t1, t2
    r1 = bytearray(b for b in reversed(t1 or b"") if b != 0)  # as appended while walking back
    r2 = bytearray(b for b in reversed(t2 or b"") if b != 0)
    last = len(r1) - 1
    for i in range(len(r1) // 2):
        r1[i], r1[last - i] = r1[last - i], r1[i]
        if last - i >= len(r2):
            raise IndexError(f"index out of range [{last - i}] with length {len(r2)}")  # the reference panics here
        r2[i], r2[last - i] = r2[last - i], r2[i]
    return (bytes(r1) or None), (bytes(r2) or None)


def AlignPairBatch(mode: str, s1s, s2s, a, indelPenalty: int, score_only: bool = False):
    """Batch form of AlignPair: (scores, t1s, t2s), one entry per pair."""
    if len(s1s) != len(s2s):
        raise ValueError("s1s and s2s must have the same length")
    if mode not in _native.MODES:
        n = len(s1s)
        return [-1] * n, [None] * n, [None] * n  # align_pair.go:14-15
    if not s1s:
        return [], [], []
    sc, t1s, t2s = _run_batch(mode, list(s1s), list(s2s), a, indelPenalty, indelPenalty, score_only)
    if not score_only:
        for p in range(len(sc)):
            t1s[p], t2s[p] = _trace_quirk(t1s[p], t2s[p])
    return sc, t1s, t2s


def AlignAffineBatch(mode: str, s1s, s2s, a, gapOpenPenalty: int, gapExtendPenalty: int,
                     score_only: bool = False):
    """Batch Gotoh in the four modes (integers): a gap of length L costs
    open + (L-1)*extend (align_affine.go:51-57).  Local / fitting / overlap affine do
    not exist in the reference: borders and end cells follow align_pair.go:109-195 on
    the middle level (parity unpinned)."""
    if len(s1s) != len(s2s):
        raise ValueError("s1s and s2s must have the same length")
    if mode not in _native.MODES:
        n = len(s1s)
        return [-1] * n, [None] * n, [None] * n
    if not s1s:
        return [], [], []
    return _run_batch(mode, list(s1s), list(s2s), a, gapOpenPenalty, gapExtendPenalty, score_only)


# ---------------------------------------------------------------------------
# the reference's single-pair functions: a batch of one
# ---------------------------------------------------------------------------
def AlignPair(mode: str, s1: bytes, s2: bytes, a, indelPenalty: int):
    """align_pair.go:3-21.  Unknown mode -> (-1, None, None)."""
    if mode not in _native.MODES:
        return -1, None, None
    sc, t1, t2 = _run_batch(mode, [bytes(s1)], [bytes(s2)], a, indelPenalty, indelPenalty, long=_long_pair(s1, s2))
    return (sc[0],) + _trace_quirk(t1[0], t2[0])


def AlignGlobal(s1: bytes, s2: bytes, a, indelPenalty: int):
    """align.go:76-145."""
    if len(s2) == 0 and len(s1) > 0:
        # The reference's own traceback indexes s2[-1] here (stride 1 makes the two gap
        # deltas equal, align.go:128-130) and panics; keep the error behaviour.
        raise IndexError("index out of range [-1]")
    sc, t1, t2 = _run_batch("global", [bytes(s1)], [bytes(s2)], a, indelPenalty, indelPenalty, long=_long_pair(s1, s2))
    return sc[0], t1[0], t2[0]  # (align.go:126-143 appends every byte: no NUL quirk here)


def AlignLocal(s1: bytes, s2: bytes, a, indelPenalty: int):
    """align.go:152-249."""
    sc, t1, t2 = _run_batch("local", [bytes(s1)], [bytes(s2)], a, indelPenalty, indelPenalty, long=_long_pair(s1, s2))
    return sc[0], t1[0], t2[0]


def AlignPairLong(mode: str, s1: bytes, s2: bytes, a, indelPenalty: int):
    """AlignPair for ONE long pair: same result, computed by the intra-sequence wavefront
    kernels (bioalign_long).  All four modes, linear gap, <= 100 distinct column symbols."""
    if mode not in _native.MODES:
        return -1, None, None
    sc, t1, t2 = _run_batch(mode, [bytes(s1)], [bytes(s2)], a, indelPenalty, indelPenalty, long=True)
    return (sc[0],) + _trace_quirk(t1[0], t2[0])


def AlignGlobalLS(s1: bytes, s2: bytes, a, indelPenalty: int):
    """align_linear.go:9-45, the reference's entry point for long inputs.  Its function is WIP
    (panics, alternate tie-breaks); this one returns exactly AlignGlobal's result, which is what
    its author's disabled test asks for (align_test.go:55-66).  Pairs of LONG_MIN_CELLS cells
    or more go to the long-mode kernels when their parameters allow it."""
    sc, t1, t2 = _run_batch("global", [bytes(s1)], [bytes(s2)], a, indelPenalty, indelPenalty, long=_long_pair(s1, s2))
    return sc[0], t1[0], t2[0]


def AlignAffinePairLong(mode: str, s1: bytes, s2: bytes, a, gapOpenPenalty: int, gapExtendPenalty: int):
    """AlignAffineBatch for ONE long pair, computed by the intra-sequence wavefront kernels with Gotoh's three
    levels in registers (bioalign_long with open != extend): same result as the batch kernels, four modes,
    <= 64 distinct column symbols."""
    if mode not in _native.MODES:
        return -1, None, None
    sc, t1, t2 = _run_batch(mode, [bytes(s1)], [bytes(s2)], a, gapOpenPenalty, gapExtendPenalty, long=True)
    return sc[0], t1[0], t2[0]


def _dyadic_scale(*vals, max_shift=6):
    """The smallest power of two that turns every value into a whole number, or None."""
    for e in range(max_shift + 1):
        if all(float(v) * (1 << e) == int(float(v) * (1 << e)) for v in vals):
            return 1 << e
    return None


def AlignGlobalAffine(s1: bytes, s2: bytes, a, gapOpenPenalty: float, gapExtendPenalty: float):
    """align_affine.go:12-167: float64 penalties and score, integer arithmetic underneath.

    Fractional penalties are not truncated: penalties that are multiples of 1/2, 1/4 ... 1/64 run
    in units of that fraction (substitution scores and penalties times 2^e, exact integer Gotoh,
    score divided back), which is what the reference's float64 arithmetic computes exactly --
    every sum it forms is a small multiple of 2^-e, so no rounding and the same comparisons.
    Any other fraction (0.1, 1/3) makes the reference's result depend on float64 rounding along
    the path; that cannot be reproduced bit for bit in integers and is refused loudly."""
    go, ge = float(gapOpenPenalty), float(gapExtendPenalty)
    scale = _dyadic_scale(go, ge)
    if scale is None:
        raise ValueError("AlignGlobalAffine on the GPU path needs penalties that are multiples of 1/64 "
                         f"(got open {go!r}, extend {ge!r}): the kernels are exact integer Gotoh")
    # (a linear gap, open == extend, stays on the batch kernels: bioalign_long would take it for AlignPair's recurrence)
    use_long = _long_pair(s1, s2, 64) if go != ge else False
    sc, t1, t2 = _run_batch("global", [bytes(s1)], [bytes(s2)], a, int(go * scale), int(ge * scale), scale=scale,
                            long=use_long)
    return sc[0] / scale, t1[0], t2[0]
