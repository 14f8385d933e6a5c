"""Scoring: the Aligner interface and SubstMatrix (reference align.go:5-8,
subst_matrix.go:8-131), host side.  The GPU never sees these objects: the batch
wrapper turns any Aligner into a dense table once per batch (score_table())."""
from __future__ import annotations

import numpy as np


class SubstMatrix:
    """subst_matrix.go:8-12.  Alphabet: bytes; Matrix: [len][len] ints; Index: byte -> row."""

    def __init__(self, alphabet: bytes, matrix, index):
        self.Alphabet = bytes(alphabet)
        self.Matrix = matrix
        self.Index = index

    def Score(self, x: int, y: int) -> int:
        # Matrix[Index[x]][Index[y]] (subst_matrix.go:15-17).  Index is zero filled up
        # to the largest alphabet byte, so smaller foreign bytes alias to symbol 0 and
        # larger ones raise IndexError where the Go code panics.
        return self.Matrix[self.Index[x]][self.Index[y]]


def NewSubstMatrix(alphabet: bytes, matrix) -> SubstMatrix:
    """subst_matrix.go:20-32."""
    alphabet = bytes(alphabet)
    index = [0] * (max(alphabet) + 1 if alphabet else 1)
    for i, b in enumerate(alphabet):
        index[b] = i
    return SubstMatrix(alphabet, matrix, index)


class MatchAligner:
    """match/mismatch scorer, as the reference's tests define locally
    (align_fitting_test.go:9-16, align_overlap_test.go:9-19)."""

    def __init__(self, match: int, mismatch: int):
        self.match, self.mismatch = match, mismatch

    def Score(self, x: int, y: int) -> int:
        return self.match if x == y else self.mismatch


def _parse(text: str) -> SubstMatrix:
    """Same text layout parseSubstMatrix accepts (subst_matrix.go:113-131): a header
    row of symbols, then one labelled row per symbol."""
    tok = text.split()
    n = 0
    while n < len(tok) and len(tok[n]) == 1 and tok[n].isalpha() and (n == 0 or tok[n] not in tok[:n]):
        n += 1
    alphabet = "".join(tok[:n]).encode()
    rows = []
    pos = n
    for _ in range(n):
        pos += 1  # row label
        rows.append([int(v) for v in tok[pos:pos + n]])
        pos += n
    return NewSubstMatrix(alphabet, rows)


# BLOSUM62, alphabet order of subst_matrix.go:64
Blosum62 = _parse("""
   A  R  N  D  C  Q  E  G  H  I  L  K  M  F  P  S  T  W  Y  V
A  4 -1 -2 -2  0 -1 -1  0 -2 -1 -1 -1 -1 -2 -1  1  0 -3 -2  0
R -1  5  0 -2 -3  1  0 -2  0 -3 -2  2 -1 -3 -2 -1 -1 -3 -2 -3
N -2  0  6  1 -3  0  0  0  1 -3 -3  0 -2 -3 -2  1  0 -4 -2 -3
D -2 -2  1  6 -3  0  2 -1 -1 -3 -4 -1 -3 -3 -1  0 -1 -4 -3 -3
C  0 -3 -3 -3  9 -3 -4 -3 -3 -1 -1 -3 -1 -2 -3 -1 -1 -2 -2 -1
Q -1  1  0  0 -3  5  2 -2  0 -3 -2  1  0 -3 -1  0 -1 -2 -1 -2
E -1  0  0  2 -4  2  5 -2  0 -3 -3  1 -2 -3 -1  0 -1 -3 -2 -2
G  0 -2  0 -1 -3 -2 -2  6 -2 -4 -4 -2 -3 -3 -2  0 -2 -2 -3 -3
H -2  0  1 -1 -3  0  0 -2  8 -3 -3 -1 -2 -1 -2 -1 -2 -2  2 -3
I -1 -3 -3 -3 -1 -3 -3 -4 -3  4  2 -3  1  0 -3 -2 -1 -3 -1  3
L -1 -2 -3 -4 -1 -2 -3 -4 -3  2  4 -2  2  0 -3 -2 -1 -2 -1  1
K -1  2  0 -1 -3  1  1 -2 -1 -3 -2  5 -1 -3 -1  0 -1 -3 -2 -2
M -1 -1 -2 -3 -1  0 -2 -3 -2  1  2 -1  5  0 -2 -1 -1 -1 -1  1
F -2 -3 -3 -3 -2 -3 -3 -3 -1  0  0 -3  0  6 -4 -2 -2  1  3 -1
P -1 -2 -2 -1 -3 -1 -1 -2 -2 -3 -3 -1 -2 -4  7 -1 -1 -4 -3 -2
S  1 -1  1  0 -1  0  0  0 -1 -2 -2  0 -1 -2 -1  4  1 -3 -2 -2
T  0 -1  0 -1 -1 -1 -1 -2 -2 -1 -1 -1 -1 -2 -1  1  5 -2 -2  0
W -3 -3 -4 -4 -2 -2 -3 -2 -2 -3 -2 -3 -1  1 -4 -3 -2 11  2 -3
Y -2 -2 -2 -3 -2 -1 -2 -3  2 -1 -1 -2 -1  3 -3 -2 -2  2  7 -1
V  0 -3 -3 -3 -1 -2 -2 -3 -3  3  1 -2  1 -1 -2 -2  0 -3 -1  4
""")

# PAM250, alphabet order of subst_matrix.go:90
Pam250 = _parse("""
   A  C  D  E  F  G  H  I  K  L  M  N  P  Q  R  S  T  V  W  Y
A  2 -2  0  0 -3  1 -1 -1 -1 -2 -1  0  1  0 -2  1  1  0 -6 -3
C -2 12 -5 -5 -4 -3 -3 -2 -5 -6 -5 -4 -3 -5 -4  0 -2 -2 -8  0
D  0 -5  4  3 -6  1  1 -2  0 -4 -3  2 -1  2 -1  0  0 -2 -7 -4
E  0 -5  3  4 -5  0  1 -2  0 -3 -2  1 -1  2 -1  0  0 -2 -7 -4
F -3 -4 -6 -5  9 -5 -2  1 -5  2  0 -3 -5 -5 -4 -3 -3 -1  0  7
G  1 -3  1  0 -5  5 -2 -3 -2 -4 -3  0  0 -1 -3  1  0 -1 -7 -5
H -1 -3  1  1 -2 -2  6 -2  0 -2 -2  2  0  3  2 -1 -1 -2 -3  0
I -1 -2 -2 -2  1 -3 -2  5 -2  2  2 -2 -2 -2 -2 -1  0  4 -5 -1
K -1 -5  0  0 -5 -2  0 -2  5 -3  0  1 -1  1  3  0  0 -2 -3 -4
L -2 -6 -4 -3  2 -4 -2  2 -3  6  4 -3 -3 -2 -3 -3 -2  2 -2 -1
M -1 -5 -3 -2  0 -3 -2  2  0  4  6 -2 -2 -1  0 -2 -1  2 -4 -2
N  0 -4  2  1 -3  0  2 -2  1 -3 -2  2  0  1  0  1  0 -2 -4 -2
P  1 -3 -1 -1 -5  0  0 -2 -1 -3 -2  0  6  0  0  1  0 -1 -6 -5
Q  0 -5  2  2 -5 -1  3 -2  1 -2 -1  1  0  4  1 -1 -1 -2 -5 -4
R -2 -4 -1 -1 -4 -3  2 -2  3 -3  0  0  0  1  6  0 -1 -2  2 -4
S  1  0  0  0 -3  1 -1 -1  0 -3 -2  1  1 -1  0  2  1 -1 -2 -3
T  1 -2  0  0 -3  0 -1  0  0 -2 -1  0  0 -1 -1  1  3  0 -5 -3
V  0 -2 -2 -2 -1 -1 -2  4 -2  2  2 -2 -1 -2 -2 -1  0  4 -6 -2
W -6 -8 -7 -7  0 -7 -3 -5 -3 -2 -4 -4 -6 -5  2 -2 -5 -6 17  0
Y -3  0 -4 -4  7 -5  0 -1 -4 -1 -2 -2 -5 -4 -4 -3 -3 -2  0 10
""")


def score_table(a, bytes1, bytes2, needed=None):
    """Materialises an Aligner for the GPU.

    bytes1 / bytes2: the distinct bytes that occur on the s1 / s2 side of the batch.
    Calls a.Score(x, y) exactly once per (s1 byte, s2 byte) pair, argument order
    (s1 byte, s2 byte) as align.go:113; symmetry is not assumed.  An exception from
    Score propagates like the reference's panic -- unless `needed` is given: the
    reference scores (s1[i], s2[j]) inside one pair only, so a batch whose pairs never
    put x against y must not fail because Score(x, y) would (an Aligner that panics on
    foreign bytes, subst_matrix.go:15-17).  `needed(x, y) -> bool` says whether some
    pair of the batch holds x on its s1 side and y on its s2 side; it is asked only
    for cells whose Score raised, and those that no pair needs are left at 0.
    Returns (table[na, nb] int32, code1[256] uint8, code2[256] uint8).
    """
    b1 = sorted(set(bytes1))
    b2 = sorted(set(bytes2))
    na, nb = max(len(b1), 1), max(len(b2), 1)
    table = np.zeros((na, nb), dtype=np.int64)
    code1 = np.zeros(256, dtype=np.uint8)
    code2 = np.zeros(256, dtype=np.uint8)
    for i, x in enumerate(b1):
        code1[x] = i
    for j, y in enumerate(b2):
        code2[y] = j
    for i, x in enumerate(b1):
        for j, y in enumerate(b2):
            try:
                table[i, j] = a.Score(x, y)
            except Exception:
                if needed is None or needed(x, y):
                    raise
    if table.size and (table.max() > 2**31 - 1 or table.min() < -(2**31)):
        raise OverflowError("Aligner scores do not fit 32 bits")
    return table.astype(np.int32), code1, code2


def cooccurrence(d1, o1, d2, o2):
    """needed(x, y) for score_table: does some pair p hold byte x in s1[p] and byte y in s2[p]?  Built on first use
    (one pass over the batch), so batches whose Aligner never raises pay nothing."""
    state = {}

    def needed(x, y):
        if "m" not in state:
            n = len(o1) - 1
            has1 = np.zeros((n, 256), dtype=bool)
            has2 = np.zeros((n, 256), dtype=bool)
            if len(d1):
                has1[np.repeat(np.arange(n), np.diff(o1).astype(np.int64)), d1] = True
            if len(d2):
                has2[np.repeat(np.arange(n), np.diff(o2).astype(np.int64)), d2] = True
            state["m"] = (has1.astype(np.int32).T @ has2.astype(np.int32)) > 0
        return bool(state["m"][x, y])

    return needed
