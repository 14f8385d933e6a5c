"""GPU parity, long mode: one pair aligned by the intra-sequence wavefront
kernels (bioalign_long) against the oracle, bit for bit on score and both traces.  Sizes are chosen
around the kernel's seams: one / several strips of 64*T rows, column counts around the 64-column
hand-over blocks, the three residue widths."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bio_amd
from bio_amd import synth
from oracle import oracle as O


def _pair(n, m, seed, div=True):
    if n == 0 or m == 0:
        rng = random.Random(seed)
        return bytes(rng.choice(b"ACGT") for _ in range(n)), bytes(rng.choice(b"ACGT") for _ in range(m))
    s1, o1, s2, o2 = synth.dna_pairs(1, n, m, config=5, first=seed, sub=0.06, ins=0.02, dele=0.02)
    return s1.tobytes(), s2.tobytes()


@pytest.mark.parametrize("mode", ["global", "fitting", "overlap"])
def test_long_small_shapes(mode):
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    shapes = [(1, 1), (5, 7), (7, 5), (63, 64), (64, 63), (65, 65), (100, 127), (128, 128), (130, 129),
              (511, 70), (512, 200), (513, 190), (700, 900), (1100, 300), (30, 1000), (0, 5), (6, 0)]
    for k, (n, m) in enumerate(shapes):
        x, y = _pair(n, m, 100 + k)
        assert bio_amd.AlignPairLong(mode, x, y, a, 1) == O.AlignPair(mode, x, y, oa, 1), (mode, n, m)


@pytest.mark.parametrize("scoring", [(1, -1, 1), (1, -2, 2), (5, -4, 3), (10, -3, 5), (2, -1, 0)])
def test_long_residue_widths(scoring):
    ma, mi, g = scoring
    a, oa = bio_amd.MatchAligner(ma, mi), O.MatchAligner(ma, mi)
    for k, (n, m) in enumerate([(300, 310), (1030, 520), (260, 777)]):
        x, y = _pair(n, m, 300 + k)
        for mode in ("global", "fitting", "overlap"):
            assert bio_amd.AlignPairLong(mode, x, y, a, g) == O.AlignPair(mode, x, y, oa, g), (scoring, mode, n, m)


def test_long_low_complexity_ties():
    rng = random.Random(9)
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    for _ in range(6):
        x = bytes(rng.choice(b"AAAC") for _ in range(rng.randrange(400, 1300)))
        y = bytes(rng.choice(b"AAAC") for _ in range(rng.randrange(400, 1300)))
        for mode in ("global", "fitting", "overlap"):
            assert bio_amd.AlignPairLong(mode, x, y, a, 1) == O.AlignPair(mode, x, y, oa, 1)


def test_long_several_strips_and_align_global_ls():
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    x, y = _pair(3000, 2500, 500)
    assert bio_amd.AlignPairLong("global", x, y, a, 1) == O.AlignPair("global", x, y, oa, 1)
    # AlignGlobalLS (align_linear.go:9) routes pairs of >= 2^24 cells to the long kernels
    x, y = _pair(4300, 4100, 501)
    assert len(x) * len(y) >= bio_amd.align.LONG_MIN_CELLS
    got = bio_amd.AlignGlobalLS(x, y, a, 1)
    assert got == O.AlignGlobal(x, y, oa, 1)
    assert bio_amd.LinearGap(got[1], got[2], a, 1) == got[0]


def test_long_unsupported_fails_loudly():
    with pytest.raises(bio_amd._native.BioAlignError) as e:
        bio_amd.AlignPairLong("local", b"ACGT" * 10, b"ACG" * 10, bio_amd.MatchAligner(1, -1), 1)
    assert e.value.code == bio_amd._native.EUNSUPPORTED
    with pytest.raises(bio_amd._native.BioAlignError):
        bio_amd.AlignPairLong("global", b"ARNDC" * 10, b"ARNDCQ" * 10, bio_amd.Blosum62, 5)  # 6 column symbols
