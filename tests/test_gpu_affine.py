"""GPU parity, affine gaps: the HIP Gotoh kernels through the C ABI against the
oracle's integer Gotoh (bo_align_affine), bit for bit on scores and both traces, in all four modes.

What is pinned by the reference: the global SCORE (align_affine.go:36-104,139-140; checked here
against the oracle's float64 restatement of exactly those lines).  Affine traces and the local /
fitting / overlap affine variants do not exist in the reference: they are the oracle's canonical
definition (PARITY UNPINNED beyond the global score), and every trace is also checked
for self-consistency: it spells the aligned substrings and rescoring it gives the reported score.
"""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bio_amd
from bio_amd import synth
from oracle import oracle as O

MODES = ("global", "local", "fitting", "overlap")


def _seqs(s, o, n):
    return [bytes(s[int(o[p]):int(o[p + 1])]) for p in range(n)]


def _self_consistent(mode, x, y, sc, t1, t2, oa, go, ge):
    if t1 is None:
        assert t2 is None
        return
    assert len(t1) == len(t2)
    g = ord("-")
    a = bytes(b for b in t1 if b != g)
    b = bytes(b for b in t2 if b != g)
    assert a in x and b in y
    if mode == "global":
        assert a == x and b == y
    if mode == "fitting":
        assert a == x
    if mode == "overlap":
        assert x.endswith(a) and y.startswith(b)
    assert O.AffineGap(t1, t2, oa, go, ge) == sc


def _check_affine(mode, s1s, s2s, pa, oa, go, ge):
    scores, t1s, t2s = bio_amd.AlignAffineBatch(mode, s1s, s2s, pa, go, ge)
    for p, (x, y) in enumerate(zip(s1s, s2s)):
        want = O.AlignAffine(mode, x, y, oa, go, ge)
        got = (scores[p], t1s[p], t2s[p])
        assert got == want, (mode, p, x, y, go, ge, got, want)
        _self_consistent(mode, x, y, *got, oa, go, ge)


@pytest.mark.parametrize("mode", MODES)
def test_affine_dna_uniform(mode):
    # config-4 scoring (+1/-1, open 5, extend 1) on 150 x 150: five strips of 32 rows, 10 rows above the matrix
    s1, o1, s2, o2 = synth.dna_pairs(200, 150, 150, config=2)
    _check_affine(mode, _seqs(s1, o1, 200), _seqs(s2, o2, 200), bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1), 5, 1)


@pytest.mark.parametrize("mode", MODES)
def test_affine_dna_ragged_with_empties(mode):
    rng = random.Random(77 + MODES.index(mode))
    alpha = b"ACGT"
    s1s, s2s = [], []
    for _ in range(260):
        n = rng.choice([0, 1, 2, 3, 31, 32, 33, 63, 64, 65, rng.randrange(0, 90)])
        m = rng.choice([0, 1, 2, 5, 17, rng.randrange(0, 90)])
        s1s.append(bytes(rng.choice(alpha) for _ in range(n)))
        s2s.append(bytes(rng.choice(alpha) for _ in range(m)))
    go, ge = rng.choice([(5, 1), (3, 2), (4, 0), (2, 1)])
    _check_affine(mode, s1s, s2s, bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1), go, ge)


@pytest.mark.parametrize("mode", MODES)
def test_affine_low_complexity_ties(mode):
    # two-letter, highly repetitive inputs: many ties between opening and extending, between levels
    rng = random.Random(5)
    s1s = [bytes(rng.choice(b"AAAC") for _ in range(rng.randrange(1, 70))) for _ in range(192)]
    s2s = [bytes(rng.choice(b"AAAC") for _ in range(rng.randrange(1, 70))) for _ in range(192)]
    _check_affine(mode, s1s, s2s, bio_amd.MatchAligner(2, -1), O.MatchAligner(2, -1), 3, 1)


@pytest.mark.parametrize("mode", MODES)
def test_affine_protein(mode):
    # config-3 scoring: Blosum62, open 11, extend 1 (16-row strips, two columns per decision word)
    rng = random.Random(11)
    s1, o1, s2, o2 = synth.protein_pairs(128, 75, 81, config=3)
    s1s, s2s = _seqs(s1, o1, 128), _seqs(s2, o2, 128)
    alpha = b"ARNDCQEGHILKMFPSTWYV"
    for _ in range(60):
        s1s.append(bytes(rng.choice(alpha) for _ in range(rng.randrange(0, 50))))
        s2s.append(bytes(rng.choice(alpha) for _ in range(rng.randrange(0, 50))))
    _check_affine(mode, s1s, s2s, bio_amd.Blosum62, O.Blosum62(), 11, 1)


def test_affine_local_best_cell_keys_int_and_float():
    # The local kernels find the first best cell through a packed key, built as a float when (score + 1) << key_shift
    # stays below 2^24 and in integers otherwise (kernels_affine.hip, FKEY): the same pairs with small scores (float
    # keys) and with scores 30 times as large (integer keys: 302 * 60 >= 2^(24 - 10)); many equal maxima (repeats).
    rng = random.Random(23)
    s1s = [bytes(rng.choice(b"ACGT") for _ in range(rng.choice([300, 290, 64]))) for _ in range(70)]
    s2s = [bytes(x[k] if rng.random() < 0.8 else rng.choice(b"ACGT") for k in range(len(x))) + b"ACGT" * 3 for x in s1s]
    s1s += [b"ACGTACGTAC" * 30] * 6
    s2s += [b"GTACGTAC" * 37] * 6
    _check_affine("local", s1s, s2s, bio_amd.MatchAligner(2, -3), O.MatchAligner(2, -3), 5, 2)
    _check_affine("local", s1s, s2s, bio_amd.MatchAligner(60, -60), O.MatchAligner(60, -60), 40, 20)


@pytest.mark.parametrize("mode", ["global", "local"])
def test_affine_scores_near_the_float_pattern_range(mode):
    # The kernels hold a score x as the float pattern 0x4B400000 + x, exact for |x| < 2^22, and the planner admits pairs
    # with (n + m + 2) * (128 + open + extend) < 2^22.  7900 x 8000 with extend 60 puts the working scores near a million
    # (bias 60 * (i + j)) and the global score far below zero; bit-exact against the oracle all the same.
    rng = random.Random(404)
    x = bytes(rng.choice(b"ACGT") for _ in range(7900))
    y = bytes(rng.choice(b"ACGT") for _ in range(8000))
    pa, oa = bio_amd.MatchAligner(6, -127), O.MatchAligner(6, -127)
    scores, t1s, t2s = bio_amd.AlignAffineBatch(mode, [x, b"ACGTTGCA"], [y, b"ACGTGCA"], pa, 61, 60)
    for p, (a, b) in enumerate([(x, y), (b"ACGTTGCA", b"ACGTGCA")]):
        assert (scores[p], t1s[p], t2s[p]) == O.AlignAffine(mode, a, b, oa, 61, 60), (mode, p)
    if mode == "global":
        assert scores[0] < -200000
    with pytest.raises(bio_amd._native.BioAlignError):  # one step past the planner's bound
        bio_amd.AlignAffineBatch(mode, [x + x + x], [y + y + y], pa, 61, 60)


def test_affine_fitting_reads_against_windows():
    # config-4 shape scaled down: read = s1, window = s2
    s1, o1, s2, o2 = synth.fitting_reads(70, 60, 400, config=4)
    s1s, s2s = _seqs(s1, o1, 70), _seqs(s2, o2, 70)
    for mode in ("fitting", "overlap"):
        _check_affine(mode, s1s, s2s, bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1), 5, 1)


def test_align_global_affine_score_is_the_references():
    # the one thing the reference pins: smm[last] of align_affine.go:36-104 in float64
    rng = random.Random(2)
    alpha = b"ARNDCQEGHILKMFPSTWYV"
    for _ in range(40):
        x = bytes(rng.choice(alpha) for _ in range(rng.randrange(1, 40)))
        y = bytes(rng.choice(alpha) for _ in range(rng.randrange(1, 40)))
        # whole and fractional penalties (multiples of 1/64 run in units of the fraction, nothing is truncated)
        go, ge = rng.choice([(11.0, 1.0), (5.0, 2.0), (7.0, 0.0), (2.5, 0.5), (10.75, 1.25), (3.5, 3.5)])
        sc, t1, t2 = bio_amd.AlignGlobalAffine(x, y, bio_amd.Blosum62, go, ge)
        assert isinstance(sc, float)
        assert sc == O.AffineGlobalScoreF64(x, y, O.Blosum62(), go, ge)
        assert bytes(b for b in t1 if b != ord("-")) == x and bytes(b for b in t2 if b != ord("-")) == y
    # a fraction whose unit makes the scores outgrow the int8 profile runs on the wide (int32) kernels, exactly
    sc, t1, t2 = bio_amd.AlignGlobalAffine(b"ACDDE", b"ACE", bio_amd.Blosum62, 1.015625, 0.5)
    assert sc == O.AffineGlobalScoreF64(b"ACDDE", b"ACE", O.Blosum62(), 1.015625, 0.5)
    assert t1.replace(b"-", b"") == b"ACDDE" and t2.replace(b"-", b"") == b"ACE"
    # a fraction that is no multiple of 1/64 is refused by the host side, not rounded
    with pytest.raises(ValueError):
        bio_amd.AlignGlobalAffine(b"ACDDE", b"ACE", bio_amd.Blosum62, 0.1, 0.5)
    # the input on which the reference's own traceback goes wrong: score as written, trace correct
    sc, t1, t2 = bio_amd.AlignGlobalAffine(b"ACDDE", b"ACE", bio_amd.Blosum62, 5, 1)
    assert sc == O.AffineGlobalScoreF64(b"ACDDE", b"ACE", O.Blosum62(), 5.0, 1.0)
    assert t1 == b"ACDDE" and t2.replace(b"-", b"") == b"ACE"


def test_affine_unsupported_parameters_fail_loudly():
    a = bio_amd.MatchAligner(1, -1)
    with pytest.raises(bio_amd._native.BioAlignError) as e:
        bio_amd.AlignAffineBatch("global", [b"ACGT"], [b"ACG"], a, 1, 3)  # open < extend
    assert e.value.code == bio_amd._native.EUNSUPPORTED
    with pytest.raises(bio_amd._native.BioAlignError):
        bio_amd.AlignAffineBatch("global", [b"ACGT"], [b"ACG"], a, -1, -2)


def test_affine_score_only_and_gap_symbol():
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    s1, o1, s2, o2 = synth.dna_pairs(64, 90, 100, config=7)
    s1s, s2s = _seqs(s1, o1, 64), _seqs(s2, o2, 64)
    sc, t1, t2 = bio_amd.AlignAffineBatch("local", s1s, s2s, a, 4, 1, score_only=True)
    assert t1 is None and t2 is None
    assert sc == [O.AlignAffine("local", x, y, oa, 4, 1)[0] for x, y in zip(s1s, s2s)]
    bio_amd.GapSymbol = ord(".")
    O.set_gap_symbol(ord("."))
    try:
        got = bio_amd.AlignAffineBatch("global", [b"ACGTTTTTACGT"], [b"ACGTACGT"], a, 3, 1)
        want = O.AlignAffine("global", b"ACGTTTTTACGT", b"ACGTACGT", oa, 3, 1)
        assert (got[0][0], got[1][0], got[2][0]) == want
        assert want[0] == 8 - 3 - 3 and b"-" not in want[2] and want[2].count(b".") == 4
    finally:
        bio_amd.GapSymbol = ord("-")
        O.set_gap_symbol(ord("-"))


@pytest.mark.parametrize("mode", ["local", "fitting"])
@pytest.mark.parametrize("go,ge", [(11, 1), (7, 2), (9, 4)])
def test_affine_protein_nibble_profile(mode, go, ge, monkeypatch):
    # large alphabet + free top row: the strip profile holds 4-bit scores (kernels_affine.hip, NIB); both launch
    # shapes (strips pipelined over the waves of a block / one wave per bundle) against the oracle and against
    # the byte profile.  ge = 4 leaves the padding nibble exactly 0.
    rng = random.Random(5)
    alpha = b"ARNDCQEGHILKMFPSTWYV"
    s1s = [bytes(rng.choice(alpha) for _ in range(rng.randrange(0, 90))) for _ in range(150)]
    s2s = [bytes(rng.choice(alpha) for _ in range(rng.randrange(0, 90))) for _ in range(150)]
    s1, o1, s2, o2 = synth.protein_pairs(64, 50, 47, config=3)
    s1s += _seqs(s1, o1, 64)
    s2s += _seqs(s2, o2, 64)
    _check_affine(mode, s1s, s2s, bio_amd.Blosum62, O.Blosum62(), go, ge)
    got = bio_amd.AlignAffineBatch(mode, s1s, s2s, bio_amd.Blosum62, go, ge)
    monkeypatch.setenv("BIOALIGN_PIPE_BELOW", "0")
    assert bio_amd.AlignAffineBatch(mode, s1s, s2s, bio_amd.Blosum62, go, ge) == got
    monkeypatch.setenv("BIOALIGN_NIBBLE_PROFILE", "0")
    assert bio_amd.AlignAffineBatch(mode, s1s, s2s, bio_amd.Blosum62, go, ge) == got


def test_affine_pam250_keeps_the_byte_profile():
    # Pam250 spans 26 values: no 4-bit profile; same answers as the oracle
    rng = random.Random(6)
    alpha = b"ARNDCQEGHILKMFPSTWYV"
    s1s = [bytes(rng.choice(alpha) for _ in range(rng.randrange(1, 60))) for _ in range(100)]
    s2s = [bytes(rng.choice(alpha) for _ in range(rng.randrange(1, 60))) for _ in range(100)]
    _check_affine("local", s1s, s2s, bio_amd.Pam250, O.Pam250(), 10, 2)


@pytest.mark.parametrize("mode", MODES)
def test_affine_hundred_column_symbols_large_lds(mode):
    # as test_gpu_parity.test_hundred_column_symbols_large_lds, through the affine kernels (open 6, extend 1)
    class Wide:
        def Score(self, x, y):
            return 4 if x == y else -((x * 5 + y * 3) % 4) - 1

    table = np.zeros((256, 256), dtype=np.int32)
    for x in range(256):
        for y in range(256):
            table[x, y] = 4 if x == y else -((x * 5 + y * 3) % 4) - 1
    oa = O.TableAligner(table)
    rng = random.Random(43)
    s1s = [bytes(rng.randrange(46, 146) for _ in range(rng.randrange(20, 90))) for _ in range(70)]
    s2s = [bytes(s1s[p][k] if rng.random() < 0.7 else rng.randrange(46, 146) for k in range(len(s1s[p]))) for p in range(70)]
    s2s[0] = bytes(range(46, 146))
    _check_affine(mode, s1s, s2s, Wide(), oa, 6, 1)
