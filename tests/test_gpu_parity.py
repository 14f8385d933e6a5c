"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle,
bit for bit on scores and both traces, plus the reference's golden vectors."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bio_amd
from bio_amd import synth
from oracle import oracle as O

MODES = ("global", "local", "fitting", "overlap")


def _oracle_aligner(spec):
    if spec == "Blosum62":
        return O.Blosum62()
    if spec == "Pam250":
        return O.Pam250()
    return O.MatchAligner(spec["match"], spec["mismatch"])


def _product_aligner(spec):
    if spec == "Blosum62":
        return bio_amd.Blosum62
    if spec == "Pam250":
        return bio_amd.Pam250
    return bio_amd.MatchAligner(spec["match"], spec["mismatch"])


def test_golden_vectors_on_gpu(golden):
    for v in golden:
        a = _product_aligner(v["aligner"])
        fn = v["fn"]
        if fn in ("LinearGap", "ConstantGap"):
            assert getattr(bio_amd, fn)(v["s"].encode(), v["t"].encode(), a, v["gap"]) == v["score"], v["id"]
            continue
        want = (v["score"], v["t1"].encode(), v["t2"].encode())
        if fn == "AlignPair":
            got = bio_amd.AlignPair(v["mode"], v["s1"].encode(), v["s2"].encode(), a, v["gap"])
        else:
            got = getattr(bio_amd, fn)(v["s1"].encode(), v["s2"].encode(), a, v["gap"])
        assert got == want, v["id"]


def _check_batch(mode, s1s, s2s, pa, oa, gap):
    scores, t1s, t2s = bio_amd.AlignPairBatch(mode, s1s, s2s, pa, gap)
    for p, (x, y) in enumerate(zip(s1s, s2s)):
        want = O.AlignPair(mode, x, y, oa, gap)
        got = (scores[p], t1s[p], t2s[p])
        assert got == want, (mode, p, x, y, gap, got, want)


@pytest.mark.parametrize("mode", MODES)
def test_uniform_dna_batch(mode):
    # config-1 shape (100 bp), smaller count; DNA +1/-1/1 as align_fitting_test.go:9-16
    s1, o1, s2, o2 = synth.dna_pairs(300, 100, 100, config=1)
    s1s = [bytes(s1[int(o1[p]):int(o1[p + 1])]) for p in range(300)]
    s2s = [bytes(s2[int(o2[p]):int(o2[p + 1])]) for p in range(300)]
    _check_batch(mode, s1s, s2s, bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1), 1)


@pytest.mark.parametrize("mode", MODES)
def test_uniform_dna_batch_second_scoring(mode):
    # +1/-2/2 as align_overlap_test.go:25; 150 bp = config-2 shape (5 strips of 32, 10 padded rows)
    s1, o1, s2, o2 = synth.dna_pairs(130, 150, 150, config=2)
    s1s = [bytes(s1[int(o1[p]):int(o1[p + 1])]) for p in range(130)]
    s2s = [bytes(s2[int(o2[p]):int(o2[p + 1])]) for p in range(130)]
    _check_batch(mode, s1s, s2s, bio_amd.MatchAligner(1, -2), O.MatchAligner(1, -2), 2)


@pytest.mark.parametrize("mode", MODES)
def test_ragged_batch_with_empty_sequences(mode):
    rng = random.Random(1234 + MODES.index(mode))
    alpha = b"ACGT"
    s1s, s2s = [], []
    for _ in range(200):
        n = rng.choice([0, 1, 2, 3, 31, 32, 33, 63, 64, 65, rng.randrange(0, 90)])
        m = rng.choice([0, 1, 2, 5, 17, rng.randrange(0, 90)])
        s1s.append(bytes(rng.choice(alpha) for _ in range(n)))
        s2s.append(bytes(rng.choice(alpha) for _ in range(m)))
    _check_batch(mode, s1s, s2s, bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1), 1)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("matrix", ["Blosum62", "Pam250"])
def test_protein_batch(mode, matrix):
    rng = random.Random(99)
    s1, o1, s2, o2 = synth.protein_pairs(100, 70, 80, config=3)
    s1s = [bytes(s1[int(o1[p]):int(o1[p + 1])]) for p in range(100)]
    s2s = [bytes(s2[int(o2[p]):int(o2[p + 1])]) for p in range(100)]
    # a few ragged ones too
    alpha = b"ARNDCQEGHILKMFPSTWYV"
    for _ in range(40):
        s1s.append(bytes(rng.choice(alpha) for _ in range(rng.randrange(0, 50))))
        s2s.append(bytes(rng.choice(alpha) for _ in range(rng.randrange(0, 50))))
    _check_batch(mode, s1s, s2s, _product_aligner(matrix), _oracle_aligner(matrix), 5)


def test_fitting_reads_against_windows():
    # config-4 shape scaled down: read = s1 (rows), window = s2 (columns), matrix wider than tall
    s1, o1, s2, o2 = synth.fitting_reads(70, 60, 400, config=4)
    s1s = [bytes(s1[int(o1[p]):int(o1[p + 1])]) for p in range(70)]
    s2s = [bytes(s2[int(o2[p]):int(o2[p + 1])]) for p in range(70)]
    for mode in ("fitting", "overlap"):
        _check_batch(mode, s1s, s2s, bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1), 1)


def test_single_pair_api_and_error_behaviour():
    b = bio_amd.Blosum62
    assert bio_amd.AlignPair("nonsense", b"AC", b"AC", b, 5) == (-1, None, None)  # align_pair.go:14-15
    assert bio_amd.AlignGlobal(b"", b"", b, 5) == (0, None, None)
    with pytest.raises(IndexError):  # the reference panics: align.go:128-130 with an empty s2
        bio_amd.AlignGlobal(b"AC", b"", b, 5)
    assert bio_amd.AlignPair("global", b"AC", b"", b, 5) == (-10, b"AC", b"--")
    assert bio_amd.AlignGlobal(b"", b"AC", b, 5) == (-10, b"--", b"AC")
    assert bio_amd.AlignLocal(b"W", b"A", b, 5) == (0, None, None)
    with pytest.raises(IndexError):  # lower case is beyond Index (subst_matrix.go:27-30): panic
        bio_amd.AlignGlobal(b"a", b"A", b, 5)
    # foreign byte below max aliases to 'A'
    assert bio_amd.AlignGlobal(b"B", b"A", b, 5)[0] == 4
    # AlignGlobalLS returns AlignGlobal's result (align_test.go:55-66)
    assert bio_amd.AlignGlobalLS(b"PLEASANTLY", b"MEANLY", b, 5) == (8, b"PLEASANTLY", b"-MEA--N-LY")
    # a non-default GapSymbol (align.go:3)
    bio_amd.GapSymbol = ord("*")
    try:
        assert bio_amd.AlignGlobal(b"PLEASANTLY", b"MEANLY", b, 5) == (8, b"PLEASANTLY", b"*MEA**N*LY")
    finally:
        bio_amd.GapSymbol = ord("-")


def test_asymmetric_user_aligner():
    class Odd:
        def Score(self, x, y):  # not symmetric; argument order is (s1 byte, s2 byte)
            return (x * 7 - y * 3) % 11 - 4

    table = np.zeros((256, 256), dtype=np.int32)
    for x in range(256):
        for y in range(256):
            table[x, y] = (x * 7 - y * 3) % 11 - 4
    oa = O.TableAligner(table)
    rng = random.Random(3)
    s1s = [bytes(rng.randrange(97, 110) for _ in range(rng.randrange(1, 60))) for _ in range(80)]
    s2s = [bytes(rng.randrange(97, 110) for _ in range(rng.randrange(1, 60))) for _ in range(80)]
    for mode in MODES:
        _check_batch(mode, s1s, s2s, Odd(), oa, 3)


@pytest.mark.parametrize("mode", MODES)
def test_hundred_column_symbols_large_lds(mode):
    # 100 distinct symbols on the column side: the strip profile of one wavefront is 100 KB of dynamic LDS, past the
    # 64 KB a launch gets without asking (hipFuncSetAttribute in the launchers); linear gaps
    class Wide:
        def Score(self, x, y):
            return 4 if x == y else -((x * 5 + y * 3) % 4) - 1

    table = np.zeros((256, 256), dtype=np.int32)
    for x in range(256):
        for y in range(256):
            table[x, y] = 4 if x == y else -((x * 5 + y * 3) % 4) - 1
    oa = O.TableAligner(table)
    rng = random.Random(41)
    s1s = [bytes(rng.randrange(46, 146) for _ in range(rng.randrange(20, 90))) for _ in range(70)]
    s2s = [bytes(s1s[p][k] if rng.random() < 0.7 else rng.randrange(46, 146) for k in range(len(s1s[p]))) for p in range(70)]
    s2s[0] = bytes(range(46, 146))  # every one of the 100 symbols occurs
    _check_batch(mode, s1s, s2s, Wide(), oa, 2)


def test_nul_bytes_follow_align_pair_trace():
    # align_pair.go:202-207: a symbol 0 is not appended to its trace, and the reversal loop then runs over t1's length
    # for both traces (panics when t2 is too short for it); AlignGlobal / AlignLocal (align.go) append every byte
    table = np.zeros((256, 256), dtype=np.int32)
    for x in range(256):
        for y in range(256):
            table[x, y] = 2 if x == y else -1

    class Eq:
        def Score(self, x, y):
            return 2 if x == y else -1

    oa = O.TableAligner(table)
    rng = random.Random(5)
    checked = panicked = 0
    for _ in range(60):
        x = bytes(rng.choice(b"\x00AC") for _ in range(rng.randrange(1, 14)))
        y = bytes(rng.choice(b"\x00AC") for _ in range(rng.randrange(1, 14)))
        for mode in MODES:
            try:
                want = O.AlignPair(mode, x, y, oa, 1)
            except O.GoPanic:
                with pytest.raises(IndexError):
                    bio_amd.AlignPair(mode, x, y, Eq(), 1)
                panicked += 1
                continue
            assert bio_amd.AlignPair(mode, x, y, Eq(), 1) == want, (mode, x, y)
            checked += 1
        assert bio_amd.AlignGlobal(x, y, Eq(), 1) == O.AlignGlobal(x, y, oa, 1)
        assert bio_amd.AlignLocal(x, y, Eq(), 1) == O.AlignLocal(x, y, oa, 1)
    assert checked > 50 and panicked > 5
