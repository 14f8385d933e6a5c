"""The bit-parallel kernels (bio_amd/csrc/kernels_bp.hip: unit scoring +1 / -1 / gap 1, four modes) against
the oracle (align.go:76-143, 152-249; align_pair.go:157-195 for fitting and overlap), bit for bit on scores and both traces, and against the general kernels."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bio_amd
from bio_amd import _native, synth, score_table
from oracle import oracle as O


def _plan_kernel(s1s, s2s, aligner, gap, mode="global"):
    syms1 = sorted(set(b"".join(s1s))) or [65]
    syms2 = sorted(set(b"".join(s2s))) or [65]
    table, c1, c2 = score_table(aligner, syms1, syms2)
    s1 = np.frombuffer(b"".join(s1s) or b"\0", dtype=np.uint8)[: sum(map(len, s1s))]
    s2 = np.frombuffer(b"".join(s2s) or b"\0", dtype=np.uint8)[: sum(map(len, s2s))]
    o1 = np.cumsum([0] + [len(x) for x in s1s]).astype(np.uint64)
    o2 = np.cumsum([0] + [len(x) for x in s2s]).astype(np.uint64)
    eng = _native.Engine([0])
    plan = eng.plan(_native.PackedBatch(s1, o1, s2, o2, mode, table, c1, c2, gap, gap))
    k = plan.kernel
    plan.close()
    eng.close()
    return k


MODES = ("global", "local", "fitting", "overlap")
A, OA = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)


def _check(mode, s1s, s2s, pa=A, oa=OA, gap=1):
    scores, t1s, t2s = bio_amd.AlignPairBatch(mode, s1s, s2s, pa, gap)
    for p, (x, y) in enumerate(zip(s1s, s2s)):
        want = O.AlignPair(mode, x, y, oa, gap)
        assert (scores[p], t1s[p], t2s[p]) == want, (mode, p, x, y)


def _rand(rnd, n, al=b"ACGT"):
    return bytes(rnd.choice(al) for _ in range(n))


def _edited(rnd, x, rate=0.12, al=b"ACGT"):
    out = bytearray()
    for c in x:
        r = rnd.random()
        if r < rate / 3:
            continue
        if r < 2 * rate / 3:
            out.append(rnd.choice(al))
        out.append(c if r > rate else rnd.choice(al))
    return bytes(out)


def test_which_plans_run_bit_parallel(monkeypatch):
    a = bio_amd.MatchAligner(1, -1)
    x = [b"ACGTACGTAC" * 15] * 3
    assert _plan_kernel(x, x, a, 1) == "bitparallel"
    assert _plan_kernel(x, x, a, 1, mode="local") == "bitparallel"
    assert _plan_kernel(x, x, a, 1, mode="fitting") == "bitparallel"
    assert _plan_kernel(x, x, a, 1, mode="overlap") == "bitparallel"
    assert _plan_kernel([b"ACGT" * 64], [b"ACGT" * 64], a, 1, mode="local") == "linear"  # a local score of 256 needs nine bits
    assert _plan_kernel([b"ACGT" * 64], [b"ACGT" * 63], a, 1, mode="local") == "bitparallel"
    assert _plan_kernel(x, x, a, 2) == "linear"
    assert _plan_kernel(x, x, bio_amd.MatchAligner(1, -2), 1) == "linear"
    assert _plan_kernel([b"ACGT" * 65], [b"ACGT"], a, 1) == "linear"  # 260 rows: more than eight words
    assert _plan_kernel([b"ACGT" * 64], [b"ACGT"], a, 1) == "bitparallel"
    assert _plan_kernel([b"ACGTN"], [b"ACGTN"], a, 1) == "linear"  # five column symbols
    assert _plan_kernel([b"ACGTN"], [b"ACGT"], a, 1) == "bitparallel"
    monkeypatch.setenv("BIOALIGN_BITPARALLEL", "0")
    assert _plan_kernel(x, x, a, 1) == "linear"


@pytest.mark.parametrize("mode", MODES)
def test_config1_shape_150bp(mode):
    # BASELINE configs[1]'s pair shape: 150 x 150, five words of rows, ten padded rows above the matrix
    s1, o1, s2, o2 = synth.dna_pairs(200, 150, 150, config=2)
    s1s = [bytes(s1[int(o1[p]):int(o1[p + 1])]) for p in range(200)]
    s2s = [bytes(s2[int(o2[p]):int(o2[p + 1])]) for p in range(200)]
    assert _plan_kernel(s1s, s2s, A, 1, mode=mode) == "bitparallel"
    _check(mode, s1s, s2s)


@pytest.mark.parametrize("mode", MODES)
def test_every_row_count_and_ragged_columns(mode):
    # every number of rows 0..256 (all word counts, all paddings) against columns of every length modulo 4, related
    # and unrelated sequences, in one ragged batch: bundles of every word count in one launch
    rnd = random.Random(11)
    s1s, s2s = [], []
    for n in range(0, 257):
        x = _rand(rnd, n)
        s1s.append(x)
        s2s.append(_edited(rnd, x)[:255] if n % 3 else _rand(rnd, rnd.randint(0, 255)))
    assert _plan_kernel(s1s, s2s, A, 1, mode=mode) == "bitparallel"
    _check(mode, s1s, s2s)


@pytest.mark.parametrize("mode", MODES)
def test_empty_sides_and_single_cells(mode):
    cases = [(b"", b""), (b"A", b""), (b"", b"C"), (b"A", b"A"), (b"A", b"C"), (b"ACGT", b""), (b"", b"ACGT" * 40),
             (b"A" * 256, b"A"), (b"A", b"A" * 300), (b"ACGT" * 64, b"ACGT" * 63), (b"A" * 33, b"C" * 31),
             (b"A" * 255, b"A" * 255), (b"ACGT" * 50, b"TGCA" * 50), (b"C" * 200, b"A" * 100 + b"C" + b"A" * 100)]
    s1s, s2s = [c[0] for c in cases], [c[1] for c in cases]
    assert _plan_kernel(s1s, s2s, A, 1, mode=mode) == "bitparallel"
    _check(mode, s1s, s2s)


def test_local_end_cell_is_the_first_maximum_in_row_major_order():
    # several cells hold the maximum: the end cell is the first of them by rows, then by columns (align.go:214-221) --
    # repeats put equal maxima in the same row, in higher rows of later columns and in lower rows of earlier ones
    rnd = random.Random(9)
    s1s, s2s = [], []
    core = b"GATTACA"
    for k in range(120):
        junk = lambda q: bytes(rnd.choice(b"CT") for _ in range(q))
        x = junk(rnd.randint(0, 30)) + core + junk(rnd.randint(0, 30)) + (core if k % 2 else b"") + junk(rnd.randint(0, 20))
        y = junk(rnd.randint(0, 30)) + core + junk(rnd.randint(0, 40)) + (core if k % 3 else b"") + junk(rnd.randint(0, 20))
        s1s.append(x)
        s2s.append(y)
    s1s += [b"ACAC" * 30, b"A" * 100, b"AC" * 50, b"TTTT"]
    s2s += [b"ACAC" * 30, b"A" * 57, b"CA" * 41, b"AAAA"]
    assert _plan_kernel(s1s, s2s, A, 1, mode="local") == "bitparallel"
    _check("local", s1s, s2s)


@pytest.mark.parametrize("mode", ("fitting", "overlap"))
def test_reads_against_windows_and_overlapping_ends(mode):
    # fitting: a short read inside a longer window (the window runs along the columns, s1 = the read is fitted whole);
    # overlap: a suffix of one sequence against a prefix of the other -- end cells anywhere along the bottom row,
    # including column 0, and ties along it (the leftmost strict maximum wins, align_pair.go:169-182)
    rnd = random.Random(17)
    s1s, s2s = [], []
    for k in range(160):
        read = _rand(rnd, rnd.randint(1, 120))
        left, right = _rand(rnd, rnd.randint(0, 150)), _rand(rnd, rnd.randint(0, 150))
        if mode == "fitting":
            s1s.append(_edited(rnd, read, 0.1) or b"A")
            s2s.append(left + read + right)
        else:
            ov = _rand(rnd, rnd.randint(0, 60))
            s1s.append((left + ov)[-250:] or b"C")
            s2s.append(_edited(rnd, ov, 0.08) + right)
    s1s += [b"ACGT" * 10, b"AAAA", b"ACGT", b"G" * 40]
    s2s += [b"ACGT" * 30, b"CCCC", b"", b"G" * 90]
    assert _plan_kernel(s1s, s2s, A, 1, mode=mode) == "bitparallel"
    _check(mode, s1s, s2s)


@pytest.mark.parametrize("mode", MODES)
def test_low_complexity_ties(mode):
    # two-letter and one-letter sequences: nearly every cell is a tie, the candidate order decides the trace
    rnd = random.Random(5)
    s1s = [_rand(rnd, rnd.randint(1, 200), b"AC") for _ in range(150)] + [b"A" * k for k in range(1, 40)]
    s2s = [_rand(rnd, rnd.randint(1, 200), b"AC") for _ in range(150)] + [b"A" * (40 - k) for k in range(1, 40)]
    _check(mode, s1s, s2s)


@pytest.mark.parametrize("mode", MODES)
def test_full_bundles_equal_the_general_kernels(monkeypatch, mode):
    # several full bundles (64 pairs a wavefront, four wavefronts a block) of one shape; strings, scores, and the
    # general kernels' output on the same batch
    rnd = random.Random(3)
    s1s = [_rand(rnd, 97) for _ in range(700)]
    s2s = [_edited(rnd, x)[:110] for x in s1s]
    got = bio_amd.AlignPairBatch(mode, s1s, s2s, A, 1)
    monkeypatch.setenv("BIOALIGN_BITPARALLEL", "0")
    ref = bio_amd.AlignPairBatch(mode, s1s, s2s, A, 1)
    assert list(got[0]) == list(ref[0]) and got[1] == ref[1] and got[2] == ref[2]
    for p in range(0, 700, 37):
        assert (got[0][p], got[1][p], got[2][p]) == O.AlignPair(mode, s1s[p], s2s[p], OA, 1)


def test_single_pair_entry_point_and_scores_only():
    x, y = b"GATTACAGATTACA" * 9, b"GATACAGGATTAC" * 9
    assert bio_amd.AlignGlobal(x, y, bio_amd.MatchAligner(1, -1), 1) == O.AlignGlobal(x, y, O.MatchAligner(1, -1), 1)
    assert bio_amd.AlignLocal(x, y, A, 1) == O.AlignLocal(x, y, OA, 1)
    for mode in MODES:
        sc = bio_amd.AlignPairBatch(mode, [x, y], [y, x], A, 1, score_only=True)
        assert list(sc[0]) == [O.AlignPair(mode, x, y, OA, 1)[0], O.AlignPair(mode, y, x, OA, 1)[0]]
