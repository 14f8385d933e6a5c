"""GPU parity, long mode: one pair aligned by the intra-sequence wavefront
kernels (bioalign_long), four modes, against the oracle, bit for bit on score and both traces.  Sizes are chosen
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


MODES4 = ["global", "local", "fitting", "overlap"]


@pytest.mark.parametrize("mode", MODES4)
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
        for mode in MODES4:
            assert bio_amd.AlignPairLong(mode, x, y, a, g) == O.AlignPair(mode, x, y, oa, g), (scoring, mode, n, m)


def test_long_low_complexity_ties():
    rng = random.Random(9)
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    for _ in range(6):
        x = bytes(rng.choice(b"AAAC") for _ in range(rng.randrange(400, 1300)))
        y = bytes(rng.choice(b"AAAC") for _ in range(rng.randrange(400, 1300)))
        for mode in MODES4:
            assert bio_amd.AlignPairLong(mode, x, y, a, 1) == O.AlignPair(mode, x, y, oa, 1)


def test_long_local_end_cells_and_empty_alignments():
    # align.go:152-249 on the long kernels: first maximum in row-major order, walk back to the first cell that scores 0
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    rng = random.Random(31)
    core = bytes(rng.choice(b"ACGT") for _ in range(200))
    junk = lambda k, al=b"ACGT": bytes(rng.choice(al) for _ in range(k))
    cases = [
        (b"A" * 300, b"C" * 500),                                  # nothing matches: score 0, empty traces
        (core, core),                                              # the whole diagonal
        (junk(400) + core + junk(300), junk(150) + core + junk(700)),  # a core inside junk
        (core + junk(500) + core, junk(300) + core + junk(100)),   # the maximum occurs twice: the first row wins
        (junk(300) + core, core + junk(300) + core),               # ... and twice in one row: the first column wins
        (b"ACGT" * 200, b"ACGT" * 150),                            # periodic: many equal maxima
        (junk(1500, b"AC"), junk(1300, b"AC")),
    ]
    for x, y in cases:
        assert bio_amd.AlignPairLong("local", x, y, a, 1) == O.AlignPair("local", x, y, oa, 1), (len(x), len(y))
    # the maximum in the last row / last column / first row, and a start at the matrix's corner
    for x, y in [(junk(700) + core, junk(90) + core), (core + junk(5), core), (b"G", junk(300) + b"G"), (core, core + junk(900))]:
        assert bio_amd.AlignPairLong("local", x, y, a, 1) == O.AlignPair("local", x, y, oa, 1), (len(x), len(y))
    # AlignLocal itself through the long kernels' size threshold is not routed: AlignPairLong is the entry point
    sc = bio_amd.AlignPairBatch("local", [cases[2][0]], [cases[2][1]], a, 1)
    assert (sc[0][0], sc[1][0], sc[2][0]) == bio_amd.AlignPairLong("local", cases[2][0], cases[2][1], a, 1)


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


def test_single_pair_entry_points_route_long_pairs():
    # AlignGlobal / AlignLocal / AlignPair of 2^24 cells or more run on the long kernels (same results); a pair they
    # refuse for its parameters stays on the batch kernels
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    x, y = _pair(4200, 4100, 91)
    assert bio_amd.align._long_pair(x, y) == "auto"
    assert bio_amd.AlignGlobal(x, y, a, 1) == O.AlignGlobal(x, y, oa, 1)
    assert bio_amd.AlignLocal(x, y, a, 1) == O.AlignLocal(x, y, oa, 1)
    assert bio_amd.AlignPair("fitting", x, y, a, 1) == O.AlignPair("fitting", x, y, oa, 1)
    # the long kernels refuse these scores and the call falls back to the batch entry point, whose wide (int32) kernels
    # hold them up to 2^26 cells a pair; beyond that the refusal is the error that reaches the caller
    big, obig = bio_amd.MatchAligner(100, -100), O.MatchAligner(100, -100)
    assert bio_amd.AlignGlobal(x, y, big, 20) == O.AlignGlobal(x, y, obig, 20)
    x, y = _pair(8300, 8200, 92)
    with pytest.raises(bio_amd._native.BioAlignError) as e:
        bio_amd.AlignGlobal(x, y, big, 20)
    assert e.value.code == bio_amd._native.EUNSUPPORTED


def test_long_unsupported_fails_loudly():
    # what the long kernels do not cover is refused, not computed some other way: more than 64 column symbols, and
    # gap penalties with open < extend
    x = bytes(range(40, 160)) * 3
    with pytest.raises(bio_amd._native.BioAlignError) as e:
        bio_amd.AlignPairLong("global", x, x[::-1], bio_amd.MatchAligner(1, -1), 1)  # 120 column symbols (linear: <= 100)
    assert e.value.code == bio_amd._native.EUNSUPPORTED
    with pytest.raises(bio_amd._native.BioAlignError) as e:
        bio_amd.AlignAffinePairLong("global", x[:240], x[:240][::-1], bio_amd.MatchAligner(1, -1), 3, 1)  # 80 (affine: <= 64)
    assert e.value.code == bio_amd._native.EUNSUPPORTED
    with pytest.raises(bio_amd._native.BioAlignError) as e:
        bio_amd.AlignAffinePairLong("global", b"ACGT" * 10, b"ACG" * 10, bio_amd.MatchAligner(1, -1), 1, 3)
    assert e.value.code == bio_amd._native.EUNSUPPORTED


@pytest.mark.parametrize("mode", MODES4)
def test_long_hundred_column_symbols(mode):
    # the widest alphabet the linear long kernels take: four strip profiles of 25 KB each in LDS
    rng = random.Random(61)
    x = bytes(rng.randrange(46, 146) for _ in range(1500))
    y = bytes(x[k] if rng.random() < 0.8 else rng.randrange(46, 146) for k in range(1400)) + bytes(range(46, 146))
    a, oa = bio_amd.MatchAligner(2, -1), O.MatchAligner(2, -1)
    assert bio_amd.AlignPairLong(mode, x, y, a, 2) == O.AlignPair(mode, x, y, oa, 2)


def _affine_same(mode, x, y, a, oa, go, ge):
    got = bio_amd.AlignAffinePairLong(mode, x, y, a, go, ge)
    want = O.AlignAffine(mode, x, y, oa, go, ge)
    assert got == want, (mode, len(x), len(y), go, ge, got[0], want[0])


@pytest.mark.parametrize("mode", MODES4)
def test_long_affine_small_shapes(mode):
    # Gotoh's three levels (align_affine.go:44-103) on the long-mode mapping, against the oracle's bo_align_affine
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    shapes = [(1, 1), (5, 7), (7, 5), (63, 64), (64, 63), (65, 65), (100, 127), (128, 128), (130, 129), (255, 70),
              (256, 200), (257, 190), (700, 900), (1100, 300), (30, 1000), (600, 3)]
    for k, (n, m) in enumerate(shapes):
        x, y = _pair(n, m, 900 + k)
        _affine_same(mode, x, y, a, oa, 5, 1)


@pytest.mark.parametrize("gaps", [(5, 1), (3, 0), (2, 2 - 1), (11, 1), (4, 3)])
def test_long_affine_penalties_and_gap_heavy_pairs(gaps):
    go, ge = gaps
    a, oa = bio_amd.MatchAligner(2, -3), O.MatchAligner(2, -3)
    rng = random.Random(go * 10 + ge)
    for _ in range(3):
        # long insertions and deletions: gap runs that cross lanes and strips
        core = bytes(rng.choice(b"ACGT") for _ in range(700))
        x = core[:200] + bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(1, 300))) + core[200:]
        y = core[:450] + core[450 + rng.randrange(1, 200):]
        for mode in MODES4:
            _affine_same(mode, x, y, a, oa, go, ge)
            _affine_same(mode, y, x, a, oa, go, ge)


def test_long_affine_protein_and_low_complexity():
    s1, o1, s2, o2 = synth.protein_pairs(1, 1500, 1700, config=3, first=11)
    x, y = s1.tobytes(), s2.tobytes()
    for mode in MODES4:
        _affine_same(mode, x, y, bio_amd.Blosum62, O.Blosum62(), 11, 1)
    rng = random.Random(17)
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    for _ in range(3):
        x = bytes(rng.choice(b"AAAC") for _ in range(rng.randrange(300, 900)))
        y = bytes(rng.choice(b"AAAC") for _ in range(rng.randrange(300, 900)))
        for mode in MODES4:
            _affine_same(mode, x, y, a, oa, 3, 1)
    # nothing matches: local scores 0 with empty traces, global is one long gap pair
    for mode in MODES4:
        _affine_same(mode, b"A" * 300, b"C" * 200, a, oa, 4, 1)


def test_long_affine_3000_blosum62_and_align_global_affine():
    s1, o1, s2, o2 = synth.protein_pairs(1, 3100, 3300, config=3, first=7)
    x, y = s1.tobytes(), s2.tobytes()
    for mode in ("global", "local"):
        _affine_same(mode, x, y, bio_amd.Blosum62, O.Blosum62(), 11, 1)
    # AlignGlobalAffine (align_affine.go:12) routes pairs of >= 2^24 cells to the long kernels: score as written
    x, y = _pair(4200, 4100, 77)
    sc, t1, t2 = bio_amd.AlignGlobalAffine(x, y, bio_amd.MatchAligner(1, -1), 5.0, 1.0)
    assert sc == O.AffineGlobalScoreF64(x, y, O.MatchAligner(1, -1), 5.0, 1.0)
    assert (int(sc), t1, t2) == O.AlignAffine("global", x, y, O.MatchAligner(1, -1), 5, 1)


@pytest.mark.parametrize("mode", MODES4)
def test_long_protein_alphabets(mode):
    # any alphabet up to 64 column symbols: the strip profile of a lane lies in LDS (20 symbols, 8 residue bits here)
    s1, o1, s2, o2 = synth.protein_pairs(1, 3100, 3300, config=3, first=7)
    x, y = s1.tobytes(), s2.tobytes()
    assert bio_amd.AlignPairLong(mode, x, y, bio_amd.Blosum62, 5) == O.AlignPair(mode, x, y, O.Blosum62(), 5)
    rng = random.Random(21)
    alpha = b"ARNDCQEGHILKMFPSTWYV"
    for n, m in [(1, 1), (70, 300), (515, 130), (260, 777)]:
        x = bytes(rng.choice(alpha) for _ in range(n))
        y = bytes(rng.choice(alpha) for _ in range(m))
        assert bio_amd.AlignPairLong(mode, x, y, bio_amd.Pam250, 7) == O.AlignPair(mode, x, y, O.Pam250(), 7), (mode, n, m)


@pytest.mark.parametrize("merged", ["1", "0"])
@pytest.mark.parametrize("stripes", [2, 3, 5])
@pytest.mark.parametrize("mode", ["global", "fitting", "overlap"])
def test_long_column_stripes(mode, stripes, merged, monkeypatch):
    # the matrix cut into column stripes, every stripe with its own residues, the last column of a
    # stripe handed to its right neighbour, the walk handed on at every stripe's column 0.  One GPU here, so all
    # stripes sit on the same device: merged = one launch over all stripes, the column handed over strip by strip on the
    # device; not merged = the host-driven row blocks that several devices use (copies become plain device copies).
    monkeypatch.setenv("BIOALIGN_LONG_STRIPES", str(stripes))
    monkeypatch.setenv("BIOALIGN_STRIPE_MERGED", merged)
    monkeypatch.setenv("BIOALIGN_STRIPE_BLOCKS", "3")
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    shapes = [(5, 7), (70, 64), (130, 129), (513, 190), (700, 900), (1100, 300), (30, 1000), (600, 3)]
    for k, (n, m) in enumerate(shapes):
        x, y = _pair(n, m, 700 + k)
        assert bio_amd.AlignPairLong(mode, x, y, a, 1) == O.AlignPair(mode, x, y, oa, 1), (mode, stripes, n, m)


def test_long_column_stripes_paths_along_the_borders(monkeypatch):
    # walks that run along the top row / the left column across several stripes, and end cells in every stripe
    monkeypatch.setenv("BIOALIGN_LONG_STRIPES", "4")
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    rng = random.Random(3)
    core = bytes(rng.choice(b"ACGT") for _ in range(150))
    junk = lambda k: bytes(rng.choice(b"ACGT") for _ in range(k))
    cases = [(core, junk(300) + core), (core, core + junk(300)), (core, junk(150) + core + junk(150)),
             (junk(200) + core, core + junk(250)), (core + junk(200), junk(250) + core), (b"A" * 90, b"C" * 500),
             (b"ACGT" * 30, b"ACGT" * 200)]
    for x, y in cases:
        for mode in ("global", "fitting", "overlap"):
            assert bio_amd.AlignPairLong(mode, x, y, a, 1) == O.AlignPair(mode, x, y, oa, 1), (mode, len(x), len(y))
    # other residue widths
    a, oa = bio_amd.MatchAligner(5, -4), O.MatchAligner(5, -4)
    x, y = _pair(900, 1300, 77)
    for mode in ("global", "fitting", "overlap"):
        assert bio_amd.AlignPairLong(mode, x, y, a, 3) == O.AlignPair(mode, x, y, oa, 3)


def test_long_column_stripes_20k_merged_and_row_blocks(monkeypatch):
    # many workgroups per stripe: the start-time order of the merged launch and the row blocks of the host-driven one
    from bio_amd import _native, synth
    s1, o1, s2, o2 = synth.dna_pairs(1, 20_000, 20_000, config=5, sub=0.06, ins=0.02, dele=0.02)
    table, c1, c2 = synth.dna_table(1, -1)
    col, _ = O.lastColumnScores(s1.tobytes(), s2.tobytes(), O.MatchAligner(1, -1), -1)
    eng = bio_amd.align.engine()
    monkeypatch.setenv("BIOALIGN_LONG_STRIPES", "8")
    ref = None
    for merged in ("1", "0"):
        monkeypatch.setenv("BIOALIGN_STRIPE_MERGED", merged)
        sc, t1, t2, toff, tlen = eng.long(_native.PackedBatch(s1, o1, s2, o2, "global", table, c1, c2, 1, 1))
        a, l = int(toff[0]), int(tlen[0])
        got = (int(sc[0]), t1[a:a + l].tobytes(), t2[a:a + l].tobytes())
        assert got[0] == int(col[-1])
        assert got[1].replace(b"-", b"") == s1.tobytes() and got[2].replace(b"-", b"") == s2.tobytes()
        ref = ref or got
        assert got == ref
    monkeypatch.delenv("BIOALIGN_LONG_STRIPES")
    sc, t1, t2, toff, tlen = eng.long(_native.PackedBatch(s1, o1, s2, o2, "global", table, c1, c2, 1, 1))
    a, l = int(toff[0]), int(tlen[0])
    assert (int(sc[0]), t1[a:a + l].tobytes(), t2[a:a + l].tobytes()) == ref  # the unstriped run gives the same traces


@pytest.mark.parametrize("budget", ["1", "3000000"])
def test_long_linear_space_traceback(budget, monkeypatch):
    # north_star's Hirschberg item (align_linear.go:9-145 is the reference's own linear-space function): when the residue
    # words do not fit the budget, the DP runs once without them and row blocks are recomputed bottom-up from the
    # checkpointed rows and walked one after the other -- the same traces as with the whole matrix in memory
    monkeypatch.setenv("BIOALIGN_LONG_RESIDUE_BYTES", budget)  # 1: one workgroup (1024 rows) per block
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    shapes = [(5, 7), (1024, 300), (1025, 1030), (3000, 2500), (2100, 4000), (4100, 700), (2500, 60)]
    for k, (n, m) in enumerate(shapes):
        x, y = _pair(n, m, 1200 + k)
        for mode in ("global", "fitting", "overlap"):
            assert bio_amd.AlignPairLong(mode, x, y, a, 1) == O.AlignPair(mode, x, y, oa, 1), (mode, n, m)
    # walks that run along the borders and leave blocks through column 0
    rng = random.Random(5)
    core = bytes(rng.choice(b"ACGT") for _ in range(300))
    junk = lambda k: bytes(rng.choice(b"ACGT") for _ in range(k))
    for x, y in [(junk(2500) + core, core + junk(400)), (core + junk(2600), junk(500) + core), (junk(1500) + core + junk(1500), core)]:
        for mode in ("global", "fitting", "overlap"):
            assert bio_amd.AlignPairLong(mode, x, y, a, 1) == O.AlignPair(mode, x, y, oa, 1), (mode, len(x), len(y))
    # an end cell ON the left border (fitting with a free gap: the bottom row's leftmost maximum is column 0): the trace is
    # the border's own moves (found by tools/fuzz_parity.py --long)
    z, oz = bio_amd.MatchAligner(0, -2), O.MatchAligner(0, -2)
    x, y = junk(780), junk(1000)
    for mode in ("global", "fitting", "overlap"):
        assert bio_amd.AlignPairLong(mode, x, y, z, 0) == O.AlignPair(mode, x, y, oz, 0), mode


@pytest.mark.parametrize("budget", ["1", "3000000"])
def test_long_linear_space_traceback_local(budget, monkeypatch):
    # local mode (align.go:152-249 on long inputs): the first pass leaves the maximum and the group of rows that holds it,
    # the block that ends there is recomputed over all columns to find the end cell (first maximum in row-major order),
    # and the walk carries its exact score upwards from block to block until it reads 0
    monkeypatch.setenv("BIOALIGN_LONG_RESIDUE_BYTES", budget)
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    rng = random.Random(11)
    junk = lambda k: bytes(rng.choice(b"ACGT") for _ in range(k))
    core = junk(2600)  # an alignment that crosses two block boundaries of 1024 rows
    mut = bytearray(core)
    for k in range(0, len(mut), 37):
        mut[k] = rng.choice(b"ACGT")
    del mut[700:709]
    mut[1500:1500] = junk(6)
    cases = [(junk(900) + core + junk(1300), junk(300) + bytes(mut) + junk(200)),  # ends mid-matrix, starts two blocks up
             (core + junk(40), bytes(mut)),                                       # starts in row 1 / column 1
             (junk(3000), junk(2500)),                                             # a short alignment somewhere
             (junk(1024) + core[:64], core[:64] + junk(500)),                      # the end cell in the first rows of a block
             (b"A" * 1500, b"C" * 2000),                                          # score 0: no alignment
             (core[:1100], core[:1100])]                                           # the whole diagonal, every block boundary crossed
    for x, y in cases:
        assert bio_amd.AlignPairLong("local", x, y, a, 1) == O.AlignPair("local", x, y, oa, 1), (len(x), len(y))
    # ties between end cells in different blocks: the first one in row-major order (BLOSUM62 as well: four-bit residues)
    rep = junk(150)
    x, y = junk(500) + rep + junk(1800) + rep + junk(300), junk(200) + rep + junk(100)
    assert bio_amd.AlignPairLong("local", x, y, a, 1) == O.AlignPair("local", x, y, oa, 1)
    prng = random.Random(3)
    px = bytes(prng.choice(b"ARNDCQEGHILKMFPSTWYV") for _ in range(2300))
    py = px[400:1900] + bytes(prng.choice(b"ARNDCQEGHILKMFPSTWYV") for _ in range(300))
    assert bio_amd.AlignPairLong("local", px, py, bio_amd.Blosum62, 5) == O.AlignPair("local", px, py, O.Blosum62(), 5)


@pytest.mark.parametrize("budget", ["1", "3000000"])
def test_long_linear_space_traceback_affine(budget, monkeypatch):
    # Gotoh's three levels (align_affine.go:12-167) past the memory budget: the blocks hold decision words, the walk is
    # handed from block to block as (level, cell) -- a gap that crosses a block boundary goes on on its level -- and in
    # local mode with its score (parity with the oracle's restatement; the reference holds no affine trace fixture)
    monkeypatch.setenv("BIOALIGN_LONG_RESIDUE_BYTES", budget)  # 1: one workgroup (512 rows at two rows per lane) per block
    a, oa = bio_amd.MatchAligner(1, -1), O.MatchAligner(1, -1)
    rng = random.Random(21)
    junk = lambda k: bytes(rng.choice(b"ACGT") for _ in range(k))
    core = junk(1700)
    gapped = core[:480] + core[560:1000] + junk(70) + core[1000:]  # a gap in each sequence, the first across row 512
    cases = [(core, gapped), (junk(300) + core + junk(200), junk(100) + gapped + junk(350)), (junk(1300), junk(900)),
             (junk(40), junk(1500)), (junk(1500), junk(40)), (core[:1030], core[:1030])]
    for x, y in cases:
        for mode in MODES4:
            for go, ge in ((3, 1), (5, 0)):
                assert bio_amd.AlignAffinePairLong(mode, x, y, a, go, ge) == O.AlignAffine(mode, x, y, oa, go, ge), (mode, go, ge, len(x), len(y))
    prng = random.Random(4)
    px = bytes(prng.choice(b"ARNDCQEGHILKMFPSTWYV") for _ in range(1500))
    py = px[200:700] + px[760:1400] + bytes(prng.choice(b"ARNDCQEGHILKMFPSTWYV") for _ in range(200))
    for mode in MODES4:
        assert bio_amd.AlignAffinePairLong(mode, px, py, bio_amd.Blosum62, 11, 1) == O.AlignAffine(mode, px, py, O.Blosum62(), 11, 1), mode


def test_long_score_only_stores_no_residues(monkeypatch):
    # score-only pairs store no residue / decision words in ANY mode, linear or affine (local mode takes its maximum from
    # the lanes' running maxima): with a one-byte budget for the traceback matrix they still run, and give the oracle's score
    from bio_amd import _native
    monkeypatch.setenv("BIOALIGN_LONG_RESIDUE_BYTES", "1")
    table, c1, c2 = synth.dna_table(1, -1)
    eng = bio_amd.align.engine()
    oa = O.MatchAligner(1, -1)
    for k, (n, m) in enumerate([(700, 900), (2300, 1500), (300, 3000)]):
        x, y = _pair(n, m, 1700 + k)
        s1, s2 = np.frombuffer(x, np.uint8), np.frombuffer(y, np.uint8)
        o1, o2 = np.array([0, n], np.uint64), np.array([0, m], np.uint64)
        for mode in MODES4:
            for go, ge in ((1, 1), (5, 1)):
                pb = _native.PackedBatch(s1, o1, s2, o2, mode, table, c1, c2, go, ge, flags=_native.SCORE_ONLY)
                sc = eng.long(pb)[0]
                want = O.AlignAffine(mode, x, y, oa, go, ge)[0] if go != ge else O.AlignPair(mode, x, y, oa, go)[0]
                assert int(sc[0]) == want, (mode, go, ge, n, m)


def test_long_linear_space_traceback_20k(monkeypatch):
    from bio_amd import _native
    s1, o1, s2, o2 = synth.dna_pairs(1, 20_000, 20_000, config=5, sub=0.06, ins=0.02, dele=0.02)
    table, c1, c2 = synth.dna_table(1, -1)
    eng = bio_amd.align.engine()
    pb = _native.PackedBatch(s1, o1, s2, o2, "global", table, c1, c2, 1, 1)
    sc, t1, t2, toff, tlen = eng.long(pb)
    a, l = int(toff[0]), int(tlen[0])
    ref = (int(sc[0]), t1[a:a + l].tobytes(), t2[a:a + l].tobytes())
    monkeypatch.setenv("BIOALIGN_LONG_RESIDUE_BYTES", str(30_000_000))  # the whole matrix needs 100 MB: four blocks
    sc, t1, t2, toff, tlen = eng.long(pb)
    a, l = int(toff[0]), int(tlen[0])
    assert (int(sc[0]), t1[a:a + l].tobytes(), t2[a:a + l].tobytes()) == ref


def test_long_column_stripes_two_devices(monkeypatch):
    # the same GPU listed twice: stripes alternate between two device entries of the context
    from bio_amd import _native, pack_sequences, score_table
    monkeypatch.setenv("BIOALIGN_LONG_STRIPES", "4")
    x, y = _pair(1500, 2100, 88)
    eng = _native.Engine([0, 0])
    d1, o1 = pack_sequences([x]); d2, o2 = pack_sequences([y])
    tbl, c1, c2 = score_table(bio_amd.MatchAligner(1, -1), np.unique(d1).tolist(), np.unique(d2).tolist())
    pb = _native.PackedBatch(d1, o1, d2, o2, "global", tbl, c1, c2, gap_open=1, gap_extend=1)
    scores, t1, t2, toff, tlen = eng.batch(pb, long=True)
    a, b = int(toff[0]), int(toff[0] + tlen[0])
    assert (int(scores[0]), t1[a:b].tobytes(), t2[a:b].tobytes()) == O.AlignPair("global", x, y, O.MatchAligner(1, -1), 1)
