"""GPU tests of the runtime around the kernels: the multi-device split of bioalign_batch (exercised with the
same GPU listed twice: two host threads, two streams, two plans), workspace chunking, the optional code paths
behind environment switches, and the limits that must be refused loudly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bio_amd
from bio_amd import _native, synth
from oracle import oracle as O


def _batch(mode, s1, o1, s2, o2, eng=None, go=1, ge=1):
    table, c1, c2 = synth.dna_table(1, -1)
    eng = eng or _native.Engine([0])
    return eng.batch(_native.PackedBatch(s1, o1, s2, o2, mode, table, c1, c2, go, ge))


def _traces(res):
    scores, t1, t2, toff, tlen = res
    return [(int(scores[p]), t1[int(toff[p]):int(toff[p] + tlen[p])].tobytes(), t2[int(toff[p]):int(toff[p] + tlen[p])].tobytes())
            for p in range(len(scores))]


def _ragged(n_pairs, seed):
    rng = np.random.default_rng(seed)
    l1 = rng.integers(0, 70, n_pairs)
    l2 = rng.integers(0, 70, n_pairs)
    o1 = np.concatenate([[0], np.cumsum(l1)]).astype(np.uint64)
    o2 = np.concatenate([[0], np.cumsum(l2)]).astype(np.uint64)
    s1 = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(o1[-1]))]
    s2 = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(o2[-1]))]
    return s1, o1, s2, o2


@pytest.mark.parametrize("mode", ["global", "local"])
def test_two_device_split_equals_one_device(mode):
    s1, o1, s2, o2 = _ragged(3000, 5)
    one = _traces(_batch(mode, s1, o1, s2, o2))
    two = _traces(_batch(mode, s1, o1, s2, o2, eng=_native.Engine([0, 0])))
    assert two == one
    oa = O.MatchAligner(1, -1)
    for p in range(0, 3000, 61):
        x, y = s1[int(o1[p]):int(o1[p + 1])].tobytes(), s2[int(o2[p]):int(o2[p + 1])].tobytes()
        sc, a, b = O.AlignPair(mode, x, y, oa, 1)
        assert one[p] == (sc, a or b"", b or b"")


def test_workspace_chunking(monkeypatch):
    s1, o1, s2, o2 = synth.dna_pairs(2000, 100, 100, config=1)
    whole_linear = _traces(_batch("fitting", s1, o1, s2, o2))
    whole_affine = _traces(_batch("global", s1, o1, s2, o2, go=4, ge=1))
    monkeypatch.setenv("BIOALIGN_WORKSPACE_BYTES", str(3 << 20))  # a few bundles per chunk
    assert _traces(_batch("fitting", s1, o1, s2, o2)) == whole_linear
    assert _traces(_batch("global", s1, o1, s2, o2, go=4, ge=1)) == whole_affine


def test_sliced_two_stream_run_matches(monkeypatch):
    # BIOALIGN_SLICES: traceback of slice k on a second stream under the DP of slice k+1
    N = 64 * 4200
    s1, o1, s2, o2 = synth.dna_pairs(N, 12, 12, config=1)
    ref = _batch("global", s1, o1, s2, o2)
    monkeypatch.setenv("BIOALIGN_SLICES", "4")
    got = _batch("global", s1, o1, s2, o2)
    assert np.array_equal(got[0], ref[0]) and _traces(got)[::997] == _traces(ref)[::997]


def test_affine_unpipelined_matches_pipelined(monkeypatch):
    s1, o1, s2, o2 = synth.fitting_reads(300, 100, 300, config=4)
    piped = _traces(_batch("fitting", s1, o1, s2, o2, go=5, ge=1))
    monkeypatch.setenv("BIOALIGN_PIPE_BELOW", "0")
    assert _traces(_batch("fitting", s1, o1, s2, o2, go=5, ge=1)) == piped


@pytest.mark.parametrize("waves,bundles", [("8", "1"), ("7", "1"), ("12", "1"), ("3", "4"), ("5", "2"), ("16", "3")])
@pytest.mark.parametrize("mode,go", [("fitting", 5), ("local", 11), ("global", 4), ("overlap", 5)])
def test_affine_pipeline_shapes(monkeypatch, waves, bundles, mode, go):
    # launch shapes of the strip pipeline that the planner does not pick by itself (kernels_affine.hip): more waves than a
    # bundle has strips (waves that own no strip of the first bundle), odd wave counts, and blocks that run several
    # bundles back to back (BIOALIGN_PIPE_BUNDLES: strip 0 of the next bundle starts under the last strips of this one,
    # local mode collects the strips' best cells through memory); ragged shapes, several bundles of each shape
    rng = np.random.default_rng(int(waves) * 31 + int(bundles))
    shapes = [(40, 90), (33, 70), (70, 33), (1, 50), (17, 0), (0, 9), (64, 64), (100, 120)]
    l1 = np.concatenate([np.full(130, a) for a, _ in shapes])
    l2 = np.concatenate([np.full(130, b) for _, b in shapes])
    o1 = np.concatenate([[0], np.cumsum(l1)]).astype(np.uint64)
    o2 = np.concatenate([[0], np.cumsum(l2)]).astype(np.uint64)
    s1 = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(o1[-1]))]
    s2 = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(o2[-1]))]
    monkeypatch.setenv("BIOALIGN_PIPE_BELOW", "0")
    ref = _traces(_batch(mode, s1, o1, s2, o2, go=go, ge=1))  # one wavefront per bundle
    monkeypatch.delenv("BIOALIGN_PIPE_BELOW")
    monkeypatch.setenv("BIOALIGN_PIPE_WAVES", waves)
    monkeypatch.setenv("BIOALIGN_PIPE_BUNDLES", bundles)
    for rows in ("16", "32"):
        monkeypatch.setenv("BIOALIGN_STRIP_ROWS", rows)
        assert _traces(_batch(mode, s1, o1, s2, o2, go=go, ge=1)) == ref, rows
    oa = O.MatchAligner(1, -1)
    for p in range(0, len(l1), 97):
        x, y = s1[int(o1[p]):int(o1[p + 1])].tobytes(), s2[int(o2[p]):int(o2[p + 1])].tobytes()
        want = O.AlignAffine(mode, x, y, oa, go, 1)
        assert (ref[p][0], ref[p][1] or None, ref[p][2] or None) == want, p


def test_context_may_be_destroyed_before_its_plans():
    # plans, stripes and parsed FASTA texts hold a reference on their context (include/bioalign.h): any destroy order
    s1, o1, s2, o2 = synth.dna_pairs(64, 30, 30, config=1)
    table, c1, c2 = synth.dna_table(1, -1)
    L = _native.lib()
    eng = _native.Engine([0])
    plan = eng.plan(_native.PackedBatch(s1, o1, s2, o2, "global", table, c1, c2, 1, 1))
    fa = _native.Fasta(eng, b">a\nACGT\n>b\nGG\n")
    h, eng._h = eng._h, None  # (the Python wrapper would close its children first: go through the C ABI)
    L.bioalign_destroy(h)
    assert int(L.bioalign_fasta_count(fa._h)) == 2
    assert fa.seq_bytes().tobytes() == b"ACGTGG"  # (a call that reaches into the context: devices, error string)
    fa.close()
    plan.close()


class _Big:
    """scores far outside the int8 profile of the fast kernels"""

    def __init__(self, match, mismatch):
        self.match, self.mismatch = match, mismatch

    def Score(self, x, y):
        return self.match if x == y else self.mismatch


def test_limits_are_refused_not_computed_elsewhere():
    # what the engine does NOT compute says so: negative penalties, open < extend (the reference's recurrences take them;
    # the kernels' border and tie arguments do not), scores beyond 32 bits
    a = bio_amd.MatchAligner(1, -1)
    for fn in (lambda: bio_amd.AlignPairBatch("global", [b"ACGT"], [b"ACG"], a, -1),
               lambda: bio_amd.AlignAffineBatch("global", [b"ACGT"], [b"ACG"], a, 1, 3),
               lambda: bio_amd.AlignPairBatch("global", [b"ACGT" * 50], [b"ACG" * 50], _Big(2_000_000, -2_000_000), 1)):
        with pytest.raises(_native.BioAlignError) as e:
            fn()
        assert e.value.code == _native.EUNSUPPORTED


def _wide_pairs(seed, n_pairs=90, lo=0, hi=60):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_pairs):
        n, m = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
        x = bytes(b"ACGT"[k] for k in rng.integers(0, 4, n))
        y = bytearray(x[:m].ljust(m, b"A")) if rng.random() < 0.6 else bytearray(bytes(b"ACGT"[k] for k in rng.integers(0, 4, m)))
        for k in range(len(y)):  # a mutated copy: long diagonal runs, ties, gaps
            if rng.random() < 0.15:
                y[k] = b"ACGT"[int(rng.integers(0, 4))]
        out.append((x, bytes(y)))
    return out


@pytest.mark.parametrize("mode", ["global", "local", "fitting", "overlap"])
def test_wide_kernels_scores_outside_the_packed_encodings(mode):
    # inputs the reference computes (align.go:76 takes any int, align_affine.go:12 any float64) and the fast kernels'
    # encodings cannot hold -- +-300 scores, a gap of 200, gap open 150, affine scores beyond 2^22 -- run on the wide
    # kernels (kernels_wide.hip): scores and both traces equal the oracle's
    pairs = _wide_pairs(11)
    xs, ys = [p[0] for p in pairs], [p[1] for p in pairs]
    for (ma, mi, go, ge) in [(300, -300, 1, 1), (1, -1, 200, 200), (300, -250, 400, 400), (5, -4, 150, 2), (300, -300, 500, 7),
                             (90_000, -70_000, 100_000, 30_000)]:
        a, oa = _Big(ma, mi), O.MatchAligner(ma, mi)
        if go == ge:
            sc, t1, t2 = bio_amd.AlignPairBatch(mode, xs, ys, a, go)
            want = [O.AlignPair(mode, x, y, oa, go) for x, y in pairs]
        else:
            sc, t1, t2 = bio_amd.AlignAffineBatch(mode, xs, ys, a, go, ge)
            want = [O.AlignAffine(mode, x, y, oa, go, ge) for x, y in pairs]
        for p in range(len(pairs)):
            assert (sc[p], t1[p], t2[p]) == want[p], (mode, ma, mi, go, ge, p)


@pytest.mark.parametrize("mode", ["global", "local", "fitting", "overlap"])
def test_wide_kernels_equal_the_fast_kernels(mode, monkeypatch):
    # BIOALIGN_WIDE=1 sends ordinary inputs through the wide kernels: same answers as the fast ones, linear and affine
    s1, o1, s2, o2 = _ragged(500, 77)
    for go, ge in ((1, 1), (3, 3), (5, 1), (11, 1)):
        fast = _traces(_batch(mode, s1, o1, s2, o2, go=go, ge=ge))
        monkeypatch.setenv("BIOALIGN_WIDE", "1")
        wide = _traces(_batch(mode, s1, o1, s2, o2, go=go, ge=ge))
        monkeypatch.delenv("BIOALIGN_WIDE")
        assert wide == fast, (mode, go, ge)


def test_share_pack_between_modes():
    # bioalign_plan_share_pack: the second mode's plan reads the packed sequences the first one made
    import torch

    s1, o1, s2, o2 = _ragged(2500, 21)
    table, c1, c2 = synth.dna_table(1, -1)
    eng = _native.Engine([0])
    pbs = {m: _native.PackedBatch(s1, o1, s2, o2, m, table, c1, c2, 1, 1) for m in ("global", "local", "overlap")}
    plans = {m: eng.plan(pb) for m, pb in pbs.items()}
    plans["local"].share_pack(plans["global"])
    plans["overlap"].share_pack(plans["global"])
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to(dev) for a in (s1, o1, s2, o2)]
    N, cap = len(o1) - 1, pbs["global"].trace_capacity
    for m in ("global", "local", "overlap"):  # the packing plan first, on one stream
        sc = torch.empty(N, dtype=torch.int64, device=dev)
        t1 = torch.zeros(cap + 16, dtype=torch.uint8, device=dev)
        t2 = torch.zeros(cap + 16, dtype=torch.uint8, device=dev)
        toff = torch.empty(N, dtype=torch.int64, device=dev)
        tlen = torch.empty(N, dtype=torch.int64, device=dev)
        plans[m].run(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), sc.data_ptr(), t1.data_ptr(),
                     t2.data_ptr(), toff.data_ptr(), tlen.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = _traces((sc.cpu().numpy(), t1.cpu().numpy(), t2.cpu().numpy(), toff.cpu().numpy(), tlen.cpu().numpy()))
        assert got == _traces(_batch(m, s1, o1, s2, o2, eng=eng)), m
    # a plan of another batch is refused
    s1b, o1b, s2b, o2b = _ragged(2500, 22)
    other = eng.plan(_native.PackedBatch(s1b, o1b, s2b, o2b, "local", table, c1, c2, 1, 1))
    with pytest.raises(Exception, match="same batch|differently"):
        other.share_pack(plans["global"])
    with pytest.raises(Exception, match="does not pack"):
        other.share_pack(plans["local"])


def test_dense_trace_layout():
    # BIOALIGN_DENSE_TRACES: the traces lie back to back in pair order; same traces as the slot layout.
    # bioalign_batch asks for it by itself: its t_off must be the running sum of its t_len.
    import torch

    s1, o1, s2, o2 = _ragged(3000, 31)
    table, c1, c2 = synth.dna_table(1, -1)
    eng = _native.Engine([0])
    for mode in ("global", "local"):
        res = _batch(mode, s1, o1, s2, o2, eng=eng)
        sc, t1, t2, toff, tlen = res
        assert np.array_equal(toff, np.concatenate([[0], np.cumsum(tlen)[:-1]]).astype(np.uint64)), mode
        dev = torch.device("cuda:0")
        d = [torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).to(dev) for a in (s1, o1, s2, o2)]
        N = len(o1) - 1
        got = {}
        for flags in (0, _native.DENSE_TRACES):
            pb = _native.PackedBatch(s1, o1, s2, o2, mode, table, c1, c2, 1, 1, flags=flags)
            plan = eng.plan(pb)
            cap = pb.trace_capacity
            o = dict(sc=torch.empty(N, dtype=torch.int64, device=dev), t1=torch.zeros(cap + 16, dtype=torch.uint8, device=dev),
                     t2=torch.zeros(cap + 16, dtype=torch.uint8, device=dev), toff=torch.empty(N, dtype=torch.int64, device=dev),
                     tlen=torch.empty(N, dtype=torch.int64, device=dev))
            plan.run(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), o["sc"].data_ptr(), o["t1"].data_ptr(),
                     o["t2"].data_ptr(), o["toff"].data_ptr(), o["tlen"].data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got[flags] = _traces(tuple(o[k].cpu().numpy() for k in ("sc", "t1", "t2", "toff", "tlen")))
            if flags:
                tl = o["tlen"].cpu().numpy()
                assert np.array_equal(o["toff"].cpu().numpy(), np.concatenate([[0], np.cumsum(tl)[:-1]]))
        assert got[0] == got[_native.DENSE_TRACES] == _traces(res), mode


@pytest.mark.parametrize("mode", ["global", "local", "fitting", "overlap"])
def test_compact_trace_records(mode):
    # BIOALIGN_TRACE_OPS: end cell + 2-bit moves per pair; expanded on the host they are the
    # reference's two strings.  Ragged batch with empty sequences; the records lie back to back in pair order.
    s1, o1, s2, o2 = _ragged(2500, 41)
    table, c1, c2 = synth.dna_table(1, -1)
    eng = _native.Engine([0])
    ref = _traces(_batch(mode, s1, o1, s2, o2, eng=eng))
    pb = _native.PackedBatch(s1, o1, s2, o2, mode, table, c1, c2, 1, 1, flags=_native.TRACE_OPS)
    sc, t1, _, toff, tlen = eng.batch(pb)
    assert np.all(toff % 4 == 0)
    sizes = 8 + 4 * ((tlen + 15) // 16)
    assert np.array_equal(toff, np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64))
    for p in range(2500):
        x, y = s1[int(o1[p]):int(o1[p + 1])].tobytes(), s2[int(o2[p]):int(o2[p + 1])].tobytes()
        a, b = _native.expand_ops(x, y, t1, int(toff[p]), int(tlen[p]))
        assert (int(sc[p]), a, b) == ref[p], (mode, p)
    # affine gaps have no move records
    with pytest.raises(_native.BioAlignError):
        eng.batch(_native.PackedBatch(s1, o1, s2, o2, mode, table, c1, c2, 3, 1, flags=_native.TRACE_OPS))


@pytest.mark.parametrize("mode", ["global", "local", "fitting", "overlap"])
@pytest.mark.parametrize("affine", [False, True])
def test_end_cells_without_traceback(mode, affine):
    """BIOALIGN_SCORE_ONLY | BIOALIGN_END_CELLS: t_off / t_len = row / column of the cell the alignment ends in.
    Checked against the traces of the ordinary run (themselves bit-exact against the oracle elsewhere): the
    symbols of each trace are the stretch of its sequence that ends there; linear gaps: also the record headers
    of BIOALIGN_TRACE_OPS."""
    s1, o1, s2, o2 = _ragged(700, 31 + int(affine))
    table, c1, c2 = synth.dna_table(1, -1)
    go, ge = (3, 1) if affine else (1, 1)
    eng = _native.Engine([0])
    full = _traces(eng.batch(_native.PackedBatch(s1, o1, s2, o2, mode, table, c1, c2, go, ge)))
    sc, _, _, ei, ej = eng.batch(_native.PackedBatch(s1, o1, s2, o2, mode, table, c1, c2, go, ge,
                                                      flags=_native.SCORE_ONLY | _native.END_CELLS))
    for p, (score, t1, t2) in enumerate(full):
        x = s1[int(o1[p]):int(o1[p + 1])].tobytes()
        y = s2[int(o2[p]):int(o2[p + 1])].tobytes()
        i, j = int(ei[p]), int(ej[p])
        a, b = t1.replace(b"-", b""), t2.replace(b"-", b"")
        assert int(sc[p]) == score
        assert 0 <= i <= len(x) and 0 <= j <= len(y), (p, i, j)
        assert x[i - len(a):i] == a and y[j - len(b):j] == b, (mode, p, i, j, x, y, t1, t2)
        if mode == "global":
            assert (i, j) == (len(x), len(y))
        if mode in ("fitting", "overlap"):
            assert i == len(x)
    if not affine:
        _, r1, _, roff, rlen = eng.batch(_native.PackedBatch(s1, o1, s2, o2, mode, table, c1, c2, go, ge, flags=_native.TRACE_OPS))
        words = r1[:len(r1) // 4 * 4].view(np.uint32)
        idx = (roff // 4).astype(np.int64)
        some = rlen > 0  # (a record without moves carries (0, 0))
        assert some.any()
        assert np.array_equal(words[idx].astype(np.uint64)[some], ei[some]) and np.array_equal(words[idx + 1].astype(np.uint64)[some], ej[some])
    # the flag needs its two arrays
    with pytest.raises(_native.BioAlignError):
        pb = _native.PackedBatch(s1, o1, s2, o2, mode, table, c1, c2, go, ge, flags=_native.SCORE_ONLY | _native.END_CELLS)
        r = _native.Result()
        scores = np.zeros(700, dtype=np.int64)
        r.scores = scores.ctypes.data
        rc = _native.lib().bioalign_batch(eng._h, _native.C.byref(pb.c), _native.C.byref(r))
        if rc != 0:
            eng._raise(rc)
