"""CPU-side tests (no GPU, no compute calls into the HIP library):
  * the C-ABI library loads and exports every symbol include/bioalign.h declares;
  * it fails loudly without a GPU (no CPU fallback);
  * host logic of the mirror: SubstMatrix quirks, score-table materialisation, rescoring,
    packing, synthetic generators, error behaviour that never reaches the device.
Reference citations are file:line of the reference tree.
"""
import ctypes
import os
import re

import numpy as np
import pytest

import bio_amd
from bio_amd import _native, shard, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "bioalign.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bioalign_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported():
    names = _header_functions()
    assert len(names) >= 12
    assert sorted(_native.EXPORTS) == names, "bio_amd._native.EXPORTS is out of step with include/bioalign.h"
    L = ctypes.CDLL(_native.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} is declared in include/bioalign.h but not exported by libbioalign.so"


def test_abi_version_matches_header():
    text = open(os.path.join(ROOT, "include", "bioalign.h")).read()
    v = int(re.search(r"#define\s+BIOALIGN_ABI_VERSION\s+(\d+)", text).group(1))
    assert _native.lib().bioalign_abi_version() == v


def test_enum_values_match_header():
    text = open(os.path.join(ROOT, "include", "bioalign.h")).read()
    vals = dict(re.findall(r"(BIOALIGN_[A-Z_]+)\s*=\s*(-?\d+)", text))
    assert int(vals["BIOALIGN_GLOBAL"]) == _native.GLOBAL
    assert int(vals["BIOALIGN_LOCAL"]) == _native.LOCAL
    assert int(vals["BIOALIGN_FITTING"]) == _native.FITTING
    assert int(vals["BIOALIGN_OVERLAP"]) == _native.OVERLAP
    for k in ("EINVAL", "ENODEVICE", "ENOMEM", "EHIP", "EUNSUPPORTED"):
        assert int(vals["BIOALIGN_" + k]) == getattr(_native, k)


def test_batch_struct_layout_matches_header():
    # field order and sizes of bioalign_batch_t / bioalign_result_t as the header lays them out (LP64)
    assert [f[0] for f in _native.Batch._fields_] == [
        "s1", "s1_off", "s2", "s2_off", "n_pairs", "mode", "score_table", "na", "nb",
        "code1", "code2", "gap_open", "gap_extend", "gap_symbol", "flags"]
    assert ctypes.sizeof(_native.Batch) == 96
    assert _native.Batch.score_table.offset == 48 and _native.Batch.code1.offset == 64
    assert _native.Batch.gap_open.offset == 80 and _native.Batch.flags.offset == 92
    assert ctypes.sizeof(_native.Result) == 40


def _no_gpu():
    import torch

    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="needs a box without a GPU")
def test_fails_loudly_without_gpu():
    with pytest.raises(_native.BioAlignError) as e:
        _native.Engine()
    assert e.value.code == _native.ENODEVICE
    with pytest.raises(_native.BioAlignError):
        bio_amd.set_engine(None)
        bio_amd.AlignPair("global", b"ACGT", b"ACG", bio_amd.MatchAligner(1, -1), 1)


def test_missing_library_is_an_import_error(monkeypatch):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libbioalign.so")
    with pytest.raises(ImportError):
        _native.lib()


# ---------------------------------------------------------------------------
# host logic
# ---------------------------------------------------------------------------
def test_subst_matrix_index_quirks():
    # subst_matrix.go:27-30: Index is zero filled up to the largest alphabet byte
    b = bio_amd.Blosum62
    assert b.Score(ord("W"), ord("W")) == 11 and b.Score(ord("A"), ord("R")) == -1
    assert b.Score(ord("B"), ord("W")) == b.Score(ord("A"), ord("W"))  # foreign byte below 'Y' aliases to 'A'
    with pytest.raises(IndexError):
        b.Score(ord("a"), ord("A"))  # above 'Y': the reference panics
    p = bio_amd.Pam250
    assert p.Score(ord("W"), ord("W")) == 17 and p.Score(ord("C"), ord("W")) == -8
    for m in (b, p):
        M = np.array(m.Matrix)
        assert M.shape == (20, 20) and np.array_equal(M, M.T)


def test_score_table_asymmetric_and_call_count():
    calls = []

    class A:
        def Score(self, x, y):
            calls.append((x, y))
            return 3 * x - y

    t, c1, c2 = bio_amd.score_table(A(), [67, 65, 65], [84, 71])
    assert t.shape == (2, 2) and len(calls) == 4 and len(set(calls)) == 4
    assert t[c1[67], c2[71]] == 3 * 67 - 71 and t[c1[65], c2[84]] == 3 * 65 - 84
    t, c1, c2 = bio_amd.score_table(A(), [], [])
    assert t.shape == (1, 1)


def test_score_table_propagates_panics():
    with pytest.raises(IndexError):
        bio_amd.score_table(bio_amd.Blosum62, [ord("a")], [ord("A")])


def test_rescoring_follows_gap_symbol():
    a = bio_amd.MatchAligner(1, -1)
    assert bio_amd.LinearGap(b"AC-GT", b"ACCG-", a, 2) == 1 + 1 - 2 + 1 - 2
    assert bio_amd.ConstantGap(b"A--GT", b"ACCG-", a, 5) == 1 - 5 + 1 - 5
    with pytest.raises(RuntimeError):
        bio_amd.LinearGap(b"AC", b"A", a, 1)  # align.go:19-21
    old = bio_amd.GapSymbol
    try:
        bio_amd.GapSymbol = ord(".")
        assert bio_amd.LinearGap(b"A.", b"AC", a, 3) == 1 - 3
        assert bio_amd.LinearGap(b"A-", b"AC", a, 3) == 1 - 1
    finally:
        bio_amd.GapSymbol = old


def test_pack_sequences():
    d, o = bio_amd.pack_sequences([b"AC", b"", b"GTT"])
    assert d.tobytes() == b"ACGTT" and o.tolist() == [0, 2, 2, 5] and o.dtype == np.uint64
    d, o = bio_amd.pack_sequences([])
    assert d.size == 0 and o.tolist() == [0]


def test_errors_that_never_reach_the_device():
    a = bio_amd.MatchAligner(1, -1)
    assert bio_amd.AlignPair("semi-global", b"AC", b"AC", a, 1) == (-1, None, None)  # align_pair.go:14-15
    assert bio_amd.AlignPairBatch("nope", [b"A"] * 3, [b"C"] * 3, a, 1) == ([-1] * 3, [None] * 3, [None] * 3)
    assert bio_amd.AlignPairBatch("global", [], [], a, 1) == ([], [], [])
    with pytest.raises(IndexError):
        bio_amd.AlignGlobal(b"AC", b"", a, 1)  # align.go:128-130 indexes s2[-1]
    with pytest.raises(ValueError):
        bio_amd.AlignPairBatch("global", [b"A"], [], a, 1)
    with pytest.raises(ValueError):
        bio_amd.AlignGlobalAffine(b"AC", b"AC", a, 0.1, 1)  # not a multiple of 1/64: float64 rounding decides in the reference
    from bio_amd.align import _dyadic_scale
    assert _dyadic_scale(11.0, 1.0) == 1 and _dyadic_scale(2.5, 1) == 2 and _dyadic_scale(0.75, 0.25) == 4
    assert _dyadic_scale(1 / 3, 1) is None and _dyadic_scale(0.1, 0.5) is None


def test_score_table_fails_only_for_cells_some_pair_needs():
    # subst_matrix.go:15-17: Score panics on a byte outside the alphabet.  The reference scores (s1[i], s2[j]) inside one
    # pair only, so a batch must not fail on byte pairs that never meet.
    from bio_amd.subst_matrix import cooccurrence, score_table

    class Strict:
        def Score(self, x, y):
            if chr(x) in "ACGT" and chr(y) in "ACGT":
                return 1 if x == y else -1
            if chr(x) in "xy" and chr(y) in "xy":
                return 2 if x == y else -2
            raise IndexError("index out of range")

    d1, o1 = bio_amd.pack_sequences([b"ACGT", b"xyx"])
    d2, o2 = bio_amd.pack_sequences([b"AGT", b"yy"])
    with pytest.raises(IndexError):
        score_table(Strict(), np.unique(d1).tolist(), np.unique(d2).tolist())
    t, c1, c2 = score_table(Strict(), np.unique(d1).tolist(), np.unique(d2).tolist(), needed=cooccurrence(d1, o1, d2, o2))
    assert t[c1[ord("A")], c2[ord("A")]] == 1 and t[c1[ord("x")], c2[ord("y")]] == -2 and t[c1[ord("A")], c2[ord("y")]] == 0
    d2b, o2b = bio_amd.pack_sequences([b"AGy", b"yy"])  # now a pair does put A against y
    with pytest.raises(IndexError):
        score_table(Strict(), np.unique(d1).tolist(), np.unique(d2b).tolist(), needed=cooccurrence(d1, o1, d2b, o2b))


def test_packed_batch_validation():
    t, c1, c2 = synth.dna_table()
    d, o = bio_amd.pack_sequences([b"ACGT"])
    with pytest.raises(ValueError):
        _native.PackedBatch(d, o, d, o, "global", t.reshape(-1), c1, c2, 1, 1)
    with pytest.raises(ValueError):
        _native.PackedBatch(d, o, d, o, "global", t, c1[:10], c2, 1, 1)
    with pytest.raises(KeyError):
        _native.PackedBatch(d, o, d, o, "glocal", t, c1, c2, 1, 1)
    pb = _native.PackedBatch(d, o, d, o, "overlap", t, c1, c2, 2, 2, gap_symbol=ord("."))
    assert pb.c.mode == _native.OVERLAP and pb.c.na == 4 and pb.c.nb == 4 and pb.trace_capacity == 8
    assert pb.c.gap_symbol == ord(".")


def test_synth_is_deterministic_and_shaped():
    s1, o1, s2, o2 = synth.dna_pairs(300, 150, 150, config=2)
    t1, p1, t2, p2 = synth.dna_pairs(100, 150, 150, config=2, first=200)
    assert np.array_equal(s1[200 * 150:], t1) and np.array_equal(s2[200 * 150:], t2)  # streams are per pair
    assert set(np.unique(s1).tolist()) <= set(b"ACGT") and o1[-1] == 300 * 150 and o2[-1] == 300 * 150
    ident = (s1.reshape(300, 150)[:, :40] == s2.reshape(300, 150)[:, :40]).mean()
    assert ident > 0.6  # a mutated copy (indels shift the tail), not an independent string (0.25)
    s1, o1, s2, o2 = synth.fitting_reads(50, 25, 200, config=4)
    assert o1[-1] == 50 * 25 and o2[-1] == 50 * 200
    s1, o1, s2, o2 = synth.protein_pairs(20, 30, 35, config=3)
    assert set(np.unique(s1).tolist()) <= set(b"ARNDCQEGHILKMFPSTWYV") and o2[1] == 35


def test_splitmix64_known_answers():
    # splitmix64 from seed 0: the published first outputs
    r = synth.SplitMix64(np.array([0], dtype=np.uint64))
    assert [int(r.next()[0]) for _ in range(3)] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_shard_ranges_balance_and_cover():
    rng = np.random.default_rng(1)
    l1 = rng.integers(0, 300, 1000)
    l2 = rng.integers(0, 300, 1000)
    for world in (1, 2, 3, 8):
        b = shard.shard_ranges(l1, l2, world)
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b) >= 0)
        cells = np.maximum(l1 * l2, 1)
        per = [cells[b[r]:b[r + 1]].sum() for r in range(world)]
        assert max(per) - min(per) <= 2 * cells.max()
    assert shard.shard_ranges([], [], 4).tolist() == [0] * 5
    assert shard.shard_ranges([5], [5], 3).tolist()[-1] == 1
    b = shard.shard_ranges([10] * 8, [10] * 8, 8)
    assert b.tolist() == list(range(9))


def test_align_pair_trace_quirk_on_nul_bytes():
    # align_pair.go:197-216 restated on aligned traces (the kernels' output); checked against the oracle's trace() on the
    # GPU (tests/test_gpu_parity.py::test_nul_bytes_follow_align_pair_trace)
    from bio_amd.align import _trace_quirk
    assert _trace_quirk(b"AC-T", b"ACGT") == (b"AC-T", b"ACGT")
    assert _trace_quirk(None, None) == (None, None)
    # one NUL in t1: walking back, t1 collects T C A and t2 collects T G C A; the loop swaps positions 0 and 2 of both
    assert _trace_quirk(b"A\x00CT", b"ACGT") == (b"ACT", b"CGTA")
    # t2 shorter than the index the loop uses: Go panics
    with pytest.raises(IndexError):
        _trace_quirk(b"ACGT", b"\x00\x00\x00T")
    assert _trace_quirk(b"\x00", b"\x00") == (None, None)
