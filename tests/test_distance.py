"""All-pairs distance matrices: bioalign_all_pairs behind DistanceMatrix.

Pinned by the reference's own examples (dna_list_test.go:320-337, str_test.go:218-235: Hamming distances of four
10-mers) and by the oracle's restatement of the loop with the oracle's aligners as the distance function.
"""
import numpy as np
import pytest

from oracle import oracle as O

EXAMPLE = [b"TTTCCATTTA", b"GATTCATTTC", b"TTTCCATTTT", b"GTTCCATTTA"]
EXAMPLE_OUT = [[0, 4, 1, 1], [4, 0, 4, 3], [1, 4, 0, 2], [1, 3, 2, 0]]


def hamming(a, b):
    return float(sum(x != y for x, y in zip(a, b)))


def test_oracle_distance_matrix_reference_example():
    assert O.DistanceMatrix(EXAMPLE, hamming) == EXAMPLE_OUT
    assert O.DistanceMatrix([b"ACGT"], hamming) == [[0.0]]
    with pytest.raises(O.GoPanic):
        O.DistanceMatrix([], hamming)


def test_oracle_hamming_as_global_alignment():
    # equal lengths, match 0 / mismatch -1, 2*gap >= length: no gapped alignment beats the ungapped one, so
    # -score is the Hamming distance; this is how the GPU test pins the engine on the reference's example
    a = O.MatchAligner(0, -1)
    f = lambda s, t: -O.AlignPair("global", s, t, a, 5)[0]  # noqa: E731
    assert O.DistanceMatrix(EXAMPLE, f) == EXAMPLE_OUT


def test_host_generic_callable_follows_reference_loop():
    import bio_amd

    assert bio_amd.DistanceMatrix(EXAMPLE, hamming) == EXAMPLE_OUT
    with pytest.raises(IndexError):
        bio_amd.DistanceMatrix([], hamming)


def _random_list(rng, n, lo, hi, alphabet=b"ACGT"):
    al = np.frombuffer(alphabet, dtype=np.uint8)
    return [al[rng.integers(0, len(al), int(rng.integers(lo, hi + 1)))].tobytes() for _ in range(n)]


@pytest.mark.gpu
def test_reference_example_through_the_engine():
    import bio_amd

    f = bio_amd.AlignDist("global", bio_amd.MatchAligner(0, -1), 5)
    assert bio_amd.DistanceMatrix(EXAMPLE, f) == EXAMPLE_OUT
    assert bio_amd.DistanceMatrix([b"ACGT"], f) == [[0.0]]
    with pytest.raises(IndexError):
        bio_amd.DistanceMatrix([], f)
    assert f(EXAMPLE[1], EXAMPLE[0]) == 4.0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["global", "local", "fitting", "overlap"])
@pytest.mark.parametrize("affine", [False, True])
def test_all_pairs_matches_oracle(mode, affine):
    import bio_amd

    rng = np.random.default_rng(["global", "local", "fitting", "overlap"].index(mode) * 2 + int(affine))
    l = _random_list(rng, 37, 0, 70) + [b"", b"A", b""]
    if affine:
        f = bio_amd.AlignDist(mode, bio_amd.MatchAligner(2, -3), 5, 1)
        oa = O.MatchAligner(2, -3)
        of = lambda s, t: -O.AlignAffine(mode, s, t, oa, 5, 1)[0]  # noqa: E731
    else:
        f = bio_amd.AlignDist(mode, bio_amd.MatchAligner(1, -1), 2)
        oa = O.MatchAligner(1, -1)
        of = lambda s, t: -O.AlignPair(mode, s, t, oa, 2)[0]  # noqa: E731
    assert bio_amd.DistanceMatrix(l, f) == O.DistanceMatrix(l, of)


@pytest.mark.gpu
def test_all_pairs_protein_asymmetric_roles():
    # sequence i is the row sequence, j the column sequence: fitting is not symmetric, so the roles show
    import bio_amd

    rng = np.random.default_rng(5)
    l = _random_list(rng, 25, 5, 90, b"ACDEFGHIKLMNPQRSTVWY")
    f = bio_amd.AlignDist("fitting", bio_amd.Blosum62, 5, to_dist=lambda s: s.astype(np.float64))
    oa = O.Blosum62()
    of = lambda s, t: O.AlignPair("fitting", s, t, oa, 5)[0]  # noqa: E731
    assert bio_amd.DistanceMatrix(l, f) == O.DistanceMatrix(l, of)


@pytest.mark.gpu
def test_all_pairs_in_chunks_and_from_fasta(monkeypatch):
    import bio_amd

    rng = np.random.default_rng(9)
    l = _random_list(rng, 60, 30, 120)
    f = bio_amd.AlignDist("local", bio_amd.MatchAligner(1, -1), 1)
    whole = bio_amd.AlignScores(l, f)
    monkeypatch.setenv("BIOALIGN_ALL_PAIRS_CHUNK", "257")  # 1770 pairs in seven plans
    assert np.array_equal(bio_amd.AlignScores(l, f), whole)
    monkeypatch.delenv("BIOALIGN_ALL_PAIRS_CHUNK")
    text = b"".join(b">r%d\n" % k + s[:50] + b"\n" + s[50:] + b"\n" for k, s in enumerate(l))
    fb = bio_amd.FastaBatch(text)
    assert fb.count == len(l)
    assert np.array_equal(bio_amd.AlignScores(fb, f), whole)
    # size-independent property: local scores are symmetric under exchange of the two sequences when the
    # score table is (the matrix is built from one triangle; the other one is computed here)
    rev = bio_amd.AlignScores(l[::-1], f)
    n = len(l)
    ii, jj = np.tril_indices(n, -1)
    back = {(n - 1 - i, n - 1 - j): v for i, j, v in zip(ii.tolist(), jj.tolist(), rev.tolist())}
    for i, j, v in zip(ii.tolist(), jj.tolist(), whole.tolist()):
        assert back[(j, i)] == v


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["overlap", "fitting", "local"])
def test_all_ordered_pairs_with_end_cells(mode, monkeypatch):
    # every (i, j), i != j: scores against the oracle, end cells against the traces' extent
    import bio_amd
    from bio_amd import _native, pack_sequences, score_table

    rng = np.random.default_rng(17)
    l = _random_list(rng, 23, 0, 60)
    n = len(l)
    data, off = pack_sequences(l)
    tbl, c1, c2 = score_table(bio_amd.MatchAligner(1, -1), np.unique(data).tolist(), np.unique(data).tolist())
    pb = _native.PackedBatch(data, off, np.zeros(0, np.uint8), off, mode, tbl, c1, c2, 2, 2)
    eng = bio_amd.engine()
    sc, ei, ej = eng.all_ordered_pairs(pb)
    monkeypatch.setenv("BIOALIGN_ALL_PAIRS_CHUNK", "100")
    sc2, ei2, ej2 = eng.all_ordered_pairs(pb)
    assert np.array_equal(sc, sc2) and np.array_equal(ei, ei2) and np.array_equal(ej, ej2)
    assert len(sc) == n * (n - 1)
    oa = O.MatchAligner(1, -1)
    p = 0
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            score, t1, t2 = O.AlignPair(mode, l[i], l[j], oa, 2)
            a = (t1 or b"").replace(b"-", b"")
            b = (t2 or b"").replace(b"-", b"")
            e_i, e_j = int(ei[p]), int(ej[p])
            assert int(sc[p]) == score, (mode, i, j)
            assert l[i][e_i - len(a):e_i] == a and l[j][e_j - len(b):e_j] == b, (mode, i, j, e_i, e_j)
            if mode != "local":
                assert e_i == len(l[i])
            p += 1
    sc3, none_i, none_j = eng.all_ordered_pairs(pb, end_cells=False)
    assert none_i is None and none_j is None and np.array_equal(sc3, sc)
