"""Fragment assembly on overlap alignments: bio_amd.AlignLongOverlap.

The reference has no test of AlignLong (align_long.go); the oracle restates it (parity unpinned beyond that reading).
Checked here: on error-free reads AlignLongOverlap = the reference's AlignLong = the genome; with read errors the
reference's exact-substring overlaps fail while the alignment-based ones still assemble; the GPU version equals the
oracle-based one (same merge, overlaps from the oracle's AlignPair).
"""
import numpy as np
import pytest

from oracle import oracle as O


def _reads(rng, genome_len=400, read_len=60, step=20, shuffle=True):
    al = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = al[rng.integers(0, 4, genome_len)].tobytes()
    starts = list(range(0, genome_len - read_len + 1, step))
    reads = [genome[s:s + read_len] for s in starts]
    if shuffle:
        order = rng.permutation(len(reads))
        reads = [reads[k] for k in order]
    return genome, reads


def _with_errors(rng, reads, every=3):
    out = []
    for k, r in enumerate(reads):
        b = bytearray(r)
        if k % every == 0:  # one substitution in the middle of the read: inside every overlap it takes part in
            p = len(b) // 2 + int(rng.integers(-5, 6))
            b[p] = ord("ACGT"[("ACGT".index(chr(b[p])) + 1) % 4])
        out.append(bytes(b))
    return out


def _oracle_cells(ss, pairs, a, indel):
    return O.overlap_end_cells(ss, pairs, O.MatchAligner(1, -1), indel)


def test_oracle_align_long_assembles_error_free_reads():
    rng = np.random.default_rng(1)
    genome, reads = _reads(rng)
    assert O.AlignLong(reads) == genome
    assert O.AlignLongOverlap(reads, O.MatchAligner(1, -1), 2) == genome
    assert O.AlignLong([]) == b"" and O.AlignLong([b"ACGT"]) == b"ACGT"
    assert O.AlignLong([b"AAAACCCC", b"GGGGTTTT"]) == b""  # nothing overlaps


def test_host_merge_logic_with_oracle_overlaps():
    # the merge (align_long.go:46-89) in the product mirror, overlaps injected from the oracle: no GPU needed
    import bio_amd

    rng = np.random.default_rng(2)
    genome, reads = _reads(rng)
    assert bio_amd.AlignLongOverlap(reads, end_cells=_oracle_cells) == genome
    noisy = _with_errors(rng, reads)
    assert O.AlignLong(noisy) == b""  # the reference's exact overlaps break on read errors
    got = bio_amd.AlignLongOverlap(noisy, end_cells=_oracle_cells)
    assert got == O.AlignLongOverlap(noisy, O.MatchAligner(1, -1), 2)
    assert len(got) == len(genome) and sum(x != y for x, y in zip(got, genome)) <= len(reads)
    assert bio_amd.AlignLongOverlap([]) == b"" and bio_amd.AlignLongOverlap([b"ACGT"]) == b"ACGT"


def test_repeated_half_fragment_records_both_behaviours():
    # A 30-base repeat R inside the genome (A + R + B + R + C, 60-base reads every 20 bases).  The reference looks for the
    # last half of a read verbatim anywhere in another read (strings.Index, align_long.go:41) and does not check the
    # prefix; AlignLongOverlap uses overlap alignments with thresholds.  On repeats the two differ, in both directions;
    # these two fixtures record what each does (the claim "same result on error-free reads" was too strong).
    import bio_amd

    al = np.frombuffer(b"ACGT", dtype=np.uint8)

    def fixture(seed, off):
        rng = np.random.default_rng(seed)
        R = al[rng.integers(0, 4, 30)].tobytes()
        A = al[rng.integers(0, 4, 130 + off)].tobytes()
        B = al[rng.integers(0, 4, 110)].tobytes()
        Cc = al[rng.integers(0, 4, 120 - off)].tobytes()
        genome = A + R + B + R + Cc
        reads = [genome[s:s + 60] for s in range(0, len(genome) - 60 + 1, 20)]
        order = rng.permutation(len(reads))
        return genome, [reads[k] for k in order]

    # the repeat is exactly the last half of a read: the reference finds it in the wrong read and gives up; the
    # alignment-based overlaps assemble the genome
    genome, reads = fixture(0, 0)
    assert O.AlignLong(reads) == b""
    assert bio_amd.AlignLongOverlap(reads, end_cells=_oracle_cells) == genome
    assert O.AlignLongOverlap(reads, O.MatchAligner(1, -1), 2) == genome
    # the repeat straddles read halves: the reference assembles the genome, the alignment-based merge does not
    genome, reads = fixture(6, 10)
    assert O.AlignLong(reads) == genome
    assert bio_amd.AlignLongOverlap(reads, end_cells=_oracle_cells) == b""
    assert O.AlignLongOverlap(reads, O.MatchAligner(1, -1), 2) == b""


@pytest.mark.gpu
def test_gpu_assembly_matches_reference_and_oracle():
    import bio_amd

    for seed in (3, 4, 5):
        rng = np.random.default_rng(seed)
        genome, reads = _reads(rng, genome_len=600 + 40 * seed, read_len=80, step=20)
        assert bio_amd.AlignLongOverlap(reads) == O.AlignLong(reads) == genome
        noisy = _with_errors(rng, reads)
        got = bio_amd.AlignLongOverlap(noisy)
        assert got == O.AlignLongOverlap(noisy, O.MatchAligner(1, -1), 2)
        assert len(got) == len(genome)
    assert bio_amd.AlignLongOverlap([b"AAAACCCC", b"GGGGTTTT"]) == b""


@pytest.mark.gpu
def test_gpu_overlap_end_cells_match_oracle():
    from bio_amd.assembly import overlap_end_cells
    import bio_amd

    rng = np.random.default_rng(8)
    _, reads = _reads(rng, genome_len=500, read_len=70, step=15)
    reads = _with_errors(rng, reads, every=2)
    pairs = [(si, ti) for si in range(len(reads)) for ti in range(len(reads)) if si != ti]
    sc, j = overlap_end_cells(reads, pairs, bio_amd.MatchAligner(1, -1), 2)
    osc, oj = O.overlap_end_cells(reads, pairs, O.MatchAligner(1, -1), 2)
    assert np.array_equal(sc, osc) and np.array_equal(j, oj)
