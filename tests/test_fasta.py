"""FASTA ingestion: ReadFASTA, fasta.go:55-84.

CPU: the oracle's restatement against the reference's own example (fasta_test.go:10-44: the only golden vector the
reference holds for this path) and the documented behaviours of fasta.go:47-54.  GPU: the parser kernels
(kernels_fasta.hip, through bioalign_fasta_parse) byte for byte against the oracle on texts built around every
rule -- crlf, blank lines, white space at both ends and inside lines, records without data, no trailing newline,
lines across the 16 KB chunks and the 64-byte thread spans, the "No header" error -- and the device-resident
result fed straight into the aligner."""
import random

import numpy as np
import pytest

from oracle import oracle as O

EXAMPLE = b">db|123|abc example sequence\nAGACCA\nTACCA"   # fasta_test.go:11-13


def test_oracle_matches_the_reference_example():
    recs = O.ReadFASTA(EXAMPLE)
    assert recs == [(b">db|123|abc example sequence", b"AGACCATACCA")]  # fasta_test.go:23-26


def test_oracle_documented_behaviour():
    # fasta.go:47-54: blank lines allowed, white space trimmed, lf and crlf, a record for every header
    assert O.ReadFASTA(b"\n  >a x \r\nAC GT \r\n\r\n>b\n>c\n  TT\n") == [(b">a x", b"AC GT"), (b">b", b""), (b">c", b"TT")]
    assert O.ReadFASTA(b"") == [] and O.ReadFASTA(b"\n\n \n") == []
    with pytest.raises(ValueError, match="No header"):
        O.ReadFASTA(b"ACGT\n>a\nAC\n")


NBSP, NEL, OGHAM, EN_QUAD, HAIR, LSEP, PSEP, NNBSP, MMSP, IDEO = (
    "\u00a0", "\u0085", "\u1680", "\u2000", "\u200a", "\u2028", "\u2029", "\u202f", "\u205f", "\u3000")


def test_oracle_trims_unicode_white_space_like_go():
    # bytes.TrimSpace (fasta.go:63) trims what unicode.IsSpace accepts, UTF-8 encoded, on both ends of a line; the
    # reference's own tests hold no such line (parity pinned by Go's documented table, not by a fixture)
    sp = "".join([NBSP, NEL, OGHAM, EN_QUAD, HAIR, LSEP, PSEP, NNBSP, MMSP, IDEO]).encode()
    text = sp + b" >a\tx " + sp + b"\n" + IDEO.encode() + b"AC" + NBSP.encode() + b"GT" + HAIR.encode() + b" \r\n" + sp + b"\n>b\nT\n"
    assert O.ReadFASTA(text) == [(b">a\tx", b"AC" + NBSP.encode() + b"GT"), (b">b", b"T")]
    # not white space: U+200B (zero width space), U+FEFF, a lone continuation byte, a cut-off sequence
    for junk in ("\u200b".encode(), "\ufeff".encode(), b"\xa0", b"\xe2\x80", b"\xc2"):
        assert O.ReadFASTA(b">h\n" + junk + b"AC" + junk + b"\n") == [(b">h", junk + b"AC" + junk)]
    # a line of Unicode spaces alone is blank; one that starts with them and then '>' is a header
    assert O.ReadFASTA(b">h\nA\n" + sp + b"\n" + NBSP.encode() + b">i\nC") == [(b">h", b"A"), (b">i", b"C")]


def test_id_and_desc_host_side():
    from bio_amd.fasta import FASTASeq
    f = FASTASeq(">db|123|abc example sequence", b"")
    assert (f.ID(), f.Desc()) == ("db|123|abc", "example sequence")   # fasta_test.go:41-43
    assert (FASTASeq(">x", b"").ID(), FASTASeq(">x", b"").Desc(), FASTASeq("", b"").ID()) == ("x", "", "")


def _texts():
    rng = random.Random(12)
    alpha = b"ACGT"
    out = [EXAMPLE, b"", b"\n", b">only", b">a\n", b">a\r\nAC\r\nGT\r\n", b" \t>a  \n\t AC GT\t\n\n\n>b\n\n>c\nT",
           b">a\n" + b"A" * 70000 + b"\n", b">a\n" + b"\n" * 40000 + b"C\n", b">h\n" + b" " * 20000 + b"G" + b" " * 20000]
    sp = [c.encode() for c in (NBSP, NEL, OGHAM, EN_QUAD, HAIR, LSEP, PSEP, NNBSP, MMSP, IDEO)]
    out.append(b"".join(sp) + b" >a x " + b"".join(sp) + b"\n" + sp[9] + b"AC" + sp[0] + b"GT" + sp[4] + b" \r\n" + b"".join(sp) + b"\n>b\nT\n")
    out.append(b">h\n\xa0AC\xa0\n\xe2\x80AC\xe2\x80\n\xc2\n" + "\u200b".encode() + b"G\n" + sp[1] + b">i\n\xc2\x85\xc2\n")
    for k in range(3):  # Unicode white space sprinkled over the line ends
        parts = []
        for r in range(rng.randrange(50, 300)):
            parts.append(rng.choice(sp + [b""]) + b">u%d" % r + rng.choice(sp + [b""]) + b"\n")
            for _ in range(rng.randrange(0, 4)):
                parts.append(rng.choice(sp + [b"", b" "]) + bytes(rng.choice(alpha) for _ in range(rng.randrange(0, 130))) +
                             rng.choice(sp + [b"", b"\t"]) * rng.randrange(0, 3) + rng.choice([b"\n", b"\r\n"]))
        out.append(b"".join(parts))
    for k in range(6):  # many records, line lengths around the 64-byte spans, text across several 16 KB chunks
        parts = []
        for r in range(rng.randrange(1, 400)):
            parts.append(b">rec%d some words %d" % (r, k) + rng.choice([b"\n", b"\r\n", b" \n"]))
            for _ in range(rng.randrange(0, 6)):
                ln = rng.choice([0, 1, 59, 60, 61, 63, 64, 65, 127, 128, 150, 250, rng.randrange(0, 700)])
                parts.append(rng.choice([b"", b" ", b"\t"]) + bytes(rng.choice(alpha) for _ in range(ln)) +
                             rng.choice([b"\n", b"\r\n", b"  \n", b"\n\n"]))
        text = b"".join(parts)
        out.append(text if k % 2 else text.rstrip(b"\r\n"))
    return out


@pytest.mark.gpu
def test_gpu_parser_equals_oracle():
    import bio_amd
    for text in _texts():
        want = O.ReadFASTA(text)
        got = bio_amd.ReadFASTA(text)
        assert [(f.Header.encode("latin-1"), f.Seq) for f in got] == want, len(text)
    with pytest.raises(ValueError, match="No header"):
        bio_amd.ReadFASTA(b"  \nACGT\n>a\nAC\n")
    with pytest.raises(ValueError, match="No header"):
        bio_amd.ReadFASTA(b">late" * 0 + b"A" * 40000 + b"\n>a\n")


@pytest.mark.gpu
def test_fasta_to_alignment_without_a_host_copy():
    # reads and references from two FASTA texts, parsed on the GPU, aligned from where the parser left them
    import torch
    import bio_amd
    from bio_amd import _native, synth

    s1, o1, s2, o2 = synth.dna_pairs(300, 90, 100, config=2)
    def fasta(s, o, tag):
        rows = []
        for p in range(len(o) - 1):
            seq = s[int(o[p]):int(o[p + 1])].tobytes()
            rows.append(b">%s%d\r\n" % (tag, p) + seq[:60] + b"\n" + seq[60:] + b"\n\n")
        return b"".join(rows)
    fa, fb = bio_amd.FastaBatch(fasta(s1, o1, b"read")), bio_amd.FastaBatch(fasta(s2, o2, b"ref"))
    assert fa.count == fb.count == 300 and np.array_equal(fa.off, o1) and np.array_equal(fb.off, o2)
    table, c1, c2 = synth.dna_table(1, -1)
    eng = bio_amd.align.engine()
    # lengths only on the host side (the plan needs them); the bytes never come back
    pb = _native.PackedBatch(np.zeros(1, np.uint8), fa.off, np.zeros(1, np.uint8), fb.off, "fitting", table, c1, c2, 1, 1)
    plan = eng.plan(pb)
    dev = torch.device("cuda:0")
    N, cap = 300, int(fa.off[-1] + fb.off[-1])
    sc = torch.empty(N, dtype=torch.int64, device=dev)
    t1 = torch.zeros(cap + 16, dtype=torch.uint8, device=dev); t2 = torch.zeros(cap + 16, dtype=torch.uint8, device=dev)
    toff = torch.empty(N, dtype=torch.int64, device=dev); tlen = torch.empty(N, dtype=torch.int64, device=dev)
    plan.run(fa.d_seq, fa.d_off, fb.d_seq, fb.d_off, sc.data_ptr(), t1.data_ptr(), t2.data_ptr(), toff.data_ptr(),
             tlen.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    sc, t1, t2, toff, tlen = (x.cpu().numpy() for x in (sc, t1, t2, toff, tlen))
    oa = O.MatchAligner(1, -1)
    for p in range(0, N, 7):
        x, y = s1[int(o1[p]):int(o1[p + 1])].tobytes(), s2[int(o2[p]):int(o2[p + 1])].tobytes()
        a, l = int(toff[p]), int(tlen[p])
        assert (int(sc[p]), t1[a:a + l].tobytes(), t2[a:a + l].tobytes()) == O.AlignPair("fitting", x, y, oa, 1)


# ---- the reference's second reader: FASTAReader.ReadSeq (fasta.go:93-142) --------------------------------------
READER_EXAMPLE = b">db|123|abc example sequence\nAGACCA\nTACCA"  # fasta_test.go:46-64


def _oracle_read_all(text, size=4096):
    r = O.FASTAReader(text, size)
    out = []
    while True:
        h, s, err = r.ReadSeq()
        if err is not None:
            return out, err
        out.append((h, s))


def test_oracle_reader_reference_example():
    r = O.FASTAReader(READER_EXAMPLE)
    assert r.ReadSeq() == (b">db|123|abc example sequence", b"AGACCATACCA", None)
    assert r.ReadSeq()[2] == "EOF"
    # rules that differ from ReadFASTA's: nothing is trimmed, crlf is a line end, leading data is ignored
    recs, err = _oracle_read_all(b"junk\n\n>a x \r\n AC GT \r\n\r\n >not a header\n>b\n>c\nTT\r")
    assert err == "EOF"
    assert recs == [(b">a x ", b" AC GT  >not a header"), (b">b", b""), (b">c", b"TT\r")]
    assert _oracle_read_all(b"") == ([], "EOF") and _oracle_read_all(b"ACGT\n") == ([], "EOF")
    # bufio's buffer: 4095 bytes + "\n" fit, 4096 + "\n" do not; the records before the line are good
    ok = b">a\n" + b"A" * 4095 + b"\n>b\nC\n"
    assert _oracle_read_all(ok)[1] == "EOF" and len(_oracle_read_all(ok)[0]) == 2
    bad = b">a\nAC\n>b\n" + b"A" * 4096 + b"\n>c\nC\n"
    assert _oracle_read_all(bad) == ([(b">a", b"AC")], "FASTA line too long")


def _gpu_read_all(text, limit=0):
    import bio_amd

    r = bio_amd.NewFASTAReader(text, limit)
    out = []
    while True:
        try:
            f = r.ReadSeq()
        except EOFError:
            return out, "EOF"
        except ValueError as e:
            return out, str(e)
        out.append((f.Header.encode("latin-1"), f.Seq))


READER_CASES = [
    READER_EXAMPLE,
    b"",
    b"ACGT\n",
    b"\n\n\n",
    b">only header",
    b"junk\n\n>a x \r\n AC GT \r\n\r\n >not a header\n>b\n>c\nTT\r",
    b">a\r\nAC\r\n\r\nGT\r\n>b\r\n\r\n",
    b"\r\n>a\nA\rC\n\r",
    b">a\n" + b"A" * 4095 + b"\n>b\nC\n",
    b">a\n" + b"A" * 4094 + b"\r\n>b\nC\n",
    b">a\nAC\n>b\n" + b"A" * 4096 + b"\n>c\nC\n",
    b">a\nAC\n>b\n" + b"A" * 4095 + b"\r\n>c\nC\n",
    b"x" * 5000 + b"\n>a\nAC\n",
    b">a\nAC\n>" + b"h" * 4200 + b"\nGG\n",
    b">a\nAC\n>b\nGG\n" + b"T" * 4096,
    b">a\nAC\n>b\nGG\n" + b"T" * 4095,
]


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(READER_CASES)))
def test_gpu_reader_matches_oracle(k):
    text = READER_CASES[k]
    assert _gpu_read_all(text) == _oracle_read_all(text)


@pytest.mark.gpu
def test_gpu_reader_random_and_small_buffer():
    rng = np.random.default_rng(3)
    pieces = [b">", b">h d", b"ACGT", b"A C", b" ", b"\r", b"\n", b"\r\n", b"\n\n", b"N" * 40, b"\t"]
    for trial in range(60):
        text = b"".join(pieces[int(rng.integers(0, len(pieces)))] for _ in range(int(rng.integers(0, 60))))
        for limit in (0, 16, 41):
            assert _gpu_read_all(text, limit) == _oracle_read_all(text, limit or 4096), (text, limit)
