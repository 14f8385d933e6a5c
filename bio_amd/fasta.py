"""Host mirror of the reference's FASTA reader (fasta.go) over the GPU parser (bioalign_fasta_parse).

ReadFASTA keeps the reference's contract (fasta.go:47-84): one FASTASeq per header line, headers stored
unparsed, blank lines and surrounding white space dropped, lf and crlf alike, ValueError("No header") where
the reference returns that error.  FastaBatch is the additive, device-resident form: the sequences stay in HBM
as concatenated bytes + offsets, the shape bioalign_plan_run takes, so reads go from file bytes to alignment
without a host pass over them.  There is no CPU parser here: without the GPU library these calls fail.
"""
from __future__ import annotations

from dataclasses import dataclass

from . import _native
from ._native import BioAlignError


@dataclass
class FASTASeq:
    """fasta.go:17-20"""
    Header: str
    Seq: bytes

    def ID(self) -> str:  # fasta.go:23-33
        if self.Header == "":
            return ""
        ident = self.Header[1:]
        sp = ident.find(" ")
        return ident[:sp] if sp >= 0 else ident

    def Desc(self) -> str:  # fasta.go:36-45
        if self.Header == "":
            return ""
        ident = self.Header[1:]
        sp = ident.find(" ")
        return ident[sp + 1:] if sp >= 0 else ""


class FastaBatch:
    """The parsed records, resident on the device: .count, .off (host offsets), .d_seq / .d_off (device
    addresses), .headers (host strings), .kernel_ms."""

    def __init__(self, text, engine=None, reader_line_limit=None):
        from .align import engine as default_engine

        self.text = bytes(text)
        self._fa = _native.Fasta(engine or default_engine(), self.text, reader_line_limit=reader_line_limit)
        self.too_long_at = self._fa.too_long_at
        self.count, self.off = self._fa.count, self._fa.off
        self.d_seq, self.d_off, self.kernel_ms = self._fa.d_seq, self._fa.d_off, self._fa.kernel_ms
        self.headers = [self.text[int(b):int(b + l)].decode("latin-1")
                        for b, l in zip(self._fa.header_begin, self._fa.header_len)]

    def sequences(self):
        raw = self._fa.seq_bytes().tobytes()  # (reader rules: ignored leading data leaves off[0] unused bytes in front)
        return [raw[int(self.off[k]):int(self.off[k + 1])] for k in range(self.count)]


def ReadFASTA(data) -> list:
    """fasta.go:55-84.  `data`: bytes, or a file-like object with .read()."""
    if hasattr(data, "read"):
        data = data.read()
    if isinstance(data, str):
        data = data.encode()
    try:
        fb = FastaBatch(data)
    except BioAlignError as e:
        if "No header" in str(e):
            raise ValueError("No header") from None  # errors.New("No header"), fasta.go:73
        raise
    return [FASTASeq(h, s) for h, s in zip(fb.headers, fb.sequences())]


def ReadFASTAFile(path: str) -> list:
    """fasta.go:88-95"""
    with open(path, "rb") as f:
        return ReadFASTA(f)


class FASTAReader:
    """fasta.go:93-142: the reference's second reader, one record per ReadSeq call.  Its rules differ from
    ReadFASTA's (no trimming, "\\r\\n" handled by bufio.ReadLine, data before the first header ignored, lines
    beyond bufio's 4096-byte buffer are an error).  Here the whole stream is parsed on the GPU at construction
    (bioalign_fasta_parse_reader) and ReadSeq hands the records out one by one: EOFError stands for io.EOF,
    ValueError("FASTA line too long") for that error -- after which the reader stays in error (the reference
    would go on from the middle of the line)."""

    def __init__(self, data, line_limit: int = 0):
        if hasattr(data, "read"):
            data = data.read()
        if isinstance(data, str):
            data = data.encode()
        self._fb = FastaBatch(data, reader_line_limit=line_limit)
        self._raw = self._fb._fa.seq_bytes().tobytes()
        self._next = 0

    def ReadSeq(self) -> FASTASeq:
        k, fb = self._next, self._fb
        if fb.too_long_at >= 0 and k >= fb.too_long_at:
            raise ValueError("FASTA line too long")  # fasta.go:125
        if k >= fb.count:
            raise EOFError("EOF")  # io.EOF, fasta.go:120-123
        self._next += 1
        return FASTASeq(fb.headers[k], self._raw[int(fb.off[k]):int(fb.off[k + 1])])


def NewFASTAReader(r, line_limit: int = 0) -> FASTAReader:
    """fasta.go:100-102"""
    return FASTAReader(r, line_limit)
