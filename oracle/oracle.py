"""ctypes binding of the CPU oracle (oracle/bio_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (bio_amd) never imports it.

Function names follow the reference's Go API (align.go, align_pair.go,
align_linear.go, align_affine.go, subst_matrix.go under /root/reference).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbio_oracle.so")

OK, PANIC, BAD_MODE, NOMEM = 0, -1, -2, -3
MODES = {"global": 0, "local": 1, "fitting": 2, "overlap": 3}


class GoPanic(RuntimeError):
    """Raised where the reference's Go code would panic."""


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "bio_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libbio_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p, i64p, szp = C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_size_t)
        L.bo_new_subst_matrix.restype = C.c_void_p
        L.bo_new_subst_matrix.argtypes = [u8p, C.c_int, C.POINTER(C.c_int32)]
        L.bo_new_match_aligner.restype = C.c_void_p
        L.bo_new_match_aligner.argtypes = [C.c_int64, C.c_int64]
        L.bo_new_table_aligner.restype = C.c_void_p
        L.bo_new_table_aligner.argtypes = [C.POINTER(C.c_int32)]
        L.bo_blosum62.restype = C.c_void_p
        L.bo_pam250.restype = C.c_void_p
        L.bo_aligner_free.argtypes = [C.c_void_p]
        L.bo_score.restype = C.c_int64
        L.bo_score.argtypes = [C.c_void_p, C.c_uint8, C.c_uint8]
        L.bo_set_gap_symbol.argtypes = [C.c_uint8]
        L.bo_get_gap_symbol.restype = C.c_uint8
        for fn in (L.bo_linear_gap, L.bo_constant_gap):
            fn.restype = C.c_int
            fn.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_void_p, C.c_int64, i64p]
        for fn in (L.bo_align_global, L.bo_align_local, L.bo_align_global_ls_as_written):
            fn.restype = C.c_int
            fn.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_void_p, C.c_int64, i64p, u8p, u8p, szp]
        L.bo_align_pair.restype = C.c_int
        L.bo_align_pair.argtypes = [C.c_int, u8p, C.c_size_t, u8p, C.c_size_t, C.c_void_p, C.c_int64,
                                    i64p, u8p, szp, u8p, szp]
        L.bo_last_column_scores.restype = C.c_int
        L.bo_last_column_scores.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_void_p, C.c_int64, i64p, i64p]
        L.bo_affine_global_score_f64.restype = C.c_int
        L.bo_affine_global_score_f64.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_void_p,
                                                 C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.bo_align_affine.restype = C.c_int
        L.bo_align_affine.argtypes = [C.c_int, u8p, C.c_size_t, u8p, C.c_size_t, C.c_void_p, C.c_int64,
                                      C.c_int64, i64p, u8p, u8p, szp]
        L.bo_affine_gap.restype = C.c_int
        L.bo_affine_gap.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_void_p, C.c_int64, C.c_int64, i64p]
        L.bo_batch.restype = C.c_int64
        L.bo_batch.argtypes = [C.c_int, C.c_int, u8p, C.POINTER(C.c_uint64), u8p, C.POINTER(C.c_uint64),
                               C.c_uint64, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                               i64p, u8p, u8p, C.POINTER(C.c_uint64)]
        L.bo_read_fasta.restype = C.c_int64
        L.bo_read_fasta.argtypes = [u8p, C.c_uint64, u8p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def _buf(b: bytes):
    arr = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b if len(b) else b"\0")
    return arr


class Aligner:
    """Owns a bo_aligner*."""

    def __init__(self, handle):
        if not handle:
            raise MemoryError("oracle aligner allocation failed")
        self._h = C.c_void_p(handle)

    def __del__(self):
        try:
            if self._h:
                lib().bo_aligner_free(self._h)
                self._h = None
        except Exception:
            pass

    def Score(self, x: int, y: int) -> int:
        return int(lib().bo_score(self._h, x, y))


def NewSubstMatrix(alphabet: bytes, matrix) -> Aligner:
    m = np.ascontiguousarray(np.asarray(matrix, dtype=np.int32).reshape(len(alphabet), len(alphabet)))
    return Aligner(lib().bo_new_subst_matrix(_buf(alphabet), len(alphabet),
                                             m.ctypes.data_as(C.POINTER(C.c_int32))))


def MatchAligner(match: int, mismatch: int) -> Aligner:
    return Aligner(lib().bo_new_match_aligner(match, mismatch))


def TableAligner(table256) -> Aligner:
    t = np.ascontiguousarray(np.asarray(table256, dtype=np.int32).reshape(256, 256))
    return Aligner(lib().bo_new_table_aligner(t.ctypes.data_as(C.POINTER(C.c_int32))))


def Blosum62() -> Aligner:
    return Aligner(lib().bo_blosum62())


def Pam250() -> Aligner:
    return Aligner(lib().bo_pam250())


def set_gap_symbol(g: int) -> None:
    lib().bo_set_gap_symbol(g)


def _check(rc: int):
    if rc == PANIC:
        raise GoPanic("reference would panic on this input")
    if rc == NOMEM:
        raise MemoryError("oracle out of memory")


def LinearGap(s: bytes, t: bytes, a: Aligner, gap: int) -> int:
    sc = C.c_int64()
    _check(lib().bo_linear_gap(_buf(s), len(s), _buf(t), len(t), a._h, gap, C.byref(sc)))
    return sc.value


def ConstantGap(s: bytes, t: bytes, a: Aligner, gap: int) -> int:
    sc = C.c_int64()
    _check(lib().bo_constant_gap(_buf(s), len(s), _buf(t), len(t), a._h, gap, C.byref(sc)))
    return sc.value


def AffineGap(s: bytes, t: bytes, a: Aligner, gap_open: int, gap_extend: int) -> int:
    sc = C.c_int64()
    _check(lib().bo_affine_gap(_buf(s), len(s), _buf(t), len(t), a._h, gap_open, gap_extend, C.byref(sc)))
    return sc.value


def _nil(b: bytes):
    """Go returns nil slices for empty traces."""
    return b if len(b) else None


def _run3(fn, s1: bytes, s2: bytes, a: Aligner, indel: int):
    sc = C.c_int64()
    cap = len(s1) + len(s2) + 1
    t1 = (C.c_uint8 * cap)()
    t2 = (C.c_uint8 * cap)()
    tl = C.c_size_t()
    _check(fn(_buf(s1), len(s1), _buf(s2), len(s2), a._h, indel, C.byref(sc), t1, t2, C.byref(tl)))
    return sc.value, _nil(bytes(t1[: tl.value])), _nil(bytes(t2[: tl.value]))


def AlignGlobal(s1: bytes, s2: bytes, a: Aligner, indel: int):
    return _run3(lib().bo_align_global, s1, s2, a, indel)


def AlignLocal(s1: bytes, s2: bytes, a: Aligner, indel: int):
    return _run3(lib().bo_align_local, s1, s2, a, indel)


def AlignGlobalLS_as_written(s1: bytes, s2: bytes, a: Aligner, indel: int):
    return _run3(lib().bo_align_global_ls_as_written, s1, s2, a, indel)


def AlignPair(mode: str, s1: bytes, s2: bytes, a: Aligner, indel: int):
    if mode not in MODES:
        return -1, None, None  # align_pair.go:14-15
    sc = C.c_int64()
    cap = len(s1) + len(s2) + 1
    t1 = (C.c_uint8 * cap)()
    t2 = (C.c_uint8 * cap)()
    l1, l2 = C.c_size_t(), C.c_size_t()
    _check(lib().bo_align_pair(MODES[mode], _buf(s1), len(s1), _buf(s2), len(s2), a._h, indel,
                               C.byref(sc), t1, C.byref(l1), t2, C.byref(l2)))
    return sc.value, _nil(bytes(t1[: l1.value])), _nil(bytes(t2[: l2.value]))


def AlignAffine(mode: str, s1: bytes, s2: bytes, a: Aligner, gap_open: int, gap_extend: int):
    if mode not in MODES:
        return -1, None, None
    sc = C.c_int64()
    cap = len(s1) + len(s2) + 1
    t1 = (C.c_uint8 * cap)()
    t2 = (C.c_uint8 * cap)()
    tl = C.c_size_t()
    _check(lib().bo_align_affine(MODES[mode], _buf(s1), len(s1), _buf(s2), len(s2), a._h,
                                 gap_open, gap_extend, C.byref(sc), t1, t2, C.byref(tl)))
    return sc.value, _nil(bytes(t1[: tl.value])), _nil(bytes(t2[: tl.value]))


def AffineGlobalScoreF64(s1: bytes, s2: bytes, a: Aligner, gap_open: float, gap_extend: float) -> float:
    sc = C.c_double()
    _check(lib().bo_affine_global_score_f64(_buf(s1), len(s1), _buf(s2), len(s2), a._h,
                                            gap_open, gap_extend, C.byref(sc)))
    return sc.value


def lastColumnScores(s1: bytes, s2: bytes, a: Aligner, indel_signed: int):
    n = len(s1)
    o0 = np.zeros(n + 1, dtype=np.int64)
    o1 = np.zeros(n + 1, dtype=np.int64)
    _check(lib().bo_last_column_scores(_buf(s1), n, _buf(s2), len(s2), a._h, indel_signed,
                                       o0.ctypes.data_as(C.POINTER(C.c_int64)),
                                       o1.ctypes.data_as(C.POINTER(C.c_int64))))
    return o0, o1


KIND_GLOBAL, KIND_LOCAL, KIND_PAIR, KIND_AFFINE = 0, 1, 2, 3


def batch(kind: int, mode: str, s1: np.ndarray, s1_off: np.ndarray, s2: np.ndarray, s2_off: np.ndarray,
          a: Aligner, gap_open: int, gap_extend: int = 0, threads: int = 1):
    """Aligns every pair; returns (scores, t1, t2, tlen, cells).

    Pair p's traces are t1[s1_off[p]+s2_off[p] : ... + tlen[p]] (left aligned in
    a slot of capacity n_p+m_p), the same slot convention as the product C ABI.
    """
    s1 = np.ascontiguousarray(s1, dtype=np.uint8)
    s2 = np.ascontiguousarray(s2, dtype=np.uint8)
    s1_off = np.ascontiguousarray(s1_off, dtype=np.uint64)
    s2_off = np.ascontiguousarray(s2_off, dtype=np.uint64)
    n_pairs = len(s1_off) - 1
    cap = int(s1_off[-1] + s2_off[-1]) + 1
    scores = np.zeros(n_pairs, dtype=np.int64)
    t1 = np.zeros(cap, dtype=np.uint8)
    t2 = np.zeros(cap, dtype=np.uint8)
    tlen = np.zeros(n_pairs, dtype=np.uint64)
    u8p = C.POINTER(C.c_uint8)
    s1p = s1.ctypes.data_as(u8p) if s1.size else C.cast(_buf(b""), u8p)
    s2p = s2.ctypes.data_as(u8p) if s2.size else C.cast(_buf(b""), u8p)
    rc = lib().bo_batch(kind, MODES[mode], s1p, s1_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                        s2p, s2_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                        n_pairs, a._h, gap_open, gap_extend, threads,
                        scores.ctypes.data_as(C.POINTER(C.c_int64)), t1.ctypes.data_as(u8p),
                        t2.ctypes.data_as(u8p), tlen.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc < 0:
        _check(int(rc))
        raise RuntimeError(f"oracle batch failed: {rc}")
    return scores, t1, t2, tlen, int(rc)


def ReadFASTA(text: bytes):
    """fasta.go:55-84 -> list of (header, seq) byte strings; raises ValueError("No header") as the reference errors."""
    text = bytes(text)
    n = len(text)
    lines = text.count(b"\n") + 2
    buf = _buf(text)
    seq = (C.c_uint8 * max(n, 1))()
    off = (C.c_uint64 * (lines + 1))()
    hb = (C.c_uint64 * (lines + 1))()
    hl = (C.c_uint64 * (lines + 1))()
    k = lib().bo_read_fasta(buf, n, seq, off, hb, hl)
    if k < 0:
        raise ValueError("No header")
    raw = bytes(seq)
    return [(text[hb[i]:hb[i] + hl[i]], raw[off[i]:off[i + 1]]) for i in range(k)]


def DistanceMatrix(l, f):
    """dna_list.go:955-969 (= str.go:454-469), statement by statement: d[0] is made first (an empty list
    panics there), then every row i >= 1 calls f(l[i], l[j]) for j < i and mirrors the value."""
    n = len(l)
    if n == 0:
        raise GoPanic("index out of range [0] with length 0")
    d = [None] * n
    d[0] = [0.0] * n
    for i in range(1, n):
        li = l[i]
        di = [0.0] * n
        d[i] = di
        for j, lj in enumerate(l[:i]):
            dij = float(f(li, lj))
            di[j] = dij
            d[j][i] = dij
    return d


class FASTAReader:
    """fasta.go:93-142 restated, with bufio.Reader.ReadLine (Go standard library, bufio.go: ReadLine / ReadSlice)
    over a buffer of `size` bytes (bufio's default 4096, fasta.go:101).  ReadSeq() -> (header, seq, err) with err
    None, "EOF" or "FASTA line too long"."""

    def __init__(self, data: bytes, size: int = 4096):
        self.data, self.pos, self.size = bytes(data), 0, size
        self.nextHeader = b""

    def _read_line(self):
        """-> (line, isPrefix, err)"""
        data, p, B = self.data, self.pos, self.size
        if p >= len(data):
            return None, False, "EOF"
        window = data[p:p + B]
        k = window.find(b"\n")
        if k >= 0:  # ReadSlice found the delimiter: drop "\n" or "\r\n"
            self.pos = p + k + 1
            line = window[:k + 1]
            drop = 2 if len(line) > 1 and line[-2] == 13 else 1
            return line[:-drop], False, None
        if len(window) == B:  # ErrBufferFull: the start of a line that does not fit
            if window[-1] == 13:  # "put the '\r' back on buf and drop it from line"
                self.pos = p + B - 1
                return window[:-1], True, None
            self.pos = p + B
            return window, True, None
        self.pos = len(data)  # the last line, without a terminator: returned as it is
        return window, False, None

    def ReadSeq(self):
        header, seq = self.nextHeader, b""
        self.nextHeader = b""
        while True:
            line, is_pre, err = self._read_line()
            if err is not None:
                if err == "EOF" and header > b"":
                    err = None
                break
            if is_pre:
                err = "FASTA line too long"
                break
            if len(line) == 0:
                continue
            if line[0] == 62:  # '>'
                if header > b"":
                    self.nextHeader = line
                    break
                header = line
            elif header > b"":
                seq += line
            # else: initial data without a header is ignored
        return header, seq, err


def AlignLong(ss):
    """align_long.go:17-90 restated statement by statement (strings as bytes).  No test of the reference covers
    it: parity unpinned beyond this reading."""
    ss = [bytes(s) for s in ss]
    if len(ss) == 0:
        return b""
    if len(ss) == 1:
        return ss[0]
    half = len(ss[0]) // 2
    overlaps = []
    for si, s in enumerate(ss):
        s_suf = s[len(s) - half:]
        for ti, t in enumerate(ss):
            if t == s:
                continue
            x = t.find(s_suf)
            if x >= 0:
                overlaps.append([si, ti, x])
    return _greedy_merge(ss, overlaps, half)


def _greedy_merge(ss, overlaps, half):
    """align_long.go:46-89"""
    left = 0
    for _ in range(len(ss), 1, -1):
        x_max, o_max = -1, 0
        for i, o in enumerate(overlaps):
            s, t = ss[o[0]], ss[o[1]]
            if s == b"" or t == b"" or t == s:
                continue
            if o[2] > x_max:
                x_max, o_max = o[2], i
        if x_max == -1:
            return b""
        left, right = overlaps[o_max][0], overlaps[o_max][1]
        ss[left] = ss[left] + ss[right][x_max + half:]
        ss[right] = b""
        i = 0
        while i < len(overlaps):
            if overlaps[i][0] == left:
                last = len(overlaps) - 1
                if i < last:
                    overlaps[i] = overlaps[last]
                del overlaps[last]
                continue
            if overlaps[i][0] == right:
                overlaps[i][0] = left
            i += 1
    return ss[left]


def overlap_end_cells(ss, pairs, a: Aligner, indel: int):
    """Score and end column of AlignPair("overlap", ss[si], ss[ti]) (align_pair.go:184-195): the end column is the
    number of symbols of t in the trace (the walk ends in column 0, align_pair.go:186-188)."""
    scores, j_end = [], []
    g = bytes([lib().bo_get_gap_symbol()])
    for si, ti in pairs:
        sc, t1, t2 = AlignPair("overlap", ss[si], ss[ti], a, indel)
        scores.append(sc)
        j_end.append(0 if t2 is None else len(t2) - t2.count(g))
    return np.array(scores, dtype=np.int64), np.array(j_end, dtype=np.int64)


def AlignLongOverlap(ss, a: Aligner, indel: int, min_score_per_column: float = 0.8):
    """AlignLong with step 1 replaced by overlap alignments (no counterpart in the reference:
    this is the checker of the product's bio_amd.AlignLongOverlap, built on the oracle's AlignPair)."""
    ss = [bytes(s) for s in ss]
    if len(ss) == 0:
        return b""
    if len(ss) == 1:
        return ss[0]
    half = len(ss[0]) // 2
    pairs = [(si, ti) for si in range(len(ss)) for ti in range(len(ss)) if ss[ti] != ss[si]]
    overlaps = []
    if pairs:
        scores, j_end = overlap_end_cells(ss, pairs, a, indel)
        for (si, ti), sc, j in zip(pairs, scores.tolist(), j_end.tolist()):
            if j >= half and j > 0 and sc >= min_score_per_column * j:
                overlaps.append([si, ti, j - half])
    return _greedy_merge(ss, overlaps, half)
