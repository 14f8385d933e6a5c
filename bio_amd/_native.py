"""ctypes binding of the C ABI in include/bioalign.h (bio_amd/lib/libbioalign.so).

There is no fallback of any kind: if the HIP library is missing or no GPU is
usable, importing works but every alignment call raises.
"""
from __future__ import annotations

import ctypes as C
import weakref
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BIOALIGN_LIB") or os.path.join(_HERE, "lib", "libbioalign.so")  # override: A/B builds

GLOBAL, LOCAL, FITTING, OVERLAP = 0, 1, 2, 3
MODES = {"global": GLOBAL, "local": LOCAL, "fitting": FITTING, "overlap": OVERLAP}
SCORE_ONLY = 1
DENSE_TRACES = 2  # plans: traces back to back in pair order instead of in per-pair slots
TRACE_OPS = 4     # compact move records in t1 instead of the two gapped strings (include/bioalign.h)
END_CELLS = 8     # with SCORE_ONLY: t_off / t_len receive the end cell (row, column) of every pair
EINVAL, ENODEVICE, ENOMEM, EHIP, EUNSUPPORTED = -1, -2, -3, -4, -5

# every symbol include/bioalign.h declares
EXPORTS = (
    "bioalign_abi_version",
    "bioalign_create",
    "bioalign_destroy",
    "bioalign_last_error",
    "bioalign_device_count",
    "bioalign_batch",
    "bioalign_long",
    "bioalign_plan_create",
    "bioalign_plan_destroy",
    "bioalign_plan_workspace_bytes",
    "bioalign_plan_cells",
    "bioalign_plan_run",
    "bioalign_plan_share_pack",
    "bioalign_plan_dp_ms",
    "bioalign_plan_kernel",
    "bioalign_run_noalloc",
    "bioalign_ops_capacity",
    "bioalign_expand_ops",
    "bioalign_fasta_parse",
    "bioalign_fasta_parse_reader",
    "bioalign_fasta_too_long_at",
    "bioalign_fasta_destroy",
    "bioalign_fasta_count",
    "bioalign_fasta_bytes",
    "bioalign_fasta_kernel_ms",
    "bioalign_fasta_device_seq",
    "bioalign_fasta_device_off",
    "bioalign_fasta_off",
    "bioalign_fasta_header_begin",
    "bioalign_fasta_header_len",
    "bioalign_fasta_copy_seq",
    "bioalign_all_pairs",
    "bioalign_all_pairs_range",
    "bioalign_all_ordered_pairs",
    "bioalign_fasta_all_pairs",
    "bioalign_stripe_create",
    "bioalign_stripe_destroy",
    "bioalign_stripe_rows",
    "bioalign_stripe_dp",
    "bioalign_stripe_blocks",
    "bioalign_stripe_block_rows",
    "bioalign_stripe_dp_block",
    "bioalign_stripe_end_cell",
    "bioalign_stripe_walk",
)


class BioAlignError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"bioalign error {code}: {msg}")
        self.code = code


class Batch(C.Structure):
    _fields_ = [
        ("s1", C.c_void_p),
        ("s1_off", C.c_void_p),
        ("s2", C.c_void_p),
        ("s2_off", C.c_void_p),
        ("n_pairs", C.c_uint64),
        ("mode", C.c_int32),
        ("score_table", C.c_void_p),
        ("na", C.c_int32),
        ("nb", C.c_int32),
        ("code1", C.c_void_p),
        ("code2", C.c_void_p),
        ("gap_open", C.c_int32),
        ("gap_extend", C.c_int32),
        ("gap_symbol", C.c_uint8),
        ("flags", C.c_uint32),
    ]


class Result(C.Structure):
    _fields_ = [
        ("scores", C.c_void_p),
        ("t1", C.c_void_p),
        ("t2", C.c_void_p),
        ("t_off", C.c_void_p),
        ("t_len", C.c_void_p),
    ]


_lib = None


def lib():
    """Loads the HIP library; raises (loudly) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950); "
                "bio_amd has no CPU fallback"
            )
        # PyTorch-ROCm ships its own HIP runtime: a process that initialises the system one first (through this
        # library) and imports torch afterwards finds "No HIP GPUs" in torch.  Let torch load first when it is there.
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        L.bioalign_abi_version.restype = C.c_int
        L.bioalign_create.restype = C.c_void_p
        L.bioalign_create.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.bioalign_destroy.argtypes = [C.c_void_p]
        L.bioalign_last_error.restype = C.c_char_p
        L.bioalign_last_error.argtypes = [C.c_void_p]
        L.bioalign_device_count.restype = C.c_int
        L.bioalign_device_count.argtypes = [C.c_void_p]
        L.bioalign_batch.restype = C.c_int
        L.bioalign_batch.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(Result)]
        L.bioalign_long.restype = C.c_int
        L.bioalign_long.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(Result), C.POINTER(C.c_float)]
        L.bioalign_plan_create.restype = C.c_void_p
        L.bioalign_plan_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(Batch)]
        L.bioalign_plan_destroy.argtypes = [C.c_void_p]
        L.bioalign_plan_workspace_bytes.restype = C.c_uint64
        L.bioalign_plan_workspace_bytes.argtypes = [C.c_void_p]
        L.bioalign_plan_cells.restype = C.c_uint64
        L.bioalign_plan_cells.argtypes = [C.c_void_p]
        L.bioalign_plan_run.restype = C.c_int
        L.bioalign_plan_run.argtypes = [C.c_void_p] + [C.c_void_p] * 9 + [C.c_void_p]
        L.bioalign_plan_share_pack.restype = C.c_int
        L.bioalign_plan_share_pack.argtypes = [C.c_void_p, C.c_void_p]
        L.bioalign_ops_capacity.restype = C.c_uint64
        L.bioalign_ops_capacity.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.bioalign_expand_ops.restype = C.c_int
        L.bioalign_expand_ops.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint8,
                                          C.c_void_p, C.c_void_p]
        L.bioalign_fasta_parse.restype = C.c_void_p
        L.bioalign_fasta_parse.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_int]
        L.bioalign_fasta_parse_reader.restype = C.c_void_p
        L.bioalign_fasta_parse_reader.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64]
        L.bioalign_fasta_too_long_at.restype = C.c_int64
        L.bioalign_fasta_too_long_at.argtypes = [C.c_void_p]
        L.bioalign_fasta_destroy.argtypes = [C.c_void_p]
        for fn in (L.bioalign_fasta_count, L.bioalign_fasta_bytes):
            fn.restype = C.c_uint64
            fn.argtypes = [C.c_void_p]
        L.bioalign_fasta_kernel_ms.restype = C.c_float
        L.bioalign_fasta_kernel_ms.argtypes = [C.c_void_p]
        for fn in (L.bioalign_fasta_device_seq, L.bioalign_fasta_device_off, L.bioalign_fasta_off,
                   L.bioalign_fasta_header_begin, L.bioalign_fasta_header_len):
            fn.restype = C.c_void_p
            fn.argtypes = [C.c_void_p]
        L.bioalign_fasta_copy_seq.restype = C.c_int
        L.bioalign_fasta_copy_seq.argtypes = [C.c_void_p, C.c_void_p]
        L.bioalign_all_pairs.restype = C.c_int
        L.bioalign_all_pairs.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p]
        L.bioalign_all_pairs_range.restype = C.c_int
        L.bioalign_all_pairs_range.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_uint64, C.c_uint64, C.c_void_p]
        L.bioalign_all_ordered_pairs.restype = C.c_int
        L.bioalign_all_ordered_pairs.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p]
        L.bioalign_fasta_all_pairs.restype = C.c_int
        L.bioalign_fasta_all_pairs.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p]
        L.bioalign_stripe_create.restype = C.c_void_p
        L.bioalign_stripe_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(Batch), C.c_uint64, C.c_uint64]
        L.bioalign_stripe_destroy.argtypes = [C.c_void_p]
        L.bioalign_stripe_rows.restype = C.c_uint64
        L.bioalign_stripe_rows.argtypes = [C.c_void_p]
        L.bioalign_stripe_dp.restype = C.c_int
        L.bioalign_stripe_dp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.bioalign_stripe_blocks.restype = C.c_uint32
        L.bioalign_stripe_blocks.argtypes = [C.c_void_p]
        L.bioalign_stripe_block_rows.restype = C.c_int
        L.bioalign_stripe_block_rows.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.bioalign_stripe_dp_block.restype = C.c_int
        L.bioalign_stripe_dp_block.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.bioalign_stripe_end_cell.restype = C.c_int
        L.bioalign_stripe_end_cell.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]
        L.bioalign_stripe_walk.restype = C.c_int
        L.bioalign_stripe_walk.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.bioalign_run_noalloc.restype = C.c_int
        L.bioalign_run_noalloc.argtypes = [C.c_void_p, C.c_int, C.POINTER(Batch)] + [C.c_void_p] * 9 + [C.c_void_p]
        L.bioalign_plan_kernel.restype = C.c_char_p
        L.bioalign_plan_kernel.argtypes = [C.c_void_p]
        L.bioalign_plan_dp_ms.restype = C.c_int
        L.bioalign_plan_dp_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data if a is not None and a.size else 0


class PackedBatch:
    """Host-side batch in the C ABI's layout; keeps every numpy buffer alive."""

    def __init__(self, s1, s1_off, s2, s2_off, mode, score_table, code1, code2,
                 gap_open, gap_extend, gap_symbol=ord("-"), flags=0):
        self.s1 = np.ascontiguousarray(s1, dtype=np.uint8)
        self.s2 = np.ascontiguousarray(s2, dtype=np.uint8)
        self.s1_off = np.ascontiguousarray(s1_off, dtype=np.uint64)
        self.s2_off = np.ascontiguousarray(s2_off, dtype=np.uint64)
        self.table = np.ascontiguousarray(score_table, dtype=np.int32)
        if self.table.ndim != 2:
            raise ValueError("score_table must be 2-D [na, nb]")
        self.code1 = np.ascontiguousarray(code1, dtype=np.uint8)
        self.code2 = np.ascontiguousarray(code2, dtype=np.uint8)
        if self.code1.size != 256 or self.code2.size != 256:
            raise ValueError("code maps must have 256 entries")
        if len(self.s1_off) != len(self.s2_off) or len(self.s1_off) < 1:
            raise ValueError("offset arrays must have n_pairs+1 entries each")
        self.n_pairs = len(self.s1_off) - 1
        self.mode = MODES[mode] if isinstance(mode, str) else int(mode)
        # 1-byte placeholders so that empty inputs still pass non-null pointers
        self._e = np.zeros(1, dtype=np.uint8)
        b = Batch()
        b.s1 = _ptr(self.s1) or self._e.ctypes.data
        b.s2 = _ptr(self.s2) or self._e.ctypes.data
        b.s1_off = self.s1_off.ctypes.data
        b.s2_off = self.s2_off.ctypes.data
        b.n_pairs = self.n_pairs
        b.mode = self.mode
        b.score_table = self.table.ctypes.data
        b.na, b.nb = self.table.shape
        b.code1 = self.code1.ctypes.data
        b.code2 = self.code2.ctypes.data
        b.gap_open = int(gap_open)
        b.gap_extend = int(gap_extend)
        b.gap_symbol = int(gap_symbol)
        b.flags = int(flags)
        self.c = b

    @property
    def trace_capacity(self) -> int:
        return int(self.s1_off[-1] + self.s2_off[-1])


class Engine:
    """Owns a bioalign_ctx."""

    def __init__(self, device_ids=None):
        L = lib()
        if device_ids:
            arr = (C.c_int * len(device_ids))(*device_ids)
            h = L.bioalign_create(arr, len(device_ids))
        else:
            h = L.bioalign_create(None, 0)
        if not h:
            raise BioAlignError(ENODEVICE, L.bioalign_last_error(None).decode())
        self._h = C.c_void_p(h)
        self._children = weakref.WeakSet()  # plans, stripes, fasta objects: they go before the context (bioalign.h)

    def close(self):
        if getattr(self, "_h", None):
            for child in list(getattr(self, "_children", ())):
                child.close()
            lib().bioalign_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _raise(self, code: int):
        raise BioAlignError(code, lib().bioalign_last_error(self._h).decode())

    def device_count(self) -> int:
        return lib().bioalign_device_count(self._h)

    def long(self, pb: PackedBatch):
        """bioalign_long on host buffers -> (scores, t1, t2, t_off, t_len); self.last_kernel_ms is set."""
        return self.batch(pb, long=True)

    def batch(self, pb: PackedBatch, long: bool = False, out=None):
        """bioalign_batch on host buffers -> (scores, t1, t2, t_off, t_len).  `out`: the tuple a previous call on a
        batch of the same size returned, to write into the same buffers again."""
        n = pb.n_pairs
        score_only = bool(pb.c.flags & SCORE_ONLY)
        cap = pb.trace_capacity
        if out is not None:
            scores, t1, t2, t_off, t_len = out
            need1 = int(lib().bioalign_ops_capacity(n, int(pb.s1_off[-1]), int(pb.s2_off[-1]))) if pb.c.flags & TRACE_OPS else cap
            need2 = 0 if pb.c.flags & TRACE_OPS else cap
            if len(scores) < n or len(t1) < need1 or len(t2) < need2 or len(t_off) < n or len(t_len) < n:
                raise ValueError("output buffers too small for this batch")
        else:
            scores = np.zeros(max(n, 1), dtype=np.int64)
            if pb.c.flags & TRACE_OPS:  # compact records: t1 only, sized by the library's rule
                cap1 = int(lib().bioalign_ops_capacity(n, int(pb.s1_off[-1]), int(pb.s2_off[-1])))
                t1 = np.zeros(cap1 + 1, dtype=np.uint8)
                t2 = np.zeros(1, dtype=np.uint8)
            else:
                t1 = np.zeros(cap + 1, dtype=np.uint8)
                t2 = np.zeros(cap + 1, dtype=np.uint8)
            t_off = np.zeros(max(n, 1), dtype=np.uint64)
            t_len = np.zeros(max(n, 1), dtype=np.uint64)
        r = Result()
        r.scores = scores.ctypes.data
        if score_only and pb.c.flags & END_CELLS:
            r.t_off, r.t_len = t_off.ctypes.data, t_len.ctypes.data
        if not score_only:
            r.t1, r.t2 = t1.ctypes.data, t2.ctypes.data
            r.t_off, r.t_len = t_off.ctypes.data, t_len.ctypes.data
        if long:
            ms = C.c_float()
            rc = lib().bioalign_long(self._h, C.byref(pb.c), C.byref(r), C.byref(ms))
            self.last_kernel_ms = ms.value
        else:
            rc = lib().bioalign_batch(self._h, C.byref(pb.c), C.byref(r))
        if rc != 0:
            self._raise(rc)
        return scores[:n], t1, t2, t_off[:n], t_len[:n]

    def all_pairs(self, pb: PackedBatch, fasta: "Fasta | None" = None, pair_range=None) -> np.ndarray:
        """bioalign_all_pairs: `pb` describes a LIST (its s1 side), or only mode / scores / gaps when the list is
        the records of `fasta`; -> scores[i*(i-1)/2 + j] of aligning sequence i with sequence j < i.
        pair_range=(first, end): only those pairs of that numbering (bioalign_all_pairs_range)."""
        n = fasta.count if fasta is not None else pb.n_pairs
        total = n * (n - 1) // 2
        first, end = (0, total) if pair_range is None else (int(pair_range[0]), int(pair_range[1]))
        scores = np.zeros(max(end - first, 1), dtype=np.int64)
        if fasta is not None:
            if pair_range is not None:
                raise ValueError("pair ranges are for host lists")
            rc = lib().bioalign_fasta_all_pairs(fasta._h, C.byref(pb.c), scores.ctypes.data)
        else:
            rc = lib().bioalign_all_pairs_range(self._h, C.byref(pb.c), first, end - first, scores.ctypes.data)
        if rc != 0:
            self._raise(rc)
        return scores[:end - first]

    def all_ordered_pairs(self, pb: PackedBatch, end_cells: bool = True):
        """bioalign_all_ordered_pairs over the list `pb` describes (its s1 side) -> (scores, end_i, end_j), pair
        number i*(n-1) + (j if j < i else j-1); end_i / end_j are None without end_cells."""
        n = pb.n_pairs
        total = n * (n - 1)
        scores = np.zeros(max(total, 1), dtype=np.int64)
        ei = np.zeros(max(total, 1), dtype=np.uint64) if end_cells else None
        ej = np.zeros(max(total, 1), dtype=np.uint64) if end_cells else None
        rc = lib().bioalign_all_ordered_pairs(self._h, C.byref(pb.c), scores.ctypes.data,
                                              ei.ctypes.data if end_cells else None, ej.ctypes.data if end_cells else None)
        if rc != 0:
            self._raise(rc)
        return scores[:total], (ei[:total] if end_cells else None), (ej[:total] if end_cells else None)

    def run_noalloc(self, pb: PackedBatch, d_s1, d_s1_off, d_s2, d_s2_off, d_scores, d_t1, d_t2, d_t_off, d_t_len,
                    stream=0, device_index: int = 0):
        """bioalign_run_noalloc: plan + run in one call on device-resident data (raw device addresses), the whole
        workspace inside the library's fixed scratch array -- no allocation of any kind.  Returns when done."""
        rc = lib().bioalign_run_noalloc(self._h, device_index, C.byref(pb.c), d_s1, d_s1_off, d_s2, d_s2_off, d_scores,
                                        d_t1, d_t2, d_t_off, d_t_len, stream)
        if rc != 0:
            self._raise(rc)

    def plan(self, pb: PackedBatch, device_index: int = 0) -> "Plan":
        return Plan(self, pb, device_index)


class Plan:
    """Owns a bioalign_plan (device workspace + bundle tables)."""

    def __init__(self, engine: Engine, pb: PackedBatch, device_index: int = 0):
        self.engine = engine
        h = lib().bioalign_plan_create(engine._h, device_index, C.byref(pb.c))
        if not h:
            raise BioAlignError(EINVAL, lib().bioalign_last_error(engine._h).decode())
        self._h = C.c_void_p(h)
        engine._children.add(self)
        self.n_pairs = pb.n_pairs

    def close(self):
        if getattr(self, "_h", None):
            lib().bioalign_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self) -> int:
        return int(lib().bioalign_plan_workspace_bytes(self._h))

    @property
    def cells(self) -> int:
        return int(lib().bioalign_plan_cells(self._h))

    def run(self, d_s1, d_s1_off, d_s2, d_s2_off, d_scores, d_t1, d_t2, d_t_off, d_t_len, stream=0):
        """All arguments are raw device addresses (ints), e.g. torch tensors' data_ptr()."""
        rc = lib().bioalign_plan_run(self._h, d_s1, d_s1_off, d_s2, d_s2_off, d_scores,
                                     d_t1, d_t2, d_t_off, d_t_len, stream)
        if rc != 0:
            self.engine._raise(rc)

    def share_pack(self, packed_by: "Plan | None"):
        """Read the packed sequence words `packed_by` (a plan of the same batch, other mode) makes instead of
        packing again; every run of this plan must then follow a run of `packed_by` on the same stream."""
        rc = lib().bioalign_plan_share_pack(self._h, packed_by._h if packed_by is not None else None)
        if rc != 0:
            self.engine._raise(rc)
        self._packed_by = packed_by  # keeps it alive

    @property
    def kernel(self) -> str:
        """the family of DP kernels the planner chose: linear, bitparallel, affine, wide"""
        return lib().bioalign_plan_kernel(self._h).decode()

    def dp_ms(self):
        """(summed ms, launches) of the DP kernel since the previous call; sync the stream first."""
        ms, n = C.c_float(), C.c_int()
        rc = lib().bioalign_plan_dp_ms(self._h, C.byref(ms), C.byref(n))
        if rc != 0:
            self.engine._raise(rc)
        return ms.value, n.value


class Stripe:
    """Owns a bioalign_stripe: columns [col_begin, col_end) of the matrix of ONE long pair, on one device of the
    engine (include/bioalign.h).  `pb` is a PackedBatch of exactly that pair."""

    def __init__(self, engine: Engine, pb: PackedBatch, col_begin: int, col_end: int, device_index: int = 0):
        self.engine = engine
        self._pb = pb
        self.col_begin, self.col_end = int(col_begin), int(col_end)
        h = lib().bioalign_stripe_create(engine._h, device_index, C.byref(pb.c), self.col_begin, self.col_end)
        if not h:
            engine._raise(EINVAL)
        self._h = C.c_void_p(h)
        engine._children.add(self)
        self.rows = int(lib().bioalign_stripe_rows(self._h))
        self.width = self.col_end - self.col_begin

    def close(self):
        if getattr(self, "_h", None):
            lib().bioalign_stripe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dp(self, d_left_col, d_right_col):
        """Device addresses (ints or None) of rows+1 int32 each; returns when the stripe's DP is done."""
        rc = lib().bioalign_stripe_dp(self._h, d_left_col, d_right_col)
        if rc != 0:
            self.engine._raise(rc)

    def blocks(self):
        """Row blocks of the stripe's DP: [(row_begin, row_end), ...], the same in every stripe of the pair."""
        out = []
        for k in range(int(lib().bioalign_stripe_blocks(self._h))):
            lo, hi = C.c_uint64(), C.c_uint64()
            if lib().bioalign_stripe_block_rows(self._h, k, C.byref(lo), C.byref(hi)) != 0:
                self.engine._raise(EINVAL)
            out.append((lo.value, hi.value))
        return out

    def dp_block(self, k, d_left_rows, d_right_rows):
        """Row block k (in order): device addresses of row_end - row_begin int32 each; returns when the block is done."""
        rc = lib().bioalign_stripe_dp_block(self._h, int(k), d_left_rows, d_right_rows)
        if rc != 0:
            self.engine._raise(rc)

    def end_cell(self):
        """(true score, column inside the stripe, score to report) of the end cell this stripe offers."""
        v, j, sc = C.c_int64(), C.c_uint64(), C.c_int64()
        rc = lib().bioalign_stripe_end_cell(self._h, C.byref(v), C.byref(j), C.byref(sc))
        if rc != 0:
            self.engine._raise(rc)
        return v.value, j.value, sc.value

    def walk(self, i: int, j_local: int):
        """-> (piece of t1, piece of t2, exit_i, exit_j)"""
        cap = self.rows + self.width + 1
        t1 = np.zeros(cap, dtype=np.uint8)
        t2 = np.zeros(cap, dtype=np.uint8)
        ln, ei, ej = C.c_uint64(), C.c_uint64(), C.c_uint64()
        rc = lib().bioalign_stripe_walk(self._h, int(i), int(j_local), t1.ctypes.data, t2.ctypes.data,
                                        C.byref(ln), C.byref(ei), C.byref(ej))
        if rc != 0:
            self.engine._raise(rc)
        return t1[:ln.value].tobytes(), t2[:ln.value].tobytes(), ei.value, ej.value


class Fasta:
    """Owns a bioalign_fasta: FASTA text parsed on the GPU into the batch format (include/bioalign.h).
    `text` is bytes / a uint8 array on the host, or a raw device address (int) with length `length`."""

    def __init__(self, engine: Engine, text, length: int | None = None, device_index: int = 0, reader_line_limit=None):
        """reader_line_limit: None = ReadFASTA's rules; an int = FASTAReader.ReadSeq's rules with a bufio buffer of
        that many bytes (0: bufio's default), bioalign_fasta_parse_reader."""
        self.engine = engine
        if isinstance(text, int):
            ptr, n, on_dev = text, int(length), 1
            self._keep = None
        else:
            arr = np.frombuffer(bytes(text), dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, dtype=np.uint8)
            self._keep = arr
            ptr, n, on_dev = (arr.ctypes.data if arr.size else None), int(arr.size), 0
        if reader_line_limit is None:
            h = lib().bioalign_fasta_parse(engine._h, device_index, ptr, n, on_dev)
        else:
            h = lib().bioalign_fasta_parse_reader(engine._h, device_index, ptr, n, on_dev, int(reader_line_limit))
        if not h:
            engine._raise(EINVAL)
        self._h = C.c_void_p(h)
        engine._children.add(self)
        L = lib()
        self.too_long_at = int(L.bioalign_fasta_too_long_at(self._h))
        self.count = int(L.bioalign_fasta_count(self._h))
        self.total_bytes = int(L.bioalign_fasta_bytes(self._h))
        self.kernel_ms = float(L.bioalign_fasta_kernel_ms(self._h))
        self.d_seq = int(L.bioalign_fasta_device_seq(self._h) or 0)
        self.d_off = int(L.bioalign_fasta_device_off(self._h) or 0)

        def host(fn, k):
            p = fn(self._h)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(k,)).copy() if k else np.zeros(0, np.uint64)

        self.off = host(L.bioalign_fasta_off, self.count + 1)
        self.header_begin = host(L.bioalign_fasta_header_begin, self.count)
        self.header_len = host(L.bioalign_fasta_header_len, self.count)

    def seq_bytes(self) -> np.ndarray:
        out = np.zeros(max(self.total_bytes, 1), dtype=np.uint8)
        rc = lib().bioalign_fasta_copy_seq(self._h, out.ctypes.data)
        if rc != 0:
            self.engine._raise(rc)
        return out[:self.total_bytes]

    def close(self):
        if getattr(self, "_h", None):
            lib().bioalign_fasta_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def expand_ops(s1: bytes, s2: bytes, t1: np.ndarray, t_off: int, moves: int, gap_symbol: int = ord("-")):
    """One pair's compact record (TRACE_OPS) -> the reference's two gapped strings (bioalign_expand_ops)."""
    a = np.frombuffer(bytes(s1) or b"\0", dtype=np.uint8)
    b = np.frombuffer(bytes(s2) or b"\0", dtype=np.uint8)
    o1 = np.zeros(max(moves, 1), dtype=np.uint8)
    o2 = np.zeros(max(moves, 1), dtype=np.uint8)
    rec = np.ascontiguousarray(t1[int(t_off):int(t_off) + 8 + 4 * ((int(moves) + 15) // 16)])
    rc = lib().bioalign_expand_ops(a.ctypes.data, len(s1), b.ctypes.data, len(s2), rec.ctypes.data, int(moves), gap_symbol,
                                   o1.ctypes.data, o2.ctypes.data)
    if rc != 0:
        raise BioAlignError(rc, "malformed compact trace record")
    return o1[:moves].tobytes(), o2[:moves].tobytes()
