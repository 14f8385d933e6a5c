"""Long mode across ranks (bio_amd/long_stripes.py): the exchange protocol on CPU with two and
three ranks over gloo, and (marked gpu) the real stripes with two ranks sharing the one GPU of the test box.

The CPU test injects a plain-Python restatement of one stripe (full matrix of its columns, the reference's
recurrences and tie order, align.go:97-143 / align_pair.go:109-216) -- test infrastructure standing in for the
GPU stripe, which the product path never does (make_stripe defaults to the HIP stripe, no CPU fallback).
Checked on rank 0: score and both gapped strings equal the oracle's AlignPair on the whole pair."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class PyStripe:
    """One column stripe in true scores; the boundary columns cross in the kernels' working space Hs = H + g*(i+j)."""

    def __init__(self, mode, s1, s2, a, gap, c0, c1, xdev):
        self.mode, self.s1, self.s2, self.g, self.c0, self.xdev = mode, s1, s2[c0:c1], gap, c0, xdev
        self.match, self.mismatch = a
        self.rows, self.width = len(s1), c1 - c0
        self.first = c0 == 0

    def _w(self, x, y):
        return self.match if x == y else self.mismatch

    def blocks(self):
        # row blocks of 16 rows (row 0 in the first one), the same on every rank
        n = self.rows
        cuts = [0] + list(range(17, n + 1, 16)) + [n + 1]
        return [(cuts[k], cuts[k + 1]) for k in range(len(cuts) - 1)]

    def dp_block(self, k, left):
        n, w, g = self.rows, self.width, self.g
        lo, hi = self.blocks()[k]
        if k == 0:
            self.H = [[0] * (w + 1) for _ in range(n + 1)]
            for j in range(1, w + 1):
                self.H[0][j] = 0 if self.mode == "fitting" else -g * (self.c0 + j)  # align_pair.go:110-112, 158, 187
        H = self.H
        for i in range(lo, hi):
            if left is None:
                H[i][0] = 0 if self.mode == "overlap" else -g * i      # align_pair.go:113, 131, 159, 188
            else:
                H[i][0] = int(left[i - lo]) - g * (i + self.c0)
        for i in range(max(lo, 1), hi):
            for j in range(1, w + 1):
                H[i][j] = max(H[i - 1][j] - g, H[i][j - 1] - g, H[i - 1][j - 1] + self._w(self.s1[i - 1], self.s2[j - 1]))
        return torch.tensor([H[i][w] + g * (i + self.c0 + w) for i in range(lo, hi)], dtype=torch.int32, device=self.xdev)

    def end_cell(self):
        n, w, H = self.rows, self.width, self.H
        if self.mode == "global":
            return H[n][w], w, H[n][w]
        best, bj = (H[n][0], 0) if self.first else (None, 0)
        for j in range(1, w + 1):  # leftmost strict maximum (align_pair.go:169-182)
            if best is None or H[n][j] > best:
                best, bj = H[n][j], j
        return best, bj, best

    def walk(self, i, j):
        H, g = self.H, self.g
        top_free, left_free = self.mode == "fitting", self.mode == "overlap"
        t1, t2 = [], []
        while i > 0 and j > 0:
            cur = H[i][j]
            if cur == H[i - 1][j] - g:                 # up first, then left, then diagonal (align.go:105-116)
                t1.append(self.s1[i - 1]); t2.append(45); i -= 1
            elif cur == H[i][j - 1] - g:
                t1.append(45); t2.append(self.s2[j - 1]); j -= 1
            else:
                t1.append(self.s1[i - 1]); t2.append(self.s2[j - 1]); i -= 1; j -= 1
        if self.first and not left_free:
            while i > 0:
                t1.append(self.s1[i - 1]); t2.append(45); i -= 1
        if not top_free and i == 0:
            while j > 0:
                t1.append(45); t2.append(self.s2[j - 1]); j -= 1
        return bytes(reversed(t1)), bytes(reversed(t2)), i, j


def _cases():
    import random
    from bio_amd import synth

    rng = random.Random(4)
    out = []
    for k, (n, m) in enumerate([(40, 57), (61, 33), (9, 80), (70, 2), (1, 1)]):
        a, _, b, _ = synth.dna_pairs(1, n, m, config=5, first=900 + k, sub=0.08, ins=0.03, dele=0.03)
        out.append((a.tobytes(), b.tobytes()))
    core = bytes(rng.choice(b"ACGT") for _ in range(30))
    junk = lambda k: bytes(rng.choice(b"ACGT") for _ in range(k))
    out += [(core, junk(40) + core), (core, core + junk(40)), (junk(25) + core, core + junk(30)), (b"A" * 20, b"C" * 50),
            (b"", b"ACGT"), (b"ACGT", b"")]
    return out


def _worker(rank, world, port, q, use_gpu):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bio_amd import long_stripes
        import bio_amd

        res = []
        for x, y in _cases():
            for mode in ("global", "fitting", "overlap"):
                if use_gpu:
                    got = long_stripes.align_long_striped(mode, x, y, bio_amd.MatchAligner(1, -1), 1)
                else:
                    got = long_stripes.align_long_striped(mode, x, y, (1, -1), 1, make_stripe=PyStripe) if x and y else \
                        ((None, None, None) if rank else ("edge", None, None))
                res.append(got)
        q.put((rank, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(world, use_gpu):
    from oracle import oracle as O

    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, use_gpu)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, res = q.get(timeout=240)
        got[rank] = res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    oa = O.MatchAligner(1, -1)
    k = 0
    for x, y in _cases():
        for mode in ("global", "fitting", "overlap"):
            want = O.AlignPair(mode, x, y, oa, 1)
            if got[0][k][0] != "edge":  # (CPU run: empty sequences go to the batch kernels, which need the GPU)
                assert got[0][k] == want, (world, mode, len(x), len(y))
                for r in range(1, world):
                    assert got[r][k][1] is None and (got[r][k][0] in (None, want[0]))
            k += 1


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_stripe_protocol_over_gloo_cpu(world):
    _run(world, use_gpu=False)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_ranks_share_one_gpu_real_stripes():
    _run(2, use_gpu=True)
