"""Long mode across the ranks of a process group (one process per GPU), SURVEY.md 8(e).

One DP matrix does not shard into independent units: it is cut into COLUMN STRIPES, rank r
holding columns [r*m/W, (r+1)*m/W) and all rows on its own GPU (bioalign_stripe_* of
include/bioalign.h).  The only data that crosses GPUs:

  1. DP: the last column of a stripe -- n+1 int32 working-space scores -- goes to the rank on
     its right in ROW BLOCKS (the stripe's blocks(): a few thousand rows each, the same on every
     rank): point-to-point isend / irecv per block, all receives posted up front (backend nccl =
     RCCL over xGMI on a GPU node; gloo in the CPU test), so that rank r works on block k while
     rank r-1 works on block k+1.
     No collective is involved in the data path.
  2. End cell: every rank offers the best end cell of its stretch of the bottom row
     (align_pair.go:169-182; global: the last rank's last cell, align.go:123-124); one
     all_gather of four integers per rank picks the leftmost best.
  3. Traceback: the walk starts on the rank that holds the end cell and is handed to the
     left neighbour at each stripe's column 0 as one (go on?, row) message; every rank
     expands its own piece of the two gapped strings.
  4. The pieces are gathered on `dst` and laid end to end.

This is the process-group form of what bioalign_long does across the devices of one context
(bio_amd/csrc/api.cpp: long_pair_striped).  At the size of BASELINE.json's configs[4]
(100k x 100k) one GPU is latency-bound, not capacity-bound: stripes are for matrices whose
traceback residues do not fit one GPU.
"""
from __future__ import annotations

import numpy as np


class GpuStripe:
    """Adapter around _native.Stripe: boundary columns travel as torch tensors on `xdev` (the device the
    process group moves tensors on: cuda for nccl, cpu for gloo)."""

    def __init__(self, mode, s1, s2, a, gap, col_begin, col_end, xdev):
        import torch
        from . import _native, pack_sequences, score_table

        d1, o1 = pack_sequences([s1])
        d2, o2 = pack_sequences([s2])
        tbl, c1, c2 = score_table(a, np.unique(d1).tolist(), np.unique(d2).tolist())
        self._torch = torch
        self.xdev = xdev
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.engine = _native.Engine([self.dev.index])
        self.pb = _native.PackedBatch(d1, o1, d2, o2, mode, tbl, c1, c2, gap_open=gap, gap_extend=gap)
        self.stripe = _native.Stripe(self.engine, self.pb, col_begin, col_end)
        self.rows, self.width = self.stripe.rows, self.stripe.width
        self._blocks = self.stripe.blocks()

    def blocks(self):
        return self._blocks

    def dp_block(self, k, left):
        """Row block k of the DP: `left` = the block's rows of the left neighbour's last column (None for the first
        stripe); returns the block's rows of this stripe's last column."""
        torch = self._torch
        lo, hi = self._blocks[k]
        right = torch.empty(hi - lo, dtype=torch.int32, device=self.dev)
        left_dev = None
        if left is not None:
            left_dev = left.to(self.dev).contiguous()
            # the library works on a stream of its own: what this stream has queued for `left` (the receive, the copy)
            # must have landed -- this stream only, not the device, and not at all for the first stripe
            torch.cuda.current_stream(self.dev).synchronize()
        self.stripe.dp_block(k, left_dev.data_ptr() if left_dev is not None else None, right.data_ptr())
        return right.to(self.xdev)

    def end_cell(self):
        return self.stripe.end_cell()

    def walk(self, i, j_local):
        return self.stripe.walk(i, j_local)


def stripe_bounds(m: int, world: int):
    """Column ranges of the stripes: min(world, m) of them, as equal as integers allow."""
    g = max(1, min(world, m))
    return [m * q // g for q in range(g + 1)]


def align_long_striped(mode: str, s1: bytes, s2: bytes, a, gap: int, *, group=None, dst: int = 0, make_stripe=None):
    """AlignPair(mode, s1, s2, a, gap) (align_pair.go:3; modes global / fitting / overlap, linear gap) with the
    matrix spread over the ranks of `group`.  Every rank calls this with the same arguments; rank `dst` returns
    (score, t1, t2) as AlignPairLong does, the others (score, None, None).

    make_stripe(mode, s1, s2, a, gap, col_begin, col_end, xdev) -> object with rows, width, blocks() -> row ranges,
    dp_block(k, left rows) -> right rows, end_cell() -> (true score, j_local, score), walk(i, j_local) ->
    (piece1, piece2, exit_i, exit_j).  The
    default is the GPU stripe; there is no CPU fallback (the CPU test injects its own restatement).
    """
    import torch
    import torch.distributed as dist

    if mode not in ("global", "fitting", "overlap"):
        raise ValueError("long mode covers global, fitting and overlap")
    s1, s2 = bytes(s1), bytes(s2)
    n, m = len(s1), len(s2)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if n == 0 or m == 0:
        # nothing to spread: the borders alone (batch kernels)
        if rank != dst:
            return None, None, None
        from .align import AlignPair
        return AlignPair(mode, s1, s2, a, gap)
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    xdev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    make_stripe = make_stripe or GpuStripe
    bounds = stripe_bounds(m, world)
    G = len(bounds) - 1
    stripe = make_stripe(mode, s1, s2, a, gap, bounds[rank], bounds[rank + 1], xdev) if rank < G else None

    # 1. DP, left to right, row block by row block: while this rank works on block k its left neighbour is on k+1.
    # Every receive is posted before the first block (the blocks are the same on every rank, so the buffers are known),
    # and a block's send is not waited for until the DP is done: a rank never stalls on a peer that is merely slow to
    # post, only on data that is not there yet.
    if stripe is not None:
        blocks = list(stripe.blocks())
        lefts, recvs, sends = [], [], []
        if rank > 0:
            for lo, hi in blocks:
                buf = torch.empty(hi - lo, dtype=torch.int32, device=xdev)
                lefts.append(buf)
                recvs.append(dist.irecv(buf, src=peer(rank - 1), group=group))
        for k, (lo, hi) in enumerate(blocks):
            left = None
            if rank > 0:
                recvs[k].wait()
                left = lefts[k]
            right = stripe.dp_block(k, left).contiguous()
            if rank + 1 < G:
                sends.append((dist.isend(right, dst=peer(rank + 1), group=group), right))  # (the buffer lives until the wait)
        for req, _ in sends:
            req.wait()

    # 2. the end cell: leftmost strict maximum over the stripes' offers (global: the last stripe's)
    offer = [0, 0, 0, 0]
    if stripe is not None and (mode != "global" or rank == G - 1):
        v, j, sc = stripe.end_cell()
        offer = [1, int(v), int(j), int(sc)]
    mine = torch.tensor(offer, dtype=torch.int64, device=xdev)
    offers = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(offers, mine, group=group)
    offers = [o.tolist() for o in offers]
    end = -1
    for q in range(G):
        if offers[q][0] and (end < 0 or offers[q][1] > offers[end][1]):
            end = q
    score, j_end = offers[end][3], offers[end][2]

    # 3. the walk, right to left from the stripe of the end cell
    piece1 = piece2 = b""
    if stripe is not None and rank <= end:
        if rank == end:
            go, i, j = 1, n, j_end
        else:
            tok = torch.empty(2, dtype=torch.int64, device=xdev)
            dist.recv(tok, src=peer(rank + 1), group=group)
            go, i = (int(x) for x in tok.tolist())
            j = stripe.width  # the cell the walk stopped in, as this stripe numbers it
        go_on, ei = 0, 0
        if go:
            piece1, piece2, ei, ej = stripe.walk(i, j)
            if rank > 0 and not (ei == 0 and mode == "fitting"):  # free top row: the alignment ends there
                if ej != 0:
                    raise RuntimeError("the walk left a stripe inside it")
                go_on = 1
        if rank > 0:
            dist.send(torch.tensor([go_on, ei], dtype=torch.int64, device=xdev), dst=peer(rank - 1), group=group)

    # 4. the pieces, laid end to end on dst (a stripe's piece goes in front of its right neighbours')
    gathered = [None] * world if rank == dst else None
    dist.gather_object((piece1, piece2), gathered, dst=peer(dst), group=group)
    if rank != dst:
        return score, None, None
    t1 = b"".join(p[0] for p in gathered[:G])
    t2 = b"".join(p[1] for p in gathered[:G])
    return score, (t1 or None), (t2 or None)
