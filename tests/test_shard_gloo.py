"""The N > 1 path on CPU: two ranks over gloo.

The rank-level sharding (bio_amd.shard) is exercised with the oracle injected as the
per-rank aligner -- the oracle stands in for the GPU here only because this test runs
without one; the product path never does that (align_fn defaults to the HIP batch).
Checked: every pair is aligned exactly once, results come back in input order and equal
the unsharded oracle run, and the timing reduction is the max over ranks.
"""
import os
import socket
import sys

import numpy as np
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


def _batch():
    from bio_amd import synth

    rng = np.random.default_rng(7)
    s1s, s2s = [], []
    for k in range(97):  # ragged, with empties
        a, _, b, _ = synth.dna_pairs(1, int(rng.integers(0, 60)) or 1, int(rng.integers(0, 60)) or 1, config=9, first=k)
        s1s.append(a.tobytes() if k % 11 else b"")
        s2s.append(b.tobytes() if k % 13 else b"")
    return s1s, s2s


def _oracle_fn(mode, s1s, s2s, a, go, ge):
    from oracle import oracle as O

    al = O.MatchAligner(1, -1)
    out = [O.AlignPair(mode, x, y, al, go) for x, y in zip(s1s, s2s)]
    return [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bio_amd import shard

        s1s, s2s = _batch()
        res = {}
        for mode in ("global", "local", "fitting", "overlap"):
            res[mode] = shard.align_sharded(mode, s1s, s2s, None, 1, align_fn=_oracle_fn)
        (lo, hi), mine = shard.align_sharded("global", s1s, s2s, None, 1, gather=False, align_fn=_oracle_fn)
        t = shard.max_over_ranks(1.0 + rank)
        q.put((rank, res, (lo, hi, len(mine[0])), t))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_gloo_shard_and_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, res, rng_, t = q.get(timeout=240)
        got[rank] = (res, rng_, t)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    s1s, s2s = _batch()
    # the two ranges tile the batch
    (lo0, hi0, n0), (lo1, hi1, n1) = got[0][1], got[1][1]
    assert lo0 == 0 and hi0 == lo1 and hi1 == len(s1s) and n0 == hi0 - lo0 and n1 == hi1 - lo1
    assert 0 < hi0 < len(s1s)
    # timing reduction = max over ranks, on every rank
    assert got[0][2] == 2.0 and got[1][2] == 2.0
    for mode in ("global", "local", "fitting", "overlap"):
        assert got[1][0][mode] is None  # only dst holds the gathered batch
        want = _oracle_fn(mode, s1s, s2s, None, 1, 1)
        assert got[0][0][mode][0] == want[0]
        assert got[0][0][mode][1] == want[1] and got[0][0][mode][2] == want[2]


def test_single_process_is_one_shard():
    from bio_amd import shard

    s1s, s2s = _batch()
    assert not dist.is_initialized()
    got = shard.align_sharded("local", s1s, s2s, None, 1, align_fn=_oracle_fn)
    assert got == _oracle_fn("local", s1s, s2s, None, 1, 1)
    assert shard.max_over_ranks(0.25) == 0.25
