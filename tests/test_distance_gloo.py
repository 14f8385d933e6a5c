"""All pairs of one list across ranks (one process per GPU): bio_amd.distance.AlignScoresSharded.

On CPU the per-rank scorer is the oracle, injected explicitly (this test runs without a GPU; the product default is
bioalign_all_pairs_range).  Checked: the ranges tile the triangle, are balanced by cells, and every rank ends up with
the whole score vector, equal to the unsharded oracle's.  With a GPU: two ranks share the one card.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _list():
    rng = np.random.default_rng(21)
    al = np.frombuffer(b"ACGT", dtype=np.uint8)
    return [al[rng.integers(0, 4, int(rng.integers(1, 80)))].tobytes() for _ in range(41)]


def _pair(p):
    i = int((1 + (1 + 8 * p) ** 0.5) / 2)
    while i * (i - 1) // 2 > p:
        i -= 1
    while (i + 1) * i // 2 <= p:
        i += 1
    return i, p - i * (i - 1) // 2


def _oracle_scores(l, f, first, end):
    from oracle import oracle as O

    a = O.MatchAligner(1, -1)
    out = []
    for p in range(first, end):
        i, j = _pair(p)
        out.append(O.AlignPair(f.mode, l[i], l[j], a, f.gap_open)[0])
    return np.array(out, dtype=np.int64)


def _worker(rank, world, port, q, backend, use_gpu):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if use_gpu:
        import torch
        torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        import bio_amd

        l = _list()
        res = {}
        for mode in ("global", "overlap"):
            f = bio_amd.AlignDist(mode, bio_amd.MatchAligner(1, -1), 1)
            res[mode] = bio_amd.AlignScoresSharded(l, f, scores_fn=None if use_gpu else _oracle_scores)
        q.put((rank, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(world, backend, use_gpu):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend, use_gpu)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, res = q.get(timeout=240)
        got[rank] = res
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import bio_amd

    l = _list()
    total = len(l) * (len(l) - 1) // 2
    for mode in ("global", "overlap"):
        want = _oracle_scores(l, bio_amd.AlignDist(mode, bio_amd.MatchAligner(1, -1), 1), 0, total)
        for r in range(world):
            assert np.array_equal(got[r][mode], want), (mode, r)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_all_pairs_over_ranks_gloo(world):
    _run(world, "gloo", False)


def test_pair_ranges_tile_and_balance():
    from bio_amd import pair_ranges

    l = _list()
    lens = [len(s) for s in l]
    total = len(l) * (len(l) - 1) // 2
    for world in (1, 2, 3, 8):
        b = pair_ranges(lens, world)
        assert b[0] == 0 and b[-1] == total and np.all(np.diff(b) >= 0)
        cost = np.zeros(world)
        for p in range(total):
            i, j = _pair(p)
            cost[np.searchsorted(b, p, side="right") - 1] += (lens[i] + 1) * (lens[j] + 1)
        assert cost.max() <= cost.sum() / world + (max(lens) + 1) ** 2 * 2
    assert list(pair_ranges([], 2)) == [0, 0, 0] and list(pair_ranges([5], 2)) == [0, 0, 0]
    assert list(pair_ranges([5, 5], 2)) in ([0, 0, 1], [0, 1, 1])


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_all_pairs_over_two_ranks_on_one_gpu():
    _run(2, "gloo", True)


@pytest.mark.gpu
def test_all_pairs_range_matches_whole():
    import bio_amd

    l = _list()
    f = bio_amd.AlignDist("fitting", bio_amd.MatchAligner(1, -1), 1)
    whole = bio_amd.AlignScores(l, f)
    total = len(whole)
    for first, end in ((0, total), (0, 0), (17, 18), (123, 700), (total - 5, total)):
        assert np.array_equal(bio_amd.AlignScores(l, f, pair_range=(first, end)), whole[first:end])
    with pytest.raises(bio_amd._native.BioAlignError):
        bio_amd.AlignScores(l, f, pair_range=(0, total + 1))
