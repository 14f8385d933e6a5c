"""GPU checks that do not need the oracle, so they run at sizes the oracle would take minutes for:
every trace spells its inputs the way its mode demands and rescoring it (LinearGap, align.go:18-31)
gives the reported score -- vectorised over a whole batch -- plus a checksum of scores that is
invariant under the order of the pairs (the planner sorts and bundles them)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from bio_amd import _native, synth


def _run(mode, s1, o1, s2, o2, go=1, ge=1):
    table, c1, c2 = synth.dna_table(1, -1)
    eng = _native.Engine([0])
    pb = _native.PackedBatch(s1, o1, s2, o2, mode, table, c1, c2, go, ge)
    return eng.batch(pb)


def _columns(t, toff, tlen):
    """All trace columns of a batch, concatenated, with the pair index of each."""
    n = len(toff)
    tl = tlen.astype(np.int64)
    pair = np.repeat(np.arange(n), tl)
    start = np.repeat(toff.astype(np.int64), tl)
    within = np.arange(tl.sum()) - np.repeat(np.cumsum(tl) - tl, tl)
    return t[start + within], pair


@pytest.mark.parametrize("mode", ["global", "local", "fitting", "overlap"])
def test_traces_spell_inputs_and_rescore(mode):
    N, L1, L2 = 60_000, 150, 150
    s1, o1, s2, o2 = synth.dna_pairs(N, L1, L2, config=2, first=12345)
    scores, t1, t2, toff, tlen = _run(mode, s1, o1, s2, o2)
    x, pair = _columns(t1, toff, tlen)
    y, _ = _columns(t2, toff, tlen)
    gap = ord("-")
    assert not np.any((x == gap) & (y == gap))
    # rescoring: +1 match, -1 mismatch, -1 per gap column
    col = np.where((x == gap) | (y == gap), -1, np.where(x == y, 1, -1))
    resc = np.bincount(pair, weights=col, minlength=N).astype(np.int64)
    assert np.array_equal(resc, scores)
    # what the traces spell
    c1 = np.bincount(pair[x != gap], minlength=N)
    c2 = np.bincount(pair[y != gap], minlength=N)
    a, b = x[x != gap], y[y != gap]
    S1, S2 = s1.reshape(N, L1), s2.reshape(N, L2)
    if mode == "global":
        assert np.array_equal(a, s1) and np.array_equal(b, s2)
    elif mode == "fitting":  # all of s1 against a piece of s2
        assert np.array_equal(a, s1) and np.all(c2 <= L2)
    elif mode == "overlap":  # a suffix of s1 against a prefix of s2
        suf = np.concatenate([S1[p, L1 - c1[p]:] for p in range(0, N, 97)])
        pre = np.concatenate([S2[p, :c2[p]] for p in range(0, N, 97)])
        sel = np.isin(pair, np.arange(0, N, 97))
        assert np.array_equal(x[sel & (x != gap)], suf) and np.array_equal(y[sel & (y != gap)], pre)
    else:
        assert np.all(scores >= 0) and np.all(c1 <= L1) and np.all(c2 <= L2)
        assert np.all((scores == 0) == (tlen == 0))


def test_scores_do_not_depend_on_batch_order_or_bundling():
    N = 20_000
    s1, o1, s2, o2 = synth.dna_pairs(N, 96, 110, config=2, first=777)
    base = _run("local", s1, o1, s2, o2)[0]
    perm = np.random.default_rng(3).permutation(N)
    p1 = s1.reshape(N, 96)[perm].reshape(-1)
    p2 = s2.reshape(N, 110)[perm].reshape(-1)
    again = _run("local", p1, o1, p2, o2)[0]
    assert np.array_equal(again, base[perm])
    # and as 3 ragged pieces
    cut = [0, 6001, 13007, N]
    pieces = []
    for a, b in zip(cut[:-1], cut[1:]):
        pieces.append(_run("local", s1[a * 96:b * 96], o1[:b - a + 1], s2[a * 110:b * 110], o2[:b - a + 1])[0])
    assert np.array_equal(np.concatenate(pieces), base)


def test_affine_traces_rescore_at_scale():
    N, L = 20_000, 120
    s1, o1, s2, o2 = synth.dna_pairs(N, L, L, config=4, first=99)
    scores, t1, t2, toff, tlen = _run("global", s1, o1, s2, o2, go=5, ge=1)
    x, pair = _columns(t1, toff, tlen)
    y, _ = _columns(t2, toff, tlen)
    gap = ord("-")
    assert np.array_equal(x[x != gap], s1) and np.array_equal(y[y != gap], s2)
    # a gap column costs `open` when it starts a run on its side, `extend` otherwise (align_affine.go:51-57)
    gx, gy = x == gap, y == gap
    first = np.ones(len(x), dtype=bool)
    first[1:] = pair[1:] != pair[:-1]
    prev_gx = np.concatenate([[False], gx[:-1]]) & ~first
    prev_gy = np.concatenate([[False], gy[:-1]]) & ~first
    col = np.where(gx, np.where(prev_gx, -1, -5), np.where(gy, np.where(prev_gy, -1, -5), np.where(x == y, 1, -1)))
    resc = np.bincount(pair, weights=col, minlength=N).astype(np.int64)
    assert np.array_equal(resc, scores)
