#!/usr/bin/env python3
"""Config 5 on one GPU: a single n x m DNA pair, global, linear gap, through bioalign_long.
   python tools/bench_long.py [--n 100000] [--m 100000] [--reps 3]
Checks: the trace spells both inputs, rescoring it gives the reported score, and (bounded) the score
equals the oracle's two-column pass (lastColumnScores, align_linear.go:122-145) on a prefix pair."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bio_amd import _native, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100_000)
ap.add_argument("--m", type=int, default=100_000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--mode", default="global")
ap.add_argument("--oracle-prefix", type=int, default=20_000)
ap.add_argument("--open", type=int, default=1, help="gap open (affine when != --extend): Gotoh's three levels in long mode")
ap.add_argument("--extend", type=int, default=1)
ap.add_argument("--score-only", action="store_true", help="DP and end cell only (no walk): the DP kernel's own time")
args = ap.parse_args()
t0 = time.time()
s1, o1, s2, o2 = synth.dna_pairs(1, args.n, args.m, config=5, sub=0.06, ins=0.02, dele=0.02)
gen = time.time() - t0
table, c1, c2 = synth.dna_table(1, -1)
eng = _native.Engine([0])
affine = args.open != args.extend
pb = _native.PackedBatch(s1, o1, s2, o2, args.mode, table, c1, c2, args.open, args.extend, flags=_native.SCORE_ONLY if args.score_only else 0)
best = None
for r in range(args.reps):
    t0 = time.time()
    scores, t1, t2, toff, tlen = eng.long(pb)
    wall = time.time() - t0
    if best is None or eng.last_kernel_ms < best[0]:
        best = (eng.last_kernel_ms, wall)
cells = args.n * args.m
import bench
# linear + direction 7 lane-ops per cell, local +2, against the full-rate int32 VALU peak
OPS = (13 if args.open != args.extend else 7) + (2 if args.mode == "local" else 0)
def roofline(ms):
    return {"bound": "valu", "ops_per_cell": OPS, "achieved": round(OPS * cells / (ms * 1e-3) / 1e12, 3),
            "peak": round(bench.VALU_PEAK_LANE_OPS / 1e12, 2), "unit": "T lane-op/s (int32 VALU)",
            "frac": round(OPS * cells / (ms * 1e-3) / bench.VALU_PEAK_LANE_OPS, 4),
            "note": "one pair is latency-bound (a strip's step is a serial chain, profiles/r02_notes.md): the fraction says how "
                    "far a single matrix is from what 1M independent pairs reach, not that issue slots are the limit"}
if args.score_only:
    print(json.dumps({"config": 5, "mode": args.mode, "n": args.n, "m": args.m, "cells": cells, "score": int(scores[0]), "score_only": True,
                      "kernel_ms": round(best[0], 3), "kernel_gcups": round(cells / (best[0] * 1e-3) / 1e9, 1),
                      "roofline": roofline(best[0])}), flush=True)
    sys.exit(0)
a, l = int(toff[0]), int(tlen[0])
x, y = t1[a:a + l], t2[a:a + l]
gapc = ord("-")
assert np.array_equal(x[x != gapc], s1) or args.mode != "global"
assert np.array_equal(y[y != gapc], s2) or args.mode != "global"
if affine:
    # a gap of length L costs open + (L-1) * extend (align_affine.go:51-57)
    g1, g2 = (x == gapc), (y == gapc)
    opens = int((g1[1:] & ~g1[:-1]).sum() + g1[0] + (g2[1:] & ~g2[:-1]).sum() + g2[0])
    gaps = int(g1.sum() + g2.sum())
    resc = int(np.where(g1 | g2, 0, np.where(x == y, 1, -1)).sum()) - opens * args.open - (gaps - opens) * args.extend
else:
    resc = int(np.where((x == gapc) | (y == gapc), -1, np.where(x == y, 1, -1)).sum())
assert resc == int(scores[0]), (resc, int(scores[0]))
res = {"config": 5, "mode": args.mode, "gap": [args.open, args.extend], "n": args.n, "m": args.m, "cells": cells, "score": int(scores[0]), "trace_len": l,
       "kernel_ms": round(best[0], 3), "kernel_gcups": round(cells / (best[0] * 1e-3) / 1e9, 1),
       "wall_ms_incl_alloc_copies": round(best[1] * 1e3, 1), "gen_s": round(gen, 1),
       "checks": "trace spells both inputs; rescoring the trace = score", "roofline": roofline(best[0])}
if args.oracle_prefix and args.mode == "global" and not affine:
    from oracle import oracle as O
    k = min(args.oracle_prefix, args.n, args.m)
    pbk = _native.PackedBatch(s1[:k], np.array([0, k], dtype=np.uint64), s2[:k], np.array([0, k], dtype=np.uint64),
                              "global", table, c1, c2, 1, 1, flags=_native.SCORE_ONLY)
    sck = int(eng.long(pbk)[0][0])
    t0 = time.time()
    col = O.lastColumnScores(s1[:k].tobytes(), s2[:k].tobytes(), O.MatchAligner(1, -1), -1)
    dt = time.time() - t0
    assert int(col[0][-1]) == sck, (int(col[0][-1]), sck)
    res["oracle_prefix"] = f"{k} x {k}: score {sck} = oracle two-column pass ({k * k / dt / 1e9:.3f} GCUPS on 1 core)"
    threads, share, host_cores, model = bench.host_cpu()
    res["cpu_baseline"] = {"value": round(k * k / dt / 1e9, 3), "unit": "GCUPS", "cores": 1, "kind": "port", "cpu_model": model,
                           "host_cores": host_cores,
                           "sample": f"score-only two-column pass (align_linear.go:122-145 restated) on the {k} x {k} prefix pair; "
                                     "the full-matrix path needs 160 GB at 100k x 100k"}
print(json.dumps(res), flush=True)
