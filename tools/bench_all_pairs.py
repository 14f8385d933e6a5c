"""Host-to-host time of bioalign_all_pairs: n sequences of `length` bases, n(n-1)/2
score-only alignments.  Prints one JSON line per mode."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bio_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--length", type=int, default=150)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--modes", default="global,local")
    ap.add_argument("--ordered", action="store_true", help="bioalign_all_ordered_pairs with end cells (overlap mode)")
    args = ap.parse_args()
    rng = np.random.default_rng(1)
    al = np.frombuffer(b"ACGT", dtype=np.uint8)
    l = [al[rng.integers(0, 4, args.length)].tobytes() for _ in range(args.n)]
    pairs = args.n * (args.n - 1) // 2
    if args.ordered:
        from bio_amd import _native, pack_sequences, score_table

        data, off = pack_sequences(l)
        tbl, c1, c2 = score_table(bio_amd.MatchAligner(1, -1), [65, 67, 71, 84], [65, 67, 71, 84])
        pb = _native.PackedBatch(data, off, np.zeros(0, np.uint8), off, "overlap", tbl, c1, c2, 2, 2)
        eng = bio_amd.engine()
        eng.all_ordered_pairs(pb)
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            sc, ei, ej = eng.all_ordered_pairs(pb)
            ts.append((time.perf_counter() - t0) * 1e3)
        cells = 2 * pairs * args.length * args.length
        print(json.dumps({"workload": f"all ordered pairs of {args.n} x {args.length} bp, overlap, scores + end cells", "pairs": 2 * pairs,
                          "ms": round(min(ts), 3), "all_ms": [round(t, 2) for t in ts],
                          "gcups": round(cells / min(ts) / 1e6, 1), "checksum": int(sc.sum() + ej.sum())}), flush=True)
        return
    for mode in args.modes.split(","):
        f = bio_amd.AlignDist(mode, bio_amd.MatchAligner(1, -1), 1)
        bio_amd.AlignScores(l, f)  # warm-up: pool, code objects
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            sc = bio_amd.AlignScores(l, f)
            ts.append((time.perf_counter() - t0) * 1e3)
        cells = pairs * args.length * args.length
        print(json.dumps({"workload": f"all pairs of {args.n} x {args.length} bp, {mode}, score only", "pairs": pairs,
                          "ms": round(min(ts), 3), "all_ms": [round(t, 2) for t in ts],
                          "gcups": round(cells / min(ts) / 1e6, 1), "checksum": int(sc.sum())}), flush=True)


if __name__ == "__main__":
    main()
