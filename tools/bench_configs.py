#!/usr/bin/env python3
"""Times the other BASELINE.json configurations on one GPU (per-GPU share of the pairs):
   python tools/bench_configs.py --config 3|4 [--pairs N] [--steps K]
config 3: protein 300 x 300, local, affine (Blosum62, open 11, extend 1)
config 4: DNA read 250 vs window 2000, fitting then overlap, affine (+1/-1, open 5, extend 1)
Prints one JSON line per mode with DP-kernel and end-to-end GCUPS, the VALU roofline of the DP kernel (SURVEY.md 8d's
algorithmic lane-ops per cell against 256 CU x 4 SIMD-32 x 2.4 GHz), a CPU leg (the oracle on the box's share of host
cores, a bounded sample of the same pairs) and checks a sample against the oracle."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bio_amd import _native, synth, score_table, Blosum62

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--pairs", type=int, default=0)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--check", type=int, default=300)
ap.add_argument("--linear", action="store_true", help="same shapes with a linear gap (gap = open)")
ap.add_argument("--cpu-sample", type=int, default=-1, help="pairs of the CPU leg (-1: about 10 s of oracle work, 0: none)")
ap.add_argument("--dna", action="store_true", help="config 3 shapes and penalties with a 4-letter alphabet (+1/-1): the LDS profile is 5x smaller")
args = ap.parse_args()
dev = torch.device("cuda", 0)
if args.config == 3:
    N = args.pairs or 125_000
    s1, o1, s2, o2 = synth.protein_pairs(N, 300, 300, config=3)
    table, c1, c2 = score_table(Blosum62, np.unique(s1).tolist(), np.unique(s2).tolist())
    modes, go, ge = ["local"], 11, 1
    if args.dna:
        s1, o1, s2, o2 = synth.dna_pairs(N, 300, 300, config=3)
        table, c1, c2 = synth.dna_table(1, -1)
else:
    N = args.pairs or 12_500
    s1, o1, s2, o2 = synth.fitting_reads(N, 250, 2000, config=4)
    table, c1, c2 = synth.dna_table(1, -1)
    modes, go, ge = ["fitting", "overlap"], 5, 1
if args.linear:
    ge = go
eng = _native.Engine([0])
d = [torch.from_numpy(x).to(dev) for x in (s1, o1.view(np.int64), s2, o2.view(np.int64))]
for mode in modes:
    pb = _native.PackedBatch(s1, o1, s2, o2, mode, table, c1, c2, go, ge)
    plan = eng.plan(pb)
    cap = pb.trace_capacity
    out = dict(scores=torch.empty(N, dtype=torch.int64, device=dev), t1=torch.empty(cap + 16, dtype=torch.uint8, device=dev),
               t2=torch.empty(cap + 16, dtype=torch.uint8, device=dev), t_off=torch.empty(N, dtype=torch.int64, device=dev),
               t_len=torch.empty(N, dtype=torch.int64, device=dev))
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: plan.run(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), out["scores"].data_ptr(),
                           out["t1"].data_ptr(), out["t2"].data_ptr(), out["t_off"].data_ptr(), out["t_len"].data_ptr(), st)
    run(); torch.cuda.synchronize(); plan.dp_ms()
    t0 = time.perf_counter()
    for _ in range(args.steps): run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / args.steps
    ms, n = plan.dp_ms()
    res = {"config": args.config, "mode": mode, "gap": [go, ge], "pairs": N, "cells": plan.cells, "ms": round(el * 1e3, 3),
           "gcups": round(plan.cells / el / 1e9, 1), "dp_ms": round(ms / max(n, 1) * (n / args.steps), 3),
           "dp_gcups": round(plan.cells / (ms / args.steps * 1e-3) / 1e9, 1), "workspace_gb": round(plan.workspace_bytes / 1e9, 2)}
    # linear + direction 7 lane-ops per cell, affine + direction 13, local +2
    ops = (13 if go != ge else 7) + (2 if mode == "local" else 0)
    dp_s = ms / args.steps * 1e-3
    import bench
    res["roofline"] = {"bound": "valu", "ops_per_cell": ops, "achieved": round(ops * plan.cells / dp_s / 1e12, 3),
                       "peak": round(bench.VALU_PEAK_LANE_OPS / 1e12, 2), "unit": "T lane-op/s (int32 VALU)",
                       "frac": round(ops * plan.cells / dp_s / bench.VALU_PEAK_LANE_OPS, 4),
                       "note": "algorithmic lane-ops (SURVEY.md 8d) over the DP kernel's HIP-event time; the affine kernels execute "
                               "15 (DNA, 32-row strips) to 20 (protein, local) instructions per cell, 9 to 12 of them full-rate "
                               "(profiles/r02_notes.md)"}
    ns = args.cpu_sample
    if ns < 0:
        ns = 6000 if args.config == 3 else 1200  # about 10 s on 16 threads
    ns = min(ns, N)
    if ns > 0:
        from oracle import oracle as O
        threads, share, host_cores, model = bench.host_cpu()
        oa = O.Blosum62() if (args.config == 3 and not args.dna) else O.MatchAligner(1, -1)
        kind = O.KIND_AFFINE if go != ge else (O.KIND_LOCAL if mode == "local" else O.KIND_PAIR)
        t0c = time.perf_counter()
        ref = O.batch(kind, mode, s1[:int(o1[ns])], o1[:ns + 1], s2[:int(o2[ns])], o2[:ns + 1], oa, go, ge, threads=threads)
        dtc = time.perf_counter() - t0c
        res["cpu_baseline"] = {"value": round(ref[4] / dtc / 1e9, 3), "unit": "GCUPS", "cores": threads, "kind": "port",
                               "cpu_model": model, "host_cores": host_cores,
                               "sample": f"{ns} of the same pairs, oracle (C restatement of the reference path), {dtc:.1f} s"}
        assert np.array_equal(out["scores"][:ns].cpu().numpy(), ref[0][:ns]), "scores differ from the oracle on the CPU sample"
    if args.check:
        from oracle import oracle as O
        oa = O.Blosum62() if (args.config == 3 and not args.dna) else O.MatchAligner(1, -1)
        k = min(args.check, N)
        sc = out["scores"][:k].cpu().numpy(); toff = out["t_off"][:k].cpu().numpy(); tlen = out["t_len"][:k].cpu().numpy()
        hi = int(o1[k] + o2[k]); t1 = out["t1"][:hi].cpu().numpy(); t2 = out["t2"][:hi].cpu().numpy()
        for p in range(k):
            x = s1[int(o1[p]):int(o1[p + 1])].tobytes(); y = s2[int(o2[p]):int(o2[p + 1])].tobytes()
            want = O.AlignAffine(mode, x, y, oa, go, ge) if go != ge else O.AlignPair(mode, x, y, oa, go)
            a, l = int(toff[p]), int(tlen[p])
            got = (int(sc[p]), t1[a:a + l].tobytes() if l else None, t2[a:a + l].tobytes() if l else None)
            assert got == want, (mode, p, got[0], want[0])
        res["parity"] = f"{k} pairs bit-exact vs oracle"
    print(json.dumps(res), flush=True)
    plan.close()
