#!/usr/bin/env python3
"""bench.py -- GCUPS of the batched alignment hot path on MI355X.

Workload (BASELINE.json configs[1]): 1,000,000 pairs of 150 bp DNA per GPU,
global THEN local alignment, linear gap (+1/-1, indel 1), score + both traces.
One "step" = one global pass + one local pass over the batch = 4.5e10 cells per GPU.

Timed region: inputs (concatenated bytes + offsets) already resident in HBM;
packing, DP and traceback kernels all inside; outputs (scores, traces) left in HBM.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank
aligns its own 1M pairs (different seeds) -- the batch shards by pair with no
data-path collective, so scaling is "weak"; the only collectives are the timing
barrier and the max-over-ranks reduction the contract asks for.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

# minimal integer lane-operations per cell
OPS_LINEAR_DIR = 7      # max(up,left), diag+w, max, two compares, two bit packs (gap folded into the bias)
OPS_LOCAL_EXTRA = 2     # clamp at 0, running-best update
# gfx950 integer-VALU peak: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (MI355X_MICROARCH.md:
# SIMD-32, a wave64 VALU instruction issues in 2 cycles, max clock 2400 MHz)
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1_000_000, help="pairs per GPU")
    ap.add_argument("--len", type=int, default=150)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000,
                    help="pairs in the CPU baseline sample (0 = skip); the default is the whole batch, about 10 s on 16 threads")
    ap.add_argument("--no-check", action="store_true", help="skip the in-run parity spot check")
    return ap.parse_args()


def cpu_baseline(s1, o1, s2, o2, n_sample):
    """Times the oracle (CPU restatement of the reference path: full matrices, 64-bit
    scores, one indirect Score call per cell) on the host cores of this box."""
    from oracle import oracle as O

    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", "16") or 16), 16))
    L1 = int(o1[1] - o1[0])
    L2 = int(o2[1] - o2[0])
    a = O.MatchAligner(1, -1)
    sub = (s1[: n_sample * L1], o1[: n_sample + 1], s2[: n_sample * L2], o2[: n_sample + 1])
    # warm the per-thread heaps
    O.batch(O.KIND_GLOBAL, "global", s1[: 512 * L1], o1[:513], s2[: 512 * L2], o2[:513], a, 1, threads=threads)
    t0 = time.time()
    g = O.batch(O.KIND_GLOBAL, "global", *sub, a, 1, threads=threads)
    l = O.batch(O.KIND_LOCAL, "local", *sub, a, 1, threads=threads)
    dt = time.time() - t0
    cells = g[4] + l[4]
    return {
        "value": round(cells / dt / 1e9, 4),
        "unit": "GCUPS",
        "cores": threads,
        "kind": "port",
        "sample": f"first {n_sample} pairs of the same batch, global + local, {cells:.3g} cells in {dt:.1f} s "
                  f"(C restatement of align.go:76-249; no Go toolchain on this image)",
    }, (g, l)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the alignment path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group(backend="nccl", device_id=dev)

    from bio_amd import _native, shard, synth

    N, L = args.pairs, args.len
    t_gen = time.time()
    s1, o1, s2, o2 = synth.dna_pairs(N, L, L, config=2, first=rank * N)
    t_gen = time.time() - t_gen
    table, code1, code2 = synth.dna_table(1, -1)

    eng = _native.Engine([local_rank])
    pb_g = _native.PackedBatch(s1, o1, s2, o2, "global", table, code1, code2, 1, 1)
    pb_l = _native.PackedBatch(s1, o1, s2, o2, "local", table, code1, code2, 1, 1)
    plan_g = eng.plan(pb_g)
    plan_l = eng.plan(pb_l)
    # both modes align the same pairs: the local pass reads the packed copy the global pass made
    if os.environ.get("BIOALIGN_BENCH_SHARE_PACK", "1") != "0":
        plan_l.share_pack(plan_g)
    cells_per_pass = plan_g.cells
    assert cells_per_pass == N * L * L

    # inputs resident in HBM; outputs stay in HBM (separate buffers per mode)
    t_h2d = time.time()
    d_s1 = torch.from_numpy(s1).to(dev)
    d_s2 = torch.from_numpy(s2).to(dev)
    d_o1 = torch.from_numpy(o1.view(np.int64)).to(dev)
    d_o2 = torch.from_numpy(o2.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    t_h2d = time.time() - t_h2d
    cap = pb_g.trace_capacity
    outs = []
    for _ in range(2):
        outs.append(dict(
            scores=torch.empty(N, dtype=torch.int64, device=dev),
            t1=torch.empty(cap + 16, dtype=torch.uint8, device=dev),
            t2=torch.empty(cap + 16, dtype=torch.uint8, device=dev),
            t_off=torch.empty(N, dtype=torch.int64, device=dev),
            t_len=torch.empty(N, dtype=torch.int64, device=dev),
        ))
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        for plan, o in ((plan_g, outs[0]), (plan_l, outs[1])):
            plan.run(d_s1.data_ptr(), d_o1.data_ptr(), d_s2.data_ptr(), d_o2.data_ptr(),
                     o["scores"].data_ptr(), o["t1"].data_ptr(), o["t2"].data_ptr(),
                     o["t_off"].data_ptr(), o["t_len"].data_ptr(), stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    plan_g.dp_ms()
    plan_l.dp_ms()  # drop warm-up timings
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, device=dev)
    ms_g, n_g = plan_g.dp_ms()
    ms_l, n_l = plan_l.dp_ms()

    total_cells = 2.0 * cells_per_pass * args.steps * world
    gcups = total_cells / elapsed / 1e9

    result = {
        "metric": "GCUPS (cell updates/sec, whole node); bit-exact score+trace match",
        "value": round(gcups, 2),
        "unit": "GCUPS",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {
            "workload": f"configs[1]: {N} pairs x {L} bp DNA per GPU, global + local, linear gap (+1/-1/1), "
                        f"score + traceback; step = both passes = {2 * cells_per_pass:.3g} cells per GPU",
            "pairs_per_gpu": N, "len": L, "parallelism": f"pairs sharded over {world} GPU(s), no collective",
        },
    }
    if rank == 0:
        # roofline of the dominant kernel (the local-mode DP: more work per cell than global)
        kern = []
        for name, ms, n, ops in (("ba_dp_linear_kernel<global,32>", ms_g, n_g, OPS_LINEAR_DIR),
                                 ("ba_dp_linear_kernel<local,32>", ms_l, n_l, OPS_LINEAR_DIR + OPS_LOCAL_EXTRA)):
            # one pass over the batch = n / steps launches (the plan may slice the batch); `pass_ms` is the
            # summed HIP-event duration of the DP launches of one pass, measured inside the timed region
            launches_per_pass = max(n // max(args.steps, 1), 1)
            pass_ms = ms / max(args.steps, 1)
            avg_ms = ms / max(n, 1)
            ach = ops * cells_per_pass / (pass_ms * 1e-3) if pass_ms > 0 else 0.0
            kern.append({
                "kernel": name, "launches": n, "launches_per_pass": launches_per_pass, "avg_ms": round(avg_ms, 4),
                "pass_ms": round(pass_ms, 4),
                "gcups": round(cells_per_pass / (pass_ms * 1e-3) / 1e9, 1) if pass_ms > 0 else None,
                "ops_per_cell": ops, "achieved": round(ach / 1e12, 3), "frac": round(ach / VALU_PEAK_LANE_OPS, 4),
                # residue words written, algorithmic
                "dir_bytes_per_pass": int(cells_per_pass // 4),
                "dir_write_gbs": round(cells_per_pass / 4 / (pass_ms * 1e-3) / 1e9, 1) if pass_ms > 0 else None,
            })
        dom = max(kern, key=lambda k: k["pass_ms"])
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(dom["kernel"])
            except Exception:
                traffic = None
        result["roofline"] = {
            "bound": "valu", "kernel": dom["kernel"], "achieved": dom["achieved"],
            "peak": round(VALU_PEAK_LANE_OPS / 1e12, 2), "unit": "T lane-op/s (int32 VALU)",
            "frac": dom["frac"], "traffic": traffic,
            "note": "integer max-plus work: neither HBM nor MFMA bounds it (SURVEY.md 8d); "
                    f"HBM side for reference: direction words {dom['dir_write_gbs']} GB/s of {HBM_PEAK_GBS} GB/s peak",
            # the HBM side of the same kernel: algorithmic bytes (K/8 per cell of residue words) and PMC-measured
            # traffic per launch over the live launch duration, against the HBM peak
            "hbm": {"algorithmic_bytes": dom["dir_bytes_per_pass"], "traffic": traffic,
                    "achieved": round((traffic or dom["dir_bytes_per_pass"]) / (dom["pass_ms"] * 1e-3) / 1e9, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round((traffic or dom["dir_bytes_per_pass"]) / (dom["pass_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "kernels": kern,
            "dp_kernel_share_of_step": round((ms_g + ms_l) / (elapsed * 1e3), 3),
        }
        if args.cpu_sample > 0:
            ns = min(args.cpu_sample, N)
            cb, (g_ref, l_ref) = cpu_baseline(s1, o1, s2, o2, ns)
            result["cpu_baseline"] = cb
            if not args.no_check:
                # parity spot check of what the timed loop left in HBM against the oracle sample
                for o, ref, nm in ((outs[0], g_ref, "global"), (outs[1], l_ref, "local")):
                    sc = o["scores"][:ns].cpu().numpy()
                    assert np.array_equal(sc, ref[0]), f"{nm}: scores differ from the oracle"
                    toff = o["t_off"][:ns].cpu().numpy().astype(np.int64)
                    tlen = o["t_len"][:ns].cpu().numpy().astype(np.int64)
                    assert np.array_equal(tlen, ref[3].astype(np.int64)), f"{nm}: trace lengths differ"
                    hi = int(o1[ns] + o2[ns])
                    t1 = o["t1"][:hi].cpu().numpy()
                    t2 = o["t2"][:hi].cpu().numpy()
                    for p in range(0, ns, max(1, ns // 2000)):
                        slot = int(o1[p] + o2[p])
                        a, b, n = int(toff[p]), slot, int(tlen[p])
                        assert np.array_equal(t1[a:a + n], ref[1][b:b + n]), f"{nm}: t1 differs at pair {p}"
                        assert np.array_equal(t2[a:a + n], ref[2][b:b + n]), f"{nm}: t2 differs at pair {p}"
                result["parity_check"] = f"scores+lengths of {ns} pairs, traces of ~2000 pairs, both modes: bit-exact vs oracle"
        result["aux"] = {"gen_s": round(t_gen, 2), "h2d_s": round(t_h2d, 3),
                         "workspace_gb": round((plan_g.workspace_bytes + plan_l.workspace_bytes) / 1e9, 2)}
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
