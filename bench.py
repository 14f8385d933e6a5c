#!/usr/bin/env python3
"""bench.py -- GCUPS of the batched alignment hot path on MI355X.

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on for one GPU):
1,000,000 pairs of 150 bp DNA per GPU, global THEN local alignment, linear gap (+1/-1, indel 1), score + both
traces.  One "step" = one global pass + one local pass over the batch = 4.5e10 cells per GPU.  With N GPUs every
rank aligns its own 1M pairs (different seeds): the batch shards by pair with no data-path collective, so
scaling is "weak".

--strong (BASELINE.json configs[2]: "1M pairs 300aa protein local alignment, BLOSUM62 affine gaps, sharded
8 GPUs"): ONE batch of --pairs protein pairs in total, cut over the N ranks by bio_amd.shard.shard_ranges (cell
count), each rank aligning its range; total work is fixed as N grows, so scaling is "strong".

Timed region: inputs (concatenated bytes + offsets) already resident in HBM; packing, DP and traceback kernels
all inside; outputs (scores, traces) left in HBM.

Launching.  `python bench.py --gpus N` starts the N ranks ITSELF (one child process per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment); the parent never touches the GPU and
rank 0 prints the one JSON line.  Under `python -m torch.distributed.run ... bench.py --gpus N` (WORLD_SIZE
already in the environment) the process IS a rank and spawns nothing.  The only collectives are the timing
barrier and the max-over-ranks reduction the contract asks for (RCCL; gloo when ranks have to share a GPU,
e.g. `--gpus 2` on a one-GPU box).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# minimal integer lane-operations per cell
OPS_LINEAR_DIR = 7      # max(up,left), diag+w, max, two compares, two bit packs (gap folded into the bias)
OPS_AFFINE_DIR = 13     # Gotoh E/F/H + four decision bits
OPS_LOCAL_EXTRA = 2     # clamp at 0, running-best update
# bit-parallel kernels (kernels_bp.hip), 150 x 150: VALU instructions of the column loop per cell (ISA count: 19.75 / 79.75
# per word of 32 rows and column, five words for 150 rows); used only when profiles/valu_insts.json has no PMC figure
BP_EXECUTED_PER_CELL = {"global": 0.66, "local": 2.66}
# gfx950 integer-VALU peak: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (MI355X_MICROARCH.md:
# SIMD-32, a wave64 VALU instruction issues in 2 cycles, max clock 2400 MHz)
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9
HBM_PEAK_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--strong", action="store_true",
                    help="configs[2]: one batch of --pairs 300 aa protein pairs (BLOSUM62, 11/1, local) cut over the ranks")
    ap.add_argument("--pairs", type=int, default=1_000_000, help="pairs per GPU (weak) or in total (--strong)")
    ap.add_argument("--len", type=int, default=0, help="sequence length (default 150 bp, --strong: 300 aa)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="pairs in the CPU baseline sample (0 = skip; default: about 10-20 s of CPU work)")
    ap.add_argument("--no-check", action="store_true", help="skip the in-run parity spot check")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise the launcher, the rendezvous, the barrier and the max-over-ranks reduction only "
                         "(gloo, no GPU, no alignment; the JSON line carries value null): the launcher's CPU test")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------
# launcher: the parent of `python bench.py --gpus N`.  Imports neither torch nor the HIP library.
# ----------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:  # one rank failed: the others would wait at the barrier for ever
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


# ----------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0 only): the oracle timed on the host cores of this box
# ----------------------------------------------------------------------------------------------------------
def host_cpu():
    try:
        share = len(os.sched_getaffinity(0))
    except Exception:
        share = os.cpu_count() or 1
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    # the harness gives one GPU's job a share of the machine (OMP_NUM_THREADS = that share); more threads than that
    # only queue behind each other
    cap = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = max(1, min(share, cap) if cap > 0 else share)
    return threads, share, os.cpu_count() or share, model


def cpu_baseline(work, n_sample):
    """Times the oracle (CPU restatement of the reference path: full matrices, 64-bit scores, one indirect Score
    call per cell) on the host cores of this box, on the first n_sample pairs of rank 0's batch."""
    from oracle import oracle as O

    threads, share, machine, model = host_cpu()
    s1, o1, s2, o2 = work["host"]
    sub = (s1[: int(o1[n_sample])], o1[: n_sample + 1], s2[: int(o2[n_sample])], o2[: n_sample + 1])
    warm = min(512, n_sample)
    refs = []
    a = O.Blosum62() if work["protein"] else O.MatchAligner(1, -1)
    for kind, mode, go, ge in work["oracle_passes"]:  # warm the per-thread heaps
        O.batch(kind, mode, s1[: int(o1[warm])], o1[: warm + 1], s2[: int(o2[warm])], o2[: warm + 1], a, go, ge, threads=threads)
    t0 = time.time()
    for kind, mode, go, ge in work["oracle_passes"]:
        refs.append(O.batch(kind, mode, *sub, a, go, ge, threads=threads))
    dt = time.time() - t0
    cells = sum(r[4] for r in refs)
    return {
        "value": round(cells / dt / 1e9, 4),
        "unit": "GCUPS",
        "cores": threads,
        "kind": "port",
        "cpu_model": model,
        "host_cores": {"used": threads, "affinity": share, "machine": machine,
                       "note": "threads = the job's CPU share (OMP_NUM_THREADS as the harness sets it), one pair per thread"},
        "sample": f"first {n_sample} pairs of rank 0's batch, {work['passes_text']}, {cells:.3g} cells in {dt:.1f} s "
                  f"(C restatement of {work['ref_cite']}; no Go toolchain on this image)",
    }, refs


# ----------------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------------
def run_rank(args):
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)

    if args.dry_run:
        import torch.distributed as dist

        if world > 1:
            dist.init_process_group(backend="gloo")
            dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "GCUPS", "value": None, "unit": "GCUPS", "n_gpus": world, "steps": 0, "warmup": 0,
                              "dry_run": True, "max_over_ranks": float(t.item())}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    n_dev = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    if n_dev == 0 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the alignment path has no CPU fallback")
    dev_index = local_rank % n_dev
    shared_gpu = world > n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    red_dev = dev
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if shared_gpu:  # RCCL refuses two ranks on one device: rehearsal on fewer GPUs than ranks
            dist.init_process_group(backend="gloo")
            red_dev = None
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from bio_amd import Blosum62, _native, score_table, shard, synth

    t_gen = time.time()
    if args.strong:
        P, L = args.pairs, (args.len or 300)
        bounds = shard.shard_ranges(np.full(P, L), np.full(P, L), world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        N = hi - lo
        s1, o1, s2, o2 = synth.protein_pairs(N, L, L, config=3, first=lo)
        table, code1, code2 = score_table(Blosum62, synth.AA.tolist(), synth.AA.tolist())
        passes = [("local", 11, 1)]
        work = {"protein": True, "oracle_passes": [(3, "local", 11, 1)], "passes_text": "local, affine 11/1",
                "ref_cite": "align_affine.go:36-104 with align_pair.go:129-153's local rules"}
        ops = [OPS_AFFINE_DIR + OPS_LOCAL_EXTRA]
        names = ["ba_dp_affine_kernel<local>"]
        dir_bytes_per_cell = 0.5
        workload = (f"configs[2]: {P} pairs x {L} aa protein in total, local, BLOSUM62 affine (open 11, extend 1), "
                    f"score + traceback; cut over {world} GPU(s) by cell count; step = one pass = {P * L * L:.3g} cells")
    else:
        N, L = args.pairs, (args.len or 150)
        s1, o1, s2, o2 = synth.dna_pairs(N, L, L, config=2, first=rank * N)
        table, code1, code2 = synth.dna_table(1, -1)
        passes = [("global", 1, 1), ("local", 1, 1)]
        work = {"protein": False, "oracle_passes": [(0, "global", 1, 0), (1, "local", 1, 0)], "passes_text": "global + local",
                "ref_cite": "align.go:76-249"}
        ops = [OPS_LINEAR_DIR, OPS_LINEAR_DIR + OPS_LOCAL_EXTRA]
        names = ["ba_dp_linear_kernel<global,32>", "ba_dp_linear_kernel<local,32>"]
        dir_bytes_per_cell = 0.25
        workload = (f"configs[1]: {N} pairs x {L} bp DNA per GPU, global + local, linear gap (+1/-1/1), "
                    f"score + traceback; step = both passes = {2 * N * L * L:.3g} cells per GPU")
    t_gen = time.time() - t_gen
    work["host"] = (s1, o1, s2, o2)

    eng = _native.Engine([dev_index])
    plans = []
    for mode, go, ge in passes:
        plans.append(eng.plan(_native.PackedBatch(s1, o1, s2, o2, mode, table, code1, code2, go, ge)))
    # several modes over the same pairs: the later passes read the packed copy the first pass made
    if len(plans) > 1 and os.environ.get("BIOALIGN_BENCH_SHARE_PACK", "1") != "0" and os.environ.get("BIOALIGN_BENCH_STREAMS", "1") != "2":
        for pl in plans[1:]:
            pl.share_pack(plans[0])
    cells_per_pass = plans[0].cells
    assert cells_per_pass == N * L * L
    # which kernels the planner chose (bioalign_plan_kernel): unit scoring runs bit-parallel (kernels_bp.hip)
    kinds = [pl.kernel for pl in plans]
    words = (L + 31) // 32
    dir_bytes = []
    for k, (mode, _, _) in enumerate(passes):
        if kinds[k] == "bitparallel":
            names[k] = f"ba_dp_bp_kernel<{mode}>"
            if mode == "global":  # match vectors + a checkpoint (3 planes) every 8 columns, per pair
                dir_bytes.append(N * (4 + 3 * ((L + 7) // 8)) * words * 4)
            else:                 # two planes of one bit per cell
                dir_bytes.append(N * 2 * words * 4 * L)
        else:
            dir_bytes.append(int(cells_per_pass * dir_bytes_per_cell))

    # inputs resident in HBM; outputs stay in HBM (separate buffers per pass)
    t_h2d = time.time()
    d_s1 = torch.from_numpy(s1).to(dev)
    d_s2 = torch.from_numpy(s2).to(dev)
    d_o1 = torch.from_numpy(o1.view(np.int64)).to(dev)
    d_o2 = torch.from_numpy(o2.view(np.int64)).to(dev)
    torch.cuda.synchronize()
    t_h2d = time.time() - t_h2d
    cap = int(o1[-1] + o2[-1])
    outs = []
    for _ in plans:
        outs.append(dict(
            scores=torch.empty(max(N, 1), dtype=torch.int64, device=dev),
            t1=torch.empty(cap + 16, dtype=torch.uint8, device=dev),
            t2=torch.empty(cap + 16, dtype=torch.uint8, device=dev),
            t_off=torch.empty(max(N, 1), dtype=torch.int64, device=dev),
            t_len=torch.empty(max(N, 1), dtype=torch.int64, device=dev),
        ))
    stream = torch.cuda.current_stream().cuda_stream
    # experiment (BIOALIGN_BENCH_STREAMS=2): every pass on a stream of its own, so that the latency-bound traceback
    # of one overlaps the VALU-bound DP of the other
    multi = os.environ.get("BIOALIGN_BENCH_STREAMS", "1") == "2" and len(plans) > 1
    side = [torch.cuda.Stream(device=dev) for _ in plans[1:]] if multi else []

    def step():
        for k, (plan, o) in enumerate(zip(plans, outs)):
            st = stream if (k == 0 or not multi) else side[k - 1].cuda_stream
            plan.run(d_s1.data_ptr(), d_o1.data_ptr(), d_s2.data_ptr(), d_o2.data_ptr(),
                     o["scores"].data_ptr(), o["t1"].data_ptr(), o["t2"].data_ptr(),
                     o["t_off"].data_ptr(), o["t_len"].data_ptr(), st)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    for pl in plans:
        pl.dp_ms()  # drop warm-up timings
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, device=red_dev)
    dp = [pl.dp_ms() for pl in plans]

    if args.strong:
        total_cells = float(args.pairs) * L * L * args.steps
    else:
        total_cells = float(len(passes)) * cells_per_pass * args.steps * world
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
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "pairs_per_gpu": N if not args.strong else None, "pairs_total": args.pairs if args.strong else N * world,
            "len": L,
            "parallelism": f"pairs sharded over {world} GPU(s), no collective"
                           + (f"; REHEARSAL: {world} ranks share {n_dev} GPU(s), barrier over gloo" if shared_gpu else ""),
        },
    }
    if rank == 0:
        # roofline of the dominant kernel
        def pmc(fn, kernel):
            tf = os.path.join(ROOT, "profiles", fn)
            if args.pairs != 1_000_000 or args.len or not os.path.exists(tf):
                return None  # the PMC files were collected on the default workloads
            try:
                return json.load(open(tf)).get(kernel)
            except Exception:
                return None

        kern = []
        for name, (ms, n), op, kind, dbytes in zip(names, dp, ops, kinds, dir_bytes):
            # one pass over the batch = n / steps launches (the plan may slice the batch); `pass_ms` is the
            # summed HIP-event duration of the DP launches of one pass, measured inside the timed region
            launches_per_pass = max(n // max(args.steps, 1), 1)
            pass_ms = ms / max(args.steps, 1)
            avg_ms = ms / max(n, 1)
            alg = op * cells_per_pass / (pass_ms * 1e-3) if pass_ms > 0 else 0.0
            executed = pmc("valu_insts.json", name)  # wave-instructions x 64 lanes / cells (SQ_INSTS_VALU)
            k = {
                "kernel": name, "family": kind, "launches": n, "launches_per_pass": launches_per_pass,
                "avg_ms": round(avg_ms, 4), "pass_ms": round(pass_ms, 4),
                "gcups": round(cells_per_pass / (pass_ms * 1e-3) / 1e9, 1) if pass_ms > 0 else None,
                # SURVEY 8d's algorithmic lane-ops per cell (one cell per lane-op) against the VALU peak
                "ops_per_cell": op, "achieved_algorithmic": round(alg / 1e12, 3),
                "frac_algorithmic": round(alg / VALU_PEAK_LANE_OPS, 4),
                "executed_valu_per_cell": executed,
                # what the kernel stores for the traceback, algorithmic
                "dir_bytes_per_pass": int(dbytes),
                "dir_write_gbs": round(dbytes / (pass_ms * 1e-3) / 1e9, 1) if pass_ms > 0 else None,
            }
            if kind == "bitparallel":
                # 32 cells per lane-op: the per-cell count does not price this kernel; what bounds it is the issue rate
                # of the instructions it executes (all VALU, nearly all full rate: v_bitop3_b32, v_addc_co_u32)
                ex = executed if executed is not None else BP_EXECUTED_PER_CELL[name.split("<")[1].rstrip(">")]
                k["executed_source"] = "SQ_INSTS_VALU (profiles/valu_insts.json)" if executed is not None else "instruction count of the column loop (ISA)"
                k["executed_valu_per_cell"] = ex
                k["achieved"] = round(ex * cells_per_pass / (pass_ms * 1e-3) / 1e12, 3) if pass_ms > 0 else 0.0
            else:
                k["achieved"] = k["achieved_algorithmic"]
            k["frac"] = round(k["achieved"] * 1e12 / VALU_PEAK_LANE_OPS, 4)
            kern.append(k)
        dom = max(kern, key=lambda k: k["pass_ms"])
        traffic = pmc("hbm_traffic.json", dom["kernel"])
        bp = dom["family"] == "bitparallel"
        result["roofline"] = {
            "bound": "valu", "kernel": dom["kernel"], "achieved": dom["achieved"],
            "peak": round(VALU_PEAK_LANE_OPS / 1e12, 2), "unit": "T lane-op/s (int32 VALU)",
            "frac": dom["frac"], "traffic": traffic,
            "note": ("bit-parallel kernel (32 rows per lane-op): `achieved` = the VALU lane-ops it EXECUTES per second "
                     "(executed_valu_per_cell x cells / the kernel's HIP-event time) against the full-rate VALU peak; "
                     "SURVEY.md 8d's per-cell count (one cell per lane-op) is kept as frac_algorithmic and exceeds 1 "
                     "because the kernel does the cell in a fraction of a lane-op; " if bp else
                     "integer max-plus work: neither HBM nor MFMA bounds it (SURVEY.md 8d); `achieved` counts SURVEY 8d's "
                     "ALGORITHMIC lane-ops per cell against the full-rate peak -- the kernel executes fewer, mostly "
                     "half-rate, instructions per cell (executed_valu_per_cell, from SQ_INSTS_VALU); ")
                    + f"HBM side for reference: traceback words {dom['dir_write_gbs']} GB/s of {HBM_PEAK_GBS} GB/s peak",
            "executed_valu_per_cell": dom["executed_valu_per_cell"],
            "frac_algorithmic": dom["frac_algorithmic"],
            # the HBM side of the same kernel: algorithmic bytes of traceback words and PMC-measured
            # traffic per launch over the live launch duration, against the HBM peak
            "hbm": {"algorithmic_bytes": dom["dir_bytes_per_pass"], "traffic": traffic,
                    "achieved": round((traffic or dom["dir_bytes_per_pass"]) / (dom["pass_ms"] * 1e-3) / 1e9, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round((traffic or dom["dir_bytes_per_pass"]) / (dom["pass_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "kernels": kern,
            "dp_kernel_share_of_step": round(sum(ms for ms, _ in dp) / (elapsed * 1e3), 3),
            # the whole step against SURVEY 8d's algorithmic count: the lane-ops of every pass over the step's wall
            # time (packing, walks and expansion included)
            "frac_end_to_end": round(sum(op * cells_per_pass for op in ops) / VALU_PEAK_LANE_OPS / (elapsed / args.steps), 4),
        }
        if bp and dom["kernel"].endswith("<local>"):
            # The local bit-parallel kernel writes two bits per cell: it is reported against the HBM roofline --
            # algorithmic bytes (the two planes) over the kernel's HIP-event time, PMC traffic next to it -- with the
            # VALU side kept as `valu`.  Measured apart (profiles/r03_notes.md): the same kernel without its stores
            # (score-only instance) 1.35 ms, the stores alone (tools/hbm_write.hip: 5.7 TB/s for this pattern) 1.07 ms,
            # together 1.58 ms -- instruction issue at three wavefronts per SIMD is the larger share, the stores cost
            # 0.2 ms on top.
            r = result["roofline"]
            r["valu"] = {"achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"],
                         "executed_valu_per_cell": r["executed_valu_per_cell"]}
            alg = dom["dir_bytes_per_pass"] / (dom["pass_ms"] * 1e-3) / 1e9
            r.update({"bound": "hbm", "achieved": round(alg, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(alg / HBM_PEAK_GBS, 4),
                      "note": "local bit-parallel kernel (32 rows per lane-op, two traceback bits stored per cell): `achieved` = "
                              "algorithmic bytes (the two planes, 0.25 B per cell) over the kernel's HIP-event time, `traffic` = "
                              "WRITE_SIZE + 2 x FETCH_SIZE per launch (PMC); the VALU side (executed lane-ops against the "
                              "full-rate issue peak) is in `valu` and is the larger share of the kernel's time (without its "
                              "stores 1.35 ms, the stores alone 1.07 ms at the 5.7 TB/s the chip takes for this pattern, "
                              "together 1.58 ms); SURVEY.md 8d's per-cell lane-op count is kept as "
                              "frac_algorithmic and exceeds 1 because the kernel does a cell in a fraction of a lane-op"})
        ns = args.cpu_sample
        if ns < 0:  # about 10-20 s of oracle work on 16 threads
            ns = 1_000_000 if not args.strong else 100_000
        ns = min(ns, N)
        if ns > 0:
            cb, refs = cpu_baseline(work, ns)
            result["cpu_baseline"] = cb
            if not args.no_check:
                # parity spot check of what the timed loop left in HBM against the oracle sample
                for o, ref, (mode, _, _) in zip(outs, refs, passes):
                    sc = o["scores"][:ns].cpu().numpy()
                    assert np.array_equal(sc, ref[0]), f"{mode}: scores differ from the oracle"
                    toff = o["t_off"][:ns].cpu().numpy().astype(np.int64)
                    tlen = o["t_len"][:ns].cpu().numpy().astype(np.int64)
                    assert np.array_equal(tlen, ref[3].astype(np.int64)), f"{mode}: trace lengths differ"
                    hi_b = int(o1[ns] + o2[ns])
                    t1 = o["t1"][:hi_b].cpu().numpy()
                    t2 = o["t2"][:hi_b].cpu().numpy()
                    for p in range(0, ns, max(1, ns // 2000)):
                        slot = int(o1[p] + o2[p])
                        a, b, n = int(toff[p]), slot, int(tlen[p])
                        assert np.array_equal(t1[a:a + n], ref[1][b:b + n]), f"{mode}: t1 differs at pair {p}"
                        assert np.array_equal(t2[a:a + n], ref[2][b:b + n]), f"{mode}: t2 differs at pair {p}"
                result["parity_check"] = (f"scores+lengths of {ns} pairs, traces of ~{min(ns, 2000)} pairs, "
                                          f"{work['passes_text']}: bit-exact vs oracle")
        result["aux"] = {"gen_s": round(t_gen, 2), "h2d_s": round(t_h2d, 3),
                         "workspace_gb": round(sum(pl.workspace_bytes for pl in plans) / 1e9, 2)}
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the parent: start one rank per GPU and wait; nothing here touches the GPU
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
