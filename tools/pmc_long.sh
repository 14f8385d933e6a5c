#!/bin/bash
# usage: tools/pmc_long.sh <tag> <n> <m>: one rocprofv3 --pmc pass (SQ counters) over a score-only long-mode run
tag=$1; n=$2; m=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d gpurun_out/$tag -- python3 tools/bench_long.py --n $n --m $m --reps 1 --oracle-prefix 0 --score-only > gpurun_out/$tag/log.txt 2>&1 || { tail -5 gpurun_out/$tag/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/$tag/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]; agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k in agg:
    if 'long_dp' not in k: continue
    print(k)
    for c in sorted(agg[k]): print('   %-22s %.6g (per dispatch, %d dispatches)' % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
PY
