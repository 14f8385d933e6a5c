#!/bin/bash
# usage: tools/pmc_cfg.sh <tag> <bench_configs.py args...> : PMC passes over tools/bench_configs.py (counters only)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  mkdir -p gpurun_out/$tag/p$i
  timeout -s KILL 240 rocprofv3 --pmc $set --output-format csv -d gpurun_out/$tag/p$i -- python3 tools/bench_configs.py "$@" --steps 1 --check 0 --cpu-sample 0 > gpurun_out/$tag/p$i/log.txt 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/$tag/p$i/log.txt; exit 1; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/$tag/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
for k in agg:
    if 'copyBuffer' in k: continue
    print(k)
    for c in sorted(agg[k]):
        print('   %-24s %.6g (per dispatch, %d dispatches)' % (c, agg[k][c] / cnt[k][c], cnt[k][c]))
PY
