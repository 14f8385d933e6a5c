#!/bin/bash
# A/B of experimental library builds: kernel stats of a short bench per variant
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  mkdir -p gpurun_out/ab_$v
  BIOALIGN_LIB=$GRAFT_REPO_ROOT/bio_amd/lib/$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$v -- python3 bench.py --steps 2 --cpu-sample 0 --no-check > gpurun_out/ab_$v/bench.log 2>&1 || { tail -5 gpurun_out/ab_$v/bench.log; exit 1; }
  echo "== $v"; cat gpurun_out/ab_$v/*/*kernel_stats.csv | python3 -c "import sys,csv
for r in csv.reader(sys.stdin):
    print(r[0][:60], r[1], r[3][:10])"
done
