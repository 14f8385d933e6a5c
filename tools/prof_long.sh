#!/bin/bash
# usage: tools/prof_long.sh <tag> [bench_long args]: rocprofv3 kernel trace + stats of one long-mode run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o long -- python3 tools/bench_long.py --reps 3 --oracle-prefix 0 "$@" > gpurun_out/$tag/log.txt 2>&1 || { tail -5 gpurun_out/$tag/log.txt; exit 1; }
tail -1 gpurun_out/$tag/log.txt | cut -c1-400
cat gpurun_out/$tag/*kernel_stats.csv | cut -c1-200 | head -12
