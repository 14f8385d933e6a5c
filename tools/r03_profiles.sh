#!/bin/bash
# usage: tools/r03_profiles.sh : the round's measurements in one call -- bench line, kernel stats of the bench and of
# configs 3 / 4 / 5, PMC passes of the bench and of config 3.  Every step under its own time limit, chained.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O/bench_stats $O/cfg3_stats $O/cfg4_stats $O/long_stats
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o bench -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-check > $O/bench_stats/log.txt 2>&1 && \
timeout -k 10 300 python3 tools/bench_configs.py --config 3 > $O/cfg3.jsonl 2> $O/cfg3.err && \
timeout -k 10 300 python3 tools/bench_configs.py --config 4 > $O/cfg4.jsonl 2> $O/cfg4.err && \
timeout -k 10 300 python3 tools/bench_configs.py --config 4 --pairs 100000 --cpu-sample 0 > $O/cfg4_100k.jsonl 2> $O/cfg4_100k.err && \
timeout -k 10 300 python3 tools/bench_configs.py --config 3 --pairs 500000 --cpu-sample 0 > $O/cfg3_500k.jsonl 2> $O/cfg3_500k.err && \
timeout -k 10 400 python3 bench.py --strong --steps 3 --cpu-sample 20000 > $O/bench_strong_1gpu.json 2> $O/bench_strong.err && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg3_stats -o cfg3 -- python3 tools/bench_configs.py --config 3 --steps 3 --check 0 --cpu-sample 0 > $O/cfg3_stats/log.txt 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4_stats -o cfg4 -- python3 tools/bench_configs.py --config 4 --steps 3 --check 0 --cpu-sample 0 > $O/cfg4_stats/log.txt 2>&1 && \
timeout -k 10 300 python3 tools/bench_long.py > $O/long.json 2> $O/long.err && \
timeout -k 10 300 python3 tools/bench_long.py --mode local --oracle-prefix 0 > $O/long_local.json 2> $O/long_local.err && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/long_stats -o long -- python3 tools/bench_long.py --reps 3 --oracle-prefix 0 > $O/long_stats/log.txt 2>&1 && \
timeout -k 10 900 bash tools/pmc.sh r03/pmc_bench > $O/pmc_bench.txt 2>&1 && \
timeout -k 10 900 bash tools/pmc_cfg.sh r03/pmc_cfg3 --config 3 > $O/pmc_cfg3.txt 2>&1 && \
timeout -k 10 900 bash tools/pmc_cfg.sh r03/pmc_cfg4 --config 4 > $O/pmc_cfg4.txt 2>&1
rc=$?
echo "rc=$rc"; ls $O
exit $rc
