#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_check.sh <tag> [bench args]
# runs the GPU parity tests, then the bench under rocprofv3 --kernel-trace --stats
tag=$1; shift
mkdir -p gpurun_out/$tag
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/$tag/tests.log 2>&1
rc=$?; tail -4 gpurun_out/$tag/tests.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python3 bench.py --steps 3 "$@" > gpurun_out/$tag/bench.log 2>&1
rc=$?
grep -h '"metric"' gpurun_out/$tag/bench.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('value', d['value'], 'ms/step', d['ms_per_step'], 'parity', d.get('parity_check'))
    for k in d['roofline']['kernels']: print('  ', k['kernel'], k['pass_ms'], 'ms/pass', k['gcups'], 'GCUPS frac', k['frac'])
"
cat gpurun_out/$tag/prof/*/*kernel_stats.csv | python3 -c "import sys,csv
for r in csv.reader(sys.stdin):
    print(r[0][:60], r[1], r[3][:10])"
exit $rc
