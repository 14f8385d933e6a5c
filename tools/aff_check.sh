#!/bin/bash
# usage: tools/aff_check.sh <tag>  -- affine GPU tests, then configs 3 and 4 (pipelined and not)
tag=$1; mkdir -p gpurun_out/$tag
timeout -k 10 300 python -u -m pytest tests/test_gpu_affine.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/$tag/tests.log 2>&1
rc=$?; tail -4 gpurun_out/$tag/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python3 tools/bench_configs.py --config 3 2>/dev/null > gpurun_out/$tag/cfg3.log || exit 1
timeout -k 10 200 python3 tools/bench_configs.py --config 4 2>/dev/null > gpurun_out/$tag/cfg4.log || exit 1
BIOALIGN_PIPE_BELOW=0 timeout -k 10 200 python3 tools/bench_configs.py --config 3 2>/dev/null > gpurun_out/$tag/cfg3_nopipe.log || exit 1
BIOALIGN_PIPE_BELOW=0 timeout -k 10 200 python3 tools/bench_configs.py --config 4 2>/dev/null > gpurun_out/$tag/cfg4_nopipe.log || exit 1
for f in cfg3 cfg4 cfg3_nopipe cfg4_nopipe; do echo "== $f"; cut -c1-330 gpurun_out/$tag/$f.log; done
