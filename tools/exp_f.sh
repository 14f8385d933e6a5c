set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp_f
mkdir -p $O/stats
export TMPDIR=/tmp
{
timeout -k 10 300 python -u -m pytest tests/test_gpu_bitparallel.py -x -q -m gpu --timeout 200 --timeout-method thread
timeout -k 10 300 python -u tools/fuzz_parity.py --unit --seconds 45 --seed 41
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 10 --warmup 2
} > gpurun_out/exp_f.log 2>&1
