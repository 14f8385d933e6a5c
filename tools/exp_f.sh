set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
timeout -k 10 300 python -u -m pytest tests/test_gpu_bitparallel.py -x -q -m gpu --timeout 200 --timeout-method thread
timeout -k 10 300 python -u tools/fuzz_parity.py --unit --seconds 60 --seed 31
timeout -k 10 300 python bench.py --steps 10 --warmup 2
} > gpurun_out/exp_f.log 2>&1
