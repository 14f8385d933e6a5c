set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "== $*" ; timeout -k 10 300 env "$@" ; }
{
run A=0 python tools/bench_configs.py --config 4 --cpu-sample 0 --check 100
run A=0 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 100
run A=0 python tools/bench_configs.py --config 3 --cpu-sample 0 --check 100
run A=0 python tools/bench_configs.py --config 3 --pairs 500000 --cpu-sample 0 --check 100
run BIOALIGN_STRIP_ROWS=16 python tools/bench_configs.py --config 4 --cpu-sample 0 --check 100
run BIOALIGN_STRIP_ROWS=16 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 100
run A=0 python -u -m pytest tests/test_gpu_affine.py tests/test_gpu_baseline_shapes.py -x -q -m gpu --timeout 300 --timeout-method thread
} > gpurun_out/exp3.log 2>&1
