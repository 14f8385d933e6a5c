set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "== $*" ; timeout -k 10 400 env "$@" ; }
{
run A=0 python -u -m pytest tests/test_gpu_affine.py tests/test_gpu_baseline_shapes.py tests/test_gpu_runtime.py -x -q -m gpu --timeout 300 --timeout-method thread
run A=0 python tools/bench_configs.py --config 4 --cpu-sample 0 --check 100
run A=0 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 100
run A=0 python tools/bench_configs.py --config 3 --cpu-sample 0 --check 100
run A=0 python tools/bench_configs.py --config 3 --pairs 500000 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_WAVES=15 python tools/bench_configs.py --config 3 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_WAVES=15 python tools/bench_configs.py --config 3 --pairs 500000 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_WAVES=10 python tools/bench_configs.py --config 3 --cpu-sample 0 --check 100
run A=0 python tools/fuzz_parity.py --seconds 100 --seed 301
} > gpurun_out/exp5.log 2>&1
