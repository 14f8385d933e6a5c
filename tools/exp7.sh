set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "== $*" ; timeout -k 10 400 env "$@" ; }
{
run A=0 python -u -m pytest tests/test_gpu_affine.py tests/test_gpu_baseline_shapes.py -x -q -m gpu --timeout 300 --timeout-method thread
for bpb in 1 8; do
run BIOALIGN_PIPE_BUNDLES=$bpb python tools/bench_configs.py --config 4 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_BUNDLES=$bpb python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 100
done
run BIOALIGN_PIPE_WAVES=12 BIOALIGN_PIPE_BUNDLES=8 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_WAVES=12 BIOALIGN_PIPE_BUNDLES=3 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_WAVES=12 BIOALIGN_PIPE_BUNDLES=2 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 100
run BIOALIGN_STRIP_ROWS=16 BIOALIGN_PIPE_BUNDLES=1 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 100
run BIOALIGN_STRIP_ROWS=16 BIOALIGN_PIPE_BUNDLES=1 BIOALIGN_PIPE_WAVES=16 python tools/bench_configs.py --config 4 --cpu-sample 0 --check 100
} > gpurun_out/exp7.log 2>&1
