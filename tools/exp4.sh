set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "== $*" ; timeout -k 10 300 env "$@" ; }
{
run BIOALIGN_STRIP_ROWS=16 BIOALIGN_PIPE_WAVES=16 python tools/bench_configs.py --config 4 --cpu-sample 0 --check 100
run BIOALIGN_STRIP_ROWS=16 BIOALIGN_PIPE_WAVES=16 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 100
run BIOALIGN_STRIP_ROWS=16 BIOALIGN_PIPE_WAVES=12 python tools/bench_configs.py --config 4 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_WAVES=10 python tools/bench_configs.py --config 3 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_WAVES=10 python tools/bench_configs.py --config 3 --pairs 500000 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_WAVES=13 python tools/bench_configs.py --config 3 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_WAVES=5 python tools/bench_configs.py --config 3 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_BELOW=0 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 100
} > gpurun_out/exp4.log 2>&1
