set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "== $*" ; timeout -k 10 200 env "$@" ; }
{
run A=0 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 50
run BIOALIGN_STRIP_ROWS=16 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 50
run BIOALIGN_STRIP_ROWS=16 BIOALIGN_PIPE_BELOW=0 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 50
run A=0 python tools/bench_configs.py --config 4 --cpu-sample 0 --check 50
run BIOALIGN_STRIP_ROWS=16 python tools/bench_configs.py --config 4 --cpu-sample 0 --check 50
run A=0 python tools/bench_configs.py --config 3 --cpu-sample 0 --check 50
run A=0 python tools/bench_configs.py --config 3 --pairs 500000 --cpu-sample 0 --check 50
} > gpurun_out/exp1.log 2>&1
