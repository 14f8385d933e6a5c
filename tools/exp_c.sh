set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "== $*" ; timeout -k 10 400 env "$@" ; }
{
run A=0 python tools/bench_configs.py --config 3 --dna --pairs 500000 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_BELOW=100000000 python tools/bench_configs.py --config 3 --dna --pairs 500000 --cpu-sample 0 --check 100
run A=0 python tools/bench_configs.py --config 4 --pairs 400000 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_BELOW=100000000 python tools/bench_configs.py --config 4 --pairs 400000 --cpu-sample 0 --check 100
run A=0 python tools/bench_configs.py --config 3 --pairs 1000000 --cpu-sample 0 --check 100
} > gpurun_out/exp_c.log 2>&1
