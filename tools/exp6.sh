set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "== $*" ; timeout -k 10 400 env "$@" ; }
{
for bpb in 1 2 4 8; do
run BIOALIGN_PIPE_BUNDLES=$bpb python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_BUNDLES=$bpb python tools/bench_configs.py --config 3 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_BUNDLES=$bpb python tools/bench_configs.py --config 3 --pairs 500000 --cpu-sample 0 --check 100
done
run BIOALIGN_PIPE_WAVES=12 BIOALIGN_PIPE_BUNDLES=1000 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_WAVES=12 BIOALIGN_PIPE_BUNDLES=3 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 100
run BIOALIGN_PIPE_WAVES=12 python tools/bench_configs.py --config 4 --cpu-sample 0 --check 100
} > gpurun_out/exp6.log 2>&1
