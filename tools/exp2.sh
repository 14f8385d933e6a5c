set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/bio_amd/lib
run() { echo "== $*" ; timeout -k 10 200 env "$@" ; }
{
VALU_OPS_FROM=72 timeout -k 10 100 tools/bin/valu_ops
for v in libbioalign aff_prio1 aff_prio2 aff_prio3 aff_dot4; do
run BIOALIGN_LIB=$L/$v.so python tools/bench_configs.py --config 4 --cpu-sample 0 --check 50
run BIOALIGN_LIB=$L/$v.so python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 50
run BIOALIGN_LIB=$L/$v.so python tools/bench_configs.py --config 3 --cpu-sample 0 --check 50
done
run BIOALIGN_PIPE_WAVES=4 python tools/bench_configs.py --config 4 --cpu-sample 0 --check 50
run BIOALIGN_PIPE_WAVES=4 python tools/bench_configs.py --config 4 --pairs 98304 --cpu-sample 0 --check 50
} > gpurun_out/exp2.log 2>&1
