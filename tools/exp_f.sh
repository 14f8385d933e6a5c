set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/exp_f
mkdir -p $O
export TMPDIR=/tmp
{
for v in bg4 bg8 bg10; do
echo "== $v"
mkdir -p $O/$v
BIOALIGN_LIB=$GRAFT_REPO_ROOT/bio_amd/lib/$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o bench -- python3 bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-check
done
} > gpurun_out/exp_f.log 2>&1
