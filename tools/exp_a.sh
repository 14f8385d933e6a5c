set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/bio_amd/lib
{
for v in old libbioalign old libbioalign; do
echo "== $v"
BIOALIGN_LIB=$L/$v.so timeout -k 10 300 python bench.py --steps 10 --cpu-sample 0 --no-check
done
} > gpurun_out/exp_a.log 2>&1
