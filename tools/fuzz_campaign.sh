#!/bin/bash
# usage: tools/fuzz_campaign.sh <part> : long randomised parity runs against the oracle (tools/fuzz_parity.py), each step
# under its own time limit, chained; prints progress every few seconds.  Parts: a (unit scoring, four modes; general
# scorings), b (big batches, wide rows, long mode).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
case "$1" in
a)
  timeout -k 10 560 python -u tools/fuzz_parity.py --unit --seconds 500 --seed 1001 2>&1 | tee gpurun_out/fuzz_unit.log | grep -E "^(done|MISMATCH)|000 batches" 
  timeout -k 10 560 python -u tools/fuzz_parity.py --seconds 500 --seed 1002 2>&1 | tee gpurun_out/fuzz_general.log | grep -E "^(done|MISMATCH)|000 batches"
  ;;
b)
  timeout -k 10 260 python -u tools/fuzz_parity.py --unit --profile big --seconds 200 --seed 1003 2>&1 | tee gpurun_out/fuzz_unit_big.log | grep -E "^(done|MISMATCH)|0 batches"
  timeout -k 10 260 python -u tools/fuzz_parity.py --profile wide --seconds 200 --seed 1004 2>&1 | tee gpurun_out/fuzz_wide.log | grep -E "^(done|MISMATCH)|0 batches"
  timeout -k 10 260 python -u tools/fuzz_parity.py --long --seconds 200 --seed 1005 2>&1 | tee gpurun_out/fuzz_long.log | grep -E "^(done|MISMATCH)|00 long"
  timeout -k 10 260 python -u tools/fuzz_parity.py --fasta --seconds 120 --seed 1006 2>&1 | tee gpurun_out/fuzz_fasta.log | tail -2
  ;;
esac
