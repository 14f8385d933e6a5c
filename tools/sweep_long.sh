#!/bin/bash
# usage: tools/sweep_long.sh : kernel time of bioalign_long over shapes that separate the step time from the hand-over lag
for shape in "512 100000" "1024 100000" "2048 100000" "8192 100000" "100000 1024" "100000 8192" "100000 100000"; do
  set -- $shape
  timeout -k 10 120 python3 tools/bench_long.py --n $1 --m $2 --reps 3 --oracle-prefix 0 --score-only 2>&1 | tail -1 | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['n'], r['m'], 'kernel_ms', r['kernel_ms'], 'gcups', r['kernel_gcups'])"
done
