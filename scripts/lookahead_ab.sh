#!/bin/bash
# what a look-ahead batch boundary costs: the 300-step bench at --lookahead 32 / 16 / 8, alternating
out=gpurun_out/r06_lookahead_ab.txt; : > $out
for r in 1 2 3; do
  for v in 32 16 8; do
    x=$(timeout -k 10 300 python bench.py --lookahead $v --no-cpu-baseline --no-latency-mode --no-roofline-pass --multi-streams= 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ate_rmse_m"], d["keyframes_timed"])') || { echo "lookahead $v FAILED" | tee -a $out; continue; }
    echo "lookahead $v: $x" | tee -a $out
  done
done
