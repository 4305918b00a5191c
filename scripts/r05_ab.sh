#!/bin/bash
# A/B of environment switches on ONE box: scripts/r05_ab.sh "VAR=val" ["VAR2=val" ...]; prints frames/s of the 300-step and the driver's 20-step form, alternating baseline / variant
B="--no-cpu-baseline --no-latency-mode --multi-streams= --no-roofline-pass"
for rep in 1 2; do
  for v in "" "$@"; do
    for form in "" "--steps 20 --warmup 5"; do
      r=$(env $v python bench.py $B $form 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ate_rmse_m'])")
      echo "[${v:-baseline}] ${form:-300 steps}: $r"
    done
  done
done
