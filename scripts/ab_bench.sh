#!/bin/bash
# same-box A/B of the single-stream figure: each variant (an environment assignment, or "-") twice, alternating
# usage: scripts/ab_bench.sh OUT "VAR1=1" "VAR2=1 VAR3=1" -
out=$1; shift
: > "$out"
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    r=$(env $e timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency-mode --multi-streams= 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d.get("upload_inclusive"), d["ate_rmse_m"])') || exit 1
    echo "[$v] $r" | tee -a "$out"
  done
done
