#!/bin/bash
# same-box A/B of the bench figures: each variant (an environment assignment, or "-") REPS times, alternating
# usage: [REPS=2] [MS=8,16] scripts/ab_bench.sh OUT "VAR1=1" "VAR2=1 VAR3=1" -
out=$1; shift
: > "$out"
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    r=$(env $e timeout -k 10 300 python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --no-latency-mode --multi-streams=${MS:-} 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d.get("upload_inclusive"), d["ate_rmse_m"], [(m["streams_per_gpu"], m["frames_per_s"]) for m in (d.get("multi_stream") or [])])') || exit 1
    echo "[$v] $r" | tee -a "$out"
  done
done
