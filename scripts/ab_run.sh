#!/bin/bash
# ab_run.sh OUT REPS TREE... -- alternate bench runs of several trees ('.' = the repo itself, NAME = ab_NAME/) on the box this runs on
# env: STEPS / WARMUP (default 300 / 30), MS (several-streams list, default none)
out=$1; reps=$2; shift 2
: > "$out"
for rep in $(seq 1 $reps); do
  for t in "$@"; do
    if [ "$t" = "." ]; then b=bench.py; else b=ab_$t/bench.py; fi
    r=$(timeout -k 10 400 python $b --steps ${STEPS:-300} --warmup ${WARMUP:-30} --no-cpu-baseline --no-latency-mode --no-roofline-pass --multi-streams=${MS:-} 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ate_rmse_m"], d["keyframes_timed"], [(m["streams_per_gpu"], m["frames_per_s"]) for m in (d.get("multi_stream") or [])])') || { echo "[$t] FAILED" | tee -a "$out"; continue; }
    echo "[$t] $r" | tee -a "$out"
  done
done
