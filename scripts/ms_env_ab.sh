#!/bin/bash
# usage: ms_env_ab.sh OUT REPS "ENV=..." ...   (alternates variants of the environment on the working tree; short form with 8/16 streams)
out=$1; reps=$2; shift 2
: > "$out"
for rep in $(seq 1 $reps); do
  for v in "" "$@"; do
    r=$(env $v timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency-mode --no-roofline-pass --multi-streams=8,16 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], [(m["streams_per_gpu"], m["frames_per_s"]) for m in (d.get("multi_stream") or [])])') || { echo "[$v] FAILED" | tee -a "$out"; continue; }
    echo "[${v:-default}] $r" | tee -a "$out"
  done
done
