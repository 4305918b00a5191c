#!/bin/bash
# several streams per GPU: device keyframes against host keyframes, same box, alternating (bench.py's multi_stream entries only)
B="--steps 20 --warmup 5 --no-cpu-baseline --no-latency-mode --no-roofline-pass"
for rep in 1 2; do
  for v in "" "--host-keyframes"; do
    r=$(VO_BENCH_NO_SEPARATE=1 python bench.py $B $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], [(m['streams_per_gpu'], m['frames_per_s']) for m in d['multi_stream']])")
    echo "[${v:-device keyframes}] $r"
  done
done
