#!/bin/bash
# GPU test suite, then the bench workload (300 steps and the driver's 20 steps) with VO_TRACE
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 || exit 1
Q="--no-cpu-baseline --no-latency-mode --multi-streams="
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["ate_rmse_m"], r["host_stage_ms"], {k: v["avg_us"] for k, v in r["roofline"]["kernels"].items() if k.startswith("k_ba") or k in ("k_match", "k_ransac_score", "k_pose_lm")})'
VO_TRACE=1 timeout -k 10 280 python bench.py $Q --steps 300 --warmup 30 2>gpurun_out/trace_full.txt | python -c "$P" || exit 1
grep "resident solve\|resident cut" gpurun_out/trace_full.txt | tail -2
timeout -k 10 280 python bench.py $Q --steps 20 --warmup 5 2>/dev/null | python -c "$P" || exit 1
