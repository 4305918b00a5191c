echo "== engine phase2"; VO_TRACE=1 timeout -k 10 100 python scripts/bench_ba.py --reps 20 --oracle --shapes bench 2>&1 | grep -v "resident" | cut -c1-900 || exit 1
timeout -k 10 100 python scripts/bench_ba.py --reps 3 --oracle --shapes small 2>&1 | cut -c1-700 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_stream_group.py -x -q -m gpu 2>&1 | tail -2
Q="--no-cpu-baseline --no-latency-mode --multi-streams= --steps 300 --warmup 30"
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["ate_rmse_m"], r["host_stage_ms"], {k: v["avg_us"] for k, v in r["roofline"]["kernels"].items() if k.startswith("k_ba") or k in ("k_match", "k_ransac_score", "k_pose_lm")})'
VO_TRACE=1 timeout -k 10 280 python bench.py $Q 2>gpurun_out/trace_launch2.txt | python -c "$P" || exit 1
grep "resident solve" gpurun_out/trace_launch2.txt | tail -1
