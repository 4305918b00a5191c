# developer tool: the bench line with the local BA on the launch-per-phase path (both generations) and on the persistent kernel
Q="--no-cpu-baseline --no-latency-mode --multi-streams= --steps 300 --warmup 30"
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["ate_rmse_m"], r["host_stage_ms"], r["ba_runs_timed"])'
echo "== launch phase2";  VO_TRACE=1 timeout -k 10 280 python bench.py $Q 2>gpurun_out/trace_launch2.txt | python -c "$P" || exit 1
grep "resident solve" gpurun_out/trace_launch2.txt | tail -1
echo "== persist 160";  VO_TRACE=1 VO_BA_PERSIST=1 VO_BA_GROUP=160 timeout -k 10 280 python bench.py $Q 2>gpurun_out/trace_p160.txt | python -c "$P" || exit 1
grep "resident solve" gpurun_out/trace_p160.txt | tail -1
