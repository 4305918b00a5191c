B="--steps 96 --warmup 32 --no-cpu-baseline --no-latency-mode --multi-streams="
for v in "" "VO_NO_STAT_POLL=1" "VO_NO_SPIN=1" "GPU_MAX_HW_QUEUES=4"; do
  r=$(env $v python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches'])")
  echo "[${v:-baseline}] $r"
done
