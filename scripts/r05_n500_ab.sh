for v in "" "VO_BA_DESC_TABLE=1" "VO_PAIRS_3=1" "VO_SCAN_2=1" "VO_CUT_SIZES_FIRST=1" ""; do
  r=$(env $v python bench.py --features 500 --no-cpu-baseline --no-latency-mode --multi-streams= --no-roofline-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ate_rmse_m'], d['ba_runs_timed'])")
  echo "[${v:-default}] $r"
done
