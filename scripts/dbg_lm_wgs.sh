# developer tool: tracking-chain parity tests, then config-5 and config-2 per-kernel timings
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_stream_group.py tests/test_pin_numerics.py -x -q -m gpu 2>&1 | tail -3 || exit 1
echo "== config5"; timeout -k 10 280 python scripts/run_config5.py --frames 40 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['frames_per_s'], r['ate_rmse_m'], {k: v['avg_us'] for k, v in r['kernels'].items() if k in ('k_pose_lm','k_match','k_ransac_score','k_ransac_hyp','k_ransac_select','k_frustum','k_match_gate','k_match_emit')}, r['avg_per_tracked_frame'])" || exit 1
