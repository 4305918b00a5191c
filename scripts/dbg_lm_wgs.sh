# developer tool: the pose LM in 1 / 4 / 8 workgroups per lane: parity tests, then config-5 timings
for k in 1 4 8; do echo "== VO_LM_WGS=$k"; VO_LM_WGS=$k timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ransac or config5_sizes or track_frame or track_batch" 2>&1 | tail -2 || exit 1; done
for k in 1 0; do echo "== config5 VO_LM_WGS=$k (0 = automatic)"; VO_LM_WGS=$k timeout -k 10 280 python scripts/run_config5.py --frames 40 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['frames_per_s'], r['ate_rmse_m'], {k: v['avg_us'] for k, v in r['kernels'].items() if k in ('k_pose_lm','k_match','k_ransac_score','k_ransac_hyp','k_ransac_select')}, r['avg_per_tracked_frame'])" || exit 1; done
