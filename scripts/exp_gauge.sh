#!/bin/bash
# does anchoring the gauge of the local BA (oldest free keyframe fixed -- NOT a reference setting) give back the accuracy of
# the run without BA?  Bench workload, seeds 0-3: ATE / RPE with the faithful free-gauge BA, with the anchored BA, and with the BA off.
Q="--no-cpu-baseline --no-latency-mode --multi-streams= --steps 300 --warmup 30"
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k: r[k] for k in ("value","ate_rmse_m","rpe_trans_rmse_m","rpe_rot_rmse_deg","ba_runs","lost")}))'
for seed in 0 1 2 3; do
  echo "seed $seed free-gauge BA (reference): $(timeout -k 10 280 python bench.py $Q --seed $seed 2>/dev/null | python -c "$P")" || exit 1
  echo "seed $seed anchored BA              : $(VO_BA_FIX_OLDEST=1 timeout -k 10 280 python bench.py $Q --seed $seed 2>/dev/null | python -c "$P")" || exit 1
  echo "seed $seed BA off                   : $(timeout -k 10 280 python bench.py $Q --seed $seed --no-ba 2>/dev/null | python -c "$P")" || exit 1
done
# follow-up: is it the reference's 1-pixel chi2 cut (config/default.yaml:29, unit information matrix whatever the keypoint's octave) that costs the accuracy?
for seed in 0 1; do for th in 9 36; do
  echo "seed $seed free-gauge BA, chi2_th $th: $(timeout -k 10 280 python bench.py $Q --seed $seed --chi2-th $th 2>/dev/null | python -c "$P")" || exit 1
done; done
