#!/bin/bash
# Round-2 variants of the bench line (GPU box; writes gpurun_out/r02v/)
R=$PWD; O=$R/gpurun_out/r02v; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-latency-mode --multi-streams ''"
eval $B --speed 1 > $O/speed1.json 2>/dev/null
eval $B --steps 1000 > $O/steps1000.json 2>/dev/null
eval $B --no-ba > $O/noba.json 2>/dev/null
eval $B --ba-lag 0 > $O/lag0.json 2>/dev/null
eval $B --hyps 2048 > $O/h2048.json 2>/dev/null
eval $B --features 500 > $O/n500.json 2>/dev/null
eval $B --steps 20 --warmup 5 > $O/driver_short.json 2>/dev/null
for s in 1 2 3; do eval $B --seed $s > $O/seed$s.json 2>/dev/null; done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"unreadable",e); continue
    print(os.path.basename(f), d["value"], "fps  ms/step", d["ms_per_step"], "ate", d["ate_rmse_m"], "rpe", d["rpe_trans_rmse_m"], "kf_timed", d["keyframes_timed"], "ba_timed", d["ba_runs_timed"], "lost", d["lost"], "hbm", d["hbm_frac_whole_frame"])
PY
