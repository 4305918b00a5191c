#!/bin/bash
# Round-1 measurement set (run on the GPU box through gpurun; writes gpurun_out/r01c/)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r01c; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"; cat $O/bench.json | head -c 400; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline > $O/bench_rocprof.json 2> $O/rocprof.err || exit 2
echo "rocprof done"
cd $R
python bench.py --no-cpu-baseline --steps 1000 > $O/bench_1000.json 2>/dev/null || exit 3
python bench.py --no-cpu-baseline --steps 1000 --no-ba > $O/bench_1000_noba.json 2>/dev/null || exit 4
python bench.py --no-cpu-baseline --features 500 > $O/bench_n500.json 2>/dev/null || exit 5
python bench.py --no-cpu-baseline --hyps 2048 > $O/bench_h2048.json 2>/dev/null || exit 6
python bench.py --no-cpu-baseline --ba-lag 0 > $O/bench_lag0.json 2>/dev/null || exit 7
echo "variants done"
for s in 1 2 3; do python bench.py --seed $s --cpu-frames 120 > $O/bench_seed$s.json 2>/dev/null || exit 8; done
echo "seeds done"
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f),"unreadable",e); continue
    c=d.get("cpu_baseline") or {}
    print(os.path.basename(f), d["value"], "fps ate", d["ate_rmse_m"], "kf", d["keyframes"], "lost", d["lost"], "| cpu", c.get("value"), c.get("ate_rmse_m"), c.get("gpu_ate_rmse_m_same_frames"))
PY
