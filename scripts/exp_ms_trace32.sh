#!/bin/bash
R=$PWD; O=$R/gpurun_out/ms32; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/scripts/exp_multistream.py --frames 330 --modes group --host-graph --streams 32 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 2; }
cd $R
python scripts/trace_busy.py $O/trace --tail-frac 0.5 --out $O/r03_multistream32_gpu_busy.json > $O/busy.txt
python3 - <<'PY'
import json
r=json.load(open("gpurun_out/ms32/r03_multistream32_gpu_busy.json"))
print({k:r[k] for k in ("window_ms","gpu_busy_frac","sum_kernel_ms","kernels")})
for q,v in r["queues"].items(): print("queue",q,v)
for k in r["top_kernels"][:22]: print("%-28s %8.1f ms %6d calls %8.1f us"%(k["name"],k["ms"],k["calls"],k["avg_us"]))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
