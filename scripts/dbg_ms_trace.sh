#!/bin/bash
# 8 streams in one group: throughput with the host and the device graph cut, then a kernel trace of the host-cut run
R=$PWD; O=$R/gpurun_out/ms8; mkdir -p $O
for g in "--host-graph" ""; do
  echo "== 8 streams, graph cut: ${g:-device}"; timeout -k 10 300 python scripts/exp_multistream.py --frames 330 --modes group $g --streams 8 2>/dev/null || exit 1
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/scripts/exp_multistream.py --frames 330 --modes group --host-graph --streams 8 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 2; }
cd $R
python scripts/trace_busy.py $O/trace --tail-frac 0.5 --out $O/r03_multistream8_gpu_busy.json | cut -c1-3000
find $O -name "*kernel_trace.csv" -size +60M -delete; find $O -name "*agent_info.csv" -delete
