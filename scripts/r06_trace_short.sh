#!/bin/bash
# kernel trace of the driver's short form -> gpurun_out/r06_short_timeline.txt
set -o pipefail
R=$PWD; O=$R/gpurun_out/r06ts; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency-mode --multi-streams= --no-roofline-pass > $O/bench_traced.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 2; }
cd $R
python scripts/trace_short_form.py $O/tr > gpurun_out/r06_short_timeline.txt 2>&1
rm -rf $O/tr
tail -c 300 $O/bench_traced.json; head -24 gpurun_out/r06_short_timeline.txt
