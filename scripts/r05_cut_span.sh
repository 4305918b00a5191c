#!/bin/bash
# the local BA's chain from a kernel trace of a 96-step bench pass: first lines of scripts/trace_ba_span.py (medians + one BA's listing)
R=$PWD; O=$R/gpurun_out/cutspan; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --steps 96 --warmup 32 --no-cpu-baseline --no-latency-mode --multi-streams= --no-roofline-pass > $O/bench.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 2; }
cd $R
python scripts/trace_ba_span.py $O/tr 8 > $O/span.txt 2>&1
python scripts/trace_queue_gaps.py $O/tr 40 40 > $O/queue_gaps.txt 2>&1
head -24 $O/span.txt; head -45 $O/queue_gaps.txt; python -c "import json; print(json.load(open('$O/bench.json'))['value'])"
find $O -name "*kernel_trace.csv" -delete
