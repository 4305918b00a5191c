#!/bin/bash
# Round 5: kernel trace of the timed pass alone (no per-kernel timing pass) -> the local BA's chain itemised (scripts/trace_ba_span.py), kernel stats.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r05t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/bench.py --steps 96 --warmup 32 --no-cpu-baseline --no-latency-mode --multi-streams= --no-roofline-pass > $O/bench_traced.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 2; }
cd $R
python scripts/trace_ba_span.py $O/tr 12 > $O/ba_span.txt 2>&1
python scripts/trace_gaps.py $O/tr > $O/ba_gaps.txt 2>&1
find $O/tr -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
head -60 $O/ba_span.txt
