#!/bin/bash
# r06_trace_ba.sh TREE TAG: kernel trace of the timed pass of TREE's bench.py ('.' or ab_NAME) -> gpurun_out/r06t_TAG/{ba_span.txt,kernel_stats.csv,bench_traced.json}
set -o pipefail
R=$PWD; T=$1; O=$R/gpurun_out/r06t_$2; mkdir -p $O
if [ "$T" = "." ]; then B=$R/bench.py; else B=$R/$T/bench.py; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $B --steps ${STEPS:-96} --warmup ${WARMUP:-32} --no-cpu-baseline --no-latency-mode --multi-streams= --no-roofline-pass > $O/bench_traced.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 2; }
cd $R
python scripts/trace_ba_span.py $O/tr ${NPRINT:-1} > $O/ba_span.txt 2>&1
find $O/tr -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; rm -rf $O/tr
head -3 $O/ba_span.txt; grep -E "k_ba_cholup|k_ba_schur2|k_ba_lin2|k_ba_round" $O/kernel_stats.csv | cut -d, -f1-5
