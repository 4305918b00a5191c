#!/bin/bash
# Round-2 measurement set (run on the GPU box through gpurun; writes gpurun_out/r02m/): bench line, rocprofv3 kernel stats of the
# same command, FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, kernel trace only).
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02m; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "bench done: $(head -c 300 $O/bench.json)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-latency-mode --multi-streams "" > $O/bench_rocprof.json 2> $O/rocprof.err || { tail -5 $O/rocprof.err; exit 2; }
echo "rocprof stats done"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -- python3 $R/bench.py --steps 96 --warmup 32 --no-cpu-baseline --no-latency-mode --multi-streams "" > $O/bench_pmc_$ctr.json 2> $O/pmc_$ctr.err || { tail -5 $O/pmc_$ctr.err; exit 3; }
  echo "pmc $ctr done"
done
cd $R
python scripts/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/r02_pmc_hbm_traffic.json "rocprofv3 --pmc <CTR> --kernel-trace --output-format csv -- python3 bench.py --steps 96 --warmup 32 --no-cpu-baseline --no-latency-mode --multi-streams '' (one pass per counter: FETCH_SIZE, WRITE_SIZE)"
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r02_bench_kernel_stats.csv
# keep the merged output small: the raw traces are not needed
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
ls -la $O
