#!/bin/bash
# Round-6 measurement set (run on the GPU box through gpurun; writes gpurun_out/r06m/): PMC passes of the bench command (FETCH_SIZE / WRITE_SIZE,
# two SQ passes: separate --pmc runs, kernel trace only), rocprofv3 kernel stats, the bench line (300 steps) and the driver's short form x 3.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r06m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--steps 96 --warmup 32 --no-cpu-baseline --no-latency-mode --multi-streams="
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_$pass -- python3 $R/bench.py $Q > $O/bench_pmc_$pass.json 2> $O/pmc_$pass.err || { tail -5 $O/pmc_$pass.err; exit 3; }
  echo "pmc $pass done"
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_c1 -- python3 $R/bench.py $Q > $O/bench_pmc_c1.json 2> $O/pmc_c1.err || { tail -5 $O/pmc_c1.err; exit 4; }
echo "pmc c1 done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F64 --kernel-trace --output-format csv -d $O/pmc_c2 -- python3 $R/bench.py $Q > $O/bench_pmc_c2.json 2> $O/pmc_c2.err || { tail -5 $O/pmc_c2.err; exit 5; }
echo "pmc c2 done"
cd $R
python scripts/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/r06_pmc_hbm_traffic.json "rocprofv3 --pmc <CTR> --kernel-trace --output-format csv -- python3 bench.py $Q (one pass per counter: FETCH_SIZE, WRITE_SIZE)"
python scripts/pmc_compute_summary.py $O/r06_pmc_compute.json "rocprofv3 --pmc <8 SQ counters + GRBM_GUI_ACTIVE> --kernel-trace --output-format csv -- python3 bench.py $Q (two passes)" $O/pmc_c1 $O/pmc_c2
cp $O/r06_pmc_hbm_traffic.json $O/r06_pmc_compute.json $R/profiles/          # the bench line below reads them (traffic, compute counters)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-latency-mode --multi-streams= > $O/r06_bench_under_rocprof.json 2> $O/rocprof.err || { tail -5 $O/rocprof.err; exit 2; }
echo "rocprof stats done"
cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r06_bench_kernel_stats.csv
python bench.py > $O/r06_bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cp bench_detail.json $O/r06_bench_detail.json
echo "bench done: $(head -c 300 $O/r06_bench.json)"
: > $O/r06_bench_driver_short_3runs.jsonl
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 >> $O/r06_bench_driver_short_3runs.jsonl 2>> $O/bench.err || { tail -5 $O/bench.err; exit 1; }; done
cp bench_detail.json $O/r06_bench_driver_short_detail.json
echo "driver-style bench done: $(cut -c1-120 $O/r06_bench_driver_short_3runs.jsonl)"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
ls -la $O
