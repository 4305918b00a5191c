#!/bin/bash
# Round-4 measurement set, part 2 (gpurun_out/r04m/): config 5, streams per GPU (host and device graph cut), kernel-trace summaries of the
# 8-stream run and of the single-stream BA step (boundaries between the step kernels), the chol_bench table, the engine batching probe.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04m; mkdir -p $O
timeout -k 10 300 python scripts/run_config5.py > $O/r04_config5.json 2> $O/config5.err || { tail -5 $O/config5.err; exit 1; }
echo "config5: $(head -c 200 $O/r04_config5.json)"
: > $O/r04_multistream.jsonl; : > $O/r04_multistream_device_graph.jsonl
for s in 1 2 4 8 16 32; do
  timeout -k 10 400 python scripts/exp_multistream.py --frames 330 --modes group --host-graph --streams $s 2>/dev/null >> $O/r04_multistream.jsonl || exit 2
done
for s in 8 16; do
  timeout -k 10 400 python scripts/exp_multistream.py --frames 330 --modes group --streams $s 2>/dev/null >> $O/r04_multistream_device_graph.jsonl || exit 2
done
echo "multistream done"; cut -c1-60 $O/r04_multistream.jsonl
(cd rgbd_visualodometry_amd/csrc && timeout -k 10 120 ./build/chol_bench 2>&1 | grep -E "^D |probe|mismatch" > $O/r04_chol_bench.txt; timeout -k 10 120 ./build/chol_bench_stamps 2>&1 | grep -E "^D |clocks" >> $O/r04_chol_bench.txt) || exit 3
VO_BA_ENGINES=1 timeout -k 10 250 python scripts/bench_ba_batch.py --threads 1,2,4,8 2>&1 | grep -v vo_trace > $O/r04_ba_engine_batching.jsonl || exit 4
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace8 -- python3 $R/scripts/exp_multistream.py --frames 330 --modes group --host-graph --streams 8 > $O/trace8.log 2>&1 || { tail -5 $O/trace8.log; exit 5; }
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -- python3 $R/bench.py --steps 96 --warmup 32 --no-cpu-baseline --no-latency-mode --multi-streams= > $O/trace1.log 2>&1 || { tail -5 $O/trace1.log; exit 6; }
cd $R
python scripts/trace_busy.py $O/trace8 --tail-frac 0.5 --out $O/r04_multistream8_gpu_busy.json > /dev/null
python scripts/trace_gaps.py $O/trace1 > $O/r04_ba_gaps.txt 2>&1 || true
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
ls -la $O | head -40
(cd rgbd_visualodometry_amd/csrc && timeout -k 10 120 ./build/queue_map > $O/r04_queue_map.txt 2>&1) || true
