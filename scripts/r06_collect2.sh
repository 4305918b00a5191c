#!/bin/bash
# Round-6 measurement set, part 2 (gpurun_out/r06m/): the local BA's chain itemised from a kernel trace of the timed pass, streams per GPU,
# the 8-stream kernel trace, config 5, row a-1 alone, chol_bench.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r06m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/bench.py --steps 96 --warmup 32 --no-cpu-baseline --no-latency-mode --multi-streams= --no-roofline-pass > $O/bench_traced.json 2> $O/trace.err || { tail -5 $O/trace.err; exit 2; }
cd $R
python scripts/trace_ba_span.py $O/tr 8 > $O/r06_ba_span.txt 2>&1
python scripts/trace_gaps.py $O/tr > $O/r06_ba_gaps.txt 2>&1
VO_TRACE=1 python bench.py --no-cpu-baseline --no-latency-mode --multi-streams= --no-roofline-pass 2> $O/vo_trace.err > /dev/null; grep "scope\|frontend ms\|BA runs\|k_pyramid" $O/vo_trace.err > $O/r06_vo_trace_scopes.txt
timeout -k 10 300 python scripts/run_config5.py > $O/r06_config5.json 2> $O/config5.err || { tail -5 $O/config5.err; exit 1; }
timeout -k 10 300 python scripts/run_config5.py --device > $O/r06_config5_device.json 2> $O/config5d.err || { tail -5 $O/config5d.err; }
echo "config5: $(head -c 200 $O/r06_config5.json)"
python scripts/bench_orb_only.py 2000 > $O/r06_orb_only.txt 2>/dev/null; python scripts/bench_orb_only.py 500 >> $O/r06_orb_only.txt 2>/dev/null; cat $O/r06_orb_only.txt
: > $O/r06_multistream.jsonl
for s in 1 2 4 8 16 32; do
  timeout -k 10 400 python scripts/exp_multistream.py --frames 330 --modes group --streams $s 2>/dev/null >> $O/r06_multistream.jsonl || exit 2
done
echo "multistream done"; cut -c1-80 $O/r06_multistream.jsonl
(cd rgbd_visualodometry_amd/csrc && timeout -k 10 120 ./build/chol_bench 2>&1 | grep -E "^D |probe|mismatch" > $O/r06_chol_bench.txt) || true
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace8 -- python3 $R/scripts/exp_multistream.py --frames 330 --modes group --streams 8 > $O/trace8.log 2>&1 || { tail -5 $O/trace8.log; exit 5; }
cd $R
python scripts/trace_busy.py $O/trace8 --tail-frac 0.5 --out $O/r06_multistream8_gpu_busy.json > /dev/null
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete
ls $O | head -50
