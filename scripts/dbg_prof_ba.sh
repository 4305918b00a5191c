cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ba2 -- python3 scripts/bench_ba.py --reps 20 --shapes bench > gpurun_out/prof_ba2.log 2>&1 || exit 1
python scripts/trace_gaps.py gpurun_out/prof_ba2
rm -rf gpurun_out/prof_ba2
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_b -- python3 bench.py --no-cpu-baseline --no-latency-mode --multi-streams= --steps 150 --warmup 30 > gpurun_out/prof_b.log 2>&1 || exit 1
python scripts/trace_gaps.py gpurun_out/prof_b
rm -rf gpurun_out/prof_b
