cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
VO_BA_PERSIST=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ba2 -- python3 scripts/bench_ba.py --reps 20 --shapes bench > gpurun_out/prof_ba2.log 2>&1 || exit 1
f=$(find gpurun_out/prof_ba2 -name "*kernel_stats.csv" | head -1); head -14 $f | cut -d, -f1-4,6-7
