echo "== engine phase2"; VO_TRACE=1 VO_BA_PERSIST=0 timeout -k 10 100 python scripts/bench_ba.py --reps 20 --oracle --shapes bench 2>&1 | grep -v "resident" | cut -c1-900 || exit 1
VO_BA_PERSIST=0 timeout -k 10 100 python scripts/bench_ba.py --reps 3 --oracle --shapes small 2>&1 | cut -c1-700 || exit 1
