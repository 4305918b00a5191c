echo "== engine phase2"; VO_TRACE=1 timeout -k 10 100 python scripts/bench_ba.py --reps 20 --oracle --shapes bench 2>&1 | grep -v "resident" | cut -c1-900 || exit 1
timeout -k 10 100 python scripts/bench_ba.py --reps 3 --oracle --shapes small 2>&1 | cut -c1-700 || exit 1
echo "== persist 128"; VO_BA_PERSIST=1 VO_BA_GROUP=128 timeout -k 10 100 python scripts/bench_ba.py --reps 20 --oracle --shapes bench 2>&1 | grep -v "resident" | cut -c1-900|| exit 1
VO_BA_PERSIST=1 timeout -k 10 100 python scripts/bench_ba.py --reps 3 --oracle --shapes small 2>&1 | cut -c1-700 || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "local_ba or resident or vo_system" 2>&1 | tail -2
