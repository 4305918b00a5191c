for its in 1,0 2,0 4,0 10,0 10,10; do timeout -k 10 100 python scripts/bench_ba.py --reps 1 --oracle --shapes small --its $its || exit 1; done
echo "== G=2"; for its in 1,0 10,10; do VO_BA_GROUP=2 timeout -k 10 100 python scripts/bench_ba.py --reps 1 --oracle --shapes small --its $its || exit 1; done
