R="timeout -k 10 500 python scripts/exp_multistream.py --frames 330 --modes group --host-graph"
for e in 2 4; do for s in 8 16; do echo "== engines $e streams $s"; VO_BA_ENGINES=$e $R --streams $s 2>/dev/null || exit 1; done; done
echo "== engines 4, 2 group chains, 16 streams"; VO_BA_ENGINES=4 VO_GROUP_CHAINS=2 $R --streams 16 2>/dev/null || exit 1
