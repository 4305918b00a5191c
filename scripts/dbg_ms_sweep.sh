#!/bin/bash
# several streams in one group, host graph cut: group chains in flight and BA engines per GPU
R="timeout -k 10 400 python scripts/exp_multistream.py --frames 330 --modes group --host-graph"
for cfg in "8 1 2" "8 2 2" "16 1 2" "16 2 2" "16 2 3"; do set -- $cfg
  echo "== streams $1 chains $2 engines $3"; VO_GROUP_CHAINS=$2 VO_BA_ENGINES=$3 $R --streams $1 2>/dev/null | cut -c1-120 || exit 1
done
