#!/bin/bash
# several streams in one group, host graph cut: BA engines per GPU and stream counts
R="timeout -k 10 400 python scripts/exp_multistream.py --frames 330 --modes group --host-graph"
for cfg in "16 1" "16 4" "32 2" "32 4"; do set -- $cfg
  echo "== streams $1 engines $2"; VO_BA_ENGINES=$2 $R --streams $1 2>/dev/null | cut -c1-120 || exit 1
done
