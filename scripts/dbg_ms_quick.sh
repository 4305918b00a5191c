#!/bin/bash
# 8 and 16 streams in one group: throughput with the host and the device graph cut
for s in 8 16; do for g in "--host-graph" ""; do
  echo "== $s streams, graph cut: ${g:-device}"; timeout -k 10 300 python scripts/exp_multistream.py --frames 330 --modes group $g --streams $s 2>/dev/null | cut -c1-330 || exit 1
done; done
