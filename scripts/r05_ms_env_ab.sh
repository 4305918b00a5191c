#!/bin/bash
# several streams per GPU under environment switches (same box, alternating): scripts/r05_ms_env_ab.sh "VAR=val" ...
for rep in 1 2; do
  for v in "" "$@"; do
    for s in 8 16; do
      r=$(env $v python scripts/exp_multistream.py --frames 330 --modes group --streams $s 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['fps'])")
      echo "[${v:-baseline}] $s streams: $r"
    done
  done
done
