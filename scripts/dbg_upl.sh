#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_run_vo.py -x -q -m gpu 2>&1 | tail -2 || exit 1
timeout -k 10 500 python bench.py --no-cpu-baseline --multi-streams= 2>/dev/null | python -c 'import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["upload_inclusive"], r["latency_mode"]["frames_per_s"])'
