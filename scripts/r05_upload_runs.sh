#!/bin/bash
# 20 consecutive 300-step runs, resident and PCIe-inclusive figure of each -> gpurun_out/r05_upload_runs.jsonl
O=gpurun_out/r05_upload_runs.jsonl; : > $O
for i in $(seq 1 ${1:-20}); do
  python bench.py --upload-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'run': $i, 'resident': d['value'], 'upload_inclusive': d['upload_inclusive']['frames_per_s'], 'vs_resident': d['upload_inclusive']['vs_resident']}))" >> $O
  tail -1 $O
done
