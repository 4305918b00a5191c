#!/bin/bash
# k_match_mfma's LDS bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) and its duration: one --pmc pass of a short bench, one timing run
R=$PWD; O=$R/gpurun_out/r05mfma; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc -- python3 $R/bench.py --steps 32 --warmup 8 --prologue 100 --no-cpu-baseline --no-latency-mode --multi-streams= --no-roofline-pass > /dev/null 2> $O/pmc.err || { tail -3 $O/pmc.err; exit 2; }
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/r05mfma/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k in ("k_match_mfma", "k_match_gate", "k_fast_nms", "k_kf_place", "k_kf_covis_tri"):
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in agg.items():
    print("%-16s launches %4d  LDS bank conflict cycles / LDS active cycles = %.3f  (LDS instructions per launch %.0f)" % (k, n[(k, "SQ_LDS_IDX_ACTIVE")], c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), c["SQ_INSTS_LDS"] / max(n[(k, "SQ_INSTS_LDS")], 1)))
PY
find $O -name "*.csv" -delete
