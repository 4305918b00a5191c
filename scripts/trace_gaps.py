"""Gaps between consecutive kernels of the local BA's LM step in a rocprofv3 --kernel-trace CSV: for every (previous kernel -> next kernel) pair on
one queue, the average of start(next) - end(previous).  usage: trace_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, re, sys
from collections import defaultdict

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((r.get("Queue_Id", "0"), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"_one$", "", re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").strip())))
byq = defaultdict(list)
for q, s, e, n in rows:
    byq[q].append((s, e, n))
gaps = defaultdict(lambda: [0, 0]); dur = defaultdict(lambda: [0, 0])
for q, lst in byq.items():
    lst.sort()
    for (s0, e0, n0), (s1, e1, n1) in zip(lst, lst[1:]):
        if n0.startswith("k_ba_") and n1.startswith("k_ba_") and s1 - e0 < 200000:
            g = gaps[(n0, n1)]; g[0] += s1 - e0; g[1] += 1
    for s, e, n in lst:
        if n.startswith("k_ba_"):
            d = dur[n]; d[0] += e - s; d[1] += 1
print("kernel durations (us):", {k: round(v[0] / v[1] / 1e3, 2) for k, v in dur.items() if v[1] > 50})
for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][1]):
    if c > 50:
        print("%-16s -> %-16s  %6d pairs  avg gap %7.2f us" % (a, b, c, t / c / 1e3))
