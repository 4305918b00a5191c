"""Gaps between consecutive kernels on every queue of a rocprofv3 --kernel-trace CSV, for ALL kernels: per (previous -> next) pair the number of
occurrences, the average start(next) - end(previous) (gaps above `cap` us are idle time, not a hand-over, and are left out) and the total.
usage: trace_queue_gaps.py <dir with *_kernel_trace.csv> [cap_us=40] [min_pairs=40]"""
import csv, glob, os, re, sys
from collections import defaultdict

cap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 40e3
minp = int(sys.argv[3]) if len(sys.argv) > 3 else 40
byq = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").strip()))
gaps = defaultdict(lambda: [0, 0]); dur = defaultdict(lambda: [0, 0])
for q, lst in byq.items():
    lst.sort()
    for (s0, e0, n0), (s1, e1, n1) in zip(lst, lst[1:]):
        if 0 <= s1 - e0 < cap:
            g = gaps[(n0, n1)]; g[0] += s1 - e0; g[1] += 1
    for s, e, n in lst:
        d = dur[n]; d[0] += e - s; d[1] += 1
tot = sum(t for (t, c) in gaps.values())
print("gaps below %.0f us between consecutive kernels of a queue: %.2f ms in total" % (cap / 1e3, tot / 1e6))
for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0]):
    if c >= minp:
        print("%-20s -> %-20s %6d pairs  avg gap %6.2f us  total %7.2f ms   (%s %.1f us, %s %.1f us)" % (a[:20], b[:20], c, t / c / 1e3, t / 1e6, a[:12], dur[a][0] / dur[a][1] / 1e3, b[:12], dur[b][0] / dur[b][1] / 1e3))
