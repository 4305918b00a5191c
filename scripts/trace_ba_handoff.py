"""Between two local BAs: what the host did while the GPU's BA chain was idle.  Input: a rocprofv3 run with --kernel-trace --hip-runtime-trace (CSV).
For the median cycle: the HIP API calls (thread, name, start, duration) from the end of a BA's last k_ba_round to the next BA's first k_ba_schur2.
usage: trace_ba_handoff.py <dir> [cycle index]"""
import csv, glob, os, re, sys
d = sys.argv[1]
kern, api = [], []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        n = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").strip()
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r["Thread_Id"]))
kern.sort(); api.sort()
rounds = [k for k in kern if k[2] == "k_ba_round"]
firsts = [k for k in kern if k[2] == "k_ba_admit"]
cycles = []
for a in firsts:
    prev = [r for r in rounds if r[1] <= a[0]]
    if prev: cycles.append((prev[-1][1], a[0]))
cycles = [c for c in cycles if 0 < c[1] - c[0] < 2_000_000]
cycles.sort(key=lambda c: c[1] - c[0])
print("cycles %d: idle between last k_ba_round and next k_ba_admit: min %.0f median %.0f max %.0f us" % (len(cycles), (cycles[0][1] - cycles[0][0]) / 1e3, (cycles[len(cycles) // 2][1] - cycles[len(cycles) // 2][0]) / 1e3, (cycles[-1][1] - cycles[-1][0]) / 1e3))
c = cycles[int(sys.argv[2]) if len(sys.argv) > 2 else len(cycles) // 2]
t0 = c[0]
tids = {}
print("--- one cycle (t = 0 at the end of the last k_ba_round) ---")
ev = [("K", k[0], k[1], k[2], "") for k in kern if c[0] - 5000 <= k[0] <= c[1] + 30000] + [("A", a[0], a[1], a[2], a[3]) for a in api if c[0] - 5000 <= a[0] <= c[1] + 30000]
ev.sort(key=lambda e: e[1])
for kind, s, e, n, tid in ev:
    if tid and tid not in tids: tids[tid] = "T%d" % len(tids)
    print("%8.1f +%7.1f  %s %s" % ((s - t0) / 1e3, (e - s) / 1e3, "GPU " if kind == "K" else tids[tid] + "  ", n))
