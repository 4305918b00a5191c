"""Timeline of the local BAs in a rocprofv3 --kernel-trace CSV of bench.py: for every BA (k_ba_admit .. the k_ba_round that follows its last step) the span,
the kernel time inside it and the largest gaps.  usage: trace_ba_span.py <dir with *_kernel_trace.csv> [n to print]"""
import csv, glob, os, re, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            n = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").strip()
            n = re.sub(r"_one$", "", n) if n in ("k_ba_schur2_one", "k_ba_cholup_one") else n      # (a lone problem's step kernels: descriptor by value)
            if n.startswith("k_ba_") or n.startswith("k_cut") or n.startswith("k_scan") or n.startswith("k_ps_"):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "0")))
rows.sort()
bas, cur = [], None
for s, e, n, q in rows:
    if n == "k_cut_init":
        if cur: bas.append(cur)
        cur = []
    if cur is not None: cur.append((s, e, n, q))
if cur: bas.append(cur)
nprint = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tot = []
for ba in bas:
    steps = [k for k in ba if k[2] in ("k_ba_schur2",)]
    if len(steps) < 5: continue
    t0 = ba[0][0]; t_admit = next((k[0] for k in ba if k[2] == "k_ba_admit"), t0); t_end = max(k[1] for k in ba)
    first_step = steps[0][0]; last_round = max((k[1] for k in ba if k[2] == "k_ba_round"), default=t_end)
    busy = sum(k[1] - k[0] for k in ba)
    tot.append(((t_end - t0) / 1e3, (t_admit - t0) / 1e3, (first_step - t_admit) / 1e3, (last_round - first_step) / 1e3, len(steps), busy / 1e3))
import statistics as st
cyc = []
for a, b in zip(bas, bas[1:]):
    ma = [k for k in a if k[2] == "k_ba_merge"]; ra = [k for k in a if k[2] == "k_ba_round"]
    if not ma or not ra or len([k for k in a if k[2] == "k_ba_schur2"]) < 5: continue
    cyc.append(((ma[-1][0] - ra[-1][1]) / 1e3, (b[0][0] - ma[-1][1]) / 1e3, (b[0][0] - a[0][0]) / 1e3))
if cyc:
    print("between BAs (median us): last round end -> merge start %.0f, merge end -> next cut start %.0f, cut start -> next cut start %.0f" % tuple(st.median(x[i] for x in cyc) for i in range(3)))
print("BAs: %d; span cut..end %.0f us (median), cut->admit %.0f, admit->first step %.0f, first step->last round %.0f, steps %.1f, kernel time %.0f" % (
    len(tot), *[st.median(x[i] for x in tot) for i in range(6)]))
for ba in bas[len(bas) // 2: len(bas) // 2 + 1]:
    t0 = ba[0][0]; prev = t0
    for s, e, n, q in ba[:nprint * 10]:
        print("  %8.1f +%6.1f gap %6.1f  q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, q, n)); prev = e
