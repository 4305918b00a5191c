"""Timeline of the timed region of `bench.py --steps 20 --warmup 5` in a rocprofv3 --kernel-trace CSV: everything from the last batched ORB launch (k_gray of the timed
batch) to the end, as events: ORB batch, every tracking chain (k_frustum .. k_pose_lm), keyframe commits (k_kf_count), cuts (k_cut_init), merges (k_ba_merge), BA ends.
usage: trace_short_form.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, re, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            n = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").strip()
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "0"), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))))
rows.sort()
orb = [i for i, r in enumerate(rows) if r[2] == "k_gray"]
# the timed batch: the LAST ORB batch that is followed by six graph cuts before the next ORB batch (the legs behind the timed pass are shorter or have no BA)
i0 = None
for a, b in zip(orb, orb[1:] + [len(rows)]):
    if sum(1 for r in rows[a:b] if r[2] == "k_cut_init") == 6: i0, i1 = a, b
if i0 is None: sys.exit("no ORB batch with six cuts behind it")
rows = rows[:i1]
t0 = rows[i0][0]
print("timed region (from the last ORB batch's first kernel): %.2f ms, %d kernels" % ((rows[-1][1] - t0) / 1e6, len(rows) - i0))
mark = {"k_gray": "ORB batch starts", "k_describe": "ORB batch ends", "k_frustum": "tracking chain starts", "k_pose_lm": "pose LM", "k_kf_count": "keyframe commit", "k_cut_init": "BA cut starts",
        "k_ba_admit": "BA admitted", "k_ba_merge": "BA merged", "k_act_mark": "local-map query"}
last_round = None
for s, e, n, q, _g in rows[i0:]:
    if n == "k_ba_round": last_round = e
    if n in mark:
        extra = ""
        if n == "k_ba_merge" and last_round: extra = "  (last round ended %.0f us before)" % ((s - last_round) / 1e3)
        print("  %9.1f us +%7.1f  q%s  %-22s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n, mark[n], extra))

# per BA of the timed region: span and the step kernels' average durations
import statistics as st
bas, cur = [], None
for r in rows[i0:]:
    if r[2] == "k_cut_init":
        if cur: bas.append(cur)
        cur = []
    if cur is not None and (r[2].startswith("k_ba_") or r[2].startswith("k_cut") or r[2].startswith("k_ps_") or r[2].startswith("k_scan")): cur.append(r)
if cur: bas.append(cur)
for ba in bas:
    ch = [(e - s) / 1e3 for s, e, n, q, _g in ba if n.startswith("k_ba_cholup")]; sc = [(e - s) / 1e3 for s, e, n, q, _g in ba if n.startswith("k_ba_schur2")]
    li = [(e - s) / 1e3 for s, e, n, q, _g in ba if n == "k_ba_lin2"]; ro = [(e - s) / 1e3 for s, e, n, q, _g in ba if n == "k_ba_round"]
    adm = next((s for s, e, n, q, _g in ba if n == "k_ba_admit"), ba[0][0]); end = max((e for s, e, n, q, _g in ba if n == "k_ba_round"), default=ba[-1][1])
    steps = [(s, e) for s, e, n, q, _g in ba if n.startswith("k_ba_schur2") or n.startswith("k_ba_cholup")]
    gaps = sum(max(0, b[0] - a[1]) for a, b in zip(steps, steps[1:])) / 1e3
    print("BA: cut->admit %.0f us, admit->last round %.0f us; %d steps: cholup %.1f us avg (min %.1f max %.1f), schur2 %.1f (min %.1f max %.1f), lin2 %s, round %s; gaps between step kernels %.0f us" % (
        (adm - ba[0][0]) / 1e3, (end - adm) / 1e3, len(ch), st.mean(ch), min(ch), max(ch), st.mean(sc), min(sc), max(sc), ["%.0f" % x for x in li], ["%.0f" % x for x in ro], gaps))
    sg = [(g, (e - s) / 1e3) for s, e, n, q, g in ba if n.startswith("k_ba_schur2")]
    print("    schur2 launches (workgroups : us): " + " ".join("%d:%.1f" % x for x in sg[:20]))
    cg = [(g, (e - s) / 1e3) for s, e, n, q, g in ba if n.startswith("k_ba_cholup")]
    print("    cholup launches (workgroups : us): " + " ".join("%d:%.1f" % x for x in cg[:6]))
