"""Summarise a rocprofv3 --kernel-trace CSV: GPU busy time (union of kernel intervals), per-queue busy time, top kernels, and the
same restricted to the last `--tail-frac` of the trace (the timed region of the experiment scripts)."""
import argparse, csv, glob, json, collections


def union(iv):
    iv.sort()
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--tail-frac", type=float, default=0.7)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    f = glob.glob(a.dir + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "0")))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t1 - int((t1 - t0) * a.tail_frac)
    sel = [r for r in rows if r[0] >= cut]
    span = t1 - cut
    busy = union([(r[0], r[1]) for r in sel])
    perq = collections.defaultdict(list)
    for r in sel: perq[r[3]].append((r[0], r[1]))
    byk = collections.defaultdict(lambda: [0, 0])
    for r in sel: byk[r[2]][0] += r[1] - r[0]; byk[r[2]][1] += 1
    top = sorted(byk.items(), key=lambda kv: -kv[1][0])[:25]
    out = {"trace": f, "window_ms": span / 1e6, "gpu_busy_ms": busy / 1e6, "gpu_busy_frac": busy / span, "sum_kernel_ms": sum(r[1] - r[0] for r in sel) / 1e6,
           "kernels": len(sel), "queues": {q: {"busy_ms": union(v) / 1e6, "kernels": len(v)} for q, v in sorted(perq.items(), key=lambda kv: -len(kv[1]))[:12]},
           "top_kernels": [{"name": k, "ms": v[0] / 1e6, "calls": v[1], "avg_us": v[0] / v[1] / 1e3} for k, v in top]}
    s = json.dumps(out, indent=1)
    print(s)
    if a.out: open(a.out, "w").write(s)


if __name__ == "__main__":
    main()
