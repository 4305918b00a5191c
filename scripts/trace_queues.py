"""Which kernels share a hardware queue in (the tail of) a rocprofv3 --kernel-trace CSV.  usage: trace_queues.py <dir> [tail_frac]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "0")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t1 - int((t1 - t0) * frac)
sel = [r for r in rows if r[0] >= cut]
q = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
for s, e, n, qi in sel:
    q[qi][n][0] += e - s; q[qi][n][1] += 1
print("window %.1f ms, %d kernels" % ((t1 - cut) / 1e6, len(sel)))
for qi, ks in sorted(q.items(), key=lambda kv: -sum(v[0] for v in kv[1].values())):
    tot = sum(v[0] for v in ks.values())
    top = sorted(ks.items(), key=lambda kv: -kv[1][0])[:5]
    print("queue %3s: %7.1f ms in %5d kernels: %s" % (qi, tot / 1e6, sum(v[1] for v in ks.values()), ", ".join("%s %.0fms/%d" % (n, v[0] / 1e6, v[1]) for n, v in top)))
