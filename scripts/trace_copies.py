"""Memory copies in a rocprofv3 --memory-copy-trace CSV around the preloads (copies >= 250 us): how long the other copies took while
preloads were in flight.  usage: trace_copies.py <dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", ""), r.get("Stream_Id", "?")))
rows.sort()
big = [r for r in rows if r[1] - r[0] >= 250000 and r[2] == "DEVICE_TO_DEVICE"]
if not big:
    print("no preloads in the trace"); sys.exit(0)
w0, w1 = big[0][0], big[-1][1]
print("preloads: %d copies, %.1f ms total, avg %.0f us, window %.1f ms" % (len(big), sum(r[1] - r[0] for r in big) / 1e6, sum(r[1] - r[0] for r in big) / len(big) / 1e3, (w1 - w0) / 1e6))
other = [r for r in rows if w0 <= r[0] <= w1 and r[1] - r[0] < 250000]
import collections
by = collections.defaultdict(list)
for s, e, d, st in other:
    by[(d, st)].append((e - s) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1]))[:8]:
    v.sort()
    print("  %-18s stream %-3s n=%5d  median %6.1f us  p95 %7.1f  max %8.1f  sum %7.2f ms" % (k[0], k[1], len(v), v[len(v) // 2], v[int(0.95 * len(v))], v[-1], sum(v) / 1e3))
# overlap: how many other copies started while a preload was running
inside = 0
j = 0
for s, e, d, st in other:
    while j < len(big) and big[j][1] < s: j += 1
    if j < len(big) and big[j][0] <= s <= big[j][1]: inside += 1
print("  copies that started while a preload was running: %d of %d" % (inside, len(other)))
