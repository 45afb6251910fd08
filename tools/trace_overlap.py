"""Concurrency seen in a rocprofv3 kernel trace: wall span, sum of kernel time, average number of kernels in flight,
per-queue kernel counts (last `frac` of the trace)."""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.5):]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
ev = sorted([(int(r["Start_Timestamp"]), 1) for r in rows] + [(int(r["End_Timestamp"]), -1) for r in rows])
cur = mx = 0
for _, d in ev:
    cur += d; mx = max(mx, cur)
q = collections.Counter(r.get("Queue_Id", "?") for r in rows)
names = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("magent_amd::", "").replace("void ", "")[:40]
    names[n][0] += 1; names[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("span %.1f us, kernel time %.1f us, avg in flight %.2f, max in flight %d, queues %s" % ((t1 - t0) / 1e3, busy / 1e3, busy / (t1 - t0), mx, dict(q)))
for n, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1]):
    print("  %-42s %6d launches  avg %7.1f us" % (n, c, t / c / 1e3))
