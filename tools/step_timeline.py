"""Timeline of ONE step out of a rocprofv3 --kernel-trace csv: start offset, duration and idle gap before every kernel.

usage: python tools/step_timeline.py gpurun_out/<dir>/bench_kernel_trace.csv [k-th step from the end, default 3] [kernel that ends a cycle]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
names = [r["Kernel_Name"].split("(")[0].replace("magent_amd::", "").replace("void ", "") for r in rows]
marks = []
# a cycle ends with clear_dead's last launch (plain games: k_clear_finish, or k_mini_norm when nobody died; the one-launch step: its own launch)
for mark in ([sys.argv[3]] if len(sys.argv) > 3 else ["k_step_report", "k_plain_commit", "k_clear_solo_all", "k_step_solo"]):
    marks = [i for i, n in enumerate(names) if n.startswith(mark)]
    if len(marks) > back:
        break
a, b = marks[-back], marks[-back + 1]
t0 = prev = int(rows[a]["End_Timestamp"])
busy = gaps = 0
for i in range(a + 1, b + 1):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    busy += e - s
    gaps += max(0, s - prev)
    print("%8.1f %7.1f %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, names[i][:60]))
    prev = max(prev, e)
print("span %.1f us  busy %.1f  gaps %.1f  launches %d" % ((prev - t0) / 1e3, busy / 1e3, gaps / 1e3, b - a))
