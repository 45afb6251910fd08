# development (GPU box): where the self-play step's wall time goes -- kernel time vs gaps on the device
out=${1:-gpurun_out/sptrace}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/t -o sp -- python -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import torch, magent_amd, bench
print(bench.selfplay_extra(torch, magent_amd, steps=8))
" > $GRAFT_REPO_ROOT/$out/run.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $out/run.log | cut -c1-200
python - <<PY
import csv, glob, collections
f = glob.glob("$out/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last 4 steps: find k_step-like boundaries by the move commit kernel
ends = [i for i, r in enumerate(rows) if "k_move_commit" in r["Kernel_Name"]]
a, b = ends[-5], ends[-1]
seg = rows[a + 1:b + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = 0; cur_end = t0; gaps = []
agg = collections.Counter()
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    agg[r["Kernel_Name"].split("(")[0][-40:]] += e - s
    if s > cur_end: gaps.append((s - cur_end, r["Kernel_Name"][:50]))
    if e > cur_end: busy += e - max(s, cur_end); cur_end = e
print("4 steps: wall %.3f ms per step, device busy %.3f ms per step, %d gaps %.3f ms per step" % ((t1 - t0) / 4e6, busy / 4e6, len(gaps), sum(g for g, _ in gaps) / 4e6))
for k, v in agg.most_common(8): print("  %-42s %.3f ms per step" % (k, v / 4e6))
for g, n in sorted(gaps, reverse=True)[:12]: print("  gap %.1f us before %s" % (g / 1e3, n))
PY
rm -rf $out/t
