R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ric; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in none moves attacks random; do
rocprofv3 --kernel-trace --output-format csv -d $O/$w -o t -- python $R/tools/render_in_cycle.py $w > $O/$w.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/$w/t_kernel_trace.csv"))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
ren=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "k_render" in r["Kernel_Name"]]
print("$w", open("$O/$w.log").read().strip().splitlines()[-1], "renders:", len(ren), "avg of launches 4.. : %.1f us" % (sum(ren[4:])/len(ren[4:])), [round(x) for x in ren[4:20]])
PY
done
