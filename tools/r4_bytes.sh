# bytes and time of the step kernels only: FETCH_SIZE / WRITE_SIZE passes + kernel stats (a subset of tools/step_pmc.sh)
R=$GRAFT_REPO_ROOT; TAG=${1:-r4bytes}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-profile --no-extras"
for c in FETCH_SIZE WRITE_SIZE; do timeout 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o bench -- $B > $O/$c.log 2>&1; done
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B > $O/stats.log 2>&1
python - <<PY
import csv, collections
def avg(path, name):
    a=collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"]==name: a[r["Kernel_Name"].split("(")[0].replace("void ","").replace("magent_amd::","")].append(float(r["Counter_Value"]))
    return {k:(sum(v)/len(v), len(v)) for k,v in a.items()}
f=avg("$O/FETCH_SIZE/bench_counter_collection.csv","FETCH_SIZE"); w=avg("$O/WRITE_SIZE/bench_counter_collection.csv","WRITE_SIZE")
t={r["Name"].split("(")[0].replace("void ","").replace("magent_amd::",""): float(r["AverageNs"])/1e3 for r in csv.DictReader(open("$O/stats/bench_kernel_stats.csv"))}
tot=0
for k in sorted(f, key=lambda k:-(f[k][0]+w.get(k,(0,0))[0])*f[k][1]):
    if "render" in k or "paint" in k: continue
    per=f[k][1]/8.0
    mb=(f[k][0]+w.get(k,(0,0))[0])*1024/1e6
    tot+=mb*per
    print("%-22s x%.1f/step  fetch %6.1f write %6.1f MB  %6.1f us"%(k[:22], per, f[k][0]*1024/1e6, w.get(k,(0,0))[0]*1024/1e6, t.get(k,0)))
print("MB per step (non-render):", round(tot,1))
PY
