# development (GPU box): memory-side counters of the policy kernels (separate passes);  usage: bash tools/policy_pmc.sh <outdir>
out=${1:-gpurun_out/polpmc}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$out/$tag -o p -- python $GRAFT_REPO_ROOT/tools/policy_rate.py 131072 30 cells > $GRAFT_REPO_ROOT/$out/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_dqn" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[1][:22] if False else r["Kernel_Name"][:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print("%-46s %-24s avg %.4g over %d launches" % (k[0], k[1], sum(v) / len(v), len(v)))
PY
rm -rf $out/*/
