# per-kernel time of the step's head against the population (rocprofv3 kernel stats): is a kernel's time a step function of
# ceil(waves / 8192 wave slots) -- wave generations -- or linear in the agents?
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/bysize; mkdir -p $O
for n in 400000 330000 280000 262000 250000 200000 131000 125000 65000; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$n -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --agents $n > /dev/null 2>&1
python - <<PY >> $O/table.txt
import csv
rows = {r["Name"].split("(")[0].replace("magent_amd::","").replace("void ",""): float(r["AverageNs"])/1e3 for r in csv.DictReader(open("$O/p_$n/b_kernel_stats.csv"))}
keys = ["k_shuffle_draw","k_plain_rank","k_plain_eval","k_strike","k_plain_init","k_plain_commit","k_clear_compact","k_clear_finish","k_set_action_a","k_get_reward","k_step_report"]
print("$n", " ".join("%s=%.1f" % (k[2:], rows.get(k, 0)) for k in keys))
PY
rm -rf $O/p_$n
done
cat $O/table.txt
