cd /tmp && export TMPDIR=/tmp
for n in 400000 320000 250000 130000; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/profn_$n -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --agents $n > /dev/null 2>&1
python - <<PY
import csv
rows = {r["Name"].split("(")[0].replace("magent_amd::",""): float(r["AverageNs"])/1e3 for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/profn_$n/b_kernel_stats.csv"))}
keys = ["k_attack_eval","k_shuffle_draw","k_move_commit","k_attack_apply","k_clear_compact","k_move_claim","k_move_prep","k_attack_rank","k_shuffle_chase","k_move_init","k_set_action_a","k_set_action_c","k_rule","k_get_reward"]
print("$n", " ".join("%s=%.1f" % (k[2:], rows.get(k, 0)) for k in keys))
PY
done
