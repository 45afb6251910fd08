R=$GRAFT_REPO_ROOT; cd /tmp
for a in "--workload gather --map-size 500 --agents 100000" "--workload battle_fill" "--map-size 3536 --agents 499849" ; do
timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --no-profile --repeats 1 --steps 40 --warmup 5 $a 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$a]', d['ms_per_step'], d['config']['attack_round_hist'], d['config']['steps_finished_by_host_driver'])"
done
