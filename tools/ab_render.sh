# development A/B (GPU box): the bench line's render figures under engine knobs;  usage: REPS=5 bash tools/ab_render.sh "VAR=val VAR=val" ...
run() {
  env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s ms/step %.4f  render avg %.4f ms  frac %.4f' % ('$1', r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))
"
}
for rep in $(seq ${REPS:-2}); do for cfg in "$@"; do run "$cfg"; done; done
