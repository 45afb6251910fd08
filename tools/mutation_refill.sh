#!/bin/bash
# Mutation check of tests/test_gpu_fullsize.py::test_long_episodes_of_the_plain_pipeline (GPU box): a copy of the tree whose engine refills
# the claim words every 64 steps instead of every 63 (Env::scratch_for) must FAIL the 200-step episode against the compiled reference's digests;
# the unmodified tree passes it.  Writes gpurun_out/mutation_refill.txt (kept as profiles/r06_raw/mutation_refill.txt).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
MUT=/tmp/magent_mut; rm -rf $MUT; mkdir -p $MUT
(cd "$ROOT" && tar cf - --exclude=gpurun_out --exclude=.git --exclude='magent_amd/lib/*.o' . ) | (cd $MUT && tar xf -)
sed -i 's/if (plain_epoch % 63u == 0) claim_epochs = false;/if (plain_epoch % 64u == 0) claim_epochs = false;/' $MUT/magent_amd/csrc/engine*.hip
grep -n 'plain_epoch % 64u' $MUT/magent_amd/csrc/engine*.hip || { echo "mutation not applied" > "$OUT/mutation_refill.txt"; exit 1; }
(cd $MUT && python __graft_entry__.py > $OUT/mutation_build.log 2>&1) || { echo "mutant build failed" > "$OUT/mutation_refill.txt"; exit 1; }
{
  echo "# mutant: claim words refilled every 64 plain steps instead of every 63 (the epoch field still wraps at 63)"
  (cd $MUT && python tools/gpu_golden_check.py battle300_long); echo "mutant exit code: $?"
  echo "# unmodified tree"
  (cd "$ROOT" && python tools/gpu_golden_check.py battle300_long); echo "unmodified exit code: $?"
} > "$OUT/mutation_refill.txt" 2>&1
cat "$OUT/mutation_refill.txt"
