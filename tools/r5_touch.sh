R=$GRAFT_REPO_ROOT; cd $R
for t in touch_map=0 touch_map=1 ""; do echo "== MAGENT_TUNE=$t"; MAGENT_TUNE=$t bash tools/measure.sh r5i_$t line -- --workload test_1m --agents 500000 --repeats 3 --steps 10 --warmup 3 | grep -v "^==\|breakdown\|host-fin"; done
K="test_1m" bash tools/measure.sh r5i parity
bash tools/measure.sh r5i_1m stats -- --workload test_1m --agents 500000 --steps 10 --warmup 3 | head -8
