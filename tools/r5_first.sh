# round 5, first call: the literal callers on the HIP engine; config 5's world measured like C3; counters of the large-map renders
R=$GRAFT_REPO_ROOT
bash $R/tools/measure.sh r5a callers
bash $R/tools/measure.sh r5a_c5 line,stats,bytes -- --workload battle_c5 --map-size 3536 --repeats 3
bash $R/tools/measure.sh r5a_c5m line,stats -- --workload battle_c5_melee --map-size 3536 --repeats 3
bash $R/tools/measure.sh r5a_1m line,bytes -- --workload test_1m --agents 500000 --repeats 3 --steps 10 --warmup 3
bash $R/tools/measure.sh r5a_c3 line
