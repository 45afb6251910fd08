R=$GRAFT_REPO_ROOT; cd $R
bash tools/measure.sh r5c_c3 stats > /dev/null; python tools/step_timeline.py gpurun_out/r5c_c3/stats/bench_kernel_trace.csv 3
bash tools/measure.sh r5c_c4 line,stats -- --workload gather --map-size 500 --agents 100000 --repeats 3 | tail -30; python tools/step_timeline.py gpurun_out/r5c_c4/stats/bench_kernel_trace.csv 3
bash tools/measure.sh r5c_c2 line,stats -- --map-size 200 --agents 2000 --steps 200 --warmup 20 --repeats 3 | tail -24; python tools/step_timeline.py gpurun_out/r5c_c2/stats/bench_kernel_trace.csv 3
