# the numbers profiles/r04_* and DESIGN.md section 6 quote, on the final tree of the round (a shorter tools/final_measure.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04last; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>$O/bench_default.err; tail -1 $O/bench_default.log | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
python $R/bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --repeats 1 2>/dev/null | tail -1 > $O/two_ranks_gloo.log
bash $R/tools/step_pmc.sh r04last_pmc > $O/pmc.log 2>&1; tail -1 $O/pmc.log
