# Development helper (GPU box, via gpurun): stall / occupancy / cache counters of the step kernels, one rocprofv3 --pmc pass per
# counter group (never combined with other trace domains), over a short bench run.  Output: gpurun_out/$1/<group>/ ; summarise with
# tools/step_pmc_summary.py into profiles/<round>_step_pmc.json.   usage: bash tools/step_pmc.sh <tag> [extra bench args]
R=$GRAFT_REPO_ROOT; TAG=${1:-steppmc}; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-profile --no-extras --preheat-ms 0 $*"
run() { name=$1; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -o bench -- $B > $O/$name.log 2>&1; echo "$name rc=$?"; }
run sq     SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM
run tcc    TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcc2   TCC_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE
run ta     TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
run fetch  FETCH_SIZE
run write  WRITE_SIZE
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B > $O/stats.log 2>&1; echo "stats rc=$?"
ls $O/*/ | head -40
