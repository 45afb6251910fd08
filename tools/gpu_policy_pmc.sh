# Development helper (GPU box, via gpurun): PMC passes over the policy kernels (SQ, TCC counters).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/policypmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|SQ_WAIT[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" | sort -u | tr '\n' ' ' > $O/avail.txt
wc -c $O/avail.txt
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/p3 -o pr -- python $R/tools/policy_rate.py 131072 3 > $O/p3.log 2>&1
tail -2 $O/p3.log; ls $O/p3
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $O/p4 -o pr -- python $R/tools/policy_rate.py 131072 3 > $O/p4.log 2>&1
ls $O/p4
