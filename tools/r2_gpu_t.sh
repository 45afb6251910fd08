#!/bin/bash
mkdir -p gpurun_out/r2t
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_[A-Z_0-9]*\|SQ_WAIT[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_ICACHE[A-Z_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/r2t/counters.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r2t/p1 -o t -- python $R/tools/many_envs_batch.py 1 1 > $R/gpurun_out/r2t/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAVES SQ_IFETCH SQC_ICACHE_MISSES SQC_ICACHE_REQ --output-format csv -d $R/gpurun_out/r2t/p2 -o t -- python $R/tools/many_envs_batch.py 1 1 > $R/gpurun_out/r2t/p2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for p in ("p1","p2"):
    fs = glob.glob("gpurun_out/r2t/%s/**/*counter_collection.csv" % p, recursive=True)
    if not fs: print(p, "no csv"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_step_solo" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(p, {k: round(sum(v)/len(v)) for k, v in agg.items()}, "launches", {k: len(v) for k,v in agg.items()})
PY
cat gpurun_out/r2t/counters.txt | head -c 1500
tail -2 gpurun_out/r2t/p2.log
