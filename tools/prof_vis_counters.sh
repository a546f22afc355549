#!/bin/bash
# SQ counters of k_vis_render (PMC passes only, no trace): tools/prof_vis_counters.sh <outdir> [envs] [HxW]
out=$1; n=${2:-256}; hw=${3:-480x640}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d gpurun_out/$out/pmc1 -o p -f csv -- python tools/prof_visual.py $n $hw > gpurun_out/$out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAVES -d gpurun_out/$out/pmc2 -o p -f csv -- python tools/prof_visual.py $n $hw > gpurun_out/$out/pmc2.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_I8 SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM -d gpurun_out/$out/pmc3 -o p -f csv -- python tools/prof_visual.py $n $hw > gpurun_out/$out/pmc3.log 2>&1
# (a fourth pass with TCC_HIT_sum / TCC_MISS_sum / TCC_REQ_sum hung rocprofv3 on this pool until the call's limit: left out)
python - <<PY
import csv, glob, collections
for d in ("pmc1","pmc2","pmc3"):
    for f in glob.glob("gpurun_out/$out/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "k_vis_render" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc: print(d, k, acc[k]/max(1,n[k]), "per launch over", n[k])
    import os
    if not glob.glob("gpurun_out/$out/%s/**/*counter_collection.csv" % d, recursive=True): print(d, "no output:", open("gpurun_out/$out/%s.log" % d).read()[-400:])
PY
