#!/bin/bash
# SQ counters of the render kernels (PMC passes only, no trace): usage tools/prof_render_counters.sh <outdir> [N]
out=$1; N=${2:-1024}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -d gpurun_out/$out/pmc1 -o p -f csv -- python tools/prof_render.py $N > gpurun_out/$out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAVES SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC -d gpurun_out/$out/pmc2 -o p -f csv -- python tools/prof_render.py $N > gpurun_out/$out/pmc2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("pmc1","pmc2"):
    for f in glob.glob("gpurun_out/$out/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "k_render_depth<false>" in r["Kernel_Name"] or "k_render_depthILb0" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc: print(d, k, acc[k]/max(1,n[k]), "per launch over", n[k])
PY
