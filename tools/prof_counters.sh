#!/bin/bash
# usage: tools/prof_counters.sh <outdir> [bench args]; PMC passes kept separate from --kernel-trace/--stats runs
out=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$out
A="--steps 3 --warmup 1 --no-cpu-baseline --no-extras $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d gpurun_out/$out/pmc1 -o p -f csv -- python bench.py $A > gpurun_out/$out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH -d gpurun_out/$out/pmc2 -o p -f csv -- python bench.py $A > gpurun_out/$out/pmc2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("pmc1","pmc2"):
    for f in glob.glob("gpurun_out/$out/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "k_phys" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc: print(d, k, acc[k]/max(1,n[k]), "per launch over", n[k])
PY
