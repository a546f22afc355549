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
# mean over the FULL-LENGTH env-step dispatches only (as tools/prof_flops.sh): a k_phys dispatch is also the forward pass of reset / FK and the
# near-empty second pass of the two-tier capacities, which are short -- averaging them in made round 5's file read 765 M VALU instructions per
# launch where kernel_flops.json has 1 161 M for the same counter
import csv, glob, collections
for d in ("pmc1","pmc2"):
    for f in glob.glob("gpurun_out/$out/%s/**/*counter_collection.csv" % d, recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "k_phys" in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        if not per:
            continue
        ref = max(next(iter(per.values())).keys(), key=lambda k: max(x.get(k, 0.0) for x in per.values()))
        top = max(x.get(ref, 0.0) for x in per.values())
        ids = [i for i in sorted(per) if per[i].get(ref, 0.0) > 0.1 * top][-3:]
        for k in sorted(per[ids[0]]):
            print(d, k, sum(per[i].get(k, 0.0) for i in ids) / len(ids), "per env-step launch, mean over dispatches", ids)
PY
