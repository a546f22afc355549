#!/bin/bash
# SQ counters of k_render_depth<false> (PMC-only passes): usage tools/prof_render_counters.sh <outdir>
out=${1:-rpmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d gpurun_out/$out/p1 -o p -f csv -- python tools/prof_render.py 4096 480 640 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM -d gpurun_out/$out/p2 -o p -f csv -- python tools/prof_render.py 4096 480 640 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for d in ("p1", "p2"):
    for f in glob.glob("gpurun_out/$out/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        disp = set()
        for r in csv.DictReader(open(f)):
            if "k_render_depth" in r["Kernel_Name"] and "Lb0" in r["Kernel_Name"] or ("k_render_depth<false>" in r["Kernel_Name"]):
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
        for k in sorted(acc): print(d, k, "%.4g per launch" % (acc[k] / max(1, len(disp))), "(%d launches)" % len(disp))
PY
