#!/bin/bash
# LDS conflict counters of k_phys (PMC pass only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/lds_pmc
mkdir -p $o
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS -d $o/p -o p -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $o/log 2>&1
python - <<PY
import csv, glob, collections
for f in glob.glob("$o/p/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "k_phys" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc: print(k, acc[k]/max(1,n[k]))
PY
tail -2 $o/log | cut -c1-200
