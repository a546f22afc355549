#!/bin/bash
# Round profile of the bench command: kernel-trace stats in one run, PMC counters in their own runs (never combined).
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/prof_$tag
mkdir -p $o
python bench.py --steps 100 --warmup 10 > $o/bench.json 2> $o/bench.err
rocprofv3 --kernel-trace --stats -d $o/trace -o t -f csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $o/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $o/fetch -o p -f csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $o/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $o/write -o p -f csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $o/write.log 2>&1
python - <<PY
import csv, glob, json, collections
o = "$o"
for f in glob.glob(o + "/trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read())
res = {}
for name in ("fetch", "write"):
    for f in glob.glob(o + "/%s/**/*counter_collection.csv" % name, recursive=True):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_phys" in r["Kernel_Name"]]
        vals = sorted(vals)[len(vals)//2:]          # the env-step launches (forward-only launches are the small half)
        res[name] = sum(vals) / max(1, len(vals))
print("PMC per k_phys launch (KB units as reported):", res)
json.dump(res, open(o + "/pmc_summary.json", "w"))
PY
cat $o/bench.json
