#!/bin/bash
# HBM-side traffic of k_phys per launch: FETCH_SIZE and WRITE_SIZE in separate PMC passes (usage: tools/prof_traffic.sh <tag>)
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/traffic_$tag
mkdir -p $o
rocprofv3 --pmc FETCH_SIZE -d $o/fetch -o p -f csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $o/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $o/write -o p -f csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $o/write.log 2>&1
python - <<PY
import csv, glob
res = {}
for name in ("fetch", "write"):
    for f in glob.glob("$o/%s/**/*counter_collection.csv" % name, recursive=True):
        vals = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_phys" in r["Kernel_Name"])
        vals = [v for v in vals if v > 0.5 * vals[-1]] if vals else vals      # the env-step launches
        res[name] = sum(vals) / max(1, len(vals))
print("$tag", "KB per k_phys launch:", res, "GB total:", (res.get("fetch",0)+res.get("write",0))*1024/1e9)
PY
tail -1 $o/write.log | cut -c1-160
