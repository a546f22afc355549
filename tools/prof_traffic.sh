#!/bin/bash
# HBM-side traffic of k_phys only (FETCH_SIZE / WRITE_SIZE, each in its own pass, no trace): tools/prof_traffic.sh <tag> [configs, default "2"]
tag=${1:-x}; cfgs=${2:-2}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/traffic_$tag; mkdir -p $o
for c in $cfgs; do
  extra=""; [ $c = 3 ] && extra="--config 3 --warmup 150"; [ $c = 4 ] && extra="--config 4 --warmup 30"
  rocprofv3 --pmc FETCH_SIZE -d $o/fetch$c -o p -f csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras $extra > $o/fetch$c.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $o/write$c -o p -f csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras $extra > $o/write$c.log 2>&1
done
python - <<PY
import csv, glob, json
o = "$o"; res = {}
for c in "$cfgs".split():
    for name in ("fetch", "write"):
        for f in glob.glob(o + "/%s%s/**/*counter_collection.csv" % (name, c), recursive=True):
            rows = sorted((int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(f)) if "k_phys" in r["Kernel_Name"])
            top = max([v for _, v in rows] or [0.0])
            vals = [v for _, v in rows if v > 0.1 * top][-5:]
            res["config%s_%s_KB_per_launch" % (c, name)] = sum(vals) / max(1, len(vals))
json.dump(res, open(o + "/pmc_summary.json", "w"), indent=1)
print(res)
PY
