#!/bin/bash
# Round profile (run on the MI355X box through gpurun): bench lines of the four measurement configurations + the f64 line,
# rocprofv3 kernel-trace stats of configs 2 and 3, PMC traffic counters in their own passes (never combined with a trace).
# usage: tools/prof_bench.sh <tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/prof_$tag
mkdir -p $o
python bench.py --steps 100 --warmup 10 > $o/bench_config2.json 2> $o/bench2.err
python bench.py --config 3 --steps 240 --warmup 5 > $o/bench_config3.json 2> $o/bench3.err
python bench.py --config 4 --steps 100 --warmup 10 > $o/bench_config4.json 2> $o/bench4.err
python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $o/bench_config5.json 2> $o/bench5.err
python bench.py --f64 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $o/bench_config2_f64.json 2> $o/bench2f64.err
rocprofv3 --kernel-trace --stats -d $o/trace2 -o t -f csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $o/trace2.log 2>&1
rocprofv3 --kernel-trace --stats -d $o/trace3 -o t -f csv -- python bench.py --config 3 --steps 60 --warmup 150 --no-cpu-baseline --no-extras > $o/trace3.log 2>&1
rocprofv3 --kernel-trace --stats -d $o/trace4 -o t -f csv -- python bench.py --config 4 --steps 20 --warmup 30 --no-cpu-baseline --no-extras > $o/trace4.log 2>&1
rocprofv3 --kernel-trace --stats -d $o/trace5 -o t -f csv -- python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $o/trace5.log 2>&1
for c in 2 3; do
  extra=""; [ $c = 3 ] && extra="--config 3 --warmup 150"
  rocprofv3 --pmc FETCH_SIZE -d $o/fetch$c -o p -f csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras $extra > $o/fetch$c.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $o/write$c -o p -f csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras $extra > $o/write$c.log 2>&1
done
python - <<PY
import csv, glob, json
o = "$o"
for c in (2, 3, 4, 5):
    for f in glob.glob(o + "/trace%d/**/*kernel_stats.csv" % c, recursive=True):
        open(o + "/kernel_stats_config%d.csv" % c, "w").write(open(f).read())
res = {}
for c in (2, 3):
    for name in ("fetch", "write"):
        for f in glob.glob(o + "/%s%d/**/*counter_collection.csv" % (name, c), recursive=True):
            rows = sorted((int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(f)) if "k_phys" in r["Kernel_Name"])
            top = max([v for _, v in rows] or [0.0])
            vals = [v for _, v in rows if v > 0.1 * top][-5:]          # the timed env-step launches (forward-only launches and the near-empty second passes of the two-tier capacities are small)
            res["config%d_%s_KB_per_launch" % (c, name)] = sum(vals) / max(1, len(vals))
json.dump(res, open(o + "/pmc_summary.json", "w"), indent=1)
print(res)
PY
head -c 600 $o/bench_config2.json; echo; for c in 2 3 4 5; do head -8 $o/kernel_stats_config$c.csv; done
