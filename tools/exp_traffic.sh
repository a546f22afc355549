#!/bin/bash
# bench config 2 (20 steps) + FETCH_SIZE / WRITE_SIZE of k_phys per env-step launch (two PMC passes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/exp_traffic; mkdir -p $o
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 2:', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_avg_ms'],3))"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 2 (100):', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_avg_ms'],3))"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $o/$c -o p -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $o/$c.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$o/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_phys" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = sorted(v)[len(v)//2:]
        print(k, "KB per env-step launch: %.4g" % (sum(v) / len(v)))
PY
