#!/bin/bash
# Experiment (GPU box): same waves per CU (8), fewer CUs busy -> smaller solver-scratch footprint per XCD's L2.  Is the env-step
# time sensitive to it?  Plus the L2 hit / miss counters of the full launch.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/exp_l2; mkdir -p $o
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_avg_ms'],3))"; }
for n in 256 512 1024 2048; do echo "wpb=8 N=$n: $(run --envs-per-gpu $n --option waves_per_block=8)"; done | tee $o/footprint.txt
for n in 256 512 1024; do echo "wpb=4 N=$n: $(run --envs-per-gpu $n --option waves_per_block=4)"; done | tee -a $o/footprint.txt
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c -d $o/$tag -o p -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $o/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$o/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_phys" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = sorted(v)[len(v)//2:]
        print(k, "per env-step launch: %.4g" % (sum(v) / len(v)))
PY
