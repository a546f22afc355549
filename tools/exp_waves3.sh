#!/bin/bash
# Round 6: the f32 physics kernel compiled for THREE waves per SIMD (168 VGPRs) with up to twelve envs per workgroup, against the product build (two waves, 256 VGPRs, eight envs).
# The LDS record only lets more than eight envs share a CU with a smaller first capacity tier (options maxefc_first / maxcon_first: 96 rows / 24 contacts = 14.6 KB = ten per CU).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/waves3; mkdir -p $o
W3='-DAVSIM_PHYS_MAXW=12 -DAVSIM_PHYS_ATTR=__attribute__((amdgpu_waves_per_eu(3)))'
b() { python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.0f env-steps/s  k_phys %.3f ms  overflow_envs %d' % (d['value'], d['roofline']['kernel_avg_ms'], d['config']['overflow_envs']))"; }
for build in two three; do
  if [ $build = two ]; then AVSIM_EXTRA_FLAGS="" python -m av_aloha_amd.build --force > /dev/null 2>&1; else AVSIM_EXTRA_FLAGS="$W3" python -m av_aloha_amd.build --force > /dev/null 2>&1; fi
  echo "== build: $build waves per SIMD" >> $o/out.txt
  echo "  config 2, one capacity tier (19.0 KB record, 8 envs per CU):" >> $o/out.txt; b >> $o/out.txt; b >> $o/out.txt
  echo "  config 2, first tier 112 rows / 32 contacts (15.6 KB: 9 envs per CU where the build allows):" >> $o/out.txt; b --option maxefc_first=112 --option maxcon_first=32 >> $o/out.txt
  echo "  config 2, first tier 96 rows / 24 contacts (14.6 KB: 10 envs per CU where the build allows):" >> $o/out.txt; b --option maxefc_first=96 --option maxcon_first=24 >> $o/out.txt; b --option maxefc_first=96 --option maxcon_first=24 >> $o/out.txt
  echo "  phases (tools/prof_phases.py 4096 maxefc_first=96 maxcon_first=24):" >> $o/out.txt
  python tools/prof_phases.py 4096 maxefc_first=96 maxcon_first=24 2>/dev/null | grep -E "kin|collide|rows|solve|euler|total|inside solve" >> $o/out.txt
  if [ $build = three ]; then
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $o/pmc -o p -f csv -- python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --option maxefc_first=96 --option maxcon_first=24 > /dev/null 2>&1
    python - >> $o/out.txt <<PY
import csv, glob, collections
for f in glob.glob("$o/pmc/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "k_phys" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    if per:
        top = max(x.get("SQ_WAVE_CYCLES", 0.0) for x in per.values())
        ids = [i for i in sorted(per) if per[i].get("SQ_WAVE_CYCLES", 0.0) > 0.1 * top][-3:]
        c = {k: sum(per[i].get(k, 0.0) for i in ids) / len(ids) for k in per[ids[0]]}
        print("  PMC (10 envs per CU, env-step launches):", {k: round(v) for k, v in c.items()}, " SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f" % (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
  fi
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
cat $o/out.txt
