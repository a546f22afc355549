#!/bin/bash
# (continuation of tools/exp_waves3.sh) the same two builds at 16384 envs, where the number of rounds of resident waves no longer quantises the result
cd $GRAFT_REPO_ROOT
o=gpurun_out/waves3; mkdir -p $o
W3='-DAVSIM_PHYS_MAXW=12 -DAVSIM_PHYS_ATTR=__attribute__((amdgpu_waves_per_eu(3)))'
b() { python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.0f env-steps/s  k_phys %.3f ms  overflow_envs %d' % (d['value'], d['roofline']['kernel_avg_ms'], d['config']['overflow_envs']))"; }
for build in two three; do
  if [ $build = two ]; then AVSIM_EXTRA_FLAGS="" python -m av_aloha_amd.build --force > /dev/null 2>&1; else AVSIM_EXTRA_FLAGS="$W3" python -m av_aloha_amd.build --force > /dev/null 2>&1; fi
  echo "== build: $build waves per SIMD, 16384 envs" >> $o/out2.txt
  echo "  one tier:" >> $o/out2.txt; b --envs-per-gpu 16384 >> $o/out2.txt
  echo "  first tier 96 / 24:" >> $o/out2.txt; b --envs-per-gpu 16384 --option maxefc_first=96 --option maxcon_first=24 >> $o/out2.txt
  echo "  first tier 96 / 24, 3072 envs (one round of twelve per CU would hold them; ten per CU: 1.2 rounds):" >> $o/out2.txt; b --envs-per-gpu 2560 --option maxefc_first=96 --option maxcon_first=24 >> $o/out2.txt
  echo "  one tier, 2560 envs:" >> $o/out2.txt; b --envs-per-gpu 2560 >> $o/out2.txt
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
cat $o/out2.txt
