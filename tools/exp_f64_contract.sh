#!/bin/bash
# f64 physics with FMA contraction (the parity build is -ffp-contract=off): throughput and what the f64 parity tests say
cd $GRAFT_REPO_ROOT
o=gpurun_out/f64c; mkdir -p $o
for f in "" "-ffp-contract=fast"; do
  AVSIM_EXTRA_FLAGS_F64="$f" python -m av_aloha_amd.build --force > /dev/null 2>&1
  echo "== f64 flags '$f'" >> $o/out.txt
  for r in 1 2; do python bench.py --f64 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   f64 config 2: %.0f env-steps/s  k_phys %.3f ms' % (d['value'], d['roofline']['kernel_avg_ms']))" >> $o/out.txt; done
  if [ -n "$f" ]; then
    python -m pytest tests/test_gpu_physics.py tests/test_gpu_boxbox.py tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -6 >> $o/out.txt
    python -m pytest tests/test_gpu_episode_parity.py -m gpu -q -k "f64" 2>&1 | tail -8 >> $o/out.txt
  fi
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
cat $o/out.txt
