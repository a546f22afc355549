#!/bin/bash
# A/B of build flags on ONE box: usage  tools/exp_ab.sh "<flags A>" "<flags B>" ... ; each build runs bench.py for configs 2 (driver flags) and 4
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for f in "$@"; do
  AVSIM_EXTRA_FLAGS="$f" python -m av_aloha_amd.build --force > /dev/null 2>&1 || echo "BUILD FAILED: $f" >> gpurun_out/ab/out.txt
  echo "== flags: '$f'" >> gpurun_out/ab/out.txt
  for rep in 1 2; do
    python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   config 2: %.0f env-steps/s  k_phys %.3f ms' % (d['value'], d['roofline']['kernel_avg_ms']))" >> gpurun_out/ab/out.txt
  done
  for c in ${CONFIGS:-4}; do
    python bench.py --config $c --steps 100 --warmup ${WARM:-10} --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   config $c: %.0f env-steps/s  k_phys %.3f ms' % (d['value'], d['roofline']['kernel_avg_ms']))" >> gpurun_out/ab/out.txt
  done
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
cat gpurun_out/ab/out.txt
