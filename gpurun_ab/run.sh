#!/bin/bash
# same-box A/B of the physics headers: old (67aa622) vs new, ABAB
cd $GRAFT_REPO_ROOT
b() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $@ 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), end=' ')"; }
for rep in 1 2; do
 for v in old new new0; do
  src=$v; flags=""; [ $v = new0 ] && { src=new; flags="-DAVS_OUTLINE_MODE=0"; }
  cp gpurun_ab/$src/*.h av_aloha_amd/csrc/
  AVSIM_EXTRA_FLAGS="$flags" AVSIM_EXTRA_FLAGS_F64="$flags" python -m av_aloha_amd.build --force > /dev/null 2>&1
  echo -n "$v: config2 "; b; b; b; echo -n " f64 "; b --f64; b --f64; echo -n " c3 "; b --config 3 --steps 240; echo -n " c4 "; b --config 4 --steps 100 --warmup 10; echo
 done
done
cp gpurun_ab/new/*.h av_aloha_amd/csrc/
