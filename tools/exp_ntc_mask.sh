#!/bin/bash
# not_tail_called per out-of-line function (avsim_math.hip.h AVS_NTC_MASK): builds on the box, same-box comparison
cd $GRAFT_REPO_ROOT
b() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $@ 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), end=' ')"; }
for rep in 1 2; do
 for m in 0 1 2 4 8 6 63; do
  AVSIM_EXTRA_FLAGS="-DAVS_NTC_MASK=$m" AVSIM_EXTRA_FLAGS_F64="-DAVS_NTC_MASK=$m" python -m av_aloha_amd.build --force > /dev/null 2>&1
  echo -n "mask $m: config2 "; b; b; b; echo -n " f64 "; b --f64; b --f64; echo -n " c3 "; b --config 3 --steps 240; echo -n " c4 "; b --config 4 --steps 100 --warmup 10; echo
 done
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
