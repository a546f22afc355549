#!/bin/bash
cd $GRAFT_REPO_ROOT
b() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $@ 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), end=' ')"; }
for rep in 1 2; do
 for m in 6 22 38 14; do
  AVSIM_EXTRA_FLAGS="-DAVS_NTC_MASK=$m" AVSIM_EXTRA_FLAGS_F64="-DAVS_NTC_MASK=$m" python -m av_aloha_amd.build --force > /dev/null 2>&1
  echo -n "mask $m: config2 "; b; b; b; echo -n " f64 "; b --f64; b --f64; echo -n " c3 "; b --config 3 --steps 240; b --config 3 --steps 240; echo -n " c4 "; b --config 4 --steps 100 --warmup 10; b --config 4 --steps 100 --warmup 10; echo
 done
done
AVSIM_EXTRA_FLAGS="-DAVS_NTC_MASK=6" AVSIM_EXTRA_FLAGS_F64="-DAVS_NTC_MASK=6" python -m av_aloha_amd.build --force > /dev/null 2>&1
bash tools/prof_traffic.sh mask6 "2 3" 2>&1 | tail -1
python -m av_aloha_amd.build --force > /dev/null 2>&1
