#!/bin/bash
# Same-box comparison of builds that differ in -D flags (built on the box): tools/exp_flags_ab.sh "<flags A>" "<flags B>" ...   ("" = the default build);
# two rounds over the list; config 2 x 3, f64 x 2, configs 3 and 4
cd $GRAFT_REPO_ROOT
b() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $@ 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), end=' ')"; }
for rep in 1 2; do
 for f in "$@"; do
  AVSIM_EXTRA_FLAGS="$f" AVSIM_EXTRA_FLAGS_F64="$f" python -m av_aloha_amd.build --force > /dev/null 2>&1
  echo -n "[$f]: config2 "; b; b; b; echo -n " f64 "; b --f64; b --f64; echo -n " c3 "; b --config 3 --steps 240; echo -n " c4 "; b --config 4 --steps 100 --warmup 10; echo
 done
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
