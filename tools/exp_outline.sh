#!/bin/bash
# bench + traffic of the three AVS_OUTLINE modes (avsim_math.hip.h), built on the box
cd $GRAFT_REPO_ROOT
for m in 0 1 2; do
  AVSIM_EXTRA_FLAGS=-DAVS_OUTLINE_MODE=$m python -m av_aloha_amd.build --force > /dev/null 2>&1
  echo "== mode $m"
  for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline'].get('kernel_ms'))"; done
  python bench.py --config 3 --steps 240 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('config3', d['value'])"
  python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('config4', d['value'])"
  bash tools/prof_traffic.sh m$m "2" 2>&1 | tail -1
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
