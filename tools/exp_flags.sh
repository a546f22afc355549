#!/bin/bash
# Experiment: bench config 2 / 3 / 4 with extra compile flags for the f32 translation unit: tools/exp_flags.sh "<flags>"
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_avg_ms'],3))"; }
echo "default: config2 $(run) | $(run)  config3 $(run --config 3 --warmup 150)  config4 $(run --config 4)"
AVSIM_EXTRA_FLAGS="$1" python -m av_aloha_amd.build --force > /dev/null 2>&1
echo "$1: config2 $(run) | $(run)  config3 $(run --config 3 --warmup 150)  config4 $(run --config 4)"
if [ -n "$2" ]; then python -m pytest tests -m gpu -q -x 2>&1 | tail -2; fi
