#!/bin/bash
# Experiment: the product kernel compiled without its profiling probes (-DAVSIM_NO_PROF) against the default build
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_avg_ms'],3))"; }
echo "default: $(run)  config3: $(run --config 3 --warmup 150)  config4: $(run --config 4)"
AVSIM_EXTRA_FLAGS="$1" python -m av_aloha_amd.build --force > /dev/null 2>&1
echo "$1: $(run)  config3: $(run --config 3 --warmup 150)  config4: $(run --config 4)"
