#!/bin/bash
# Experiment: config 2 (the driver's flags) under a list of extra compile flags for the f32 translation unit, each next to the
# default build in the same call: tools/exp_flags_multi.sh "<flags 1>" "<flags 2>" ...
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_avg_ms'],3))"; }
echo "default: $(run) | $(run) | $(run)"
for f in "$@"; do
  if AVSIM_EXTRA_FLAGS="$f" python -m av_aloha_amd.build --force > /tmp/build.log 2>&1; then
    echo "$f: $(run) | $(run)   $(tools/kernel_resources.sh | grep 'k_physIfLi64ELi8ELb0' | grep -o 'private_segment_fixed_size: [0-9]*.*')"
  else echo "$f: build failed: $(grep -m1 error /tmp/build.log)"; fi
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
echo "default again: $(run) | $(run)"
