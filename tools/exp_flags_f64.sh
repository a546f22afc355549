#!/bin/bash
# Experiment: extra compile flags for the f64 translation unit (AVSIM_EXTRA_FLAGS_F64): tools/exp_flags_f64.sh "<flags 1>" ...
cd $GRAFT_REPO_ROOT
run() { python bench.py --f64 --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_avg_ms'],3))"; }
echo "default: $(run) | $(run)"
for f in "$@"; do
  if AVSIM_EXTRA_FLAGS_F64="$f" python -m av_aloha_amd.build --force > /tmp/build.log 2>&1; then
    echo "$f: $(run) | $(run)   $(tools/kernel_resources.sh | grep 'k_physIdLi64ELi4ELb0' | grep -o 'private_segment_fixed_size: [0-9]*.*')"
  else echo "$f: build failed: $(grep -m1 error /tmp/build.log)"; fi
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
