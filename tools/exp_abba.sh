#!/bin/bash
# A/B of the headline configuration between the working tree and a build of another revision, in ABBA order inside ONE gpurun call (the
# boxes differ by +-0.7 %, and the first run of a pair is not the faster one by order alone): env-steps/s and ms per k_phys launch.
#   rm -rf ab_old; mkdir ab_old; git archive HEAD av_aloha_amd bench.py models include | tar -x -C ab_old; (cd ab_old; python -m av_aloha_amd.build)
#   gpurun -- 'bash tools/exp_abba.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
runo() { (cd ab_old && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old', round(d['value']), d['roofline']['kernel_avg_ms'])"); }
runn() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', round(d['value']), d['roofline']['kernel_avg_ms'])"; }
runn; runo; runo; runn; runn; runo; runo; runn
