#!/bin/bash
# tools/exp_abba.sh for another bench configuration: CFG (default 3), STEPS (default 60); prints env-steps/s, ms per step and ms per k_phys launch.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CFG=${CFG:-3}; STEPS=${STEPS:-60}
one() { timeout 300 python bench.py --config $CFG --steps $STEPS --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'], 3), d['roofline']['kernel_avg_ms'])"; }
runo() { (cd ab_old && one old); }
runn() { one new; }
runn; runo; runo; runn
