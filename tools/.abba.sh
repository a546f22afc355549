cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
runo() { (cd ab_old && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old', round(d['value']), d['roofline']['kernel_avg_ms'])"); }
runn() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', round(d['value']), d['roofline']['kernel_avg_ms'])"; }
runn; runo; runo; runn; runn; runo; runo; runn
