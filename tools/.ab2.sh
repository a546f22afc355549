cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do for c in "--steps 20 --warmup 5" "--steps 40 --warmup 55"; do
 timeout 300 python bench.py $c --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new $c', round(d['value']), d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
done; done
