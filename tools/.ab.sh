cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_physics.py tests/test_gpu_bench_path.py tests/test_gpu_configs.py -q -x 2>&1 | tail -5
for i in 1 2; do
 (cd ab_old && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old', round(d['value']), d['ms_per_step'], d['roofline']['kernel_avg_ms'])")
 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', round(d['value']), d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
done
