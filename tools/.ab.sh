cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_physics.py tests/test_gpu_bench_path.py tests/test_gpu_configs.py tests/test_gpu_episode_parity.py tests/test_gpu_scripted.py -q -x 2>&1 | tail -4
for c in "--steps 20 --warmup 5" "--config 3 --steps 240 --warmup 5" "--config 4 --steps 100 --warmup 10"; do
 (cd ab_old && timeout 300 python bench.py $c --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old $c', round(d['value']), d['ms_per_step'])")
 timeout 300 python bench.py $c --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new $c', round(d['value']), d['ms_per_step'])"
done
