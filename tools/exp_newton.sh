cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_avg_ms'],3), d['config']['newton_iters_per_substep'], d['config']['newton_iters_max'])"; }
echo "base: $(run)"
echo "newton_iters=1: $(run --option newton_iters=1)"
echo "newton_tol=3e-6: $(run --option newton_tol=3e-6)"
echo "newton_tol=1e-5: $(run --option newton_tol=1e-5)"
echo "newton_tol=1e-7: $(run --option newton_tol=1e-7)"
echo "f64 base: $(run --f64)"
echo "f64 tol 1e-6: $(run --f64 --option newton_tol=1e-6)"
