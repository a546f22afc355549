cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/s4_call2; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_configs.py -q -s -k "tridiag" 2>&1 | tail -8 > $o/tridiag.txt
for v in 1 2; do
 timeout 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline --no-extras --option qcqp_tridiag=$v > $o/bench4_q$v.json 2>$o/err
 { echo "== qcqp_tridiag=$v"; TASK=hook_package ARMS=2 timeout 300 python tools/prof_phases.py 4096 qcqp_tridiag=$v 2>/dev/null; } > $o/phases4_q$v.txt
done
cat $o/tridiag.txt; for v in 1 2; do head -c 330 $o/bench4_q$v.json; echo; cat $o/phases4_q$v.txt | grep -E "percentiles|probe|slowest"; done
