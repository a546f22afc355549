cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/s4_call3; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_configs.py -q -s -k "tridiag" 2>&1 | grep -E "secular|Mismatch|Max abs|Max rel|passed|failed" > $o/tridiag.txt
cat $o/tridiag.txt
