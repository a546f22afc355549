python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/prof_bench.sh r01 2>&1 | tail -30
