#!/bin/bash
# k_render_depth: arbitrary -D flag sets A/B on one box.  usage: tools/exp_render_flags.sh "<flags>" "<flags>" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rdbg
for f in "$@"; do
  AVSIM_EXTRA_FLAGS="$f" python -m av_aloha_amd.build --force > /dev/null 2>&1 || echo "BUILD FAILED" >> gpurun_out/rdbg/out.txt
  echo "== flags '$f'" >> gpurun_out/rdbg/out.txt
  rm -rf gpurun_out/rdbg/prof
  rocprofv3 --kernel-trace --stats -d gpurun_out/rdbg/prof -o p -f csv -- python tools/prof_render.py 4096 480 640 2>/dev/null | grep "N=" >> gpurun_out/rdbg/out.txt
  python - >> gpurun_out/rdbg/out.txt <<PY
import csv,glob
for f in glob.glob("gpurun_out/rdbg/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render_" in r["Name"] and "<true>" not in r["Name"]:
            print("   ", r["Name"][:40], r["Calls"], "avg ms %.3f" % (float(r["AverageNs"]) / 1e6))
PY
  python -m pytest tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -1 >> gpurun_out/rdbg/out.txt
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
cat gpurun_out/rdbg/out.txt
