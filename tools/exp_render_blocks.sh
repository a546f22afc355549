#!/bin/bash
# k_render_depth: bin shapes (tiles per block) A/B on one box.  LIST = "BIN_TX,BIN_TY ..."
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rdbg
for v in ${LIST:-10,2 20,2 20,4 10,4}; do
  AVSIM_EXTRA_FLAGS="-DAVSIM_BIN_TX=${v%,*} -DAVSIM_BIN_TY=${v#*,} $EXTRA" python -m av_aloha_amd.build --force > /dev/null 2>&1
  echo "== AVSIM_BIN_TX,TY=$v $EXTRA" >> gpurun_out/rdbg/out.txt
  rm -rf gpurun_out/rdbg/prof
  rocprofv3 --kernel-trace --stats -d gpurun_out/rdbg/prof -o p -f csv -- python tools/prof_render.py 4096 480 640 2>/dev/null | grep "N=" >> gpurun_out/rdbg/out.txt
  python - >> gpurun_out/rdbg/out.txt <<PY
import csv,glob
for f in glob.glob("gpurun_out/rdbg/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render_" in r["Name"] and "<true>" not in r["Name"]:
            print("   ", r["Name"][:40], r["Calls"], "avg ms %.3f" % (float(r["AverageNs"]) / 1e6))
PY
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
python -m pytest tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -2 >> gpurun_out/rdbg/out.txt
cat gpurun_out/rdbg/out.txt
