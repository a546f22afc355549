#!/bin/bash
# k_render_depth / k_render_geoms taken apart (round 6): builds on the box with -DAVSIM_RDBG=n and profiles tools/prof_render.py (rocprofv3 --kernel-trace --stats).
#   1: no list at all (pure far-plane stores)   2: list made, nothing staged or cast   3: list + staging, nothing cast   4: no bin masks   0: the product
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rdbg
for v in ${RDBG_LIST:-1 2 3 4 0}; do
  AVSIM_EXTRA_FLAGS="-DAVSIM_RDBG=$v" python -m av_aloha_amd.build --force > /dev/null 2>&1
  echo "== AVSIM_RDBG=$v" >> gpurun_out/rdbg/out.txt
  rm -rf gpurun_out/rdbg/prof
  rocprofv3 --kernel-trace --stats -d gpurun_out/rdbg/prof -o p -f csv -- python tools/prof_render.py 4096 480 640 2>/dev/null | grep "N=" >> gpurun_out/rdbg/out.txt
  python - >> gpurun_out/rdbg/out.txt <<PY
import csv,glob
for f in glob.glob("gpurun_out/rdbg/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render_depth<false>" in r["Name"] or "k_render_geoms" in r["Name"]:
            print("   ", r["Name"][:40], r["Calls"], "avg ms %.3f" % (float(r["AverageNs"]) / 1e6))
PY
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
cat gpurun_out/rdbg/out.txt
