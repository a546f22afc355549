#!/bin/bash
# k_vis_render variants built on the box: tools/exp_vis.sh "<flags A>" "<flags B>" ...   (each a set of -D flags; "" = the default build)
cd $GRAFT_REPO_ROOT
for f in "$@"; do
  AVSIM_EXTRA_FLAGS="$f" python -m av_aloha_amd.build --force > /dev/null 2>&1
  echo "== flags: $f"
  python tools/prof_visual.py 1024 480x640 2>/dev/null | grep -v "^scene" | tail -6
  SHADOWS=1 SAMPLES=4 python tools/prof_visual.py 1024 480x640 2>/dev/null | sed -n 3p
  python tools/prof_visual.py 4096 120x160 2>/dev/null | sed -n 3p
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
