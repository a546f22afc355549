#!/bin/bash
# timing of the depth rasteriser under different tile / bin shapes (build flags); run on the GPU box
for v in "$@"; do
  AVSIM_EXTRA_FLAGS="$v" python -m av_aloha_amd.build --force > /dev/null 2>&1
  echo "== $v"; python tools/prof_render.py 1024 2>&1 | tail -2 | head -1
done
python -m av_aloha_amd.build --force > /dev/null 2>&1
