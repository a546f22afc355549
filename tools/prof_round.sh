#!/bin/bash
# Everything profiles/ holds for a round, in one call on the GPU box: tools/prof_round.sh <tag>
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/round_$tag; mkdir -p $o
[ -z "$SKIP_GPUTEST" ] && python -m pytest tests -m gpu -q 2>&1 | tail -4 > $o/gputest.txt
bash tools/prof_bench.sh $tag > $o/prof_bench.log 2>&1
# (the per-phase profile is taken on the build that keeps kinematics .. smooth apart: -DAVS_NO_SPLIT_PRE; the default build runs them as one function)
AVSIM_EXTRA_FLAGS=-DAVS_NO_SPLIT_PRE python -m av_aloha_amd.build --force > /dev/null 2>&1
{ echo "(build: -DAVS_NO_SPLIT_PRE)"; echo "== config 2: SlotInsertion-3Arms resting scene, 4096 envs (tools/prof_phases.py 4096) =="; python tools/prof_phases.py 4096 2>/dev/null
  echo; echo "== config 3 model: SewNeedle-3Arms, random-walk actions, 4096 envs (TASK=sew_needle ARMS=3) =="; TASK=sew_needle ARMS=3 python tools/prof_phases.py 4096 2>/dev/null
  echo; echo "== config 4: HookPackage-2Arms random walk, 4096 envs (TASK=hook_package ARMS=2) =="; TASK=hook_package ARMS=2 python tools/prof_phases.py 4096 2>/dev/null; } > $o/phases.txt
python -m av_aloha_amd.build --force > /dev/null 2>&1
bash tools/prof_counters.sh round_$tag/counters > $o/sq_counters.txt 2>&1
bash tools/prof_flops.sh $tag > $o/prof_flops.log 2>&1; cp gpurun_out/flops_$tag/kernel_flops.json $o/ 2>/dev/null
{ python tools/prof_visual.py 1024 480x640; python tools/prof_visual.py 4096 120x160; } > $o/visual.txt 2>&1
rocprofv3 --kernel-trace --stats -d $o/trace_vis -o t -f csv -- python tools/prof_visual.py 256 480x640 > $o/trace_vis.log 2>&1
cp $(find $o/trace_vis -name "*kernel_stats.csv" | head -1) $o/kernel_stats_visual.csv 2>/dev/null
python tools/prof_rerender.py 4 > $o/rerender.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_driver_style.json 2> $o/bench_driver_style.err
tail -3 $o/gputest.txt 2>/dev/null; tail -8 $o/rerender.txt; head -c 400 $o/bench_driver_style.json; echo; tail -5 $o/visual.txt
