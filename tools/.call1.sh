cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/s4_call1; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $o/gputest.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench20.json 2> $o/bench20.err
timeout 300 python tools/prof_render.py 4096 > $o/render.txt 2>&1
AVSIM_RENDER_STATS=1 python -m av_aloha_amd.build --force > /dev/null 2>&1
timeout 300 python tools/prof_render.py 1024 > $o/render_stats.txt 2>&1
python -m av_aloha_amd.build --force > /dev/null 2>&1
cat $o/gputest.txt; head -c 600 $o/bench20.json; echo; cat $o/render.txt $o/render_stats.txt
