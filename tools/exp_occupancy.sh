#!/bin/bash
# Experiment (GPU box): one-round launches of 256 x k envs with k waves per workgroup (k = 4..8) -- how the env-step time grows with
# the number of resident envs per CU -- and the same with the kernel built for three waves per SIMD (168 VGPRs).
cd $GRAFT_REPO_ROOT
o=gpurun_out/exp_occ; mkdir -p $o
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_avg_ms'],3))"; }
for k in 4 5 6 7 8; do echo "base k=$k N=$((256*k)): $(run --envs-per-gpu $((256*k)) --option waves_per_block=$k)"; done | tee $o/base.txt
echo "base N=4096: $(run)" | tee -a $o/base.txt
AVSIM_EXTRA_FLAGS="-DAVSIM_PHYS_ATTR=__attribute__((amdgpu_waves_per_eu(3)))" python -m av_aloha_amd.build --force > $o/build3.log 2>&1
for k in 4 6 8; do echo "v168 k=$k N=$((256*k)): $(run --envs-per-gpu $((256*k)) --option waves_per_block=$k)"; done | tee $o/v168.txt
echo "v168 N=4096: $(run)" | tee -a $o/v168.txt
