#!/bin/bash
# Round-4 experiment (GPU box): what would a split of the substep at the solver boundary (k_pre at more waves per SIMD, k_solve as
# now) buy?  Measured without writing the split: the per-phase cycles of one wave's substep (profile_phases, s_memtime per phase)
#   (a) at one and at two waves per SIMD (256 x 4 envs with 4 waves per workgroup, 256 x 8 with 8): what a second resident wave costs
#       EACH PHASE -- the phases that barely slow down are the ones further waves would speed up;
#   (b) with the kernel compiled for three and for four waves per SIMD (168 / 128 VGPRs) at the SAME eight envs per CU: what the
#       smaller register budget costs each phase.
# -> profiles/r04_experiments.txt
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4/exp_split; mkdir -p $o
ph() { python tools/prof_phases.py "$@" 2>&1 | grep -v "^$"; }
res() { tools/kernel_resources.sh 2>/dev/null | grep "k_physIfLi64ELi8ELb0" | head -2; }
echo "== base build (waves_per_eu 2) ==" | tee $o/base.txt; res | tee -a $o/base.txt
echo "-- 1 wave per SIMD: 1024 envs, 4 waves per workgroup" | tee -a $o/base.txt; ph 1024 waves_per_block=4 | tee -a $o/base.txt
echo "-- 2 waves per SIMD: 2048 envs, 8 waves per workgroup" | tee -a $o/base.txt; ph 2048 waves_per_block=8 | tee -a $o/base.txt
echo "-- 4096 envs (two rounds of 8 per CU)" | tee -a $o/base.txt; ph 4096 | tee -a $o/base.txt
for w in 3 4; do
  AVSIM_EXTRA_FLAGS="-DAVSIM_PHYS_ATTR=__attribute__((amdgpu_waves_per_eu($w)))" python -m av_aloha_amd.build --force > $o/build$w.log 2>&1
  echo "== build for $w waves per SIMD ==" | tee $o/w$w.txt; res | tee -a $o/w$w.txt
  echo "-- 2048 envs, 8 waves per workgroup (same residency as the base build)" | tee -a $o/w$w.txt; ph 2048 waves_per_block=8 | tee -a $o/w$w.txt
  echo "-- 1024 envs, 4 waves per workgroup" | tee -a $o/w$w.txt; ph 1024 waves_per_block=4 | tee -a $o/w$w.txt
done
python -m av_aloha_amd.build --force > $o/build_restore.log 2>&1
