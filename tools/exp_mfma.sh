#!/bin/bash
# Experiment (GPU box): the Newton Hessian's contact blocks J^T diag(w) J through v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64
# (tools/microbench/newton_mfma_hessian.patch) against the 16-lane-group path of the product: bench, the Hessian probe
# of tools/prof_phases.py, and the parity tests with the variant.  The patch is applied to the box's scratch copy only.
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), 'env-steps/s,', round(d['roofline']['kernel_avg_ms'],3), 'ms per k_phys launch')"; }
hess() { python tools/prof_phases.py 4096 2>/dev/null | grep "inside solve" | sed 's/.*hess \([0-9]*\).*/Hessian assembly \1 cycles per substep/'; }
echo "product (16-lane groups): config 2 $(run); config 3 $(run --config 3 --warmup 150); $(hess)"
patch -s av_aloha_amd/csrc/avsim_newton.hip.h < tools/microbench/newton_mfma_hessian.patch || exit 1
python -m av_aloha_amd.build --force > /dev/null 2>&1
bash tools/kernel_resources.sh | grep -c "k_physIf" > /dev/null
echo "MFMA variant:             config 2 $(run); config 3 $(run --config 3 --warmup 150); $(hess)"
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading av_aloha_amd/libavsim.so > /dev/null 2>&1
for f in av_aloha_amd/libavsim.so.*gfx950; do echo "v_mfma instructions in $(basename $f): $(/opt/rocm/lib/llvm/bin/llvm-objdump -d $f | grep -c v_mfma)"; done; rm -f av_aloha_amd/libavsim.so.*
python -m pytest tests/test_gpu_physics.py tests/test_gpu_configs.py -q 2>&1 | tail -1
