#!/bin/bash
# Soak / size checks on one MI355X (rounds 4, 5): ten whole episodes of config 2 with their auto-resets, config 4 at the 8-GPU job's TOTAL env count
# on one GPU (32768 envs), config 3 over four scripted episodes, the depth images at 16384 views -> profiles/r0N_soak.txt
cd $GRAFT_REPO_ROOT
o=gpurun_out/${SOAK_TAG:-r5}; mkdir -p $o
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'envs', c['envs_total'], 'steps', d['steps'], 'resets', c['resets_in_timed_region'], 'nan_envs', c['nan_envs'], 'overflow_envs', c['overflow_envs'], 'mean_return', round(c['mean_return'],3), 'success_rate', c['success_rate'])"; }
{
echo "config 2, 3000 steps (ten 300-step episodes, auto-reset): $(python bench.py --steps 3000 --warmup 0 --no-extras --no-cpu-baseline 2>/dev/null | pick)"
echo "config 4, 32768 envs on one GPU, 300 steps: $(python bench.py --config 4 --envs-per-gpu 32768 --steps 300 --warmup 0 --no-cpu-baseline 2>/dev/null | pick)"
echo "config 3, 1000 steps (four scripted episodes): $(python bench.py --config 3 --steps 1000 --warmup 0 --no-cpu-baseline 2>/dev/null | pick)"
echo "config 2, 16384 envs, 100 steps: $(python bench.py --envs-per-gpu 16384 --steps 100 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | pick)"
echo "config 2, 256 envs, 100 steps: $(python bench.py --envs-per-gpu 256 --steps 100 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | pick)"
echo "config 2, 1024 envs (BASELINE configs[1] as written), 100 steps: $(python bench.py --envs-per-gpu 1024 --steps 100 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | pick)"
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -2
} | tee $o/soak.txt
