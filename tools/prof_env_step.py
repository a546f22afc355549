"""Wall time of the gym facade's step() with pixel observations (the reference's own use: env.py:180-188), per env-step:
    python tools/prof_env_step.py [num_envs] [steps]
run under `rocprofv3 --kernel-trace --stats` for the kernels behind it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from av_aloha_amd.env import make
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
env = make("gym_guided_vision/SlotInsertion-3Arms-v0", num_envs=N)
obs, info = env.reset(seed=0)
a = np.tile(obs["agent_pos"][:1], (N, 1)).astype(np.float32) if obs["agent_pos"].ndim == 2 else obs["agent_pos"]
for _ in range(2):
    obs, r, term, trunc, info = env.step(a)
t = time.time()
for _ in range(K):
    obs, r, term, trunc, info = env.step(a)
dt = (time.time() - t) / K
cams = list(obs["pixels"].keys())
px = sum(v.nbytes for v in obs["pixels"].values())
print(f"{N} envs, cameras {cams}: {dt * 1e3:.1f} ms per step() = {N / dt:.0f} env-steps/s with {px / 1e6:.0f} MB of pixels to the host per step ({px / dt / 1e9:.2f} GB/s)")
