import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from av_aloha_amd.sim import BatchedSim
from av_aloha_amd import workloads as W
from test_oracle_physics import model_dict
N=1024; task, arms = sys.argv[1], int(sys.argv[2])
md = model_dict(task, arms)
sim = BatchedSim(task, arms, N, options={"profile_phases": 1, "export_contacts": 0})
seed = {"hook_package": 3000, "sew_needle": 2000}[task]
sim.reset(W.object_poses(task, np.arange(N), seed))
nj = 21 if arms == 3 else 14
acts = W.walk_actions(md["qpos_home"], md["act_ctrlrange"], np.arange(N), 30, nj, seed)
for t in range(30): sim.step(acts[t])
out = np.zeros((N, 26), dtype=np.int64)
sim.h.check(sim.h.L.avsim_get_phase_cycles(sim.h.h, out.ctypes.data))
code = out[:, 22]     # prof[4]
print(task, "why bits histogram:", np.bincount((code % 1000000) // 1000, minlength=128).nonzero()[0], np.bincount((code % 1000000) // 1000, minlength=128)[np.bincount((code % 1000000) // 1000, minlength=128).nonzero()[0]])
