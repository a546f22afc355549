"""Trace the first failing env of the soak scenario and replay it in the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim import BatchedSim
from orc_env import OrcEnv
from test_gpu_configs import poses_for, walk_actions
from test_oracle_physics import model_dict
task, na, N, T = "slot_insertion", 3, 256, 150
md = model_dict(task, na)
gids = np.arange(N)
sim = BatchedSim(task, na, N, options={"export_contacts": 0})
poses = poses_for(task, gids, 7000)
sim.reset(poses)
acts = walk_actions(md, gids, T, 21, 7000)
first = {}
A = []
for t in range(T):
    a = acts[t].copy()
    a[:, 6] = a[:, 13] = 1.0 if (t // 25) % 2 == 0 else 0.0
    a[:, 1] += 0.004 * t; a[:, 8] += 0.004 * t
    A.append(a)
    sim.step(a)
    d = sim.diag()
    for e in range(N):
        flags = (int(d[e, 2]), int(d[e, 3] & 1), int((d[e, 3] >> 28) & 0xf))
        if flags[1] and e not in first:
            first[e] = (t, flags, int(d[e, 0]), int(d[e, 1]))
print("envs with an event:", len(first), "earliest:", sorted(first.items(), key=lambda kv: kv[1][0])[:8])
e0 = min(first, key=lambda e: first[e][0])
t0 = first[e0][0]
print("replaying env", e0, "event at step", t0, first[e0])
o = OrcEnv(task, na); o.d.solver = 1; o.reset(poses[e0])
sim2 = BatchedSim(task, na, 1, options={"solver": 1})
sim2.reset(poses[e0][None])
for t in range(min(T, t0 + 2)):
    a = A[t][e0].astype(np.float64)
    ap, r, s = o.env_step(a)
    sim2.step(A[t][e0][None])
    q = sim2.get_state()[0][0]
    dd = sim2.diag()[0]
    if t >= t0 - 30:
        print(t, "oracle ncon", o.d.ncon, "nefc", o.d.nefc, "overflow", o.d.overflow, "newton its", o.d.stat_sweeps, "max|qvel|", np.abs(o.qvel).max(),
              "| gpu ncon", dd[0], "nefc", dd[1], "flags", dd[2], "it max", (dd[3] >> 28) & 0xf, "dq", np.abs(q - o.qpos).max(), "gpu max|qvel|", np.abs(sim2.get_state()[1][0]).max(), "at dof", int(np.argmax(np.abs(sim2.get_state()[1][0]))), "nan", dd[3] & 1)
