"""f64 device against the oracle under several line-search settings (HookPackage-2Arms random walk, one env): max |dq| per env-step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from av_aloha_amd.sim import BatchedSim
from av_aloha_amd.workloads import object_poses, walk_actions
from orc_env import OrcEnv
from test_oracle_physics import model_dict
task, na, T = "hook_package", 2, 10
md = model_dict(task, na)
gid = np.array([int(sys.argv[1]) if len(sys.argv) > 1 else 5])
pose = object_poses(task, gid, 3000)[0]
acts = walk_actions(md["qpos_home"], md["act_ctrlrange"], gid, T, 14, 3000)[:, 0].astype(np.float64)
for tol, it in ((1e-10, 50), (1e-6, 50), (1e-10, 8), (1e-4, 8), (1e-2, 8), (1e-2, 3), (1e-2, 50)):
    e = OrcEnv(task, na); e.d.solver = 1; e.d.ls_tol, e.d.ls_iters = tol, it; e.reset(pose)
    sim = BatchedSim(task, na, 1, f64=True, options={"solver": 1, "ls_tolerance": tol, "ls_iterations": it}); sim.reset(pose[None])
    d = []
    for a in acts:
        e.env_step(a); sim.step(a[None])
        d.append(np.abs(sim.get_state()[0][0] - e.qpos).max())
    print(f"ls_tolerance {tol:g} ls_iterations {it}: max |dq| per step", " ".join(f"{x:.1e}" for x in d), " ncon", e.d.ncon, "device newton iters max", (sim.diag()[0, 3] >> 28) & 0xf)
    e.close(); sim.close()
