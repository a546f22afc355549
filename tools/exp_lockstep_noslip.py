"""Lock-step comparison of the noslip pass per tree (noslip_trees = 1) with the wave-wide groups (0) on the scripted SlotInsertion episode:
every env-step both variants start from the SAME state (the reference's), so a deviation is that step's own and not an earlier one's
amplified.  f32 by default, any argument: f64 (with qcqp_tridiag = 2 on both sides).  Prints the steps whose largest |dv| exceeds 1e-2
(1e-6 in f64) or that reset an env.  Run on the GPU box: python tools/exp_lockstep_noslip.py [f64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import episode_util as U
from av_aloha_amd import workloads as W
from av_aloha_amd.sim_env import make_sim_env
task, n = "slot_insertion", 32
F64 = len(sys.argv) > 1
envA = make_sim_env("sim_" + task, cameras=[], num_envs=n, f64=F64); envA.sim.set_option("noslip_trees", 0); envA.sim.set_option("qcqp_tridiag", 2)
envB = make_sim_env("sim_" + task, cameras=[], num_envs=n, f64=F64); envB.sim.set_option("noslip_trees", 1); envB.sim.set_option("qcqp_tridiag", 2)
poses = W.object_poses(task, np.arange(n), 1000)
envA.sim.reset(poses); envB.sim.reset(poses)
obs = envA.get_obs(); q = obs["qpos"].reshape(n, -1)
home = {k: obs["poses"][k].reshape(n, 7).copy() for k in ("left", "right", "middle")}
script = U.make_script(task, home, q)
worst = []
for t in range(script.steps()):
    qa, va, ca, wa = envA.sim.get_state()
    envB.sim.set_state(qa, va, ca, wa)
    a = script.action(q)
    envA.sim.step_cartesian(a); envB.sim.step_cartesian(a)
    q, v, _, _ = envA.sim.get_state()
    qb, vb, _, _ = envB.sim.get_state()
    dA, dB = envA.sim.diag(), envB.sim.diag()
    dq = np.abs(qb - q).max(1); dv = np.abs(vb - v).max(1)
    k = int(np.argmax(dv))
    if dv[k] > (1e-6 if F64 else 1e-2) or (dB[:, 3] & 1).any():
        print(f"step {t}: env {k} |dq| {dq[k]:.3e} |dv| {dv[k]:.3e} ncon A/B {dA[k,0]}/{dB[k,0]} nefc {dA[k,1]}/{dB[k,1]} diverged B {int((dB[:,3]&1).sum())} A {int((dA[:,3]&1).sum())}; envs with |dv|>1e-2: {int((dv>1e-2).sum())}")
    worst.append(dv.max())
print("max |dv| over the episode:", max(worst), "at step", int(np.argmax(worst)))
