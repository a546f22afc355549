import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim_env import make_sim_env
from av_aloha_amd import workloads as W
from scripted import SlotInsertionScript
n = 128
kw = {k: float(v) for k, v in (a.split("=") for a in sys.argv[1:])}
env = make_sim_env("sim_slot_insertion", cameras=[], num_envs=n)
env.sim.reset(W.object_poses("slot_insertion", np.arange(n), 1000))
obs = env.get_obs()
home = {k: obs["poses"][k].copy() for k in ("left", "right", "middle")}
q0 = obs["qpos"].copy()
script = SlotInsertionScript(home, obs["qpos"], **kw)
for t in range(script.steps()):
    q = env.sim.get_state()[0]
    _, rw, su = env.sim.step_cartesian(script.action(q))
q = env.sim.get_state()[0]
yaw = 2 * np.arctan2(q[:, 36], q[:, 33])
print("reward hist", np.bincount(rw, minlength=5))
for r in (0, 3, 4):
    m = rw == r
    if not m.any(): continue
    print(f"reward {r}: n {m.sum()} use_left {script.use_left[m].mean():.2f} |stick0 x| {np.abs(q0[m,30]).mean():.3f} stick0 y {q0[m,31].mean():.3f} slot x {q0[m,23].mean():.3f} slot y {q0[m,24].mean():.3f} "
          f"|yaw| p50 {np.percentile(np.abs(yaw[m]),50):.3f} p90 {np.percentile(np.abs(yaw[m]),90):.3f} stick z p50 {np.percentile(q[m,32],50):.3f} dxy p50 {np.percentile(np.linalg.norm(q[m,30:32]-q[m,23:25],axis=1),50):.3f} yawcorr p90 {np.percentile(np.abs(script.yaw[m]),90):.2f}")
bad = np.nonzero(rw != 4)[0][:12]
for e in bad:
    print(e, "rw", rw[e], "left" if script.use_left[e] else "right", "stick0", q0[e, 30:32].round(3), "slot", q0[e, 23:25].round(3), "final stick", q[e, 30:33].round(3), "yaw %.3f corr %.3f" % (yaw[e], script.yaw[e]), "xycorr", script.corr[e].round(3))
import json
man = env.sim.manifest
names = man["geom_names"]
env.sim.set_option("export_contacts", 1)
env.sim.step_cartesian(script.action(env.sim.get_state()[0]))
ncon, pairs, dist = env.sim.contacts()
for e in (1, 2):
    print("env", e, "fingers", q[e, 6:8].round(4), q[e, 14:16].round(4), "ncon", ncon[e])
    for k in range(ncon[e]):
        a, b = pairs[e, k]
        print("   ", names[a], "|", names[b], "%.5f" % dist[e, k])
    print("   stick pose", q[e, 30:37].round(3), "ctrl fingers", env.sim.get_state()[2][e, [6, 13]].round(4))
