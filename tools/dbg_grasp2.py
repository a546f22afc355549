import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim_env import make_sim_env
from scripted import grasp_lift_targets
from test_gpu_configs import poses_for
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f64 = len(sys.argv) > 2 and sys.argv[2] == "f64"
env = make_sim_env("sim_sew_needle", cameras=[], num_envs=n, f64=f64)
poses = poses_for("sew_needle", np.arange(n), 2000)
env.sim.reset(poses)
obs = env.get_obs()
home = {k: obs["poses"][k].copy() for k in ("left", "right", "middle")}
needle0 = obs["qpos"][:, 30:33].copy()
names = env.sim.manifest["geom_names"]
rich = np.zeros(n, dtype=np.int64)
prevq = None
for t, a in enumerate(grasp_lift_targets(home, needle0 + np.array([0.0, 0.0, 0.01]))):
    _, rw, _ = env.sim.step_cartesian(a)
    d = env.sim.diag()
    rich += d[:, 0] >= 8
    bad = np.nonzero(d[:, 3] & 1)[0]
    q, v, _, _ = env.sim.get_state()
    if len(bad):
        print("step", t, "diverged envs", bad[:10], "prev needle", None if prevq is None else prevq[bad[0], 30:37].round(3), "prev |v| max", None if prevv is None else np.abs(prevv[bad[0]]).max().round(2),
              "nit", (dprev[bad[0], 3] >> 28) & 15, "ncon", dprev[bad[0], 0])
        nc, pr, ds = cprev
        print("   prev contacts:", sorted(set((names[x] or f"g{x}", names[y] or f"g{y}") for x, y in pr[bad[0]][:nc[bad[0]]])))
    prevq, prevv, dprev = q, v, d
    cprev = env.sim.contacts()
lifted = q[:, 32] - needle0[:, 2] > 0.08
print("rich", (rich >= 100).mean(), "lifted", lifted.mean(), "reward hist", np.bincount(rw, minlength=6), "max |v|", np.abs(v).max())
env.close()
