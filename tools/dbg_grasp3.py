import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim_env import make_sim_env
from scripted import grasp_lift_targets
from test_gpu_configs import poses_for
ids = np.array([int(x) for x in sys.argv[1].split(",")] + [0])
n = len(ids)
env = make_sim_env("sim_sew_needle", cameras=[], num_envs=n, f64=True)
poses = poses_for("sew_needle", ids, 2000)
env.sim.reset(poses)
obs = env.get_obs()
home = {k: obs["poses"][k].copy() for k in ("left", "right", "middle")}
needle0 = obs["qpos"][:, 30:33].copy()
names = env.sim.manifest["geom_names"]
for t, a in enumerate(grasp_lift_targets(home, needle0 + np.array([0.0, 0.0, 0.01]))):
    if 118 <= t <= 150:
        for s in range(20 if t >= 124 else 1):
            if t >= 124:
                env.sim.step_cartesian(a, nsub=1)
            else:
                env.sim.step_cartesian(a)
            q, v, _, _ = env.sim.get_state()
            nc, pr, ds = env.sim.contacts()
            k = 0
            cl = [(names[x] or f"g{x}", names[y] or f"g{y}", round(float(dd) * 1000, 2)) for (x, y), dd in zip(pr[k][:nc[k]], ds[k][:nc[k]]) if "needle" == names[x] or "needle" == names[y]]
            print(t, s, "needle", q[k, 30:33].round(4), "v", v[k, 29:32].round(2), "w", v[k, 32:35].round(1), "fingers", q[k, 14:16].round(4), cl)
            if np.abs(v[k, 29:35]).max() > 50:
                sys.exit(0)
    else:
        env.sim.step_cartesian(a)
