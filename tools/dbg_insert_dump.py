import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim_env import make_sim_env
from scripted import SlotInsertionScript
from test_gpu_configs import poses_for
n = 2
env = make_sim_env("sim_slot_insertion", cameras=[], num_envs=n, f64=True)
env.sim.reset(poses_for("slot_insertion", np.arange(n), 1000))
obs = env.get_obs()
home = {k: obs["poses"][k].copy() for k in ("left", "right", "middle")}
sc = SlotInsertionScript(home, obs["qpos"])
rec = {}
for t in range(140):
    q, v, c, w = env.sim.get_state()
    if t >= 100:
        rec[f"q{t}"], rec[f"v{t}"], rec[f"w{t}"] = q[0].copy(), v[0].copy(), w[0].copy()
    a = sc.action(q)
    env.sim.step_cartesian(a)
    if t >= 100:
        rec[f"c{t}"] = env.sim.get_state()[2][0].copy()      # the ctrl this step used
np.savez(sys.argv[1], **rec)
