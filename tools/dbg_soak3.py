"""Same soak scenario, f32 vs f64 device physics: number of envs that diverge."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim import BatchedSim
from test_gpu_configs import poses_for, walk_actions
from test_oracle_physics import model_dict
task, na, N, T = "slot_insertion", 3, 256, 150
md = model_dict(task, na)
gids = np.arange(N)
for f64 in (False, True):
    sim = BatchedSim(task, na, N, f64=f64, options={"export_contacts": 0})
    sim.reset(poses_for(task, gids, 7000))
    acts = walk_actions(md, gids, T, 21, 7000)
    div = np.zeros(N, dtype=bool); ovf = np.zeros(N, dtype=bool); vmax = 0
    for t in range(T):
        a = acts[t].copy()
        a[:, 6] = a[:, 13] = 1.0 if (t // 25) % 2 == 0 else 0.0
        a[:, 1] += 0.004 * t; a[:, 8] += 0.004 * t
        sim.step(a)
        d = sim.diag()
        div |= (d[:, 3] & 1) != 0; ovf |= d[:, 2] != 0
        v = sim.get_state()[1]
        vmax = max(vmax, float(np.abs(v[:, 23:]).max()))
    print("f64" if f64 else "f32", "diverged envs", int(div.sum()), "overflow envs", int(ovf.sum()), "max object |qvel|", vmax, flush=True)
    sim.close()
