"""Debug: device Newton (f64 / f32) vs the oracle's Newton on the wiggle rollout."""
import sys, time
import numpy as np
sys.path.insert(0, "tests")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from orc_env import OrcEnv
from test_oracle_physics import OBJ, home_action, model_dict
from test_gpu_physics import actions_wiggle, make

md = model_dict()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 12
acts = actions_wiggle(md, T)
e = OrcEnv("slot_insertion", 3)
e.d.solver = 1
e.reset(OBJ)
ref = []
for a in acts:
    ap, r, s = e.env_step(a)
    ref.append((e.qpos.copy(), e.qvel.copy(), r, e.d.ncon))
for f64 in (True, False):
    sim = make(f64=f64, solver=1)
    sim.reset(OBJ[None])
    for t, a in enumerate(acts):
        ap, rw, su = sim.step(a[None])
        qpos, qvel, _, _ = sim.get_state()
        print("f64" if f64 else "f32", t, "ncon", int(sim.contacts()[0][0]), ref[t][3], "dq %.2e dv %.2e" % (np.abs(qpos[0] - ref[t][0]).max(), np.abs(qvel[0] - ref[t][1]).max()),
              "rw", rw[0], ref[t][2], "diag", sim.diag()[0])
    sim.close()
