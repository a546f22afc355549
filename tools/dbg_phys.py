import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from orc_env import OrcEnv
from test_oracle_physics import OBJ, home_action, model_dict
from av_aloha_amd.sim import BatchedSim
md = model_dict()
f64 = '--f32' not in sys.argv
sim = BatchedSim("slot_insertion", 3, 1, f64=f64, options={"pgs_iters": 20})
sim.reset(OBJ[None])
e = OrcEnv(); e.d.pgs_iters = 20; e.reset(OBJ)
q, v, c, w = sim.get_state()
print('reset qpos err', np.abs(q[0]-e.qpos).max(), 'ctrl err', np.abs(c[0]-e.ctrl).max())
print('ncon after reset gpu', sim.contacts()[0], 'orc', e.d.ncon)
a = home_action(md)
for nsub in (1,):
    ap, rw, su = sim.step(a[None], nsub=nsub)
    e.env_step(a, nsub=nsub)
    q, v, c, w = sim.get_state()
    d = sim.diag()[0]
    print(f'nsub {nsub}: qpos err {np.abs(q[0]-e.qpos).max():.3e} at {np.abs(q[0]-e.qpos).argmax()} qvel err {np.abs(v[0]-e.qvel).max():.3e} at {np.abs(v[0]-e.qvel).argmax()} warm err {np.abs(w[0]-e.arr("qacc_warmstart",35)).max():.3e} diag {d} orc ncon {e.d.ncon} nefc {e.d.nefc}')


np.set_printoptions(linewidth=200, precision=3)
print('warm diff', (w[0]-e.arr("qacc_warmstart",35)))
print('warm orc', e.arr("qacc_warmstart",35))
print('efc_force orc', np.array(e.d.efc_force[:8]), 'R', np.array(e.d.efc_R[:8]), 'aref', np.array(e.d.efc_aref[:8]))
