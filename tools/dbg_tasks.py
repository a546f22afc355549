import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from orc_env import OrcEnv
from test_oracle_physics import home_action, model_dict
from test_gpu_physics import actions_wiggle
from av_aloha_amd.sim import BatchedSim
for task in sys.argv[1:]:
    md = model_dict(task)
    obj = md["qpos_home"][md["objects_qposadr"][0]:].reshape(-1, 7).copy()
    acts = actions_wiggle(md, 3)
    sim = BatchedSim(task, 3, 1, f64=True, options={"pgs_iters": 20})
    sim.reset(obj[None])
    e = OrcEnv(task, 3); e.d.pgs_iters = 20; e.reset(obj)
    for a in acts:
      for k in range(4):
        sim.step(a[None], nsub=5); e.env_step(a, nsub=5)
        q, v, c, w = sim.get_state()
        i = np.abs(q[0]-e.qpos).argmax()
        print(task, 'diag', sim.diag()[0], 'orc ncon', e.d.ncon, 'nefc', e.d.nefc, 'ovf', e.d.overflow, 'qerr %.2e at %d' % (np.abs(q[0]-e.qpos).max(), i))
    nc, pairs, dist = sim.contacts()
    names = e.man['geom_names']
    print('gpu', [(names[a], names[b], round(d, 6)) for (a, b), d in zip(pairs[0][:nc[0]], dist[0][:nc[0]])])
    print('orc', [(c[0], c[1], round(c[2], 6)) for c in e.contacts()])
