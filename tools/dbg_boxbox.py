import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from orc_env import OrcEnv
from test_oracle_physics import model_dict
from test_gpu_boxbox import rand_quat
from av_aloha_amd.sim import BatchedSim
md = model_dict(); n = 256; rng = np.random.default_rng(5)
q = np.repeat(md["qpos_home"][None], n, 0).copy()
for i in range(n):
    small = i % 2 == 0
    q[i, 23:26] = [rng.uniform(-0.05, 0.05), rng.uniform(0.08, 0.14), rng.uniform(-0.005, 0.03)]
    q[i, 26:30] = rand_quat(rng, small)
    q[i, 30:33] = q[i, 23:26] + rng.uniform(-0.04, 0.04, 3) + [0, 0, rng.uniform(0, 0.03)]
    q[i, 33:37] = rand_quat(rng, small)
sim = BatchedSim("slot_insertion", 3, n, f64=True)
sim.set_qpos(q)
rw = np.empty(n, dtype=np.int32); su = np.empty(n, dtype=np.uint8)
sim.h.check(sim.h.L.avsim_observe(sim.h.h, None, rw.ctypes.data, su.ctypes.data))
ncon, pairs, dist = sim.contacts()
e = OrcEnv(); e.L.orc_set_qpos.argtypes = [C.c_void_p, C.c_void_p]
names = e.man["geom_names"]
bad = 0
for i in range(n):
    e.L.orc_set_qpos(e.dptr, q[i].ctypes.data)
    cs = list(e.d.contact)[: e.d.ncon]
    dev = [(names[a], names[b], round(float(d) * 1e3, 4)) for (a, b), d in zip(pairs[i][:ncon[i]], dist[i][:ncon[i]])]
    orc = [(names[c.geom1], names[c.geom2], round(c.dist * 1e3, 4)) for c in cs]
    if dev != orc:
        bad += 1
        if bad <= 4:
            print("env", i, "small" if i % 2 == 0 else "large")
            print("  dev", dev)
            print("  orc", orc)
print("mismatching envs", bad, "of", n)
