"""Per-step cost over a whole episode of a bench workload (debug aid; GPU box): wall time per step, Newton iteration statistics,
contact counts, the most expensive envs.  usage: python tools/dbg_episode_cost.py [config] [steps] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd import workloads as W, _ffi
from av_aloha_amd.sim import BatchedSim
from test_oracle_physics import model_dict
cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
cfg = W.CONFIGS[cfg_id]
md = model_dict(cfg["task"], cfg["arms"])
ids = np.arange(N)
sim = BatchedSim(cfg["task"], cfg["arms"], N, options={"solver": 1, "export_contacts": 0})
sim.reset(W.object_poses(cfg["task"], ids, cfg["seed"]))
ch = np.asarray(md["ctrl_home"], dtype=np.float64)
Ts = []
for arm, sl in ((0, slice(0, 6)), (1, slice(7, 13)), (2, slice(14, 21))):
    q = np.ascontiguousarray(ch[sl])[None]; Tm = np.empty((1, 16))
    sim.h.check(sim.h.L.avsim_fk_jac(sim.h.h, arm, 1, q.ctypes.data, Tm.ctypes.data, None)); Ts.append(Tm)
home = W.home_poses(Ts)
if cfg_id == 4:
    acts = W.walk_actions(md["qpos_home"], md["act_ctrlrange"], ids, T, 14, cfg["seed"])
for t in range(T):
    t0 = time.perf_counter()
    if cfg_id == 4: sim.step(acts[t])
    else: sim.step_cartesian(W.sinusoid_actions(home, ids, N, t), _ffi.IK_DLS)
    dt = time.perf_counter() - t0
    if t % 10 == 0 or dt > 0.03:
        d = sim.diag()
        nit = (d[:, 3] >> 16) & 0xfff
        worst = np.argsort(-nit)[:3]
        print(f"step {t:3d} {dt * 1e3:7.2f} ms  newton it/substep mean {nit.mean() / 20:.2f} max {nit.max() / 20:.1f} (envs {worst.tolist()})  ncon mean {d[:, 0].mean():.1f} max {d[:, 0].max()}  rows max {d[:, 1].max()}  flags {np.bitwise_or.reduce(d[:, 2])} diverged {(d[:, 3] & 1).sum()}", flush=True)
