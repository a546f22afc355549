"""How often the per-tree noslip pass takes an env-substep (debug aid): config 2's workload at a given step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd import _ffi, workloads as W
from av_aloha_amd.sim import BatchedSim
from av_aloha_amd.compiler.compile import read_blob
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sim = BatchedSim("slot_insertion", 3, N, options={"profile_phases": 1, "export_contacts": 0})
md = read_blob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models", "slot_insertion_3arms.avm"))
ch = np.asarray(md["ctrl_home"], dtype=np.float64)
T_home = []
for arm, sl in ((0, slice(0, 6)), (1, slice(7, 13)), (2, slice(14, 21))):
    q = np.ascontiguousarray(ch[sl])[None]
    Tm = np.empty((1, 16))
    sim.h.check(sim.h.L.avsim_fk_jac(sim.h.h, arm, 1, q.ctypes.data, Tm.ctypes.data, None))
    T_home.append(Tm)
home = W.home_poses(T_home)
ids = np.arange(N)
sim.reset(W.object_poses("slot_insertion", ids, 1000))
for t in range(T):
    sim.step_cartesian(W.sinusoid_actions(home, ids, N, t), _ffi.IK_DLS)
out = np.zeros((N, 26), dtype=np.int64)
sim.h.check(sim.h.L.avsim_get_phase_cycles(sim.h.h, out.ctypes.data))
p = out[:, 18:26].mean(0) / 20
print(f"step {T}: per substep: noslip_trees refused at entry {p[1]:.2f}, gave up on a sliding contact {p[2]:.2f}, done {p[3]:.2f}; pgs_groups steps {p[0]:.1f}; noslip cycles {out[:, 16].mean() / 20:.0f}; ncon {sim.diag()[:, 0].mean():.1f}")
