"""Per-phase cycle breakdown of the physics kernel (debug aid; run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim import BatchedSim
from test_oracle_physics import OBJ, home_action, model_dict
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
opts = {"pgs_iters": 20, "profile_phases": 1, "export_contacts": 0}
for a in sys.argv[2:]:
    k, v = a.split("="); opts[k] = float(v)
sim = BatchedSim("slot_insertion", 3, N, options=opts)
sim.reset(np.repeat(OBJ[None], N, 0))
md = model_dict()
a = np.repeat(home_action(md)[None], N, 0)
for _ in range(3):
    sim.step(a)
out = np.zeros((N, 18), dtype=np.int64)
sim.h.check(sim.h.L.avsim_get_phase_cycles(sim.h.h, out.ctypes.data))
names = ["kinematics", "crb", "rne", "smooth", "collide", "rows", "solve", "euler"]
print("broad/narrow per collide call:", out[:, 8].mean() / 21, out[:, 9].mean() / 21)
m = out[:, :8].mean(0) / 20
print("cycles per substep per wave (mean over envs):")
for n, v in zip(names, m):
    print(f"  {n:10s} {v:10.0f}  {100 * v / m.sum():5.1f}%")
print(f"  total      {m.sum():10.0f}  -> {m.sum() * 20 / 2.4e6:.2f} ms per env-step per wave at 2.4 GHz;  diag {sim.diag()[0]}")
nn = ["init", "grad", "hess", "chol", "search", "final", "noslip", "backsub"]
mn = out[:, 10:18].mean(0) / 20
print("inside solve (Newton):", "  ".join(f"{n} {v:.0f}" for n, v in zip(nn, mn)), f"  iterations/substep {((sim.diag()[:, 3] >> 16) & 0xfff).mean() / 20:.2f}")
