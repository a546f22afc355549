"""Debug aid: one substep from a saved state (gpurun_out/dbg_state.npz of tools/dbg_newton_state.py) on the f64 device under several solver
options, against the oracle.   usage: python tools/dbg_newton_options.py <task> [state.npz]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import episode_util as U
from av_aloha_amd.sim import BatchedSim
task = sys.argv[1]
z = np.load(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "dbg_state.npz"))
tol = float(os.environ.get("NEWTON_TOL", "1e-13"))
U.NEWTON_TOL = tol
model = U.MODEL_OF.get(task, task)
e = U._new_env(task, z["pose"])
e.qpos[:] = z["q"]; e.qvel[:] = z["v"]; e.arr("qacc_warmstart", e.nv)[:] = z["w"]; e.ctrl[:] = z["c"]
e.step(1)
qa = np.array(e.arr("qacc_warmstart", e.nv))
print(f"oracle: newton its {e.d.stat_sweeps}, ncon {e.d.ncon}, nefc {e.d.nefc}")
for f64 in (True, False):
    for opts in ({}, {"newton_component": 0}, {"noslip_trees": 0}, {"noslip_per_tree": 0}, {"solver": 0, "pgs_iters": 200}):
        o = dict(opts)
        if f64:
            o.setdefault("newton_tol", tol)
        sim = BatchedSim(model, 3, 1, f64=f64, variant=U.VARIANT, options=o)
        sim.reset(z["pose"][None])
        sim.set_state(z["q"][None], z["v"][None], z["c"][None], z["w"][None])
        sim.step_ctrl(1)
        q, v, _, w = sim.get_state()
        d = sim.diag()[0]
        print(f"  {'f64' if f64 else 'f32'} {str(opts):36s} newton its {(d[3] >> 16) & 0xfff:3d}  |dqacc| vs oracle {np.abs(w[0] - qa).max():.3e}  |dv| {np.abs(v[0] - np.array(e.qvel)).max():.3e}")
        sim.close()
