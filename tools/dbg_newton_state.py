"""Debug aid: one substep of one env of a scripted f64 episode with the Newton solver's per-iteration trace (a build with
AVSIM_EXTRA_FLAGS_F64=-DAVSIM_DEBUG_NEWTON prints it), next to the oracle's iteration count.
usage: NEWTON_TOL=1e-13 python tools/dbg_newton_state.py <task> <env> <step> [n_envs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import episode_util as U
from av_aloha_amd.sim import BatchedSim
task, k, t = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]); n = int(sys.argv[4]) if len(sys.argv) > 4 else 16
if os.environ.get("NEWTON_TOL"):
    U.NEWTON_TOL = float(os.environ["NEWTON_TOL"])
dev = U.device_episode(task, n, f64=True, record_state=True)
np.savez("gpurun_out/dbg_state.npz", q=dev["q0"][t, k], v=dev["v0"][t, k], w=dev["w0"][t, k], c=dev["ctrl"][t, k], pose=dev["poses"][k])
model = U.MODEL_OF.get(task, task)
sim = BatchedSim(model, 3, 1, f64=True, variant=U.VARIANT, options={"newton_tol": U.NEWTON_TOL} if U.NEWTON_TOL is not None else None)
sim.reset(dev["poses"][k][None])
sim.set_state(dev["q0"][t, k][None], dev["v0"][t, k][None], dev["ctrl"][t, k][None], dev["w0"][t, k][None])
sim.set_option("newton_iters", 99)         # (the debug build's trace is keyed to this cap)
print("--- device substep 0 ---", flush=True)
sim.step_ctrl(1)
q, v, _, w = sim.get_state()
d = sim.diag()[0]
e = U._new_env(task, dev["poses"][k])
e.qpos[:] = dev["q0"][t, k]; e.qvel[:] = dev["v0"][t, k]; e.arr("qacc_warmstart", e.nv)[:] = dev["w0"][t, k]; e.ctrl[:] = dev["ctrl"][t, k]
e.step(1)
print(f"device newton its {(d[3] >> 16) & 0xfff}, oracle {e.d.stat_sweeps}; |dq| {np.abs(q[0] - np.array(e.qpos)).max():.3e} |dv| {np.abs(v[0] - np.array(e.qvel)).max():.3e} |dqacc| {np.abs(w[0] - np.array(e.arr('qacc_warmstart', e.nv))).max():.3e}")
