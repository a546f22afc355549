"""Newton iterations per substep: the f32 product kernel against the f64 oracle FROM THE SAME STATES (config 4: HookPackage-2Arms joint random walk).
The device walks N envs; at every env-step its state is copied out, the device then takes ONE substep (avsim_step with nsub = 1) and so does the oracle from the
device's state (exact in double) with the same action: iteration counts side by side, for all envs and for the heavy ones.
usage: python tools/exp_newton_iters.py [N] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from av_aloha_amd import workloads as W
from av_aloha_amd.compiler.compile import read_blob
from av_aloha_amd.sim import BatchedSim
from orc_env import OrcEnv
import episode_util as U

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
opts = dict(kv.split("=") for kv in sys.argv[3:])
cfg = W.CONFIGS[4]
md = read_blob(os.path.join(ROOT, "models", "hook_package_2arms.avm"))
ids = np.arange(N)
poses = W.object_poses(cfg["task"], ids, cfg["seed"])
acts = W.walk_actions(md["qpos_home"], md["act_ctrlrange"], ids, T, 14, cfg["seed"])
sim = BatchedSim("hook_package", 2, N, options={k: float(v) for k, v in opts.items()})
sim.reset(poses)
rows = []


def oracle_iters(args):
    pose, q, v, w, a = args
    e = OrcEnv("hook_package", 2)
    e.d.solver = 1
    e.reset(pose)
    e.qpos[:] = q; e.qvel[:] = v; e.arr("qacc_warmstart", e.nv)[:] = w
    e.env_step(a.astype(np.float64), 1)
    n, nc = e.d.stat_sweeps, e.d.ncon
    e.close()
    return n, nc


for t in range(T):
    if t >= T // 2:                     # second half: one substep on both sides from the device's state, then the device goes on with the remaining 19
        q, v, c, w = sim.get_state()
        sim.step(acts[t], nsub=1)
        d = sim.diag()
        dev = (d[:, 3] >> 16) & 0xfff
        orc = U.pool_map(oracle_iters, [(poses[k], q[k], v[k], w[k], acts[t, k]) for k in range(N)])
        rows.append((dev.copy(), np.array([o[0] for o in orc]), d[:, 0].copy()))
        sim.step(acts[t], nsub=19)
    else:
        sim.step(acts[t])
dev = np.concatenate([r[0] for r in rows]); orc = np.concatenate([r[1] for r in rows]); ncon = np.concatenate([r[2] for r in rows])
print(f"{len(dev)} substeps (N {N}, options {opts}): Newton iterations device f32 mean {dev.mean():.2f} p50/p90/p99/max {np.percentile(dev, [50, 90, 99, 100])}; oracle f64 mean {orc.mean():.2f} {np.percentile(orc, [50, 90, 99, 100])}")
heavy = orc >= 5
print(f"substeps where the oracle needs >= 5: {heavy.sum()}; device there mean {dev[heavy].mean():.2f}, oracle {orc[heavy].mean():.2f}; device - oracle histogram {np.bincount(np.clip(dev[heavy] - orc[heavy] + 5, 0, 15), minlength=16).tolist()} (index 5 = equal)")
hd = dev >= 8
print(f"substeps where the DEVICE needs >= 8: {hd.sum()}; oracle there mean {orc[hd].mean() if hd.any() else 0:.2f}; ncon there mean {ncon[hd].mean() if hd.any() else 0:.1f}")
