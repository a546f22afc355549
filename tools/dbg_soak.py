"""Soak run (debug aid, GPU box): every task x arm count, random-walk actions with gripper toggles, diagnostics per task."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim import BatchedSim
from test_gpu_configs import poses_for, walk_actions
from test_oracle_physics import model_dict
N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 150
for task in ("insert_peg", "slot_insertion", "sew_needle", "tube_transfer", "hook_package"):
    for na in (2, 3):
        md = model_dict(task, na)
        nj = 21 if na == 3 else 14
        gids = np.arange(N)
        sim = BatchedSim(task, na, N, options={"export_contacts": 0})
        sim.reset(poses_for(task, gids, 7000))
        acts = walk_actions(md, gids, T, nj, 7000)
        # grippers: square wave, arms wander towards the table
        mx_con = mx_efc = mx_it = 0
        ovf = nan = 0
        nan_envs = np.zeros(N, dtype=bool); ovf_envs = np.zeros(N, dtype=bool); first_nan = None
        rsum = 0
        t0 = time.perf_counter()
        for t in range(T):
            a = acts[t].copy()
            a[:, 6] = a[:, 13] = 1.0 if (t // 25) % 2 == 0 else 0.0
            a[:, 1] += 0.004 * t; a[:, 8] += 0.004 * t          # shoulders lean forward over time -> arms reach the table
            ap, rw, su = sim.step(a)
            d = sim.diag()
            mx_con = max(mx_con, int(d[:, 0].max())); mx_efc = max(mx_efc, int(d[:, 1].max()))
            mx_it = max(mx_it, int(((d[:, 3] >> 28) & 0xf).max()))
            ovf |= int(np.bitwise_or.reduce(d[:, 2])); nan += int((d[:, 3] & 1).sum())
            nan_envs |= (d[:, 3] & 1) != 0; ovf_envs |= d[:, 2] != 0
            if first_nan is None and nan_envs.any(): first_nan = (t, int(np.argmax(nan_envs)))
            rsum += int(rw.sum())
        dt = time.perf_counter() - t0
        print(f"{task:15s} {na}arms: max ncon {mx_con:3d}/{sim.maxcon} max nefc {mx_efc:3d}/{sim.maxefc} newton max it {mx_it} overflow flags {ovf} in {int(ovf_envs.sum())} envs, NaN in {int(nan_envs.sum())} envs (first {first_nan}), reward sum {rsum} ({N * T / dt:.0f} env-steps/s host mode)", flush=True)
        sim.close()
