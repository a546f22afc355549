"""Success statistics of the scripted policies (tests/scripted.py, av_aloha_amd/workloads.py) on the device: needle lift of
config 3 and the SlotInsertion script.  usage: python tools/report_scripted.py [n] [key=value script options]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim_env import make_sim_env
from av_aloha_amd import workloads as W
from scripted import SlotInsertionScript
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kw = {k: float(v) for k, v in (a.split("=") for a in sys.argv[2:])}
if "T" in os.environ: SlotInsertionScript.T = tuple(int(x) for x in os.environ["T"].split(","))
which = os.environ.get("WHICH", "both")
if which in ("both", "lift"):
    env = make_sim_env("sim_sew_needle", cameras=[], num_envs=n)
    poses = W.object_poses("sew_needle", np.arange(n), 2000)
    env.sim.reset(poses)
    obs = env.get_obs()
    home = {k: obs["poses"][k].copy() for k in ("left", "right", "middle")}
    needle0 = obs["qpos"][:, 30:33].copy()
    flagged = np.zeros(n, bool); capped = np.zeros(n, bool)
    for a in W.grasp_lift_targets(home, needle0 + np.array([0.0, 0.0, 0.01])):
        _, rw, _ = env.sim.step_cartesian(a)
        d = env.sim.diag(); flagged |= (d[:, 3] & 1) != 0; capped |= d[:, 2] != 0
    q = env.sim.get_state()[0]
    lifted = q[:, 32] - needle0[:, 2] > 0.08
    print(f"needle lift: lifted {lifted.mean():.3f} reward>=2 {(rw >= 2).mean():.3f} diverged {flagged.mean():.3f} capped {capped.mean():.3f} mean ncon {d[:,0].mean():.1f}")
    env.close()
if which in ("both", "slot"):
    env = make_sim_env("sim_slot_insertion", cameras=[], num_envs=n)
    env.sim.reset(W.object_poses("slot_insertion", np.arange(n), 1000))
    obs = env.get_obs()
    home = {k: obs["poses"][k].copy() for k in ("left", "right", "middle")}
    script = SlotInsertionScript(home, obs["qpos"], **kw)
    best = np.zeros(n, np.int32); flagged = np.zeros(n, bool); capped = np.zeros(n, bool)
    hist = []
    for t in range(script.steps()):
        q = env.sim.get_state()[0]
        _, rw, su = env.sim.step_cartesian(script.action(q), int(os.environ.get("IK", "0")))
        best = np.maximum(best, rw)
        d = env.sim.diag(); flagged |= (d[:, 3] & 1) != 0; capped |= d[:, 2] != 0
        if t % 20 == 0:
            q2 = env.sim.get_state()[0]
            err = q2[:, 23:25] - q2[:, 30:32]
            # measured hand orientation: rotation about the world z axis relative to the commanded top-down orientation
            qh = q2[:, 0:6] if False else None
            arm_q = np.where(script.use_left[:, None], q2[:, 0:6], q2[:, 8:14])
            Tl = np.empty((n, 16)); Tr = np.empty((n, 16))
            h = env.sim.h
            ql = np.ascontiguousarray(q2[:, 0:6]); qr = np.ascontiguousarray(q2[:, 8:14])
            h.check(h.L.avsim_fk_jac(h.h, 0, n, ql.ctypes.data, Tl.ctypes.data, None)); h.check(h.L.avsim_fk_jac(h.h, 1, n, qr.ctypes.data, Tr.ctypes.data, None))
            T = np.where(script.use_left[:, None], Tl, Tr).reshape(n, 4, 4)
            if t == 0: T0 = None
            if script.phase()[0] == 2 and "Tg" not in globals(): Tg = T.copy()
            hyaw = 0.0
            if "Tg" in globals():
                Rrel = np.einsum("nij,nkj->nik", T[:, :3, :3], Tg[:, :3, :3])
                hyaw = float(np.percentile(np.abs(np.arctan2(Rrel[:, 1, 0], Rrel[:, 0, 0])), 90).round(3))
            hist.append((t, script.phase()[0], "hand yaw p90", hyaw, np.bincount(rw, minlength=5).tolist(), "stick z p10/50/90", np.percentile(q2[:, 32], [10, 50, 90]).round(3).tolist(),
                         "held", float((q2[:, 32] > 0.03).mean().round(2)), "|xy err| p50/p90", np.percentile(np.linalg.norm(err, axis=1), [50, 90]).round(3).tolist(),
                         "yaw p90", float(np.percentile(np.abs(2 * np.arctan2(q2[:, 36], q2[:, 33])), 90).round(3))))
    q = env.sim.get_state()[0]
    print(f"slot insertion: best==4 {(best == 4).mean():.3f} final==4 {(rw == 4).mean():.3f} final hist {np.bincount(rw, minlength=5).tolist()} diverged {flagged.mean():.3f} capped {capped.mean():.3f}")
    for h in hist: print("   step", h)
    r3 = rw == 3
    if r3.any():
        dy, dx, z = q[r3, 31] - q[r3, 24], q[r3, 30] - q[r3, 23], q[r3, 32]
        print("   reward-3 envs: dy", np.percentile(dy, [5, 50, 95]).round(4), "dx", np.percentile(dx, [5, 50, 95]).round(4), "stick z", np.percentile(z, [5, 50, 95]).round(4))
        qw = q[r3, 33:37]; print("   stick quat (median)", np.median(qw, 0).round(3))
    env.close()
