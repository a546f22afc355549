import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim_env import make_sim_env
from scripted import SlotInsertionScript
from test_gpu_configs import poses_for
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
TRACE = os.environ.get('TRACE') == '1'
env = make_sim_env("sim_slot_insertion", cameras=[], num_envs=n, f64=os.environ.get("F64") == "1")
env.sim.reset(poses_for("slot_insertion", np.arange(n), 1000))
obs = env.get_obs()
home = {k: obs["poses"][k].copy() for k in ("left", "right", "middle")}
sc = SlotInsertionScript(home, obs["qpos"], drop=float(os.environ.get("DROP", 0.03)), clip=float(os.environ.get("CLIP", 0.03)), gain=float(os.environ.get("GAIN", 0.15)))
best = np.zeros(n, dtype=np.int32)
last = -1
for t in range(sc.steps()):
    q = env.sim.get_state()[0]
    k, f = sc.phase()
    a = sc.action(q)
    _, rw, su = env.sim.step_cartesian(a)
    best = np.maximum(best, rw)
    if TRACE and 100 <= t <= 170 and t % 3 == 0:
        q2 = env.sim.get_state()[0]
        nc, pr, ds = env.sim.contacts()
        names = env.sim.manifest["geom_names"]
        cl = sorted(set((names[x] or f"g{x}") for (x, y), dd in zip(pr[0][:nc[0]], ds[0][:nc[0]]) if names[y] == "stick") | set((names[y] or f"g{y}") for (x, y), dd in zip(pr[0][:nc[0]], ds[0][:nc[0]]) if names[x] == "stick"))
        print("  ", t, "stick", q2[0, 30:37].round(3), "fingersL", q2[0, 6:8].round(4), "fingersR", q2[0, 14:16].round(4), "rw", rw[0], cl)
    if k != last or t == sc.steps() - 1:
        eL = env._fk_pose(0, np.ascontiguousarray(q[:, 0:6])); eR = env._fk_pose(1, np.ascontiguousarray(q[:, 8:14]))
        print(f"t={t} phase {k}: stick {q[0, 30:33].round(3)} slot {q[0, 23:26].round(3)} eeL {eL[0, :3].round(3)} eeR {eR[0, :3].round(3)} tgtL {a[0, 0:3].round(3)} "
              f"reward hist {np.bincount(rw, minlength=5)} best {np.bincount(best, minlength=5)} dy mean {np.abs(q[:, 31] - q[:, 24]).mean():.4f} stick z {q[:, 32].mean():.3f}")
        last = k
d = env.sim.diag()
print("final reward hist", np.bincount(rw, minlength=5), "best", np.bincount(best, minlength=5), "success", su.mean(), "flagged", (d[:, 3] & 1).sum())
if len(sys.argv) > 2:
    from dbg_render_png import write_png
    img = env.sim.render_rgb(["overhead_cam", "zed_cam_left", "worms_eye_cam"], 240, 320)[0]
    write_png(sys.argv[2], np.concatenate(list(img), axis=1))
env.close()
