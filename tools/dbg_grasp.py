"""Debug: scripted top-down grasp-and-lift of the needle by the right arm (BASELINE config 3 flavour)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim_env import make_sim_env


def qmul(a, b):
    w1, x1, y1, z1 = a; w2, x2, y2, z2 = b
    return np.array([w1*w2 - x1*x2 - y1*y2 - z1*z2, w1*x2 + x1*w2 + y1*z2 - z1*y2, w1*y2 - x1*z2 + y1*w2 + z1*x2, w1*z2 + x1*y2 - y1*x2 + z1*w2])


def qrot_y(th):
    return np.array([np.cos(th / 2), 0, np.sin(th / 2), 0])


N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
zg = float(sys.argv[2]) if len(sys.argv) > 2 else 0.14
env = make_sim_env("sim_sew_needle", cameras=[], num_envs=N)
np.random.seed(0)
obs, _ = env.reset()
sim = env.sim
home = {k: obs["poses"][k].copy() for k in ("left", "right", "middle")}
qpos = obs["qpos"]
needle = qpos[:, 30:33].copy()           # second free joint: needle (wall first)
print("home right pose", home["right"][0], "needle", needle[0], "wall", qpos[0, 23:26])
down = np.stack([qmul(qrot_y(-np.pi / 2), home["right"][i, 3:]) for i in range(N)])


def act(rpos, rquat, grip):
    a = np.zeros((N, 23))
    a[:, 0:7] = home["left"]; a[:, 7] = 0
    a[:, 8:11] = rpos; a[:, 11:15] = rquat; a[:, 15] = grip
    a[:, 16:23] = home["middle"]
    return a


def rewards():
    rw = np.empty(N, dtype=np.int32); su = np.empty(N, dtype=np.uint8)
    sim.h.check(sim.h.L.avsim_observe(sim.h.h, None, rw.ctypes.data, su.ctypes.data))
    return rw


def report(tag, o):
    q = o["qpos"]
    ncon, pairs, dist = sim.contacts()
    names = sim.manifest["geom_names"]
    print("   contacts env0:", sorted(set((names[a] or f"g{a}", names[b] or f"g{b}") for a, b in pairs[0][:ncon[0]])))
    ee = env._fk_pose(1, np.ascontiguousarray(q[:, 8:14]))
    print(f"{tag}: ee {ee[0, :3].round(3)} quat {ee[0, 3:].round(3)} fingers {q[0, 14:16].round(4)} needle {q[0, 30:33].round(3)} "
          f"reward hist {np.bincount(rewards(), minlength=6)} needle z mean {q[:, 32].mean():.3f} max {q[:, 32].max():.3f}")


above = needle + np.array([0, 0, zg + 0.10])
grasp = needle + np.array([0, 0, zg])
for t in range(70):
    o, *_ = env.step(act(above, down, 0.0))
report("above", o)
for t in range(50):
    o, *_ = env.step(act(above + (grasp - above) * min(1, (t + 1) / 35), down, 0.0))
report("down", o)
for t in range(30):
    o, *_ = env.step(act(grasp, down, min(1.0, (t + 1) / 15)))
report("closed", o)
for t in range(60):
    o, *_ = env.step(act(grasp + np.array([0, 0, 0.12]) * min(1, (t + 1) / 40), down, 1.0))
report("lifted", o)
for t in range(40):
    o, *_ = env.step(act(grasp + np.array([0, 0, 0.12]), down, 1.0))
report("held", o)
d = sim.diag()
print("ncon mean", d[:, 0].mean(), "diverged", (d[:, 3] & 1).sum())
if len(sys.argv) > 3:
    from dbg_render_png import write_png
    img = sim.render_rgb(["overhead_cam", "wrist_cam_right", "zed_cam_left"], 240, 320)[0]
    write_png(sys.argv[3], np.concatenate(list(img), axis=1))
env.close()
