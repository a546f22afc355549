"""Debug aid: where does the teacher-forced oracle step leave the device's f32 step (tests/episode_util.py lockstep_worker)?
usage: python tools/dbg_lockstep.py <task> [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import episode_util as U

task = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = U.device_episode(task, n, f64=False, record_state=True)
c = np.ascontiguousarray
np.set_printoptions(precision=4, suppress=True, linewidth=220)


def worker(args):
    task, pose, q0, v0, w0, l0, ctrls, q1 = args
    e = U._new_env(task, pose)
    T = ctrls.shape[0]
    warm = e.arr("qacc_warmstart", e.nv)
    out = []
    for t in range(T):
        e.qpos[:] = q0[t]; e.qvel[:] = v0[t]; warm[:] = w0[t]; e.d.threaded = int(l0[t])
        rw, su = U._step_ctrl(e, ctrls[t])
        d = np.abs(np.array(e.qpos) - q1[t])
        out.append((d.max(), int(d.argmax()), rw, e.d.ncon, np.array(e.qpos).copy()))
    e.close()
    return out


res = U.pool_map(worker, [(task, dev["poses"][k], c(dev["q0"][:, k]), c(dev["v0"][:, k]), c(dev["w0"][:, k]), c(dev["l0"][:, k]), c(dev["ctrl"][:, k]), c(dev["qpos"][:, k])) for k in range(n)])
for k, r in enumerate(res):
    errs = np.array([x[0] for x in r])
    if errs.max() > 0.03:
        bad = np.nonzero(errs > 0.03)[0]
        t = int(bad[0])
        print(f"env {k}: {len(bad)} steps with one-step err > 0.03, first at t={t} (err {errs[t]:.3f} at qpos[{r[t][1]}]); dev reward {dev['reward'][t, k]} orc {r[t][2]}; ncon dev {dev['ncon'][t, k]} orc {r[t][3]}")
        print("   start  q0[23:]", dev["q0"][t, k, 23:], " v0[21:] max", np.abs(dev["v0"][t, k]).max())
        print("   device q1[23:]", dev["qpos"][t, k, 23:])
        print("   oracle q1[23:]", r[t][4][23:])
        print("   errs around", errs[max(0, t - 3):t + 6])
