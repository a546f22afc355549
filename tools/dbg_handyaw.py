"""One env of the SlotInsertion script: yaw (rotation about world z relative to the grasp orientation) of the commanded target,
of FK(ctrl) = the IK's answer, and of the measured hand, plus the arm's joint angles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim_env import make_sim_env
from av_aloha_amd import workloads as W
from scripted import SlotInsertionScript
n = 8
env = make_sim_env("sim_slot_insertion", cameras=[], num_envs=n)
env.sim.reset(W.object_poses("slot_insertion", np.arange(n), 1000))
obs = env.get_obs()
home = {k: obs["poses"][k].copy() for k in ("left", "right", "middle")}
script = SlotInsertionScript(home, obs["qpos"])
def q2R(q):
    w, x, y, z = q
    return np.array([[1-2*(y*y+z*z), 2*(x*y-z*w), 2*(x*z+y*w)], [2*(x*y+z*w), 1-2*(x*x+z*z), 2*(y*z-x*w)], [2*(x*z-y*w), 2*(y*z+x*w), 1-2*(x*x+y*y)]])
def yaw_rel(R, R0):
    Rr = R @ R0.T
    return np.arctan2(Rr[1, 0], Rr[0, 0])
e = int(sys.argv[1]) if len(sys.argv) > 1 else 0
left = bool(script.use_left[e])
R0 = q2R(script.down_l[e] if left else script.down_r[e])
h = env.sim.h
for t in range(script.steps()):
    q = env.sim.get_state()[0]
    a = script.action(q)
    env.sim.step_cartesian(a)
    if t % 10 == 0 or t > 230 and t % 5 == 0:
        q, v, c, _ = env.sim.get_state()
        tq = a[e, 3:7] if left else a[e, 11:15]
        arm = 0 if left else 1
        jc = np.ascontiguousarray(c[e:e+1, 0:6] if left else c[e:e+1, 7:13]); jm = np.ascontiguousarray(q[e:e+1, 0:6] if left else q[e:e+1, 8:14])
        Tc = np.empty((1, 16)); Tm = np.empty((1, 16))
        h.check(h.L.avsim_fk_jac(h.h, arm, 1, jc.ctypes.data, Tc.ctypes.data, None)); h.check(h.L.avsim_fk_jac(h.h, arm, 1, jm.ctypes.data, Tm.ctypes.data, None))
        Tc = Tc.reshape(4, 4); Tm = Tm.reshape(4, 4)
        tp = a[e, 0:3] if left else a[e, 8:11]
        print(t, script.phase()[0], "yaw target %.3f ik %.3f meas %.3f | pos err ik %.4f meas %.4f | tilt ik %.3f | joints" % (
            yaw_rel(q2R(tq), R0), yaw_rel(Tc[:3, :3], R0), yaw_rel(Tm[:3, :3], R0), np.linalg.norm(Tc[:3, 3] - tp), np.linalg.norm(Tm[:3, 3] - tp),
            np.arccos(np.clip((np.trace(Tc[:3, :3] @ q2R(tq).T) - 1) / 2, -1, 1))), jm[0].round(2), "stick yaw %.3f" % (2*np.arctan2(q[e, 36], q[e, 33])))
