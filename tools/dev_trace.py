"""Development aid: one env of a scripted policy on the CPU oracle with a per-step trace (object poses, hand poses, reward, contacts).
usage: python tools/dev_trace.py <task> [seed index] [every] [key=value ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import episode_util as U
from av_aloha_amd import workloads as W
from av_aloha_amd.build import build_oracle
from orc_ffi import dp

if __name__ == "__main__":
    build_oracle()
    task = sys.argv[1]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    U.SCRIPT_KW = {a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[4:]}
    home = U.oracle_home(task)
    pose = W.object_poses(U.MODEL_OF.get(task, task), np.array([k]), U.TASK_SEED[task])[0]
    e = U._new_env(task, pose)
    script = U.make_script(task, home, np.array(e.qpos)[None])
    a21 = np.zeros(21)
    lo, hi = U.GRIP_RANGE
    np.set_printoptions(precision=3, suppress=True, linewidth=200)
    for t in range(script.steps()):
        ph = script.phase()[0]
        a = np.ascontiguousarray(script.action(np.array(e.qpos)[None])[0])
        e.L.orc_cart_to_ctrl(e.dptr, dp(a), 0, dp(a21))
        c = a21.copy()
        for j in (6, 13):
            c[j] = a21[j] * (hi - lo) + lo
        rw, su = U._step_ctrl(e, c)
        if t % every == 0 or t == script.steps() - 1:
            Ts = []
            for arm, sl in ((0, slice(0, 6)), (1, slice(8, 14))):
                T = np.zeros(16); q = np.ascontiguousarray(np.array(e.qpos)[sl]); e.L.orc_fk(e.m, arm, dp(q), dp(T)); Ts.append(T.reshape(4, 4))
            print(f"t {t:3d} ph {ph} rw {rw} ncon {e.d.ncon:2d} | L cmd {a[0:3]} is {Ts[0][:3, 3]} | R cmd {a[8:11]} is {Ts[1][:3, 3]} | grip {np.array(e.qpos)[[6, 14]]}")
            print(f"       objs {np.array(e.qpos)[23:]}")
            if "-c" in os.environ.get("TRACE", ""):
                print("       ", [(a_, b_) for a_, b_, *_ in e.contacts() if not (a_ == "table" or b_ == "table") or True][:24])
