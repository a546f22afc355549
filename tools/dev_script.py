"""Development aid: run a scripted policy closed loop on the CPU oracle for a few seeds (one process per env) and print the reward
time line.  usage: python tools/dev_script.py <task> [n] [seed0] [key=value ...]  (test infrastructure; the oracle is the checker)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import episode_util as U
from av_aloha_amd import workloads as W
from av_aloha_amd.build import build_oracle

if __name__ == "__main__":
    build_oracle()
    task = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else U.TASK_SEED.get(task, 4000)
    U.SCRIPT_KW = {a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[4:]}
    home = U.oracle_home(task)
    poses = W.object_poses(U.MODEL_OF.get(task, task), np.arange(n), seed0)
    t0 = time.time()
    res = U.pool_map(U.closed_loop_worker, [(task, poses[k], home) for k in range(n)])
    print(f"{task}: {n} envs, {time.time() - t0:.0f} s")
    for k, (rw, su, q, cs) in enumerate(res):
        ch = np.nonzero(np.diff(rw, prepend=0))[0]
        print(f"env {k}: final {rw[-1]} max {rw.max()} success {bool(su[-1])}  changes " + " ".join(f"{t}:{rw[t]}" for t in ch[:40]))
        print("     obj", np.round(q[23:], 3))
