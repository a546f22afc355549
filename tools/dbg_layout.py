"""Prints the LDS layout of every task (AVSIM_DEBUG_LAYOUT) and the envs per block the launch picks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AVSIM_DEBUG_LAYOUT"] = "1"
from av_aloha_amd.sim import BatchedSim, TASK_KEYS
for task in TASK_KEYS:
    for f64 in (False, True):
        s = BatchedSim(task, 3, 8, f64=f64)
        print(task, "f64" if f64 else "f32", "lds_bytes(1 env + tables)", s.h.lds_bytes, "maxcon", s.maxcon, "maxefc", s.maxefc, flush=True)
        s.close()
