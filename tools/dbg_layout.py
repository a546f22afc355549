"""Prints the LDS layout figures of a task's handle (AVSIM_DEBUG_LAYOUT): python tools/dbg_layout.py [task] [arms] [f64]"""
import os, sys
os.environ["AVSIM_DEBUG_LAYOUT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from av_aloha_amd.sim import BatchedSim
task = sys.argv[1] if len(sys.argv) > 1 else "slot_insertion"
arms = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f64 = len(sys.argv) > 3 and sys.argv[3] == "1"
s = BatchedSim(task, arms, 64, f64=f64)
print(task, arms, "f64" if f64 else "f32", {k: getattr(s, k) for k in ("nq", "nv", "nu", "maxcon", "maxefc")})
