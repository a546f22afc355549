"""The Cartesian (data-collection) path of a scripted recording without images, a few envs: wall time per step; run under
rocprofv3 --kernel-trace --stats for the kernels.  python tools/prof_cart_step.py [episodes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from av_aloha_amd import harness
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
t = time.time()
eps = harness.record_scripted("sim_slot_insertion", n, cameras=[], seed=0)
T = eps[0]["data"]["/action"].shape[0]
print(f"{n} episodes side by side, {T} steps: {(time.time() - t) / T * 1e3:.2f} ms per step (handle creation included)")
