"""Experiment: time of ONE round (one block per CU) vs waves per block."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim import BatchedSim
from test_oracle_physics import OBJ, home_action, model_dict
md = model_dict()
for wpb in (1, 2, 4, 6, 8):
    N = 256 * wpb
    a = np.repeat(home_action(md)[None], N, 0)
    sim = BatchedSim("slot_insertion", 3, N, options={"export_contacts": 0, "waves_per_block": wpb})
    sim.reset(np.repeat(OBJ[None], N, 0))
    for _ in range(3):
        sim.step(a)
    t0 = time.perf_counter()
    for _ in range(10):
        sim.step(a)
    dt = (time.perf_counter() - t0) / 10
    print(f"wpb {wpb}: N={N} {dt * 1e3:.2f} ms/step -> {N / dt:.0f} env-steps/s", flush=True)
    sim.close()
