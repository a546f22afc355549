"""Experiment: throughput vs waves (envs) per block when the LDS record is made small enough (home-pose scene)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim import BatchedSim
from test_oracle_physics import OBJ, home_action, model_dict
N = 8192
md = model_dict()
a = np.repeat(home_action(md)[None], N, 0)
for efc, con, wpb in ((176, 48, 4), (96, 24, 4), (96, 24, 6), (88, 16, 6)):
    sim = BatchedSim("slot_insertion", 3, N, options={"export_contacts": 0, "maxefc": efc, "maxcon": con, "waves_per_block": wpb})
    sim.reset(np.repeat(OBJ[None], N, 0))
    for _ in range(2):
        sim.step(a)
    t0 = time.perf_counter()
    for _ in range(5):
        sim.step(a)
    dt = (time.perf_counter() - t0) / 5
    d = sim.diag()
    print(flush=True); print(f"maxefc {efc} maxcon {con} wpb {wpb}: {dt * 1e3:.1f} ms/step -> {N / dt:.0f} env-steps/s (host-pointer mode), overflow {int((d[:, 2] != 0).sum())}, dims {sim.h.dims if hasattr(sim.h, 'dims') else ''}")
    sim.close()
