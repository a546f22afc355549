"""Experiment: script parameters against success rate and the reference's open-loop data-set check (256 episodes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from av_aloha_amd import harness
for task, gid, kws in (("sim_slot_insertion", "SlotInsertion", ({"gain": 0.08}, {"yaw_gain": 0.08}, {"side": 0.05}, {"side": 0.03}, {"drop": 0.045}, {"drop": 0.065})),
                       ("sim_insert_peg", "InsertPeg", ({},))):
    for kw in kws:
        eps = harness.record_scripted(task, 256, seed=7, **kw)
        ok, _ = harness.check_dataset_reward(f"gym_guided_vision/{gid}-3Arms-v0", [e["data"] for e in eps])
        print(task, kw, "success", sum(e["success"] for e in eps), "open-loop", int(ok.sum()), "of 256", np.bincount([int(e["rewards"][-1]) for e in eps], minlength=5).tolist(), flush=True)
