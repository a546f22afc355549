"""Experiment: which InsertPeg script parameters make the recorded episodes pass the reference's open-loop data-set check most often."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from av_aloha_amd import harness
for kw in ({}, {"side": 0.02, "depth": 0.035}, {"depth": 0.035}, {"pitch": 1.0}, {"side": 0.02, "depth": 0.035, "pitch": 1.0}, {"gain": 0.08}, {"carry": 0.08}):
    eps = harness.record_scripted("sim_insert_peg", 64, seed=7, **kw)
    ok, _ = harness.check_dataset_reward("gym_guided_vision/InsertPeg-3Arms-v0", [e["data"] for e in eps])
    print(kw, "success", sum(e["success"] for e in eps), "open-loop", int(ok.sum()), flush=True)
