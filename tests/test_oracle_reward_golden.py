"""Reward predicates: the oracle's class-bit restatement vs truth tables produced by the reference's
own get_reward methods (env.py) on random contact sets.  Integer path: bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

from orc_ffi import ip, lib, load_model

G = os.path.join(os.path.dirname(__file__), "golden")
TASKS = ["insert_peg", "slot_insertion", "sew_needle", "tube_transfer", "hook_package"]


@pytest.mark.parametrize("task", TASKS)
def test_reward_tables(task):
    L = lib()
    m = load_model(task, 3)
    r = np.load(os.path.join(G, "reward_tables.npz"))
    pairs, want = r[f"{task}_pairs"], r[f"{task}_reward"]
    S, Ln, Cn, _ = pairs.shape
    assert len(np.unique(want)) >= 4
    for s in range(S):
        latch = C.c_int(0)          # the SewNeedle latch is reset with the episode (env.py:631)
        for l in range(Ln):
            p = np.ascontiguousarray(pairs[s, l].astype(np.int32))
            got = L.orc_reward_from_pairs(m, ip(p), Cn, C.byref(latch))
            assert got == want[s, l], (task, s, l, got, want[s, l])
