"""Device reward predicates (the ones the step kernel applies to its contacts) against truth tables produced by the
reference's own get_reward methods (env.py:425-863) on random contact sets (tests/golden/gen_golden.py).
Integer path: bit-exact, through the C-ABI (avsim_reward_from_pairs)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
TASKS = ["insert_peg", "slot_insertion", "sew_needle", "tube_transfer", "hook_package"]


@pytest.mark.parametrize("f64", [False, True])
@pytest.mark.parametrize("task", TASKS)
def test_device_reward_tables(task, f64):
    from av_aloha_amd.sim import BatchedSim
    r = np.load(os.path.join(G, "reward_tables.npz"))
    pairs, want = r[f"{task}_pairs"], r[f"{task}_reward"]
    S, Ln, Cn, _ = pairs.shape
    sim = BatchedSim(task, 3, 1, f64=f64)
    try:
        latch = np.zeros(S, dtype=np.int32)      # the SewNeedle latch is reset with the episode (env.py:631)
        for l in range(Ln):                      # sequences in parallel, their steps in order (the latch carries over)
            got = sim.reward_from_pairs(pairs[:, l], latch)
            assert np.array_equal(got, want[:, l]), (task, l, np.nonzero(got != want[:, l])[0][:8])
        # without a latch buffer every list is judged on its own
        got = sim.reward_from_pairs(pairs[:, 0])
        assert np.array_equal(got, want[:, 0])
        # empty lists and lists with only empty slots: reward 0
        assert np.array_equal(sim.reward_from_pairs(np.zeros((3, 0, 2), np.int32)), np.zeros(3, np.int32))
        assert np.array_equal(sim.reward_from_pairs(np.full((3, 4, 2), -1, np.int32)), np.zeros(3, np.int32))
    finally:
        sim.close()
