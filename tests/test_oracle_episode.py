"""The scripted policies solve their tasks on the CPU oracle too (closed loop on the oracle's own state, its own GradIK / DiffIK):
the success flags the GPU episode tests compare are those of real episodes on both sides (tests/test_gpu_episode_parity.py)."""
import numpy as np

import episode_util as U
from av_aloha_amd import workloads as W


def test_oracle_scripted_slot_insertion_succeeds():
    task, n = "slot_insertion", 4
    home = U.oracle_home(task)
    poses = W.object_poses(task, np.arange(n), U.TASK_SEED[task])
    res = U.pool_map(U.closed_loop_worker, [(task, poses[k], home) for k in range(n)], 4)
    for rw, su, q, cs in res:
        assert rw.max() == 4 and su[-1] and rw[-1] == 4          # env.py:584-587 pins touch, :224 is_success
        assert np.all(np.diff(np.maximum.accumulate(rw)) >= 0)
    # and an open-loop replay of the recorded ctrl sequence is the same episode bit for bit (determinism of the oracle)
    rw2, su2, qs, nc = U.replay_worker((task, poses[0], res[0][3]))
    assert np.array_equal(rw2, res[0][0]) and np.array_equal(qs[-1], res[0][2])


def test_oracle_scripted_needle_lift():
    task, n = "sew_needle", 2
    home = U.oracle_home(task)
    poses = W.object_poses(task, np.arange(n), U.TASK_SEED[task])
    res = U.pool_map(U.closed_loop_worker, [(task, poses[k], home) for k in range(n)], 2)
    for k, (rw, su, q, cs) in enumerate(res):
        assert rw[-1] >= 2 and q[32] - poses[k, 1, 2] > 0.08      # held by the gripper, off the table (env.py:666-671)
