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


def _solves(task, n, max_reward):
    home = U.oracle_home(task)
    poses = W.object_poses(U.MODEL_OF.get(task, task), np.arange(n), U.TASK_SEED[task])
    res = U.pool_map(U.closed_loop_worker, [(task, poses[k], home) for k in range(n)], n)
    for rw, su, q, cs in res:
        assert rw.max() == max_reward and su[-1] and rw[-1] == max_reward, (task, rw.max(), rw[-1])
    return res


def test_oracle_scripted_insert_peg_succeeds():
    """env.py:453-462: both grasped and lifted (2), the peg in the tube touches the pin (4 = success)."""
    for rw, su, q, cs in _solves("insert_peg", 2, 4):
        assert (rw == 2).sum() > 50


def test_oracle_scripted_sew_needle_threads_and_hands_over():
    """env.py:676-689: grasp (1, 2), the threading latch (4, env.py:673), the left gripper alone holds the threaded needle clear of the
    table and of pin-wall (5 = success) -- the stages the lift of BASELINE config 3 never reaches."""
    for rw, su, q, cs in _solves("sew_needle_thread", 2, 5):
        first = {v: int(np.argmax(rw == v)) for v in (1, 2, 4, 5)}
        assert first[1] < first[2] < first[4] < first[5]
        assert q[30] < q[23] - 0.05          # the needle ends up beyond the wall (qpos[23:26] wall, [30:33] needle)


def test_oracle_scripted_hook_package_succeeds():
    """env.py:851-862: both hands on the package (1), lifted (2), on the hook with the pins meeting (4 = success), still 4 after both let go."""
    for rw, su, q, cs in _solves("hook_package", 2, 4):
        assert (rw[-30:] == 4).all() and q[31] > 0.22          # hanging on the hook near the wall, hands gone


def test_oracle_scripted_tube_transfer_succeeds():
    """env.py:771-778: both tubes grasped (1) and lifted (2), the ball poured from tube1 into tube2 meets the pin (3 = success)."""
    for rw, su, q, cs in _solves("tube_transfer", 2, 3):
        assert (rw == 2).sum() > 100
