"""The device against the FAITHFUL oracle, under the driver (round 5 had this as a builder-run report only: tools/fidelity.py ->
profiles/r05_fidelity.json).

The reference collides the FULL convex hull of every collision mesh (gym_guided_vision/assets/aloha_sim.xml:106-111, class "collision":
mesh geoms, which MuJoCo collides as the hull of all their vertices [EXT]); the device keeps at most 128 vertices per hull behind
support tables (compiler/hull.py).  The oracle's faithful mode (tests/orc_ffi.py load_model(hulls="full"): models/oracle_full_hulls.avh,
18 032 vertices over 27 meshes) is what MuJoCo's geometry is; the parity tests elsewhere run the oracle on the device's hulls.  Here the
scripted episodes of all five tasks (grasp, carry, insert / thread / hang / pour: 350 - 536 env-steps) run closed loop on the device
(f64 physics) and the faithful oracle looks at them twice:
  * teacher-forced -- from the device's state at every env-step, the device's ctrl for one step: the contact COUNT may differ in at most
    1 % of the env-steps per task (observed 0 - 0.8 %, profiles/r05_fidelity.json; 17 % for HookPackage with the 20 / 32-vertex hulls of
    rounds 1-4), success flags in none, rewards in at most 0.5 %;
  * open-loop replay of the whole ctrl sequence -- the final success flag and the largest reward reached identical in EVERY env."""
import numpy as np
import pytest

import episode_util as U

pytestmark = pytest.mark.gpu

N_ENVS = 8
TASKS = ["slot_insertion", "insert_peg", "sew_needle_thread", "hook_package", "tube_transfer"]


@pytest.mark.parametrize("task", TASKS)
def test_device_against_the_full_hull_oracle(task):
    dev = U.device_episode(task, N_ENVS, f64=True, record_state=True)
    assert not dev["diverged"].any() and not dev["capped"].any()
    old = dict(U.ORACLE_MODE)
    U.ORACLE_MODE.update(hulls="full", boxbox_points=8)
    try:
        ls = U.compare_lockstep(task, dev)
        rows = U.compare_with_replay(task, dev)
    finally:
        U.ORACLE_MODE.update(old)
    steps = sum(r["steps"] for r in ls)
    ncon_diff = sum(r["ncon_diff_steps"] for r in ls)
    rew_diff = sum(r["reward_diff_steps"] for r in ls)
    succ_diff = sum(r["success_diff_steps"] for r in ls)
    print(f"{task}: faithful oracle, {N_ENVS} envs x {steps // N_ENVS} env-steps teacher-forced: ncon differs in {ncon_diff} ({ncon_diff / steps:.4f}), reward in {rew_diff}, "
          f"success flag in {succ_diff}; one-step |dq| max {max(r['max_step_err'] for r in ls):.2e}; open-loop replay: "
          f"{sum(r['dev_success'] != r['orc_success'] for r in rows)} final-flag, {sum(r['dev_max_reward'] != r['orc_max_reward'] for r in rows)} largest-reward mismatches")
    assert ncon_diff <= 0.01 * steps, (ncon_diff, steps)
    assert succ_diff == 0 and rew_diff <= 0.005 * steps, (succ_diff, rew_diff, steps)
    for r in rows:
        assert r["dev_success"] == r["orc_success"] and r["dev_max_reward"] == r["orc_max_reward"], r
    assert np.mean([r["dev_success"] for r in rows]) >= 0.75          # real episodes: they end at max_reward on the device
