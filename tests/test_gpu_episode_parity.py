"""Episode-length parity of the device against the CPU oracle, success flags included, for the four task families north_star names
(north_star: "task-success flags match bit-exact"; env.py:224 is_success; reward stages env.py:425-472 InsertPeg, :546-589
SlotInsertion, :640-690 SewNeedle, :820-863 HookPackage) and for the fifth task of the registry, TubeTransfer (:738-779).

The scripted policies (tests/scripted.py, av_aloha_amd/workloads.py) run closed loop on the device for whole episodes that REACH
max_reward: grasp - carry - insert (SlotInsertion, 350 env-steps = 7000 substeps), two pitched grasps and the peg into the tube
(InsertPeg, 350), grasp - thread through the wall's window - hand over to the left gripper (SewNeedle, all five stages, 535), two-arm
carry onto the hook and release (HookPackage, 410), two side grasps and the ball poured from one tube into the other (TubeTransfer,
515); plus BASELINE config 3's reach - grasp - lift (SewNeedle, 250).  The ctrl vector
the device's IK produced at every step is recorded and the oracle steps the SAME ctrl sequence from the same reset state
(tests/episode_util.py).

  * f64 device mode vs the oracle, open-loop replay of whole episodes: at MuJoCo's default Newton tolerance the reward of every step, the
    final is_success and the contact count of every step are identical in every env and positions stay within 1e-6 (1e-4 HookPackage)
    for the whole episode; with the tolerance at 1e-12 on both sides everything agrees to 1e-7 (observed 1e-9), TubeTransfer's ball
    included.  (Rounds 3-4 saw millimetres to centimetres in single envs and blamed chaos; it was a line search that cycled, see below.)
  * f32 product mode vs the f64 oracle, 128 seeds per task, two comparisons:
      - teacher-forced (`lockstep`): at every env-step the oracle is put into the device's f32 state and steps the device's ctrl once.
        Per-step reward and success flags without the divergence of two chaotic trajectories in between: a differing flag needs a
        contact within f32 rounding of its margin in that step.  Observed (profiles/r04_episode_parity.json): 0 differing success
        flags in 224 000 env-steps of SlotInsertion / InsertPeg / SewNeedle / TubeTransfer and 0 - 15 differing rewards per task
        (< 0.03 %; round 6: one differing flag-step in SewNeedle); HookPackage 23 flag-steps of 52 480 (0.04 %) while the released package is knocked about.
      - open-loop replay of the whole ctrl sequence: the final is_success per env.  The f64 replay of controls that were computed
        in closed loop on the f32 trajectory has no feedback: a millimetre of difference in how the object sits in the gripper, and
        the replayed peg meets the tube's rim (clearance 8 mm).  The mismatch count is stated and bounded per task
        (observed 0 / 0 / 3 / 0 / 0 of 128 for SlotInsertion / InsertPeg / SewNeedle / HookPackage / TubeTransfer in round 6; 0 / 1 / 1 / 0 / 5 in round 5).
"""
import numpy as np
import pytest

import episode_util as U

pytestmark = pytest.mark.gpu

# ---- f64 device mode against the oracle -------------------------------------------------------------------------------------------
# Round 5 found what rounds 3-4 had taken for chaos.  Both sides' exact line search was a Newton iteration on phi' that only fell back on
# bisection when a step LEFT the bracket; the elliptic cone's middle-zone cost is not quadratic, and on a sigmoid-shaped phi' that
# iteration cycles between the two ends of its bracket until the evaluation cap, ending on whichever end the cap's parity chooses
# (oracle/orc_newton.c has the trace).  Device (cap 40) and oracle (cap 50) then took different steps from identical states -- a
# millimetre in one env-step, centimetres by the end of the episode.  With the rtsafe safeguard on both sides the episodes agree as the
# tables below say; what is left at MuJoCo's default tolerance is the solver's own freedom (both sides stop within 1e-8 of the minimiser,
# each on its own side of it), and at a tolerance of 1e-12 both converge to the minimiser and whole episodes agree to 1e-9.
#
# (script, envs, max_reward the episode reaches, bound on the largest position difference over the WHOLE episode at MuJoCo's default Newton
# tolerance 1e-8 -- observed with 16 envs, profiles/r05_episode_parity.json: 6e-11, 3e-13, 2.5e-12, 1.6e-6, (tube: see below), 1.2e-4)
# (config 3's lift holds the needle in a gripper that GradIK steers: the secant descent amplifies a difference by 1.5 - 2 per iteration, 50 iterations a step --
# DESIGN.md 2 --, so the solver's 1e-8 becomes 1e-4 in single envs)
F64_CASES = [("slot_insertion", 8, 4, 1e-6), ("insert_peg", 8, 4, 1e-6), ("sew_needle_thread", 8, 5, 1e-6), ("hook_package", 8, 4, 1e-4), ("tube_transfer", 8, 3, None),
             ("sew_needle", 8, None, 1e-3)]


@pytest.mark.parametrize("task,n,max_reward,pos_tol", F64_CASES)
def test_f64_full_episode_rewards_and_success_identical(task, n, max_reward, pos_tol):
    """MuJoCo's default solver tolerance (the product's setting): the oracle replays the device's ctrl sequence open loop for the whole
    episode.  Reward of EVERY step, final is_success and the contact count of every step identical in every env, positions within
    pos_tol over the whole episode; TubeTransfer's 0.5 g ball (a billiard inside the carried tube, 2.3 cm of clearance) turns the
    solver's 1e-8 into another bounce in single envs AFTER the pour: identical up to the pour phase, flags and final rewards identical."""
    dev = U.device_episode(task, n, f64=True)
    assert not dev["diverged"].any() and not dev["capped"].any()
    rows = U.compare_with_replay(task, dev)
    differing = [r for r in rows if r["first_reward_diff"] != -1]
    for r in rows:
        assert r["dev_success"] == r["orc_success"], r
        assert r["dev_final_reward"] == r["orc_final_reward"], r
        assert r["held_reward_diff"] == 0, r                       # (the steps before the script lets go / pours)
    if task == "tube_transfer":
        # (round 6: bounds at three times the observed maxima -- n // 4 envs, 3e-3 and 2e-2 until then)
        assert len(differing) <= 1 and all(r["n_reward_diff"] <= 10 for r in differing), differing               # observed: 0 of 16 envs
        assert all(r["held_max_qpos_err"] < 1e-3 and r["arm_max_qpos_err"] < 6e-3 for r in rows), rows          # observed: 2.9e-4 up to the pour, 1.8e-3 for the arms over the whole episode
        assert np.median([r["max_qpos_err"] for r in rows]) < 1e-6                                              # observed: 4e-10
    else:
        assert not differing, f"{task}: reward sequences differ: {differing}"
        for r in rows:
            assert r["ncon_diff_steps"] == 0 and r["max_qpos_err"] < pos_tol, r
    if max_reward is not None:            # the episodes are real ones: they end at max_reward = success (env.py:224) on both sides
        assert sum(r["dev_success"] and r["dev_final_reward"] == max_reward for r in rows) >= n - 1, rows
        assert sum(r["orc_success"] for r in rows) >= n - 1, rows
    else:                                  # config 3's lift: the needle is held off the table by the gripper (reward 2, env.py:666-671)
        assert sum(r["dev_final_reward"] >= 2 for r in rows) >= n - 1, rows


@pytest.mark.parametrize("task,n", [(c[0], c[1]) for c in F64_CASES])
def test_f64_whole_episodes_agree_to_1e7_when_the_solver_converges(task, n):
    """The IMPLEMENTATION against the oracle: Newton tolerance 1e-12 on both sides (tests/episode_util.py NEWTON_TOL; MuJoCo's default 1e-8
    leaves each side within 1e-8 of the minimiser, on its own side of it).  Open-loop replay of whole episodes that grasp, carry, thread,
    hang and pour (250 - 536 env-steps = 5 000 - 10 720 substeps): the reward of every step, the success flag and the CONTACT COUNT of every
    step identical in every env, every joint and object position within 1e-7 for the whole episode (observed: 2e-13 ... 9e-10,
    TubeTransfer's ball included); teacher-forced, one env-step from the device's state: within 1e-8 (observed 1e-13 ... 3e-10)."""
    U.NEWTON_TOL = 1e-12
    try:
        dev = U.device_episode(task, n, f64=True, record_state=True)
        assert not dev["diverged"].any() and not dev["capped"].any()
        rows = U.compare_with_replay(task, dev)
        ls = U.compare_lockstep(task, dev)
    finally:
        U.NEWTON_TOL = None
    for r in rows:
        assert r["first_reward_diff"] == -1 and r["dev_success"] == r["orc_success"] and r["ncon_diff_steps"] == 0, r
        assert r["max_qpos_err"] < 1e-7, r
    for r in ls:
        assert r["reward_diff_steps"] == 0 and r["success_diff_steps"] == 0 and r["ncon_diff_steps"] == 0 and r["max_step_err"] < 1e-8, r
    assert np.mean([r["dev_success"] for r in rows]) >= (0.8 if task != "sew_needle" else 0.0)


# task: (open-loop replay: bound on final-flag mismatches of 128; lockstep: bounds on differing success-flag steps, differing reward
# steps (fraction of all env-steps), final-flag mismatches).  Round 6: the bounds are the maxima observed on the driver's kind of box (in the
# comments: replay mismatches; success-flag steps, reward-step fraction, final flags) times three, at least observed + 1 -- they were 2 / 4 / 4 / 2 / 10
# replay mismatches and 160 HookPackage flag-steps until round 5.
F32_CASES = {
    "slot_insertion":    dict(replay_mismatch=1, ls_success_steps=2, ls_reward_frac=5e-4, ls_final=0, min_success=0.9),     # 0; 0, 0, 0
    "insert_peg":        dict(replay_mismatch=1, ls_success_steps=2, ls_reward_frac=5e-4, ls_final=0, min_success=0.95),    # 0; 0, 0, 0
    "sew_needle_thread": dict(replay_mismatch=4, ls_success_steps=3, ls_reward_frac=7e-4, ls_final=0, min_success=0.95),    # 3; 1, 2.2e-4, 0
    "hook_package":      dict(replay_mismatch=1, ls_success_steps=70, ls_reward_frac=4e-3, ls_final=1, min_success=0.9),    # 0; 23, 1.3e-3, 0
    "tube_transfer":     dict(replay_mismatch=2, ls_success_steps=2, ls_reward_frac=5e-4, ls_final=0, min_success=0.93),    # 0 (5 in round 5: the ball's billiard in the carried tube); 0, 0, 0
}


@pytest.mark.parametrize("task", list(F32_CASES))
def test_f32_product_mode_success_flags_vs_f64_oracle_128_seeds(task):
    """The f32 product path's whole episodes against the f64 oracle, 128 seeds: per-step flags teacher-forced from the device's own
    states, and the final is_success of the open-loop replay (see the module docstring for what each can and cannot show)."""
    n, B = 128, F32_CASES[task]
    dev = U.device_episode(task, n, f64=False, record_state=True)
    assert dev["diverged"].mean() <= 0.01 and dev["capped"].mean() <= 0.01
    assert dev["success"][-1].mean() >= B["min_success"], f"{task}: the scripted episodes succeed on the device in {dev['success'][-1].mean():.2f} of the envs"
    ls = U.compare_lockstep(task, dev)
    steps = sum(r["steps"] for r in ls)
    succ_steps, rew_steps = sum(r["success_diff_steps"] for r in ls), sum(r["reward_diff_steps"] for r in ls)
    ls_final = sum(r["dev_success"] != r["orc_success"] for r in ls)
    rows = U.compare_with_replay(task, dev)
    mism = [r for r in rows if r["dev_success"] != r["orc_success"]]
    print(f"{task} f32 vs f64 oracle, {n} seeds: teacher-forced {succ_steps} success-flag / {rew_steps} reward differences in {steps} env-steps, {ls_final} final flags; "
          f"open-loop replay {len(mism)} final-flag mismatches; device success {np.mean([r['dev_success'] for r in rows]):.3f}, oracle replay {np.mean([r['orc_success'] for r in rows]):.3f}")
    assert succ_steps <= B["ls_success_steps"], (succ_steps, [r for r in ls if r["success_diff_steps"]][:8])
    assert rew_steps <= B["ls_reward_frac"] * steps, (rew_steps, steps)
    assert ls_final <= B["ls_final"], [r for r in ls if r["dev_success"] != r["orc_success"]]
    assert len(mism) <= B["replay_mismatch"], mism
    assert np.mean([r["orc_success"] for r in rows]) >= 0.85


@pytest.mark.parametrize("f64", [False, True])
def test_coupled_component_factorisation_is_bit_identical(f64):
    """Newton in a scene where a contact couples two kinematic trees (the needle in the gripper: right arm + needle, 14 of 35 dofs): the
    dense factorisation and the two substitutions over the coupled component only, the other trees inside their lane octets (the default),
    against the same over all nv columns (option newton_component = 0).  The entries between the component and the other trees are zeros
    that stay zeros and every other operation is the same one: whole closed-loop episodes (250 env-steps = 5000 substeps; GradIK amplifies a
    rounding-level difference to 1e-5 within a step) agree BIT FOR BIT -- ctrl, joint and object positions, rewards, contact counts."""
    a = U.device_episode("sew_needle", 8, f64=f64)
    b = U.device_episode("sew_needle", 8, f64=f64, options={"newton_component": 0})
    assert a["ncon"].max() >= 30 and (a["reward"].max(axis=0) >= 2).sum() >= 7          # the needle is grasped and lifted: coupled scenes
    for k in ("ctrl", "qpos", "reward", "ncon"):
        assert np.array_equal(a[k], b[k]), k
