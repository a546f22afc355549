"""Behaviour of the gym facades that callers of the reference rely on beyond reset / step: what happens on a diverged state
(the reference's physics.step raises dm_control's PhysicsError, env.py:218) and hide_middle_arm / show_middle_arm (env.py:394-398)."""
import numpy as np
import pytest

from test_oracle_physics import model_dict

pytestmark = pytest.mark.gpu


def _home_action(env):
    from av_aloha_amd.constants import LEFT_ARM_POSE, MIDDLE_ARM_POSE, RIGHT_ARM_POSE
    a = np.concatenate([LEFT_ARM_POSE[:6], [1.0], RIGHT_ARM_POSE[:6], [1.0], MIDDLE_ARM_POSE]).astype(np.float32)
    return a[:env.num_joints]


def test_single_env_raises_physics_error_on_divergence():
    from av_aloha_amd.env import PhysicsError, make
    env = make("gym_guided_vision/SlotInsertion-3Arms-v0", cameras=[])
    np.random.seed(3)
    env.reset()
    a = _home_action(env)
    env.step(a)
    q, v, c, w = env.sim.get_state()
    v[0, 30] = 1e9                                   # the stick with an absurd spin
    env.sim.set_state(qvel=v)
    with pytest.raises(PhysicsError):
        env.step(a)
    obs, info = env.reset()                          # the env is usable again after reset()
    obs, r, term, trunc, info = env.step(a)
    assert np.isfinite(obs["agent_pos"]).all() and trunc is False
    env.close()


def test_batch_flags_the_diverged_env_and_truncates_it():
    from av_aloha_amd.env import make
    from av_aloha_amd.sim_env import make_sim_env
    env = make("gym_guided_vision/SlotInsertion-3Arms-v0", cameras=[], num_envs=4)
    np.random.seed(3)
    env.reset()
    a = np.repeat(_home_action(env)[None], 4, 0)
    obs, r, term, trunc, info = env.step(a)
    assert not trunc.any() and not info["diverged"].any() and term.shape == (4,) and not term.any()
    q, v, c, w = env.sim.get_state()
    v[2, 30] = 1e9
    env.sim.set_state(qvel=v)
    obs, r, term, trunc, info = env.step(a)
    assert trunc.tolist() == [False, False, True, False] and info["diverged"].tolist() == trunc.tolist()
    assert np.isfinite(obs["agent_pos"]).all()
    env.close()
    # the Cartesian-action env reports it the same way
    cenv = make_sim_env("sim_slot_insertion", cameras=[], num_envs=2)
    cenv.reset()
    o = cenv.get_obs()
    a23 = np.concatenate([o["poses"]["left"], np.zeros((2, 1)), o["poses"]["right"], np.zeros((2, 1)), o["poses"]["middle"]], axis=1)
    _, _, _, trunc, _ = cenv.step(a23)
    assert not np.any(trunc)
    q, v, c, w = cenv.sim.get_state()
    v[1, 30] = 1e9
    cenv.sim.set_state(qvel=v)
    _, _, _, trunc, _ = cenv.step(a23)
    assert trunc.tolist() == [False, True]
    cenv.close()


def test_hide_and_show_middle_arm_move_the_camera_arm_base():
    """env.py:394-398.  After hide_middle_arm a 3-arm env behaves like the 2-arm model (camera arm parked at (0, -2.4, -0.4))
    but keeps its 21-D action / agent_pos; show_middle_arm brings it back; the joint state carries over both ways."""
    from av_aloha_amd.env import make
    env = make("gym_guided_vision/SlotInsertion-3Arms-v0", cameras=["overhead_cam"], observation_height=60, observation_width=80)
    np.random.seed(4)
    obs, _ = env.reset()
    a = _home_action(env)
    obs, *_ = env.step(a)
    shown = env.sim.render_depth(["overhead_cam"], 60, 80)[0, 0]
    q0 = env.sim.get_state()[0].copy()
    env.hide_middle_arm()
    q1 = env.sim.get_state()[0]
    assert np.array_equal(q0, q1) and env.num_joints == 21 and env.sim.nj == 21
    hidden = env.sim.render_depth(["overhead_cam"], 60, 80)[0, 0]
    assert (np.abs(hidden - shown) > 0.02).mean() > 0.01, "the overhead camera must lose sight of the camera arm"
    # same thing as an env built with num_arms=2 from the same state, on the shared 14 joints
    env2 = make("gym_guided_vision/SlotInsertion-2Arms-v0", cameras=[])
    env2.reset()
    env2.sim.set_state(*env.sim.get_state())
    obs, r, *_ = env.step(a)
    assert obs["agent_pos"].shape == (21,) and obs["pixels"]["overhead_cam"].shape == (60, 80, 3)
    obs2, r2, *_ = env2.step(a[:14])
    assert obs2["agent_pos"].shape == (14,) and np.array_equal(obs2["agent_pos"], obs["agent_pos"][:14]) and r == r2
    assert np.array_equal(env.sim.get_state()[0], env2.sim.get_state()[0])
    env.hide_middle_arm()                            # idempotent
    env.show_middle_arm()
    back = env.sim.render_depth(["overhead_cam"], 60, 80)[0, 0]
    obs, *_ = env.step(a)
    assert obs["agent_pos"].shape == (21,)
    assert (np.abs(back - shown) > 0.02).mean() < 0.02          # the arm is back in view (one env-step later)
    env2.hide_middle_arm()                           # no-op on a 2-arm env
    env2.close()
    env.close()


def test_hide_middle_arm_keeps_the_sew_needle_latch():
    """The staged SewNeedle reward remembers that the needle was threaded (env.py:602, :673, :686-689).  hide_middle_arm only moves a
    body in the reference, so the memory must survive it here too (the env continues on the task's other compiled model)."""
    from av_aloha_amd.env import make
    env = make("gym_guided_vision/SewNeedle-3Arms-v0", cameras=[], num_envs=3)
    np.random.seed(5)
    env.reset()
    assert env.sim.get_latch().tolist() == [0, 0, 0]
    env.sim.set_latch(np.array([0, 1, 0], dtype=np.int32))
    fallback = env.sim.get_reset_poses()                        # where a diverged env's objects go back to: this episode's reset poses
    assert fallback.shape == (3, 2, 7) and np.allclose(fallback[:, :, :3], env.sim.get_state()[0][:, 23:].reshape(3, 2, 7)[:, :, :3], atol=0.02)
    assert np.ptp(fallback[:, 1, 0]) > 1e-4                     # (sampled per env, not the model's default poses)
    env.hide_middle_arm()
    assert env.sim.get_latch().tolist() == [0, 1, 0]
    assert np.array_equal(env.sim.get_reset_poses(), fallback)
    env.show_middle_arm()
    assert env.sim.get_latch().tolist() == [0, 1, 0]
    assert np.array_equal(env.sim.get_reset_poses(), fallback)
    env.reset()
    assert env.sim.get_latch().tolist() == [0, 0, 0]           # env.py:631: reset clears it
    env.close()
