"""The Cartesian-action env (av_aloha_amd/sim_env.py, counterpart of data_collection_scripts/sim_env.py:160-312) against the
oracle's composite of the same path: orc_cart_to_ctrl (GradIK / GradIK / DiffIK on the measured joints, gripper
unnorm(1 - trigger)) + 20 substeps, and the observation layout of get_obs (:205-218)."""
import ctypes as C

import numpy as np
import pytest

from orc_env import OrcEnv
from orc_ffi import dp
from test_oracle_physics import OBJ, model_dict

pytestmark = pytest.mark.gpu


def fk_pose_oracle(e, arm, q):
    from av_aloha_amd.sim_env import mat2quat_xyzw
    T = np.zeros(16)
    e.L.orc_fk(e.m, arm, dp(np.ascontiguousarray(q, dtype=np.float64)), dp(T))
    T = T.reshape(4, 4)
    qx = mat2quat_xyzw(T[:3, :3])
    return np.concatenate([T[:3, 3], qx[[3, 0, 1, 2]]])


def test_cartesian_step_and_obs_layout_match_the_oracle():
    from av_aloha_amd.sim_env import make_sim_env
    md = model_dict()
    env = make_sim_env("sim_slot_insertion", cameras=[], num_envs=2, f64=True, options={"solver": 1})
    np.random.seed(5)
    obs, info = env.reset()
    assert info == "Resetting arms..." and obs["joints"]["position"].shape == (2, 21) and obs["qpos"].shape == (2, 37)
    np.random.seed(5)
    from av_aloha_amd.env import sample_object_poses
    poses = [sample_object_poses("slot_insertion") for _ in range(2)]
    e = OrcEnv()
    e.d.solver = 1
    e.reset(poses[0])
    # targets: the home end-effector poses, nudged a little every step (kept well inside limit_pose's 0.1 m / 0.3 rad clamp)
    home = {k: obs["poses"][k][0].copy() for k in ("left", "right", "middle")}
    lo, span = md["grip_range"][0], md["grip_range"][1] - md["grip_range"][0]
    a21 = np.zeros(21)
    for t in range(3):
        a = np.concatenate([home["left"], [0.3 * t], home["right"], [1.0 - 0.3 * t], home["middle"]])
        a[0] += 0.01 * (t + 1)
        a[9] -= 0.008 * (t + 1)
        a[18] += 0.005 * (t + 1)
        obs, rew, term, trunc, info = env.step(np.repeat(a[None], 2, 0))
        assert rew == 0 and term is False and not np.any(trunc) and info == ""
        # the oracle runs its own IK on its own measured joints ...
        e.L.orc_cart_to_ctrl(e.dptr, dp(np.ascontiguousarray(a)), 0, dp(a21))
        dev21 = obs["control"][0].copy()                 # the device's command in action-21 form (grippers normalised)
        # GradIK (left / right arm) is a chaotic iteration (tests/test_gpu_ik.py bounds device vs oracle
        # at 1e-3 max, 2e-6 median after its 50 iterations), DiffIK (middle arm) and the gripper mapping are not
        np.testing.assert_allclose(dev21, a21, atol=2e-3)
        np.testing.assert_allclose(dev21[14:], a21[14:], atol=1e-7)
        assert abs(dev21[6] - a21[6]) < 1e-12 and abs(dev21[13] - a21[13]) < 1e-12
        # ... and is then stepped with the DEVICE's command, so that the physics and the observation layout are compared
        # free of the IK's chaos
        e.env_step(dev21)
        q, v, c = e.qpos, e.qvel, e.ctrl
        pos = q[md["obs_qposadr"]].copy(); vel = v[md["obs_dofadr"]].copy(); con = c.copy()
        for k in (6, 13):
            pos[k] = (pos[k] - lo) / span; vel[k] = vel[k] / span; con[k] = (con[k] - lo) / span
        np.testing.assert_allclose(obs["control"][0], con, atol=1e-12)
        np.testing.assert_allclose(obs["joints"]["position"][0], pos, atol=1e-7)
        np.testing.assert_allclose(obs["joints"]["velocity"][0], vel, atol=1e-5)
        np.testing.assert_allclose(obs["qpos"][0], q, atol=1e-7)
        # gripper convention (:300-301): trigger 0.3 t -> commanded opening 1 - trigger, normalised back by get_obs
        assert abs(obs["control"][0][6] - (1 - 0.3 * t)) < 1e-12 and abs(obs["control"][0][13] - 0.3 * t) < 1e-12
        for arm, name, sl in ((0, "left", slice(0, 6)), (1, "right", slice(7, 13)), (2, "middle", slice(14, 21))):
            np.testing.assert_allclose(obs["poses"][name][0], fk_pose_oracle(e, arm, c[sl]), atol=1e-9)
            assert obs["poses"][name].shape == (2, 7) and abs(np.linalg.norm(obs["poses"][name][0][3:]) - 1) < 1e-12
    obs, *_ = env.step_joints(np.repeat(np.concatenate([md["qpos_home"][:6], [1.0], md["qpos_home"][8:14], [1.0], md["qpos_home"][16:23]])[None], 2, 0))
    assert obs["images"] == {} and abs(obs["control"][0][6] - 1.0) < 1e-6
    env.close()
    e.close()
    with pytest.raises(NotImplementedError):
        make_sim_env("sim_unknown_task")
