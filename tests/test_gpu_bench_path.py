"""Parity of the paths bench.py times and of the BASELINE.json configurations that had no oracle test:

* configs[1]: avsim_step_cartesian(AVSIM_IK_DLS) in the f32 PRODUCT mode at the bench's own size (4096
  SlotInsertion-3Arms envs, av_aloha_amd/workloads.py inputs) against orc_cart_to_ctrl(mode 1) + orc_env_step;
* configs[3] (config 4): HookPackage-2Arms with the 14-D joint-space random walk, f64 and f32, against the oracle;
* configs[0] (config 1): gym_guided_vision/InsertPeg-2Arms-v0, one env, 300 steps of the home action with the grippers closing
  at step 50: API shapes / dtypes of env.py:203-226, two runs bit-identical, trajectory against the oracle.
Stated f32 tolerances (product mode vs the f64 oracle): ctrl 1e-6 rad, joint angles 2e-5 rad / m over 25 env-steps."""
import numpy as np
import pytest

from av_aloha_amd import workloads as W
from orc_env import OrcEnv
from orc_ffi import dp
from test_oracle_physics import model_dict

pytestmark = pytest.mark.gpu


def _home_poses(sim, md):
    ch = np.asarray(md["ctrl_home"], dtype=np.float64)
    Ts = []
    for arm, sl in ((0, slice(0, 6)), (1, slice(7, 13)), (2, slice(14, 21))):
        q = np.ascontiguousarray(ch[sl])[None]
        T = np.empty((1, 16))
        sim.h.check(sim.h.L.avsim_fk_jac(sim.h.h, arm, 1, q.ctypes.data, T.ctypes.data, None))
        Ts.append(T)
    return W.home_poses(Ts)


def test_config2_benchmarked_dls_f32_path_vs_oracle_at_bench_size():
    from av_aloha_amd import _ffi
    from av_aloha_amd.sim import BatchedSim
    cfg = W.CONFIGS[2]
    N, T = 4096, 25
    ids = np.arange(N)
    md = model_dict(cfg["task"], 3)
    sim = BatchedSim(cfg["task"], 3, N, options={"solver": 1, "export_contacts": 0})       # f32 product mode, as bench.py
    poses = W.object_poses(cfg["task"], ids, cfg["seed"])
    sim.reset(poses)
    home = _home_poses(sim, md)
    # SURVEY App. A known answer: FK(home) of the left control site
    np.testing.assert_allclose(home["left"][:3], [-0.196896, 0.032, 0.200362], atol=1e-6)
    track = (0, 2047, 4095)
    orcs, shadows = [], []
    for k in track:
        e = OrcEnv(cfg["task"], 3)
        e.d.solver = 1
        e.reset(poses[k])
        orcs.append(e)
        s = OrcEnv(cfg["task"], 3)          # IK-only twin: gets the DEVICE's measured joints, so that the controller is compared
        s.reset(poses[k])                   # free of the f32 drift of the physics
        shadows.append(s)
    a21 = np.zeros(21)
    lo, span = md["grip_range"][0], md["grip_range"][1] - md["grip_range"][0]
    over = np.zeros(N, dtype=bool)
    nan = np.zeros(N, dtype=bool)
    worst_q = worst_c = 0.0
    for t in range(T):
        a = W.sinusoid_actions(home, ids, N, t)
        q_before = sim.get_state()[0]
        ap, rw, su = sim.step_cartesian(a, _ffi.IK_DLS)
        q, v, c, _ = sim.get_state()
        d = sim.diag()
        over |= d[:, 2] != 0
        nan |= (d[:, 3] & 1) != 0
        for k, e, s in zip(track, orcs, shadows):
            # controller on identical inputs: DLS IK of the oracle on the device's measured joints -> ctrl within f32 rounding
            s.qpos[:] = q_before[k]
            s.L.orc_cart_to_ctrl(s.dptr, dp(np.ascontiguousarray(a[k])), 1, dp(a21))
            want = a21.copy()
            want[[6, 13]] = lo + span * want[[6, 13]]
            worst_c = max(worst_c, np.abs(c[k] - want).max())
            np.testing.assert_allclose(c[k], want, atol=1e-6, err_msg=f"ctrl env {k} step {t}")
            # the whole composite, oracle on its own f64 state
            e.L.orc_cart_to_ctrl(e.dptr, dp(np.ascontiguousarray(a[k])), 1, dp(a21))
            apo, ro, so = e.env_step(a21)
            worst_q = max(worst_q, np.abs(q[k] - e.qpos).max())
            np.testing.assert_allclose(q[k], e.qpos, atol=2e-5, err_msg=f"qpos env {k} step {t}")
            np.testing.assert_allclose(ap[k], apo, atol=1e-3)          # grippers are normalised by 0.035 m: 2e-5 m -> 6e-4
            assert rw[k] == ro and bool(su[k]) == so
    assert not over.any(), f"row / contact caps overflowed in {over.sum()} envs"
    assert not nan.any(), f"{nan.sum()} envs diverged"
    assert np.isfinite(q).all() and np.isfinite(v).all()
    print(f"config 2 f32 DLS path: max |ctrl - oracle| {worst_c:.2e}, max |qpos - oracle| {worst_q:.2e} over {T} env-steps")
    for e in orcs + shadows:
        e.close()
    sim.close()


@pytest.mark.parametrize("f64", [True, False])
def test_config4_hook_package_2arms_vs_oracle(f64):
    from av_aloha_amd.sim import BatchedSim
    cfg = W.CONFIGS[4]
    n, T = 8, 12
    ids = np.arange(n)
    md = model_dict(cfg["task"], 2)
    sim = BatchedSim(cfg["task"], 2, n, f64=f64, options={"solver": 1})
    assert sim.nj == 14
    poses = W.object_poses(cfg["task"], ids, cfg["seed"])
    sim.reset(poses)
    acts = W.walk_actions(md["qpos_home"], md["act_ctrlrange"], ids, T, 14, cfg["seed"])
    orcs = []
    for k in range(3):
        e = OrcEnv(cfg["task"], 2)
        e.d.solver = 1
        e.reset(poses[k])
        orcs.append(e)
    tol_q, tol_v = (1e-10, 1e-8) if f64 else (2e-5, 1e-3)
    for t in range(T):
        ap, rw, su = sim.step(acts[t])
        q, v, _, _ = sim.get_state()
        assert ap.shape == (n, 14)
        for k, e in enumerate(orcs):
            apo, ro, so = e.env_step(acts[t, k].astype(np.float64))
            np.testing.assert_allclose(q[k], e.qpos, atol=tol_q, err_msg=f"f64={f64} qpos env {k} step {t}")
            np.testing.assert_allclose(v[k], e.qvel, atol=tol_v, err_msg=f"f64={f64} qvel env {k} step {t}")
            np.testing.assert_allclose(ap[k], apo, atol=1e-7 if f64 else 1e-3)
            assert rw[k] == ro and bool(su[k]) == so
    d = sim.diag()
    assert (d[:, 2] == 0).all() and ((d[:, 3] & 1) == 0).all()
    for e in orcs:
        e.close()
    sim.close()


def _config1_actions():
    from av_aloha_amd.constants import LEFT_ARM_POSE, RIGHT_ARM_POSE
    a = np.concatenate([LEFT_ARM_POSE[:6], [1.0], RIGHT_ARM_POSE[:6], [1.0]]).astype(np.float32)
    out = np.repeat(a[None], 300, 0)
    out[50:, [6, 13]] = 0.0                  # grippers toggled 1 -> 0 at step 50
    return out


def _config1_run(**kw):
    from av_aloha_amd.env import make
    env = make("gym_guided_vision/InsertPeg-2Arms-v0", cameras=[], **kw)
    np.random.seed(0)
    obs, info = env.reset()
    assert set(obs) == {"pixels", "agent_pos"} and obs["pixels"] == {}
    assert obs["agent_pos"].shape == (14,) and obs["agent_pos"].dtype == np.float64 and info == {"is_success": False}
    assert env.action_space.shape == (14,) and env.action_space.dtype == np.float32 and env.max_reward == 4 and env.num_arms == 2
    traj, rewards = [obs["agent_pos"].copy()], []
    for a in _config1_actions():
        obs, reward, terminated, truncated, info = env.step(a)
        assert isinstance(reward, int) and terminated is False and truncated is False and isinstance(info["is_success"], bool)
        assert obs["agent_pos"].shape == (14,) and obs["agent_pos"].dtype == np.float64
        assert info["is_success"] == (reward == env.max_reward)                                  # env.py:224
        traj.append(obs["agent_pos"].copy())
        rewards.append(reward)
    q = env.sim.get_state()[0][0].copy()
    env.close()
    return np.stack(traj), np.array(rewards), q


def test_config1_insert_peg_2arms_plumbing_determinism_and_oracle():
    from av_aloha_amd.env import sample_object_poses
    traj, rewards, q = _config1_run()
    traj2, rewards2, q2 = _config1_run()
    assert np.array_equal(traj, traj2) and np.array_equal(rewards, rewards2) and np.array_equal(q, q2)       # bit-identical
    np.random.seed(0)
    poses = sample_object_poses("insert_peg")
    e = OrcEnv("insert_peg", 2)
    e.d.solver = 1
    e.reset(poses)
    ref, rref = [], []
    for a in _config1_actions():
        ap, r, s = e.env_step(a.astype(np.float64))
        ref.append(ap)
        rref.append(r)
    ref = np.stack(ref)
    # f32 product mode vs the f64 oracle over 300 env-steps (6000 substeps): arms at rest under gravity and servo, grippers closing
    # on nothing; agent_pos within 1e-4 (the normalised grippers within 1e-3), objects within 1e-4 m, rewards exact
    err = np.abs(traj[1:] - ref)
    assert err[:, [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12]].max() < 1e-4, err.max(0)
    assert err[:, [6, 13]].max() < 1e-3
    assert np.array_equal(rewards, np.array(rref))
    assert np.abs(q[23:] - e.qpos[23:]).max() < 1e-4
    # the grippers close from the home opening until the finger hulls meet (normalised opening about 0.17)
    assert traj[40, 6] > 0.9 and traj[-1, 6] < 0.25 and traj[-1, 13] < 0.25
    e.close()
    # f64 device mode follows the oracle at rounding level
    trajd, rewardsd, qd = _config1_run(f64=True)
    assert np.abs(trajd[1:] - ref).max() < 1e-8 and np.array_equal(rewardsd, np.array(rref))
