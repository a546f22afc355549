"""Invariants that pin the CPU oracle's physics restatement (MuJoCo itself is not available here, so
these and the independent numpy routines in av_aloha_amd/compiler/refdyn.py are the pins; DESIGN.md
states 'parity unpinned' at the MuJoCo boundary)."""
import ctypes as C

import numpy as np
import pytest

from av_aloha_amd.compiler import refdyn
from av_aloha_amd.compiler.compile import read_blob
from av_aloha_amd.compiler.mjcf import quat_mul
from orc_env import OrcEnv
from orc_ffi import ROOT, dp

OBJ = np.array([[0, 0.12, 0, 1, 0, 0, 0], [0, -0.05, 0, 1, 0, 0, 0.0]])


def model_dict(task="slot_insertion", na=3):
    md = read_blob(f"{ROOT}/models/{task}_{na}arms.avm")
    for k in ("nbody", "nv", "njnt", "nq", "nu"):
        md[k] = int(md[k].reshape(-1)[0])
    return md


def home_action(md):
    h = md["qpos_home"]
    return np.concatenate([h[:6], [1.0], h[8:14], [1.0], h[16:23]])


def rand_state(md, rng):
    q = md["qpos_home"].copy()
    q[:23] += rng.normal(scale=0.3, size=23)
    for a in (23, 30):
        q[a:a + 3] = rng.uniform(-0.3, 0.3, 3) + [0, 0, 0.5]
        qq = rng.normal(size=4)
        q[a + 3:a + 7] = qq / np.linalg.norm(qq)
    # keep fingers inside their range
    for k in (6, 7, 14, 15):
        q[k] = np.clip(q[k], 0.005, 0.035)
    return q


def test_mass_matrix_and_kinematics_vs_numpy():
    md = model_dict()
    e = OrcEnv()
    rng = np.random.default_rng(1)
    for _ in range(10):
        q = rand_state(md, rng)
        e.qpos[:] = q
        e.L.orc_kinematics(e.dptr)
        e.L.orc_crb(e.dptr)
        kin = refdyn.kinematics(md, q)
        np.testing.assert_allclose(e.arr("xpos", 31 * 3).reshape(31, 3), kin["xpos"], atol=1e-13)
        np.testing.assert_allclose(e.arr("cdof", 35 * 6).reshape(35, 6), kin["cdof"], atol=1e-13)
        M = e.arr("M", 35 * 35).reshape(35, 35)
        np.testing.assert_allclose(M, refdyn.mass_matrix(md, kin), atol=1e-13)
        assert np.all(np.linalg.eigvalsh(M) > 0)
    e.close()


def integrate_q(md, q, v, eps):
    q2 = q.copy()
    q2[:23] += eps * v[:23]
    for qa, da in ((23, 23), (30, 29)):
        q2[qa:qa + 3] += eps * v[da:da + 3]
        w = v[da + 3:da + 6]
        ang = eps * np.linalg.norm(w)
        if ang != 0:
            ax = w / np.linalg.norm(w)
            dq = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
            q2[qa + 3:qa + 7] = quat_mul(q[qa + 3:qa + 7], dq)
    return q2


def potential(md, q):
    kin = refdyn.kinematics(md, q)
    pe = 0.0
    for b in range(md["nbody"]):
        c = kin["xpos"][b] + kin["xmat"][b] @ md["body_ipos"][b]
        pe += md["body_mass"][b] * 9.81 * c[2]
    return pe


def test_rne_bias_gravity_and_coriolis():
    """gravity part: bias(q,0) = dPE/dq (finite differences); velocity part: qd.c(q,qd) = 1/2 qd^T Mdot qd."""
    md = model_dict()
    e = OrcEnv()
    rng = np.random.default_rng(2)
    for _ in range(5):
        q = rand_state(md, rng)
        v = rng.normal(scale=1.0, size=35)
        e.qpos[:] = q
        e.qvel[:] = 0
        e.L.orc_kinematics(e.dptr)
        e.L.orc_rne_bias(e.dptr)
        g = e.arr("qfrc_bias", 35).copy()
        eps = 1e-6
        for k in range(35):
            dv = np.zeros(35)
            dv[k] = 1
            fd = (potential(md, integrate_q(md, q, dv, eps)) - potential(md, integrate_q(md, q, dv, -eps))) / (2 * eps)
            assert abs(fd - g[k]) < 1e-6 * max(1, abs(g[k])), (k, fd, g[k])
        e.qvel[:] = v
        e.L.orc_rne_bias(e.dptr)
        c = e.arr("qfrc_bias", 35).copy() - g

        def Mat(qq):
            return refdyn.mass_matrix(md, refdyn.kinematics(md, qq))
        Md = (Mat(integrate_q(md, q, v, eps)) - Mat(integrate_q(md, q, v, -eps))) / (2 * eps)
        lhs, rhs = v @ c, 0.5 * v @ Md @ v
        assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(rhs)), (lhs, rhs)
    e.close()


def test_free_fall_matches_semi_implicit_euler():
    e = OrcEnv()
    obj = OBJ.copy()
    obj[1, 2] = 0.8      # stick well above the table
    e.reset(obj)
    z0, h, g = 0.8, 0.002, 9.81
    n = 40
    e.step(n)
    want = z0 - g * h * h * n * (n + 1) / 2
    assert abs(e.qpos[30 + 2] - want) < 1e-12
    assert np.abs(e.qpos[30:32] - obj[1, :2]).max() < 1e-14
    e.close()


def test_objects_rest_on_table_and_arms_hold_pose():
    md = model_dict()
    e = OrcEnv()
    e.reset(OBJ)
    a = home_action(md)
    for _ in range(25):
        ap, r, s = e.env_step(a)
    names = [c[:2] for c in e.contacts()]
    assert ("table", "stick") in names and ("table", "slot-1") in names and ("table", "slot-2") in names
    assert r == 0 and not s and e.d.overflow == 0
    # table top is at z=-0.0009 (scene.xml:55); bodies sink by a few 1e-5 m into the soft contact
    assert -0.0011 < e.qpos[25] < -0.0009 and -0.0011 < e.qpos[32] < -0.0009
    p0 = e.qpos.copy()
    for _ in range(25):
        e.env_step(a)
    # 20 PGS sweeps are not converged and the noslip pass stops at MuJoCo's noslip_tolerance: the stick creeps by < 0.1 mm/s
    # (with the Newton solver the resting objects do not move at all)
    assert np.abs(e.qpos[23:26] - p0[23:26]).max() < 2e-5 and np.abs(e.qpos[30:33] - p0[30:33]).max() < 1e-4
    assert np.abs(e.qvel).max() < 1e-4
    # static sag: actuator torque balances gravity where there is no dry friction / limit
    e.L.orc_forward(e.dptr)
    bias, act = e.arr("qfrc_bias", 35), e.arr("qfrc_actuator", 35)
    for k in (0, 3, 4, 5, 8, 11, 12, 13, 16, 19, 20, 21, 22):
        assert abs(bias[k] - act[k]) < 5e-3, (k, bias[k], act[k])
    # finger coupling (aloha_sim.xml:376-379): equality residual stays tiny
    assert abs(e.qpos[6] - e.qpos[7]) < 1e-4 and abs(e.qpos[14] - e.qpos[15]) < 1e-4
    e.close()


def test_gripper_closes_and_coupling_holds():
    md = model_dict()
    e = OrcEnv()
    e.reset(OBJ)
    a = home_action(md)
    a[6] = 0.0
    a[13] = 0.0
    for _ in range(30):
        ap, r, s = e.env_step(a)
    # ctrl 0 -> 0.002 m (env.py:210); the r=0.6 mm pad spheres of opposing fingers meet at ~8 mm opening
    # (aloha_sim.xml:181-183, 194-196); the driven finger loads the soft coupling with ~12 N
    assert ap[6] < 0.2 and ap[13] < 0.2
    names = [c[:2] for c in e.contacts()]
    assert ("left_left_g0", "left_right_g0") in names
    assert abs(e.qpos[6] - e.qpos[7]) < 1.5e-3
    e.close()


def test_determinism():
    md = model_dict()
    outs = []
    for _ in range(2):
        e = OrcEnv()
        e.reset(OBJ)
        a = home_action(md)
        a[0] = 0.3
        for _ in range(10):
            e.env_step(a)
        outs.append(np.concatenate([e.qpos.copy(), e.qvel.copy()]))
        e.close()
    assert np.array_equal(outs[0], outs[1])


def test_newton_solver_reaches_the_pgs_fixed_point():
    """Both solvers minimise the same convex problem (primal Newton / dual PGS): from identical contact-rich states the
    Newton acceleration must equal the one of PGS run to convergence; 20 sweeps (the north_star setting) are not converged."""
    md = model_dict()
    e = OrcEnv()
    e.d.solver = 1
    e.reset(OBJ)
    a = home_action(md)
    a[6] = 0.0
    a[0] += 0.2
    for _ in range(8):
        e.env_step(a)
    assert e.d.ncon >= 8
    accs = {}
    for name, solver, sweeps in (("newton", 1, 0), ("pgs20", 0, 20), ("pgs_conv", 0, 20000)):
        e.d.solver = solver
        e.d.pgs_iters = max(1, sweeps)
        e.d.pgs_tol = 0.0
        e.L.orc_forward(e.dptr)
        accs[name] = e.arr("qacc", 35).copy()
        if name == "newton":
            assert 1 <= e.d.stat_sweeps <= 8          # Newton iterations used
    scale = np.abs(accs["pgs_conv"]).max()
    assert np.abs(accs["newton"] - accs["pgs_conv"]).max() < 1e-6 * max(1.0, scale)
    assert np.abs(accs["pgs20"] - accs["pgs_conv"]).max() > np.abs(accs["newton"] - accs["pgs_conv"]).max()
    e.close()
