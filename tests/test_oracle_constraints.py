"""A second, independent statement of MuJoCo's documented constraint model (numpy, av_aloha_amd/compiler/refdyn.py: rows, solref /
solimp -> reference acceleration and regulariser, elliptic cones with impratio, the primal objective) against the C oracle's stages
on random contact states, a generic convex solve (scipy) against the oracle's Newton solver, and closed-form friction scenes.
MuJoCo itself cannot run here (mujoco ^3.2.2, gym_guided_vision/pyproject.toml:11, is not importable): these narrow what a future
MuJoCo trajectory (tests/golden/gen_mujoco_traj.py) can still find; the physics stays "parity unpinned" until then."""
import ctypes as C

import numpy as np
import pytest
from scipy.optimize import minimize

from av_aloha_amd.compiler import refdyn
from orc_env import OrcEnv
from test_oracle_physics import OBJ, home_action, model_dict

GRAV_OFFSET = 56        # orc_model: 12 ints, double timestep, double gravity[3] (oracle/orc.h)


def contact_dicts(e):
    out = []
    for c in list(e.d.contact)[:e.d.ncon]:
        out.append(dict(dist=c.dist, pos=list(c.pos), frame=list(c.frame), geom1=c.geom1, geom2=c.geom2, dim=c.dim, friction=list(c.friction),
                        solref=list(c.solref), solimp=list(c.solimp), includemargin=c.includemargin, efc_adr=c.efc_adr))
    return out


def contact_states(n, seed=0, steps=(1, 12)):
    """Oracle states with the objects on the table / in the grippers' way and non-zero velocities: (env, md) after a few steps of
    wiggling arms with closing grippers."""
    md = model_dict()
    rng = np.random.default_rng(seed)
    for k in range(n):
        e = OrcEnv()
        e.d.solver = 1
        obj = OBJ.copy()
        obj[:, :2] += rng.uniform(-0.03, 0.03, (2, 2))
        obj[1, 2] = rng.choice([0.0, 0.0, 0.03])            # resting, or dropped from 3 cm
        e.reset(obj)
        a = home_action(md)
        a[:6] += rng.normal(scale=0.15, size=6)
        a[7:13] += rng.normal(scale=0.15, size=6)
        a[6], a[13] = rng.choice([0.0, 1.0]), rng.choice([0.0, 1.0])
        if k % 3 == 0:                                        # an arm pressed onto the table: arm-table contacts, joint limits
            a[1] += 0.9
            a[2] += 0.4
        for _ in range(int(rng.integers(*steps))):
            e.env_step(a)
        e.ctrl[:] = np.concatenate([a[:6], [0.002 + 0.035 * a[6]], a[7:13], [0.002 + 0.035 * a[13]], a[14:21]])
        if k % 3 == 1:                                        # a joint beyond its range: limit rows (right forearm roll, a finger)
            rg = md["jnt_range"].reshape(-1, 2)
            for j, over in ((11, 0.02), (6, 0.001)):
                e.qpos[md["jnt_qposadr"][j]] = rg[j, 1] + over
        yield e, md


def stage(e):
    L = e.L
    L.orc_kinematics(e.dptr); L.orc_crb(e.dptr); L.orc_collide(e.dptr); L.orc_rne_bias(e.dptr); L.orc_smooth(e.dptr); L.orc_make_constraints(e.dptr)


def test_constraint_rows_match_the_numpy_statement():
    seen_types, ncon_total = set(), 0
    for e, md in contact_states(12, seed=3):
        stage(e)
        n, nv = e.d.nefc, e.nv
        q, v = np.array(e.qpos), np.array(e.qvel)
        Cn = refdyn.constraints(md, q, v, contact_dicts(e))
        assert len(Cn["aref"]) == n
        J = np.ctypeslib.as_array(e.d.efc_J, shape=(n * nv,)).reshape(n, nv)
        np.testing.assert_allclose(Cn["J"], J, atol=1e-12)
        np.testing.assert_allclose(Cn["R"], np.array(e.d.efc_R[:n]), rtol=1e-12)
        np.testing.assert_allclose(Cn["aref"], np.array(e.d.efc_aref[:n]), rtol=1e-10, atol=1e-9)
        assert np.array_equal(Cn["type"], np.array(e.d.efc_type[:n]))
        for cc, c in zip(Cn["con"], contact_dicts(e)):
            assert (cc is None) == (c["efc_adr"] < 0) and (cc is None or cc[0] == c["efc_adr"])
        seen_types |= set(Cn["type"].tolist())
        ncon_total += sum(cc is not None for cc in Cn["con"])
        e.close()
    assert seen_types == {0, 1, 2, 3} and ncon_total > 100          # equalities, dry friction, joint limits and contacts all occurred


def test_newton_solution_minimises_the_primal_objective_a_generic_solver_agrees():
    """qacc of the oracle's Newton solver (before the noslip pass) against scipy's BFGS on the numpy statement of MuJoCo's primal
    cost built from the numpy rows: same minimiser to 1e-6 relative, zero gradient, and no lower cost found."""
    worst = 0.0
    for e, md in contact_states(10, seed=5):
        stage(e)
        nv = e.nv
        e.L.orc_solve_newton(e.dptr)
        a_newton = e.arr("qacc", nv).copy()
        M = e.arr("M", nv * nv).reshape(nv, nv).copy()
        a_s = e.arr("qacc_smooth", nv).copy()
        Cn = refdyn.constraints(md, np.array(e.qpos), np.array(e.qvel), contact_dicts(e))
        c_newton, g_newton, f = refdyn.primal_cost(Cn, M, a_s, a_newton, grad=True)
        scale = 1.0 / (float(md["opt"][7]) * nv)                       # MuJoCo's 1 / (meaninertia * nv) scaling of the termination tests
        assert np.linalg.norm(g_newton) * scale < 1e-7, np.linalg.norm(g_newton) * scale
        np.testing.assert_allclose(f, np.array(e.d.efc_force[:e.d.nefc]), rtol=1e-7, atol=1e-7 * max(1.0, np.abs(f).max()))
        # the analytic gradient is the gradient (central differences along random directions)
        rng = np.random.default_rng(1)
        for _ in range(3):
            dvec = rng.normal(size=nv)
            h = 1e-6
            fd = (refdyn.primal_cost(Cn, M, a_s, a_newton + h * dvec) - refdyn.primal_cost(Cn, M, a_s, a_newton - h * dvec)) / (2 * h)
            assert abs(fd - g_newton @ dvec) < 1e-4 * max(1.0, abs(fd), np.linalg.norm(g_newton))
        res = minimize(lambda a: refdyn.primal_cost(Cn, M, a_s, a, grad=True)[:2], a_s, jac=True, method="BFGS", options={"gtol": 1e-9 / scale * 1e-2, "maxiter": 4000})
        assert c_newton <= res.fun + 1e-9 * max(1.0, abs(res.fun))
        rel = np.linalg.norm(res.x - a_newton) / max(1.0, np.linalg.norm(a_newton))
        worst = max(worst, rel)
        assert rel < 1e-4, rel          # (the bound is BFGS's own accuracy on this ill-conditioned problem -- impratio 100 --; the zero gradient above is the sharper statement)
        e.close()
    print("largest relative distance BFGS - Newton:", worst)


def set_gravity(e, g):
    (C.c_double * 3).from_address(e.m.value + GRAV_OFFSET)[:] = list(g)


@pytest.mark.parametrize("deg,slides", [(35.0, False), (55.0, True)])
def test_stick_on_the_tilted_table_holds_below_the_friction_angle_and_slides_above(deg, slides):
    """Coulomb friction known answer: the stick lies on the table (pair friction 1.0, task_slot_insertion.xml / aloha_sim.xml
    defaults), gravity is tilted by theta about y instead of the table.  tan(theta) < mu: it stays (noslip_iterations = 3 removes
    the creep of the soft constraint); tan(theta) > mu: it accelerates down the slope with g (sin(theta) - mu cos(theta))."""
    e = OrcEnv()
    e.d.solver = 1
    th = np.deg2rad(deg)
    assert np.allclose((C.c_double * 3).from_address(e.m.value + GRAV_OFFSET)[:], [0, 0, -9.81])
    e.reset(OBJ)
    md = model_dict()
    a = home_action(md)
    for _ in range(5):
        e.env_step(a)                                   # settle
    x0 = e.qpos[30]
    set_gravity(e, [9.81 * np.sin(th), 0.0, -9.81 * np.cos(th)])
    try:
        T = 10                                          # 0.4 s
        for _ in range(T):
            e.env_step(a)
        dx, t = e.qpos[30] - x0, T * 0.04
        if slides:
            acc = 9.81 * (np.sin(th) - 1.0 * np.cos(th))
            assert abs(dx - 0.5 * acc * t * t) < 0.15 * 0.5 * acc * t * t, (dx, 0.5 * acc * t * t)
        else:
            assert abs(dx) < 2e-4, dx
    finally:
        set_gravity(e, [0, 0, -9.81])
        e.close()


def test_dry_joint_friction_holds_until_the_torque_exceeds_it():
    """frictionloss 2.0 N m on the shoulders (aloha_sim.xml:40): without gravity, a position-servo offset that asks for less than
    2 N m (kp 265) leaves the joint where it is, one that asks for more moves it."""
    md = model_dict()
    for offset, moves in ((0.8 * 2.0 / 265.0, False), (1.6 * 2.0 / 265.0, True)):
        e = OrcEnv()
        e.d.solver = 1
        obj = OBJ.copy()
        e.reset(obj)
        set_gravity(e, [0, 0, 0])
        try:
            q0 = np.array(e.qpos)
            ctrl = np.array(e.ctrl)
            for k, qa in enumerate(md["act_qposadr"]):
                ctrl[k] = q0[qa]                          # every servo at rest ...
            ctrl[1] = q0[1] + offset                      # ... but the left shoulder
            e.ctrl[:] = ctrl
            e.step(50)
            dq = e.qpos[1] - q0[1]
            if moves:
                assert dq > 0.2 * offset, dq
            else:
                assert abs(dq) < 1e-7, dq
        finally:
            set_gravity(e, [0, 0, -9.81])
            e.close()
