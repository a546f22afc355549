"""Pin the CPU oracle's IK path to golden vectors produced by the reference's own Python
(tests/golden/gen_golden.py).  Tolerances: the reference rounds target quaternions to float32
(transform_utils.py:66), so rotation-dependent quantities agree to ~1e-6, everything else ~1e-12."""
import ctypes as C
import os

import numpy as np
import pytest

from orc_ffi import dp, lib, load_model

G = os.path.join(os.path.dirname(__file__), "golden")
ARMS = {"left": 0, "right": 1, "middle": 2}


def test_so3_helpers():
    L = lib()
    d = np.load(os.path.join(G, "so3_helpers.npz"))
    n = len(d["q_xyzw"])
    R = np.zeros(9)
    q = np.zeros(4)
    v = np.zeros(3)
    for i in range(n):
        L.orc_quat2mat(dp(d["q_xyzw"][i].copy()), dp(R))
        np.testing.assert_allclose(R.reshape(3, 3), d["quat2mat"][i], atol=1e-6)
        L.orc_mat2quat(dp(np.ascontiguousarray(d["quat2mat"][i].reshape(-1))), dp(q))
        np.testing.assert_allclose(q, d["mat2quat"][i], atol=1e-9)
        L.orc_quat2axisangle(dp(d["mat2quat"][i].copy()), dp(v))
        np.testing.assert_allclose(v, d["quat2axisangle"][i], atol=1e-12)
        L.orc_axisangle2quat(dp(d["aa_in"][i].copy()), dp(q))
        np.testing.assert_allclose(q, d["axisangle2quat"][i], atol=1e-14)
        a = np.ascontiguousarray(d["quat2mat"][i].reshape(-1))
        b = np.ascontiguousarray(d["quat2mat"][(i + 1) % n].reshape(-1))
        L.orc_angular_error(dp(a), dp(b), dp(v))
        np.testing.assert_allclose(v, d["angular_error"][i], atol=1e-14)
        T = np.zeros(16)
        L.orc_exp2mat(dp(d["exp_w"][i].copy()), dp(d["exp_v"][i].copy()), C.c_double(d["exp_th"][i]), dp(T))
        np.testing.assert_allclose(T.reshape(4, 4), d["exp2mat"][i], atol=1e-13)
        A = np.zeros(36)
        L.orc_adjoint(dp(T), dp(A))
        np.testing.assert_allclose(A.reshape(6, 6), d["adjoint"][i], atol=1e-13)
        a4, b4 = d["exp2mat"][i], d["exp2mat"][(i + 7) % n]
        op, oR = np.zeros(3), np.zeros(9)
        L.orc_limit_pose(dp(np.ascontiguousarray(a4[:3, 3] * 0.05)), dp(np.ascontiguousarray(a4[:3, :3].reshape(-1))),
                         dp(np.ascontiguousarray(b4[:3, 3] * 0.05)), dp(np.ascontiguousarray(b4[:3, :3].reshape(-1))),
                         C.c_double(0.1), C.c_double(0.3), dp(op), dp(oR))
        np.testing.assert_allclose(op, d["limit_pose_pos"][i], atol=1e-14)
        np.testing.assert_allclose(oR.reshape(3, 3), d["limit_pose_mat"][i], atol=2e-6)


@pytest.mark.parametrize("arm", ["left", "right", "middle"])
def test_fk_jac(arm):
    L = lib()
    m = load_model()
    d = np.load(os.path.join(G, f"fk_jac_{arm}.npz"))
    n = d["q"].shape[1]
    T = np.zeros(16)
    J = np.zeros(6 * n)
    for i in range(len(d["q"])):
        q = d["q"][i].copy()
        L.orc_fk(m, ARMS[arm], dp(q), dp(T))
        np.testing.assert_allclose(T.reshape(4, 4), d["fk"][i], atol=1e-13)
        L.orc_jac(m, ARMS[arm], dp(q), dp(J))
        np.testing.assert_allclose(J.reshape(6, n), d["jac"][i], atol=1e-13)


def run_diffik(L, m, arm, d, i, use_mat):
    n = d["q"].shape[1]
    out = np.zeros(n)
    common = (C.c_double(0.9), C.c_double(0.9), C.c_double(1e-4), dp(d["k_null"].copy()), dp(d["q0"].copy()),
              C.c_double(3.14), C.c_double(0.04), 10, dp(out))
    if use_mat:
        L.orc_diffik_R(m, ARMS[arm], dp(d["q"][i].copy()), dp(d["target_pos"][i].copy()),
                       dp(np.ascontiguousarray(d["target_mat"][i].reshape(-1))), *common)
    else:
        L.orc_diffik(m, ARMS[arm], dp(d["q"][i].copy()), dp(d["target_pos"][i].copy()),
                     dp(d["target_quat_wxyz"][i].copy()), *common)
    return out


@pytest.mark.parametrize("arm", ["left", "right", "middle"])
def test_diffik(arm):
    """diff_ik.py:51-85, 10 iterations, sim_env.py:125-138 parameters."""
    L = lib()
    m = load_model()
    d = np.load(os.path.join(G, f"diffik_{arm}.npz"))
    N = len(d["q"])
    # (a) fed the reference's own float32-rounded target matrix: the algorithm itself must agree tightly
    err = np.array([np.abs(run_diffik(L, m, arm, d, i, True) - d["q_out"][i]).max() for i in range(N)])
    assert err.max() < 1e-8, (err.max(), int(np.argmax(err)))
    # (b) through the quaternion entry: orc_quat2mat reproduces NumPy's float32 arithmetic op for op (transform_utils.py:64-79;
    # the norm is OpenBLAS sdot: float products accumulated in double), so the target matrix is the reference's bit for bit
    err = np.array([np.abs(run_diffik(L, m, arm, d, i, False) - d["q_out"][i]).max() for i in range(N)])
    assert err.max() < 1e-8, (np.median(err), err.max())


@pytest.mark.parametrize("arm", ["left", "right"])
def test_gradik(arm):
    """grad_ik.py:8-99 with the sim_env.py:89-122 parameters.  The secant descent is chaotic (rounding
    noise grows ~x3-10 per iteration; the reference itself runs under numba fastmath), so the algorithm
    is pinned on runs truncated to 1/4/8 iterations and the 50-iteration result statistically."""
    L = lib()
    m = load_model()
    d = np.load(os.path.join(G, f"gradik_{arm}.npz"))
    max_it = C.c_int.in_dll(L, "orc_gradik_max_it")

    def run(K):
        max_it.value = K
        err = []
        for i in range(len(d["q"])):
            out = np.zeros(6)
            L.orc_gradik_R(m, ARMS[arm], dp(d["q"][i].copy()), dp(d["target_pos"][i].copy()),
                           dp(np.ascontiguousarray(d["target_mat"][i].reshape(-1))), dp(out))
            err.append(np.abs(out - d[f"q_out_it{K}"][i]).max())
        return np.array(err)

    try:
        # (limit_pose re-enters the float32 quat2mat when it clamps the rotation, transform_utils.py:283: reproduced bit for bit)
        e1 = run(1)
        assert np.median(e1) < 1e-12 and e1.max() < 1e-10, (np.median(e1), e1.max())
        e4 = run(4)
        assert np.median(e4) < 1e-11 and e4.max() < 1e-8, (np.median(e4), e4.max())
        e8 = run(8)
        assert np.median(e8) < 1e-10 and e8.max() < 1e-6, (np.median(e8), e8.max())
        e50 = run(50)
        assert np.median(e50) < 1e-4 and e50.max() < 0.1, (np.median(e50), e50.max())
    finally:
        max_it.value = 50


@pytest.mark.parametrize("arm", ["left", "right"])
def test_gradik_error_grows_exponentially_with_the_iteration_count(arm):
    """Where does the 50-iteration GradIK leave the reference?  tests/golden/gradik_iters.npz holds the reference's answer for a
    ladder of max_iterations (gen_gradik_iters.py, 64 inputs per arm).  The oracle starts 1e-13 away (rounding) and the distance
    grows by a steady factor of 1.4 - 2.2 per iteration -- the secant step on a cost of magnitude 1e4 amplifies rounding noise --
    with no jump at any particular iteration, which is what a logic difference in the stopping rule or the best-iterate tracking
    (grad_ik.py:60-99) would produce.  Stated bounds: 1e-10 up to 8 iterations, 1e-8 up to 16, 1e-3 up to 32."""
    L = lib()
    m = load_model()
    g = np.load(os.path.join(G, "gradik_iters.npz"))
    max_it = C.c_int.in_dll(L, "orc_gradik_max_it")
    q, pos, quat = g[f"{arm}_q"], g[f"{arm}_pos"], g[f"{arm}_quat_wxyz"]
    its = [int(k) for k in g["iters"]]
    med, mx = [], []
    try:
        for ki, k in enumerate(its):
            max_it.value = k
            err = []
            for i in range(len(q)):
                o = np.zeros(6)
                L.orc_gradik(m, ARMS[arm], dp(q[i].copy()), dp(pos[i].copy()), dp(quat[i].copy()), dp(o))
                err.append(np.abs(o - g[f"{arm}_q_out"][ki, i]).max())
            med.append(np.median(err))
            mx.append(np.max(err))
    finally:
        max_it.value = 50
    med, mx = np.array(med), np.array(mx)
    for k, bound in ((8, 1e-10), (16, 1e-8), (32, 1e-3)):
        assert mx[its.index(k)] < bound, (k, mx[its.index(k)])
    assert med[0] < 1e-12
    # growth per iteration between the rungs of the ladder (median over the inputs): never a step
    rate = (med[1:] / med[:-1]) ** (1.0 / np.diff(its))
    assert rate.max() < 3.0, (its, rate)
    overall = (med[its.index(40)] / med[its.index(4)]) ** (1.0 / 36)
    assert 1.2 < overall < 2.5, overall


def test_quat2mat_reproduces_the_reference_bit_for_bit():
    """transform_utils.py:64-79 in NumPy float32 arithmetic: 256 random quaternions, every matrix entry identical."""
    L = lib()
    g = np.load(os.path.join(G, "so3_helpers.npz"))
    for i in range(len(g["q_xyzw"])):
        R = np.zeros(9)
        L.orc_quat2mat(dp(g["q_xyzw"][i].copy()), dp(R))
        assert np.array_equal(R.reshape(3, 3), g["quat2mat"][i]), i
