"""HIP IK kernels (through the C-ABI) vs the golden vectors of the reference's Python and vs the CPU
oracle.  f64 on the device, so parity is at rounding level."""
import ctypes as C
import os

import numpy as np
import pytest

from avsim_test_util import blob

pytestmark = pytest.mark.gpu
POSE_TOL_M, POSE_TOL_RAD = 0.02, 0.08       # measured: 7.3e-3 m, 3.4e-2 rad (left), 3.6e-3 m, 2.1e-2 rad (right) at worst over 256 targets; medians equal to three digits
G = os.path.join(os.path.dirname(__file__), "golden")
ARMS = {"left": 0, "right": 1, "middle": 2}


@pytest.fixture(scope="module")
def sim():
    from av_aloha_amd._ffi import Handle
    h = Handle(blob(), 8)
    yield h
    h.close()


@pytest.mark.parametrize("arm", ["left", "right", "middle"])
def test_fk_jac_golden(sim, arm):
    d = np.load(os.path.join(G, f"fk_jac_{arm}.npz"))
    q = np.ascontiguousarray(d["q"])
    n, nj = q.shape
    T = np.zeros((n, 16))
    J = np.zeros((n, 6, nj))
    sim.check(sim.L.avsim_fk_jac(sim.h, ARMS[arm], n, q.ctypes.data, T.ctypes.data, J.ctypes.data))
    np.testing.assert_allclose(T.reshape(n, 4, 4), d["fk"], atol=1e-12)
    np.testing.assert_allclose(J, d["jac"], atol=1e-12)


@pytest.mark.parametrize("arm", ["left", "right", "middle"])
def test_diffik_golden_and_oracle(sim, arm):
    from orc_ffi import dp, lib, load_model
    d = np.load(os.path.join(G, f"diffik_{arm}.npz"))
    q = np.ascontiguousarray(d["q"])
    n, nj = q.shape
    out = np.zeros((n, nj))
    pos = np.ascontiguousarray(d["target_pos"])
    quat = np.ascontiguousarray(d["target_quat_wxyz"])
    sim.check(sim.L.avsim_ik(sim.h, ARMS[arm], 0, 0, n, q.ctypes.data, pos.ctypes.data, quat.ctypes.data, out.ctypes.data))
    # vs the reference's own outputs: the device's quat2mat reproduces NumPy's float32 arithmetic op for op (avsim_ik.hip.h), so the
    # target matrix is the reference's; what is left is Cholesky vs LU / eigen-pinv in the damped solve near singular poses
    err = np.abs(out - d["q_out"]).max(axis=1)
    assert np.median(err) < 1e-9 and err.max() < 1e-5, (np.median(err), err.max())
    # vs the CPU oracle on identical inputs (same float32 quat2mat): Cholesky vs LU / eigen-pinv only
    L, m = lib(), load_model()
    ref = np.zeros((n, nj))
    for i in range(n):
        o = np.zeros(nj)
        L.orc_diffik(m, ARMS[arm], dp(q[i].copy()), dp(pos[i].copy()), dp(quat[i].copy()), C.c_double(0.9), C.c_double(0.9),
                     C.c_double(1e-4), dp(d["k_null"].copy()), dp(d["q0"].copy()), C.c_double(3.14), C.c_double(0.04), 10, dp(o))
        ref[i] = o
    err = np.abs(out - ref).max(axis=1)
    assert np.median(err) < 1e-10 and err.max() < 1e-4, (np.median(err), err.max())


@pytest.mark.parametrize("arm", ["left", "right"])
def test_gradik_truncated(sim, arm):
    d = np.load(os.path.join(G, f"gradik_{arm}.npz"))
    q = np.ascontiguousarray(d["q"])
    n, nj = q.shape
    pos = np.ascontiguousarray(d["target_pos"])
    quat = np.ascontiguousarray(d["target_quat_wxyz"])
    for K, med, mx in ((1, 1e-11, 1e-9), (4, 1e-10, 1e-7), (8, 1e-9, 1e-5)):
        out = np.zeros((n, nj))
        sim.check(sim.L.avsim_ik(sim.h, ARMS[arm], 1, K, n, q.ctypes.data, pos.ctypes.data, quat.ctypes.data, out.ctypes.data))
        err = np.abs(out - d[f"q_out_it{K}"]).max(axis=1)
        assert np.median(err) < med and err.max() < mx, (K, np.median(err), err.max())
    out = np.zeros((n, nj))
    sim.check(sim.L.avsim_ik(sim.h, ARMS[arm], 1, 0, n, q.ctypes.data, pos.ctypes.data, quat.ctypes.data, out.ctypes.data))
    err = np.abs(out - d["q_out"]).max(axis=1)
    assert np.median(err) < 5e-3 and err.max() < 0.2, (np.median(err), err.max())  # chaotic beyond ~20 iterations
    # What a caller of the controller can rely on at the reference's own 50 iterations is the POSE the joints reach, not the joints: the
    # device's answers get as close to the target pose as the reference's do (FK of both through avsim_fk_jac, itself pinned at 1e-12).
    def pose_err(qs):
        T = np.zeros((n, 16))
        J = np.zeros((n, 6, nj))
        qs = np.ascontiguousarray(qs)
        sim.check(sim.L.avsim_fk_jac(sim.h, ARMS[arm], n, qs.ctypes.data, T.ctypes.data, J.ctypes.data))
        T = T.reshape(n, 4, 4)
        dp_ = np.linalg.norm(T[:, :3, 3] - pos, axis=1)
        R = np.einsum("nij,nkj->nik", T[:, :3, :3], d["target_mat"])              # R_reached R_target^T
        ang = np.arccos(np.clip((np.trace(R, axis1=1, axis2=2) - 1) / 2, -1, 1))
        return dp_, ang
    dp_dev, ang_dev = pose_err(out)
    dp_ref, ang_ref = pose_err(d["q_out"])
    print(arm, "pose error at 50 iterations, device vs reference: position median %.2e / %.2e m, max %.2e / %.2e; angle median %.2e / %.2e rad, max %.2e / %.2e; |difference| max %.2e m %.2e rad"
          % (np.median(dp_dev), np.median(dp_ref), dp_dev.max(), dp_ref.max(), np.median(ang_dev), np.median(ang_ref), ang_dev.max(), ang_ref.max(),
             np.abs(dp_dev - dp_ref).max(), np.abs(ang_dev - ang_ref).max()))
    assert np.median(dp_dev) <= 1.05 * np.median(dp_ref) + 1e-6 and np.median(ang_dev) <= 1.05 * np.median(ang_ref) + 1e-6
    assert np.abs(dp_dev - dp_ref).max() < POSE_TOL_M and np.abs(ang_dev - ang_ref).max() < POSE_TOL_RAD


@pytest.mark.parametrize("arm", ["left", "right"])
def test_gradik_ladder_vs_reference(sim, arm):
    """Device GradIK against the reference's answers for a ladder of iteration counts (tests/golden/gradik_iters.npz, 64 inputs):
    the distance starts at rounding level and grows by a steady factor per iteration (chaos of the secant descent, see
    tests/test_oracle_ik_golden.py) -- no step at any iteration.  Bounds: 1e-9 at 12 iterations, 1e-7 at 16, 1e-5 at 24, 1e-2 at 32."""
    g = np.load(os.path.join(G, "gradik_iters.npz"))
    q = np.ascontiguousarray(g[f"{arm}_q"])
    pos = np.ascontiguousarray(g[f"{arm}_pos"])
    quat = np.ascontiguousarray(g[f"{arm}_quat_wxyz"])
    n = q.shape[0]
    its = [int(k) for k in g["iters"]]
    med, mx = [], []
    for ki, k in enumerate(its):
        out = np.zeros((n, 6))
        sim.check(sim.L.avsim_ik(sim.h, ARMS[arm], 1, k, n, q.ctypes.data, pos.ctypes.data, quat.ctypes.data, out.ctypes.data))
        err = np.abs(out - g[f"{arm}_q_out"][ki]).max(axis=1)
        med.append(np.median(err))
        mx.append(err.max())
    med, mx = np.array(med), np.array(mx)
    print(arm, "median", dict(zip(its, med.round(14))), "max", dict(zip(its, mx.round(12))))
    for k, bound in ((1, 1e-10), (12, 1e-9), (16, 1e-7), (24, 1e-5), (32, 1e-2)):
        assert mx[its.index(k)] < bound, (k, mx[its.index(k)])
    rate = (np.maximum(med[1:], 1e-15) / np.maximum(med[:-1], 1e-15)) ** (1.0 / np.diff(its))
    assert rate.max() < 3.5, (its, rate)
