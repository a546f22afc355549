"""The device's narrow-phase source (av_aloha_amd/csrc/avsim_collide.hip.h: box_box clipping, sphere / cylinder / hull pairs
through MPR) compiled for the HOST by g++ with the oracle's floating-point rules (tests/hostshim/) and run against the oracle's
orc_collide on the poses of tests/test_gpu_boxbox.py -- no GPU needed:

* identical inputs -> identical contacts, bit for bit (same expressions in the same order);
* inputs perturbed in the last bits (the device multiplies its kinematic chains out in a different order than the oracle, so its
  geom poses differ at 1e-16) -> the same contacts to 1e-9: every choice among candidates that are equal in exact arithmetic
  (support vertices of a face, box corners / cylinder caps the direction is normal to, clipped vertices on an edge parallel to the
  base line, equal depths) is made with a margin (TieTol) instead of following the sign of the noise.  Without the margins 6 % of
  the hull contacts and 0.3 % of the box pairs flip under this perturbation;
* the float instantiation (the product kernel's arithmetic) makes the same choices in all but a few per cent of the contacts."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from orc_env import OrcEnv
from orc_ffi import ROOT, dp
from test_oracle_physics import model_dict


def rand_quat(rng, small):
    if small:
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        ang = rng.uniform(-0.3, 0.3)
        return np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hostshim") / "libhostcollide.so")
    shim = os.path.join(ROOT, "tests", "hostshim")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I" + shim,
                           "-o", so, os.path.join(shim, "host_collide.cpp")])
    return C.CDLL(so)


def box_poses(n):
    md = model_dict()
    rng = np.random.default_rng(5)
    q = np.repeat(md["qpos_home"][None], n, 0).copy()
    for i in range(n):
        small = i % 2 == 0
        q[i, 23:26] = [rng.uniform(-0.05, 0.05), rng.uniform(0.08, 0.14), rng.uniform(-0.005, 0.03)]
        q[i, 26:30] = rand_quat(rng, small)
        q[i, 30:33] = q[i, 23:26] + rng.uniform(-0.04, 0.04, 3) + [0, 0, rng.uniform(0, 0.03)]
        q[i, 33:37] = rand_quat(rng, small)
    return q


def hull_poses(n):
    md = model_dict()
    rng = np.random.default_rng(9)
    q = np.repeat(md["qpos_home"][None], n, 0).copy()
    for i in range(n):
        q[i, 0:6] += rng.uniform(-0.15, 0.15, 6)
        q[i, 8:14] += rng.uniform(-0.15, 0.15, 6)
        side = 1.0 if i % 2 else -1.0
        q[i, 23:26] = [side * 0.26 + rng.uniform(-0.05, 0.05), 0.03 + rng.uniform(-0.05, 0.05), 0.2 + rng.uniform(-0.06, 0.06)]
        q[i, 26:30] = rand_quat(rng, False)
        q[i, 30:33] = [-side * 0.26 + rng.uniform(-0.05, 0.05), 0.03 + rng.uniform(-0.05, 0.05), 0.2 + rng.uniform(-0.06, 0.06)]
        q[i, 33:37] = rand_quat(rng, False)
    return q


def compare(host, q, fn, pert, tol, only_boxes):
    """Every contact pair the oracle finds at the poses q, re-run through the host-compiled device routine `fn` on the oracle's geom
    poses (times 1 + pert * noise).  Returns (pairs compared, pairs whose contact count or a distance differs by more than tol)."""
    md = model_dict()
    e = OrcEnv()
    e.L.orc_set_qpos.argtypes = [C.c_void_p, C.c_void_p]
    ng = md["geom_type"].shape[0]
    gh, hv, bc = md["geom_hull"].reshape(-1, 2), md["hull_vert"].reshape(-1, 3), md["geom_bcenter"].reshape(-1, 3)
    rng = np.random.default_rng(1)
    f = getattr(host, fn)
    tot = bad = 0
    for i in range(q.shape[0]):
        e.L.orc_set_qpos(e.dptr, q[i].ctypes.data)
        gx = np.ctypeslib.as_array(e.d.geom_xpos, shape=(ng, 3)).copy()
        gm = np.ctypeslib.as_array(e.d.geom_xmat, shape=(ng, 9)).copy()
        cs = list(e.d.contact)[: e.d.ncon]
        seen = set()
        for c in cs:
            g1, g2 = c.geom1, c.geom2
            if (g1, g2) in seen:
                continue
            seen.add((g1, g2))
            t1, t2 = int(md["geom_type"][g1]), int(md["geom_type"][g2])
            boxes = t1 == 6 and t2 == 6
            if boxes != only_boxes or (not boxes and not (t1 in (5, 7) or t2 in (5, 7))):
                continue
            ref = np.array([x.dist for x in cs if (x.geom1, x.geom2) == (g1, g2)])

            def shape(g):
                h = np.ascontiguousarray(hv[gh[g, 0]:gh[g, 0] + gh[g, 1]]).reshape(-1) if gh[g, 1] > 0 else np.zeros(3)
                p = gx[g] * (1 + rng.normal(size=3) * pert)
                m = gm[g] * (1 + rng.normal(size=9) * pert)
                cen = p + m.reshape(3, 3) @ bc[g] if pert else gx[g] + gm[g].reshape(3, 3) @ bc[g]
                return np.ascontiguousarray(md["geom_size"][g]), p, m, h, int(gh[g, 1]), np.ascontiguousarray(cen)
            a, b = shape(g1), shape(g2)
            dist, pos, nrm = np.zeros(8), np.zeros(24), np.zeros(3)
            nn = f(t1, dp(a[0]), dp(a[1]), dp(a[2]), dp(a[3]), a[4], dp(a[5]), t2, dp(b[0]), dp(b[1]), dp(b[2]), dp(b[3]), b[4], dp(b[5]),
                   dp(dist), dp(pos), dp(nrm))
            tot += 1
            bad += nn != len(ref) or np.abs(dist[:nn] - ref).max() > tol
    e.close()
    return tot, bad


def test_device_box_box_source_equals_the_oracle_and_ignores_rounding_noise(host):
    q = box_poses(192)
    tot, bad = compare(host, q, "dev_narrow_f64", 0.0, 0.0, True)
    assert tot > 900 and bad == 0, (tot, bad)
    for _ in range(2):
        tot, bad = compare(host, q, "dev_narrow_f64", 2e-16, 1e-9, True)
        assert bad == 0, (tot, bad)
    tot, bad = compare(host, q, "dev_narrow_f32", 0.0, 2e-5, True)
    assert bad <= 0.03 * tot, (tot, bad)


def test_device_mpr_source_equals_the_oracle_and_ignores_rounding_noise(host):
    q = hull_poses(160)
    tot, bad = compare(host, q, "dev_narrow_f64", 0.0, 0.0, False)
    assert tot > 900 and bad == 0, (tot, bad)
    tot, bad = compare(host, q, "dev_narrow_f64", 2e-16, 1e-9, False)
    assert bad == 0, (tot, bad)
    tot, bad = compare(host, q, "dev_narrow_f32", 0.0, 1e-4, False)
    assert bad <= 0.05 * tot, (tot, bad)
