"""The device's narrow-phase source (av_aloha_amd/csrc/avsim_collide.hip.h: box_box clipping, sphere / cylinder / hull pairs
through MPR, the multiccd perturbation contacts) compiled for the HOST by g++ with the oracle's floating-point rules (tests/hostshim/) and run against the oracle's
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


def ipv(a):
    """int32 array (possibly a view into a larger one) -> pointer to its first element"""
    assert a.dtype == np.int32
    return a.ctypes.data_as(C.POINTER(C.c_int))


def expand_tables(md):
    """The model's support tables as the device holds them (PhysHost::build): per cube-map cell a record of eight (x, y, z, w) entries --
    the first eight candidates, a shorter list padded with its last vertex; w of entry 0 = the count, w of entry 1 = where the cell's
    further candidates start in the overflow part behind the records.  -> (flat float64 array, entry index of the overflow part)"""
    gh, hv = md["geom_chull"].reshape(-1, 2), md["chull_vert"].reshape(-1, 3)
    ctab, cells, cand = md["geom_ctab"].reshape(-1, 2), md["chull_cells"], md["chull_cand"]
    tab = np.zeros((len(cells), 8, 4))
    ovf, done = [], np.zeros(len(cells), dtype=bool)
    for g in range(len(gh)):
        if gh[g, 1] == 0:
            continue
        cb, R = (int(x) for x in ctab[g])
        for c in range(cb, cb + 6 * R * R):
            if done[c]:
                continue
            done[c] = True
            off, cnt = int(cells[c]) >> 8, int(cells[c]) & 255
            idx = cand[off:off + cnt]
            first = np.concatenate([idx[:8], np.repeat(idx[-1:], max(0, 8 - cnt))])
            tab[c, :, :3] = hv[gh[g, 0] + first]
            tab[c, :, 3] = first
            tab[c, 0, 3], tab[c, 1, 3] = cnt, len(ovf)
            for v in idx[8:]:
                ovf.append(np.concatenate([hv[gh[g, 0] + v], [v]]))
    flat = np.concatenate([tab.reshape(-1), np.array(ovf).reshape(-1)]) if ovf else tab.reshape(-1)
    return np.ascontiguousarray(flat), tab.size // 4


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
    gh, hv, bc = md["geom_chull"].reshape(-1, 2), md["chull_vert"].reshape(-1, 3), md["geom_bcenter"].reshape(-1, 3)
    ctab = md["geom_ctab"].reshape(-1, 2)
    tab, hovf = expand_tables(md)
    rng = np.random.default_rng(1)
    f = getattr(host, fn)
    tot = bad = 0
    multi = np.zeros(9, dtype=np.int64)          # histogram of contacts per pair
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
                p = gx[g] * (1 + rng.normal(size=3) * pert)
                m = gm[g] * (1 + rng.normal(size=9) * pert)
                cen = p + m.reshape(3, 3) @ bc[g] if pert else gx[g] + gm[g].reshape(3, 3) @ bc[g]
                return np.ascontiguousarray(md["geom_size"][g]), p, m, int(ctab[g, 0]), int(ctab[g, 1]), np.ascontiguousarray(cen)
            a, b = shape(g1), shape(g2)
            dist, pos, nrm = np.zeros(8), np.zeros(24), np.zeros(3)
            nn = f(t1, dp(a[0]), dp(a[1]), dp(a[2]), a[3], a[4], dp(a[5]), t2, dp(b[0]), dp(b[1]), dp(b[2]), b[3], b[4], dp(b[5]),
                   dp(tab), len(tab), hovf, C.c_double(float(md["geom_rbound"][g1])), C.c_double(float(md["geom_rbound"][g2])), dp(dist), dp(pos), dp(nrm))
            multi[min(nn, 8)] += 1
            tot += 1
            bad += nn != len(ref) or np.abs(dist[:nn] - ref).max() > tol
    e.close()
    compare.multi = multi
    return tot, bad


def test_device_box_box_source_equals_the_oracle_and_ignores_rounding_noise(host):
    q = box_poses(192)
    tot, bad = compare(host, q, "dev_narrow_f64", 0.0, 0.0, True)
    assert tot > 900 and bad == 0, (tot, bad)
    print("box-box contacts per pair:", compare.multi.tolist())
    assert compare.multi[5:].sum() >= 20 and compare.multi[6:].sum() >= 5       # manifolds of five and six points occur and agree (<= 4 kept until round 4)
    for _ in range(2):
        tot, bad = compare(host, q, "dev_narrow_f64", 2e-16, 1e-9, True)
        assert bad == 0, (tot, bad)
    tot, bad = compare(host, q, "dev_narrow_f32", 0.0, 2e-5, True)
    assert bad <= 0.03 * tot, (tot, bad)


def test_device_mpr_source_equals_the_oracle_and_ignores_rounding_noise(host):
    q = hull_poses(160)
    tot, bad = compare(host, q, "dev_narrow_f64", 0.0, 0.0, False)
    assert tot > 900 and bad == 0, (tot, bad)
    # (these are deep corner-first overlaps: a unique deepest point, so the multiccd perturbations rarely find a second contact)
    tot, bad = compare(host, q, "dev_narrow_f64", 2e-16, 1e-9, False)
    assert bad == 0, (tot, bad)
    tot, bad = compare(host, q, "dev_narrow_f32", 0.0, 1e-4, False)
    assert bad <= 0.05 * tot, (tot, bad)


def test_multiccd_gives_a_flat_contact_its_rim(host):
    """aloha_sim.xml:5 multiccd: a cylinder standing on a box (flat cap on flat face, 0.1 mm deep) gets the first contact plus the
    four perturbation contacts on the rim of the cap, about +-x and +-y of the contact frame; tilting the cylinder by 0.01 rad
    leaves the contacts of the low side only.  Device source == oracle."""
    from orc_ffi import lib
    L = lib()
    z, I = np.zeros(3), np.eye(3).reshape(-1).copy()
    cyl, box = np.array([0.02, 0.03, 0.0]), np.array([0.1, 0.1, 0.02])
    seen = {}
    for tilt in (0.0, 0.01):
        c, s = np.cos(tilt), np.sin(tilt)
        Rc = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]).reshape(-1).copy()
        cp, bp = np.array([0.01, 0.02, 0.02 + 0.03 - 1e-4]), np.zeros(3)
        dist, pos, nrm = np.zeros(8), np.zeros(24), np.zeros(3)
        c4 = np.zeros(32)
        n = host.dev_narrow_f64(5, dp(cyl), dp(cp), dp(Rc), 0, 1, dp(cp.copy()), 6, dp(box), dp(bp), dp(I), 0, 1, dp(bp.copy()),
                                dp(c4), 32, 8, C.c_double(0.05), C.c_double(0.05), dp(dist), dp(pos), dp(nrm))
        do, po, no = np.zeros(8), np.zeros(24), np.zeros(24)
        m = L.orc_narrow(5, dp(cyl), dp(cp), dp(Rc), dp(z), 0, 6, dp(box), dp(bp), dp(I), dp(z), 0, dp(do), dp(po), dp(no))
        assert n == m and np.array_equal(dist[:n], do[:n]) and np.array_equal(pos[:3 * n], po[:3 * n])
        seen[tilt] = (n, dist[:n].copy(), pos[:3 * n].reshape(-1, 3).copy())
        assert abs(nrm[2] + 1) < 1e-6                      # normal from the cylinder (geom 1) down into the box
    n, dist, pos = seen[0.0]
    assert n == 5 and abs(dist[0] + 1e-4) < 1e-9 and (dist[1:] < -1e-4).all() and (dist[1:] > -2e-4).all()
    rad = np.linalg.norm(pos[1:, :2] - [0.01, 0.02], axis=1)
    assert np.abs(rad - 0.02).max() < 3e-4                 # the four extra contacts sit on the rim
    dirs = (pos[1:, :2] - [0.01, 0.02]) / rad[:, None]
    assert np.abs(dirs @ dirs.T).round(2).tolist().count([1.0, 1.0, 0.0, 0.0]) >= 1 or len({tuple(np.sign(d.round(1))) for d in dirs}) == 4
    n, dist, pos = seen[0.01]
    assert 1 <= n <= 4 and (pos[:, 0] > 0.025).all()          # only the low side (+x) of the tilted cap touches
