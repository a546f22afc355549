"""The 16-lane box-box routine against the oracle's serial one on random poses: the two task objects of SlotInsertion are thrown
into / onto each other and the table in random orientations (face, edge and corner configurations, clipped polygons with more
than four vertices), the device's contact list after a forward pass must equal the oracle's: same geom pairs in the same
order, every distance to 1e-12.  The f64 kernel is compiled without FMA contraction (avsim_phys_f64.hip), as the oracle is: the
same expressions round the same way, so the tie-breaks of the clipping (which four vertices of a larger polygon are kept:
farthest point, largest cross product -- ties up to rounding for symmetric polygons) fall the same way on both sides."""
import ctypes as C

import numpy as np
import pytest

from orc_env import OrcEnv
from test_oracle_physics import model_dict

pytestmark = pytest.mark.gpu


def rand_quat(rng, small):
    if small:
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        ang = rng.uniform(-0.3, 0.3)
        return np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * ax])
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def test_box_box_contacts_match_the_oracle_on_random_poses():
    from av_aloha_amd.sim import BatchedSim
    md = model_dict()
    n = 256
    rng = np.random.default_rng(5)
    q = np.repeat(md["qpos_home"][None], n, 0).copy()
    for i in range(n):
        small = i % 2 == 0
        q[i, 23:26] = [rng.uniform(-0.05, 0.05), rng.uniform(0.08, 0.14), rng.uniform(-0.005, 0.03)]      # slot
        q[i, 26:30] = rand_quat(rng, small)
        q[i, 30:33] = q[i, 23:26] + rng.uniform(-0.04, 0.04, 3) + [0, 0, rng.uniform(0, 0.03)]             # stick, overlapping the slot
        q[i, 33:37] = rand_quat(rng, small)
    sim = BatchedSim("slot_insertion", 3, n, f64=True)
    sim.set_qpos(q)
    rw = np.empty(n, dtype=np.int32)
    su = np.empty(n, dtype=np.uint8)
    sim.h.check(sim.h.L.avsim_observe(sim.h.h, None, rw.ctypes.data, su.ctypes.data))      # zero-substep pass: contacts of this state
    ncon, pairs, dist = sim.contacts()
    e = OrcEnv()
    e.L.orc_set_qpos.argtypes = [C.c_void_p, C.c_void_p]
    kinds = set()
    total = 0
    for i in range(n):
        e.L.orc_set_qpos(e.dptr, q[i].ctypes.data)
        cs = list(e.d.contact)[: e.d.ncon]
        assert ncon[i] == e.d.ncon, (i, ncon[i], e.d.ncon)
        for k, c in enumerate(cs):
            assert (pairs[i, k, 0], pairs[i, k, 1]) == (c.geom1, c.geom2), (i, k)
            assert abs(dist[i, k] - c.dist) < 1e-12, (i, k, dist[i, k], c.dist)
        total += e.d.ncon
        kinds.add(min(e.d.ncon, 12))
    assert total > 4 * n and len(kinds) >= 5          # plenty of contacts, and many different contact counts
    sim.close()
    e.close()


def test_hull_contacts_match_the_oracle_near_the_grippers():
    """The same comparison for the convex-hull pairs (MPR, one contact per pair): the objects are dropped around the open
    grippers of randomly perturbed arms.  MPR's portal walk branches on signs of small dot products and its support function picks
    the hull vertex with the largest projection (ties up to rounding whenever the search direction is a face normal); without FMA
    contraction on either side every such decision agrees: pairs, counts and every distance to 1e-12."""
    from av_aloha_amd.sim import BatchedSim
    md = model_dict()
    n = 256
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
    sim = BatchedSim("slot_insertion", 3, n, f64=True)
    sim.set_qpos(q)
    rw = np.empty(n, dtype=np.int32)
    su = np.empty(n, dtype=np.uint8)
    sim.h.check(sim.h.L.avsim_observe(sim.h.h, None, rw.ctypes.data, su.ctypes.data))
    ncon, pairs, dist = sim.contacts()
    e = OrcEnv()
    e.L.orc_set_qpos.argtypes = [C.c_void_p, C.c_void_p]
    names = e.man["geom_names"]
    hull_contacts = seen = 0
    for i in range(n):
        e.L.orc_set_qpos(e.dptr, q[i].ctypes.data)
        cs = list(e.d.contact)[: e.d.ncon]
        assert ncon[i] == e.d.ncon, (i, ncon[i], e.d.ncon)
        for k, c in enumerate(cs):
            assert (pairs[i, k, 0], pairs[i, k, 1]) == (c.geom1, c.geom2), (i, k)
            assert abs(dist[i, k] - c.dist) < 1e-12, (i, k, names[c.geom1], names[c.geom2], dist[i, k], c.dist)
            seen += 1
            hull_contacts += names[c.geom1] not in ("table",) and not names[c.geom1].startswith(("slot", "pin", "stick"))
    assert hull_contacts > 50, hull_contacts
    assert seen > 500
    sim.close()
    e.close()
