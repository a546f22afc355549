"""Support tables of the collision hulls (av_aloha_amd/compiler/hull.py support_table; the device's support() in avsim_collide.hip.h
looks at the candidates of the direction's cube-map cell only): for every hull of a compiled model and 12 000 directions -- random ones,
the hull's face normals (every vertex of the face ties), edge directions, the geom's own axes (cell centres of the odd resolutions) --
the cell's candidate list holds the vertex a scan over all vertices finds, and every vertex that ties with it."""
import os

import numpy as np

from test_oracle_physics import ROOT


def test_every_support_point_is_in_its_cell():
    from av_aloha_amd.compiler import hull as H
    from av_aloha_amd.compiler.compile import read_blob
    md = read_blob(os.path.join(ROOT, "models", "hook_package_3arms.avm"))
    gh, hv = md["geom_chull"].reshape(-1, 2), md["chull_vert"].reshape(-1, 3)
    ctab, cells, cand = md["geom_ctab"].reshape(-1, 2), md["chull_cells"], md["chull_cand"]
    rng = np.random.default_rng(3)
    seen, nh = set(), 0
    for g in range(len(gh)):
        if gh[g, 1] == 0 or int(gh[g, 0]) in seen:
            continue
        seen.add(int(gh[g, 0]))
        nh += 1
        V = hv[gh[g, 0]:gh[g, 0] + gh[g, 1]]
        cb, R = (int(x) for x in ctab[g])
        assert R % 2 == 1 and len(V) <= 128
        fn = np.array([n for n, _ in H.merged_faces(V)])
        D = np.concatenate([rng.normal(size=(12000, 3)), fn, fn + rng.normal(size=fn.shape) * 1e-7, fn[:-1] + fn[1:], np.eye(3), -np.eye(3)])
        D = D[np.abs(D).sum(axis=1) > 1e-3]                  # (the sum of two opposite normals is no direction)
        P = D @ V.T
        mx = P.max(axis=1)
        ci = H.cube_cell(D, R)
        recs = cells[cb + ci]
        for k in range(len(D)):
            off, cnt = int(recs[k]) >> 8, int(recs[k]) & 255
            c = cand[off:off + cnt]
            assert np.all(np.diff(c) > 0)                                   # sorted by index: the lowest index of a tie is found first
            tied = np.nonzero(P[k] >= mx[k] - 1e-10 * np.abs(D[k]).sum())[0]
            assert np.isin(tied, c).all(), (g, D[k], tied, c)
    assert nh >= 20


def test_collision_hulls_are_close_to_the_full_hulls():
    """The device's collision hulls keep <= 128 vertices: at most 0.3 mm of any mesh sticks out (0.15 mm for the gripper parts, 0.02 mm for
    the fingers), against 8.9 / 2.3 / 0.4 mm for the 20 / 32-vertex hulls of rounds 1-4 (manifest "hulls")."""
    import json
    man = json.load(open(os.path.join(ROOT, "models", "hook_package_3arms.json")))
    for name, h in man["hulls"].items():
        assert h["collision_nvert"] <= 128 and h["collision_err_m"] <= 3.0e-4, (name, h)
        if "gripper" in name or "d405" in name:
            assert h["collision_err_m"] <= 1.5e-4, (name, h)
        if "finger" in name:
            assert h["collision_err_m"] <= 2.5e-5, (name, h)
