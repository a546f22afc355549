"""Convex-hull extraction and vertex-budget decimation for collision meshes.

MuJoCo collides mesh geoms as the convex hull of their vertices [EXT]; the
reference's robot links and frame extrusions are such meshes
(aloha_sim.xml:106-111, scene.xml:43-47).  The HIP narrow phase evaluates
support functions by brute force over a hull's vertices, so each hull is
reduced to at most ``kmax`` vertices with a greedy outer-distance criterion:
starting from the axis-extreme points, repeatedly add the original hull vertex
that lies farthest outside the current sub-hull.  The reported ``err`` is the
largest distance by which a dropped vertex still sticks out (metres).
"""
from __future__ import annotations

import struct

import numpy as np
from scipy.spatial import ConvexHull


def read_stl(path):
    b = open(path, "rb").read()
    n = struct.unpack("<I", b[80:84])[0]
    if len(b) != 84 + 50 * n:
        raise ValueError(f"{path}: not a binary STL")
    rec = np.dtype([("n", "<3f4"), ("v", "<9f4"), ("a", "<u2")])
    a = np.frombuffer(b[84:], dtype=rec)
    return a["v"].reshape(-1, 3).astype(np.float64)


def decimate_hull(points, kmax):
    hull = ConvexHull(points)
    hv = points[hull.vertices]
    if len(hv) <= kmax:
        return hv, 0.0
    sel = set()
    for ax in range(3):
        sel.add(int(np.argmin(hv[:, ax])))
        sel.add(int(np.argmax(hv[:, ax])))
    sel = list(sel)
    # make sure the seed is full-dimensional
    k = 0
    while True:
        try:
            sub = ConvexHull(hv[sel])
            break
        except Exception:
            cand = [i for i in range(len(hv)) if i not in sel]
            sel.append(cand[k])
            k += 1
    err = 0.0
    while True:
        sub = ConvexHull(hv[sel])
        eq = sub.equations  # n.x + d <= 0 inside
        out = (hv @ eq[:, :3].T + eq[:, 3]).max(axis=1)
        out[sel] = -1.0
        i = int(np.argmax(out))
        err = float(max(out[i], 0.0))
        if len(sel) >= kmax or err <= 1e-9:
            break
        sel.append(i)
    sub = ConvexHull(hv[sel])
    return hv[sel][sub.vertices], err


def hull_planes(verts, tol=1e-9):
    """Facet planes [n, d] (n.x <= d inside, |n| = 1) of the convex hull of `verts`, coplanar triangles merged."""
    from scipy.spatial import ConvexHull
    eq = ConvexHull(np.asarray(verts, dtype=np.float64)).equations      # n.x + off <= 0 inside
    out = []
    for e in eq:
        row = np.array([e[0], e[1], e[2], -e[3]])
        if not any(np.abs(row - o).max() < 1e-7 for o in out):
            out.append(row)
    return np.array(out)
