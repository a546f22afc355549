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


def hull_topology(verts):
    """Facet planes as hull_planes(), plus what a rasteriser needs to find a hull's outline: the vertices of every (merged) face
    and the edges between two different faces.  Taken from qhull's triangulation itself (simplices and their neighbours), not
    from vertex / plane distances: end caps made of almost-coplanar triangles would fool a tolerance.
    -> planes [F, 4], face_verts (list of F sorted vertex-index lists), edges [E, 4] = (v0, v1, f0, f1)."""
    h = ConvexHull(np.asarray(verts, dtype=np.float64))
    assert len(h.vertices) == len(verts), "hull_topology expects the vertices of a convex hull"
    planes, tri_face = [], []
    for e in h.equations:
        row = np.array([e[0], e[1], e[2], -e[3]])
        for k, o in enumerate(planes):
            if np.abs(row - o).max() < 1e-7:
                tri_face.append(k)
                break
        else:
            tri_face.append(len(planes))
            planes.append(row)
    face_verts = [set() for _ in planes]
    for t, f in enumerate(tri_face):
        face_verts[f].update(int(v) for v in h.simplices[t])
    edges = {}
    for t, nb in enumerate(h.neighbors):
        for k, u in enumerate(nb):              # neighbour k lies opposite vertex k: the shared edge is the other two vertices
            if u < t or tri_face[u] == tri_face[t]:
                continue
            a, b = sorted(int(v) for j, v in enumerate(h.simplices[t]) if j != k)
            edges[(a, b)] = (tri_face[t], tri_face[u])
    E = np.array([[a, b, f0, f1] for (a, b), (f0, f1) in sorted(edges.items())], dtype=np.int32).reshape(-1, 4)
    return np.array(planes), [sorted(s) for s in face_verts], E
