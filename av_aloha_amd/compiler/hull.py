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


# ---- support tables: a convex hull of ANY size behind a constant-cost support function ------------------------------------
# MuJoCo collides a mesh geom as the full convex hull of its vertices [EXT]; its support function walks the hull's vertex graph.  A
# lane of the HIP narrow phase cannot afford a scan over hundreds of vertices (the hulls were decimated to 20 / 32 vertices until
# round 4) -- but for a given direction only the few vertices whose normal cones the direction can lie in matter.  The unit sphere of
# directions is cut into the 6 R^2 cells of a cube map; a cell's record lists every vertex that is a support point for SOME direction
# of the (slightly padded) cell.  The support function is then: cell of the direction (three compares, two divisions) -> the cell's
# candidates (typically 2 .. 8) -> the best of them.  Exact: the result is the vertex a scan over all vertices finds.
def cube_cell(l, R):
    """Cell index of direction(s) l [..., 3] in the cube map of resolution R: major axis a (ties to the lower axis), face 2 a + (l_a < 0),
    (u, v) = (l_{a+1}, l_{a+2}) / |l_a| in [-1, 1] cut into R x R squares.  The device's support() computes the same."""
    l = np.asarray(l, dtype=np.float64)
    ab = np.abs(l)
    a = np.where((ab[..., 0] >= ab[..., 1]) & (ab[..., 0] >= ab[..., 2]), 0, np.where(ab[..., 1] >= ab[..., 2], 1, 2))
    la = np.take_along_axis(l, a[..., None], -1)[..., 0]
    inv = 1.0 / np.maximum(np.abs(la), 1e-300)
    u = np.take_along_axis(l, ((a + 1) % 3)[..., None], -1)[..., 0] * inv
    v = np.take_along_axis(l, ((a + 2) % 3)[..., None], -1)[..., 0] * inv
    iu = np.clip(((u + 1) * 0.5 * R).astype(np.int64), 0, R - 1)
    iv = np.clip(((v + 1) * 0.5 * R).astype(np.int64), 0, R - 1)
    return ((2 * a + (la < 0)) * R + iu) * R + iv


def merged_faces(verts, tol=1e-7):
    """(unit normal, sorted vertex indices) of every face of the hull, coplanar neighbouring triangles of qhull's triangulation merged
    (union-find over the triangles' neighbours: linear in the number of triangles)."""
    h = ConvexHull(np.asarray(verts, dtype=np.float64))
    par = list(range(len(h.simplices)))

    def find(x):
        while par[x] != x:
            par[x] = par[par[x]]
            x = par[x]
        return x
    for t, nb in enumerate(h.neighbors):
        for u in nb:
            if u > t and np.abs(h.equations[t] - h.equations[u]).max() < tol:
                par[find(u)] = find(t)
    groups = {}
    for t in range(len(h.simplices)):
        groups.setdefault(find(t), []).append(t)
    return [(h.equations[root][:3].copy(), sorted({int(v) for t in ts for v in h.simplices[t]})) for root, ts in groups.items()]


def support_table(verts, R, eps=5e-6):
    """-> list of 6 R^2 sorted index arrays: the candidates of every cube-map cell.  Vertex p supports direction d iff (p - w) . d >= 0 for
    every hull neighbour w of p.  On a cube face the directions are d = s e_a + u e_b + v e_c, affine in (u, v): p's normal cone is the
    intersection of half-planes c0 + c1 u + c2 v >= 0 there, a cell is a square.  p is listed in a cell unless ONE of its half-planes has
    all four corners of the square on the wrong side by more than eps (metres per unit |d_a|: beyond the f32 tie margin of the device's
    support function) -- a superset of the cells the cone really meets (a cone can miss a square without any single half-plane saying so:
    near a corner; the list then holds a vertex too many), never a subset: directions ON a cone's boundary -- two vertices of an edge, all
    vertices of a face the direction is normal to tie -- find every tied vertex in their cell."""
    V = np.asarray(verts, dtype=np.float64)
    n = len(V)
    h = ConvexHull(V)
    assert len(h.vertices) == n, "support_table expects the vertices of a convex hull"
    nbr = [set() for _ in range(n)]
    for tri in h.simplices:
        for i in range(3):
            nbr[int(tri[i])].update((int(tri[(i + 1) % 3]), int(tri[(i + 2) % 3])))
    g = -1.0 + 2.0 / R * np.arange(R + 1)                      # cell borders
    uu, vv = np.meshgrid(g, g, indexing="ij")                   # corner grid [R + 1, R + 1]
    cells = [[] for _ in range(6 * R * R)]
    for p in range(n):
        W = V[p] - V[sorted(nbr[p])]                            # [K, 3] rows p - w
        W = W / np.linalg.norm(W, axis=1, keepdims=True)
        for a in range(3):
            b, c = (a + 1) % 3, (a + 2) % 3
            for sg in (1.0, -1.0):
                val = sg * W[:, a, None, None] + W[:, b, None, None] * uu[None] + W[:, c, None, None] * vv[None]      # [K, R + 1, R + 1] at the corners
                cmax = np.maximum(np.maximum(val[:, :-1, :-1], val[:, 1:, :-1]), np.maximum(val[:, :-1, 1:], val[:, 1:, 1:]))   # per cell, per half-plane
                ok = (cmax >= -eps).all(axis=0)                 # [R, R]
                face = 2 * a + (sg < 0)
                for iu, iv in zip(*np.nonzero(ok)):
                    cells[(face * R + int(iu)) * R + int(iv)].append(p)
    return [np.array(sorted(c), dtype=np.int32) for c in cells]


def support_table_for(verts, max_count=12, resolutions=(3, 5, 7, 9, 13, 17, 25)):
    """Smallest odd resolution (the geom's own axes -- the normals of a CAD part's flat faces -- are cell CENTRES then) whose 99th-percentile
    cell holds at most max_count candidates -> (R, cells)."""
    for R in resolutions:
        cells = support_table(verts, R)
        if np.percentile([len(c) for c in cells], 99) <= max_count or len(verts) <= max_count:
            break
    return R, cells
