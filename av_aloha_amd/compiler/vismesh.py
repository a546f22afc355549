"""Visual scene of the colour renderer: the meshes MuJoCo's OpenGL pipeline draws for the reference's images
(gym_guided_vision/gym_guided_vision/env.py:180-188 get_obs "pixels", :195-200 render; assets/aloha_sim.xml class "visual",
assets/scene.xml class "frame" and the table), brought down to a triangle budget a per-env software rasteriser can afford.

A mesh LIBRARY shared by every model (models/visual_meshes.avv): the STL / OBJ meshes decimated by vertex clustering (one grid
cell size for the whole scene, found by bisection so that the instances of the biggest scene stay under the budget), unit
primitives for the task objects (box, cylinder, sphere), texture coordinates of the OBJ meshes and the table texture
(small_meta_table_diffuse.png, box-filtered to 256 x 256).  Every compiled model carries its INSTANCES of it (vis_inst_*: mesh,
body, pose in the body frame, scale, colour, textured or not): the geoms of groups 0-2, which is what MuJoCo draws by default
[EXT mjvOption.geomgroup]; group 3 (collision hulls, reward pins) is not drawn.
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np

from .mjcf import quat_to_mat

PRIM_BOX, PRIM_CYLINDER, PRIM_SPHERE = "__box", "__cylinder", "__sphere"
TEX_N = 256


# ---------------------------------------------------------------------------------------------
# file formats
# ---------------------------------------------------------------------------------------------
def weld(tri_verts):
    """(m, 3, 3) triangle corner coordinates -> (V, F): identical corners become one vertex."""
    flat = tri_verts.reshape(-1, 3)
    V, inv = np.unique(flat, axis=0, return_inverse=True)
    return V, inv.reshape(-1, 3).astype(np.int32)


def read_stl(path):
    b = open(path, "rb").read()
    n = struct.unpack("<I", b[80:84])[0]
    if 84 + 50 * n != len(b):
        raise ValueError(f"{path}: not a binary STL")
    rec = np.frombuffer(b, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
    return weld(rec["v"].astype(np.float64))


def read_obj(path):
    """Vertices, texture coordinates, triangles (fan-triangulated polygons) with their per-corner uv indices (-1: none)."""
    V, T, F, FT = [], [], [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            V.append([float(x) for x in p[1:4]])
        elif p[0] == "vt":
            T.append([float(x) for x in p[1:3]])
        elif p[0] == "f":
            idx = []
            for c in p[1:]:
                q = c.split("/")
                idx.append((int(q[0]) - 1, int(q[1]) - 1 if len(q) > 1 and q[1] else -1))
            for k in range(1, len(idx) - 1):
                F.append([idx[0][0], idx[k][0], idx[k + 1][0]])
                FT.append([idx[0][1], idx[k][1], idx[k + 1][1]])
    return np.array(V), np.array(T) if T else np.zeros((0, 2)), np.array(F, dtype=np.int32), np.array(FT, dtype=np.int32)


def read_png_rgb8(path):
    """8-bit RGB / RGBA, non-interlaced PNG -> uint8 (h, w, 3)."""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    i, idat, hdr = 8, [], None
    while i < len(b):
        n, t = struct.unpack(">I4s", b[i:i + 8])
        if t == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", b[i + 8:i + 21])
        elif t == b"IDAT":
            idat.append(b[i + 8:i + 8 + n])
        i += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or ctype not in (2, 6) or interlace:
        raise NotImplementedError("PNG variant")
    bpp = 3 if ctype == 2 else 4
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * bpp)
    out = np.zeros((h, w * bpp), dtype=np.int32)
    prev = np.zeros(w * bpp, dtype=np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:
            # sub / average / paeth run left to right through the scanline
            cur = np.zeros(w * bpp, dtype=np.int32)
            ln, pv = line.tolist(), prev.tolist()
            c = [0] * (w * bpp)
            for x in range(w * bpp):
                a = c[x - bpp] if x >= bpp else 0
                bb = pv[x]
                cc = pv[x - bpp] if x >= bpp else 0
                if f == 1:
                    pr = a
                elif f == 3:
                    pr = (a + bb) >> 1
                else:
                    p_ = a + bb - cc
                    pa, pb, pc = abs(p_ - a), abs(p_ - bb), abs(p_ - cc)
                    pr = a if (pa <= pb and pa <= pc) else (bb if pb <= pc else cc)
                c[x] = (ln[x] + pr) & 255
            cur = np.array(c, dtype=np.int32)
        out[y] = cur
        prev = cur
    return out.reshape(h, w, bpp)[:, :, :3].astype(np.uint8)


# ---------------------------------------------------------------------------------------------
# geometry
# ---------------------------------------------------------------------------------------------
def clean(V, F, FT=None):
    """Drop degenerate and repeated triangles and unreferenced vertices."""
    ok = (F[:, 0] != F[:, 1]) & (F[:, 1] != F[:, 2]) & (F[:, 0] != F[:, 2])
    F = F[ok]
    if FT is not None:
        FT = FT[ok]
    key = np.sort(F, axis=1)
    _, first = np.unique(key, axis=0, return_index=True)
    first = np.sort(first)
    F = F[first]
    if FT is not None:
        FT = FT[first]
    used = np.unique(F)
    remap = -np.ones(len(V), dtype=np.int64)
    remap[used] = np.arange(len(used))
    return V[used], remap[F].astype(np.int32), FT


def cluster_decimate(V, F, cell):
    """Vertex clustering: the vertices of a grid cell become their mean; triangles that lose a corner disappear."""
    if cell <= 0 or len(F) <= 12:
        return V, F
    k = np.floor((V - V.min(0)) / cell).astype(np.int64)
    key = (k[:, 0] * 73856093) ^ (k[:, 1] * 19349663) ^ (k[:, 2] * 83492791)
    _, inv = np.unique(np.stack([k[:, 0], k[:, 1], k[:, 2]], 1), axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    n = inv.max() + 1
    cnt = np.bincount(inv, minlength=n).astype(np.float64)
    V2 = np.stack([np.bincount(inv, weights=V[:, c], minlength=n) / cnt for c in range(3)], 1)
    V2, F2, _ = clean(V2, inv[F].astype(np.int32))
    del key
    return V2, F2


CREASE_COS = 0.8


def corner_normals(V, F, cos_thr=CREASE_COS):
    """Per triangle and corner, the normal the lighting uses there (unit, mesh frame, [nt, 3, 3]): the area-weighted mean of the normals of
    the faces around that vertex whose own normal lies within the crease angle (cos >= 0.8) of THIS face's -- smooth over curved surfaces,
    faceted across hard edges.  MuJoCo's compiler generates vertex normals for meshes that bring none (the STL files here) by area-weighted
    averaging and, with <compiler smoothnormal="false"> (the default), leaves large-angle faces out of a vertex's average [EXT: from the
    documentation and memory of user_mesh.cc; the 0.8 is the uncertain part]; its fixed-function GL pipeline then lights per VERTEX and
    interpolates (Gouraud)."""
    V = np.asarray(V, dtype=np.float64)
    F = np.asarray(F, dtype=np.int64)
    cr = np.cross(V[F[:, 1]] - V[F[:, 0]], V[F[:, 2]] - V[F[:, 0]])          # 2 x area x unit normal
    ln = np.linalg.norm(cr, axis=1)
    fn = cr / np.maximum(ln, 1e-300)[:, None]
    order = np.argsort(F.reshape(-1), kind="stable")
    vs = F.reshape(-1)[order]
    fs = order // 3
    start = np.searchsorted(vs, np.arange(len(V)))
    end = np.searchsorted(vs, np.arange(len(V)), side="right")
    out = np.zeros((len(F), 3, 3))
    for f in range(len(F)):
        for c in range(3):
            v = F[f, c]
            adj = fs[start[v]:end[v]]
            keep = adj[fn[adj] @ fn[f] >= cos_thr]
            n = cr[keep].sum(0)
            l = np.linalg.norm(n)
            out[f, c] = n / l if l > 1e-300 else (fn[f] if ln[f] > 0 else np.array([0.0, 0.0, 1.0]))      # (a degenerate triangle is never drawn)
    return out


def unit_box():
    V = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float64)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    F = []
    for a, b, c, d in quads:
        F += [[a, b, c], [a, c, d]]
    return V, np.array(F, dtype=np.int32)


def unit_cylinder(n=24):
    """Radius 1 about z, half height 1."""
    ang = 2 * np.pi * np.arange(n) / n
    ring = np.stack([np.cos(ang), np.sin(ang)], 1)
    V = np.concatenate([np.c_[ring, -np.ones(n)], np.c_[ring, np.ones(n)], [[0, 0, -1], [0, 0, 1]]])
    F = []
    for i in range(n):
        j = (i + 1) % n
        F += [[i, j, n + j], [i, n + j, n + i], [2 * n, j, i], [2 * n + 1, n + i, n + j]]
    return V, np.array(F, dtype=np.int32)


def unit_sphere(sub=2):
    t = (1 + 5 ** 0.5) / 2
    V = [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]]
    F = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    V = [np.array(v, dtype=np.float64) / np.linalg.norm(v) for v in V]
    for _ in range(sub):
        cache, F2 = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                p = V[a] + V[b]
                V.append(p / np.linalg.norm(p))
                cache[k] = len(V) - 1
            return cache[k]
        for a, b, c in F:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            F2 += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        F = F2
    return np.array(V), np.array(F, dtype=np.int32)


# ---------------------------------------------------------------------------------------------
# instances of a parsed model, the library
# ---------------------------------------------------------------------------------------------
def visible_geoms(m):
    """Geoms MuJoCo draws by default: groups 0..2, alpha > 0."""
    return [g for g in m.geoms if g["group"] <= 2 and g["rgba"][3] > 0 and g["type"] in ("mesh", "box", "cylinder", "sphere")]


def instance_table(m, colour_of):
    """Per visible geom: library mesh name, body, pose in the body frame, scale, colour, texture flag."""
    rows = []
    for g in visible_geoms(m):
        s = np.zeros(3)
        s[:min(3, len(g["size"]))] = g["size"][:3]
        if g["type"] == "mesh":
            name, scale = m.meshes[g["mesh"]]["file"], np.asarray(m.meshes[g["mesh"]]["scale"], dtype=float)
        elif g["type"] == "box":
            name, scale = PRIM_BOX, s
        elif g["type"] == "cylinder":
            name, scale = PRIM_CYLINDER, np.array([s[0], s[0], s[1]])
        else:
            name, scale = PRIM_SPHERE, np.array([s[0], s[0], s[0]])
        mat = g.get("material")
        textured = mat is not None and mat in m.materials and m.materials[mat] is None and not g.get("rgba_given")
        rows.append({"mesh": name, "body": g["body"], "pos": np.asarray(g["pos"], dtype=float), "mat": quat_to_mat(g["quat"]),
                     "scale": scale, "rgba": colour_of(g), "tex": int(textured)})
    return rows


def build_library(models, texture_png, budget=20000, verbose=False):
    """models: parsed Models whose scenes must fit the budget.  Returns (arrays of the library file, {mesh name: id})."""
    files = sorted({r["mesh"] for m in models for r in instance_table(m, lambda g: np.ones(4)) if not r["mesh"].startswith("__")})
    raw = {}
    for f in files:
        if f.lower().endswith(".obj"):
            V, T, F, FT = read_obj(f)
            raw[f] = (V, F, T, FT)
        else:
            V, F = read_stl(f)
            raw[f] = (V, F, None, None)
    prim = {PRIM_BOX: unit_box(), PRIM_CYLINDER: unit_cylinder(), PRIM_SPHERE: unit_sphere()}
    # the grid cell is a length in the scene (metres); the files come in millimetres or metres (mesh scale attribute)
    unit = {}
    for m in models:
        for r in instance_table(m, lambda g: np.ones(4)):
            if not r["mesh"].startswith("__"):
                unit[r["mesh"]] = max(unit.get(r["mesh"], 0.0), float(np.abs(r["scale"]).max()))

    def scene_tris(dec, m):
        return sum(len(dec[r["mesh"]][1]) if r["mesh"] in dec else len(prim[r["mesh"]][1]) for r in instance_table(m, lambda g: np.ones(4)))

    def decimate_all(cell):
        out = {}
        for f, (V, F, T, FT) in raw.items():
            out[f] = (V, F) if T is not None else cluster_decimate(V, F, cell / unit[f])      # the small textured OBJ meshes keep their uv mapping
        return out
    lo, hi = 0.0, 0.05
    for _ in range(18):
        mid = 0.5 * (lo + hi)
        worst = max(scene_tris(decimate_all(mid), m) for m in models)
        if worst > budget:
            lo = mid
        else:
            hi = mid
    dec = decimate_all(hi)
    names = list(prim) + files
    ids = {n: i for i, n in enumerate(names)}
    vadr, vnum, tadr, tnum, Vs, Fs, UVs, TNs = [], [], [], [], [], [], [], []
    for n in names:
        if n in prim:
            V, F = prim[n]
            uv = np.zeros((len(F), 6))
        else:
            V, F = dec[n]
            _, _, T, FT = raw[n]
            uv = np.zeros((len(F), 6))
            if T is not None and len(T):
                uv = T[np.maximum(FT, 0)].reshape(len(F), 6)
        vadr.append(sum(len(v) for v in Vs)); vnum.append(len(V)); tadr.append(sum(len(f) for f in Fs)); tnum.append(len(F))
        Vs.append(V); Fs.append(F); UVs.append(uv); TNs.append(corner_normals(V, F).reshape(len(F), 9))
        if verbose:
            print(f"  {os.path.basename(n):40s} {len(raw[n][1]) if n in raw else len(F):7d} -> {len(F):6d} triangles")
    img = read_png_rgb8(texture_png).astype(np.float64)
    h, w, _ = img.shape
    fy, fx = h // TEX_N, w // TEX_N
    tex = img[:fy * TEX_N, :fx * TEX_N].reshape(TEX_N, fy, TEX_N, fx, 3).mean((1, 3))
    arrays = {
        "lib_cell": np.array([hi]), "lib_nmesh": np.array([len(names)], dtype=np.int32),
        "lib_vadr": np.array(vadr, dtype=np.int32), "lib_vnum": np.array(vnum, dtype=np.int32),
        "lib_tadr": np.array(tadr, dtype=np.int32), "lib_tnum": np.array(tnum, dtype=np.int32),
        "lib_vert": np.concatenate(Vs), "lib_tri": np.concatenate(Fs).astype(np.int32), "lib_uv": np.concatenate(UVs),
        "lib_tnorm": np.concatenate(TNs),       # per triangle the three corners' lighting normals, mesh frame (corner_normals; round 6: smooth shading)
        "lib_tex": (lambda t: (t[:, 0] | (t[:, 1] << 8) | (t[:, 2] << 16)).astype(np.int32))(np.round(tex).astype(np.int64).reshape(-1, 3)),   # r | g << 8 | b << 16, row-major from the top
    }
    info = {"cell_m": hi, "meshes": {os.path.basename(n): int(t) for n, t in zip(names, tnum)},
            "scene_triangles": {i: int(scene_tris(dec, m)) for i, m in enumerate(models)}}
    return arrays, {os.path.basename(n) if not n.startswith("__") else n: i for n, i in ids.items()}, info


def expand_instances(lib, inst):
    """Instances x library -> the scene's triangles in body frames (what the device's loader and the oracle draw):
    vert (nv, 3), vbody (nv,), tri (nt, 3), rgb (nt, 3), uv (nt, 6), tex (nt,), tnorm (nt, 9): the corners' lighting normals, body frame."""
    Vs, Bs, Fs, Cs, Us, Ts, Ns = [], [], [], [], [], [], []
    nv = 0
    for k in range(len(inst["vis_inst_mesh"])):
        mid = int(inst["vis_inst_mesh"][k])
        va, vn, ta, tn = (int(lib[a][mid]) for a in ("lib_vadr", "lib_vnum", "lib_tadr", "lib_tnum"))
        V = lib["lib_vert"][va:va + vn] * inst["vis_inst_scale"][k]
        V = V @ inst["vis_inst_mat"][k].reshape(3, 3).T + inst["vis_inst_pos"][k]
        F = lib["lib_tri"][ta:ta + tn]
        # normals go with the inverse transpose: n / scale, then the instance's rotation
        TN = lib["lib_tnorm"][ta:ta + tn].reshape(tn, 3, 3) / inst["vis_inst_scale"][k]
        TN = TN @ inst["vis_inst_mat"][k].reshape(3, 3).T
        TN = TN / np.maximum(np.linalg.norm(TN, axis=2, keepdims=True), 1e-300)
        if np.prod(inst["vis_inst_scale"][k]) < 0:
            F = F[:, ::-1]
            TN = TN[:, ::-1]
        Vs.append(V); Bs.append(np.full(vn, int(inst["vis_inst_body"][k]), dtype=np.int32)); Fs.append(F + nv)
        Cs.append(np.repeat(inst["vis_inst_rgba"][k][None, :3], tn, 0)); Us.append(lib["lib_uv"][ta:ta + tn])
        Ts.append(np.full(tn, int(inst["vis_inst_tex"][k]), dtype=np.int32)); Ns.append(TN.reshape(tn, 9))
        nv += vn
    return (np.concatenate(Vs), np.concatenate(Bs), np.concatenate(Fs).astype(np.int32), np.concatenate(Cs), np.concatenate(Us),
            np.concatenate(Ts), np.concatenate(Ns))
