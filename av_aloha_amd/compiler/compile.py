"""Offline model compiler: MJCF scene -> compact binary model blob (.avm).

Replaces the MuJoCo model compiler the reference invokes through
``mjcf.Physics.from_mjcf_model`` (gym_guided_vision/gym_guided_vision/env.py:53-56).
Run where the reference assets exist:

    python -m av_aloha_amd.compiler.compile --assets <assets dir> --out models/

Emits ``models/<task>_<n>arms.avm`` (named little-endian arrays, see
``write_blob``) and a JSON manifest with the element names the Python host
needs.  The blob is the only thing the C-ABI library and the C oracle read.
"""
from __future__ import annotations

import argparse
import json
import os
import struct

import numpy as np

from . import hull as hullmod
from . import refdyn
from . import vismesh
from .mjcf import parse, quat_mul, quat_to_mat

GEOM_SPHERE, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 2, 5, 6, 7
GTYPE = {"sphere": GEOM_SPHERE, "cylinder": GEOM_CYLINDER, "box": GEOM_BOX, "mesh": GEOM_MESH}


# material colours of the drawn proxies: an explicit rgba wins, then the material's rgba; the textured table material
# (scene.xml:39, small_meta_table_diffuse.png) is drawn in one flat colour, and collision geoms without either (robot
# links, aloha_sim.xml:106-108) take the colour of the visual meshes they stand in for (material "black", aloha_sim.xml:9)
TABLE_RGBA = (0.55, 0.50, 0.44, 1.0)


def geom_colour(m, g):
    if g.get("rgba_given"):
        return np.asarray(g["rgba"], dtype=float)
    mat = g.get("material")
    if mat is not None and mat in m.materials:
        return np.asarray(m.materials[mat] if m.materials[mat] is not None else TABLE_RGBA, dtype=float)
    if g["name"] == "table":
        return np.asarray(TABLE_RGBA)
    return np.asarray(m.materials.get("black", (0.15, 0.15, 0.15, 1.0)), dtype=float)
JTYPE = {"free": 0, "ball": 1, "slide": 2, "hinge": 3}

TASKS = {
    # task key -> (xml, task id, object free joints in qpos order)
    "insert_peg": ("task_insert_peg.xml", 0),
    "slot_insertion": ("task_slot_insertion.xml", 1),
    "sew_needle": ("task_sew_needle.xml", 2),
    "tube_transfer": ("task_tube_transfer.xml", 3),
    "hook_package": ("task_hook_package.xml", 4),
}

# reward geom classes (env.py:425-472, 546-589, 640-690, 738-779, 820-863)
C_LEFT, C_RIGHT, C_TABLE, C_A, C_B, C_C, C_D = 1, 2, 4, 8, 16, 32, 64


def geom_class(task, name):
    c = 0
    if name.startswith("left"):
        c |= C_LEFT
    if name.startswith("right"):
        c |= C_RIGHT
    if name == "table":
        c |= C_TABLE
    if task == "insert_peg":
        if name == "peg":
            c |= C_A
        if name.startswith("hole-"):
            c |= C_B
        if name == "pin":
            c |= C_C
    elif task == "slot_insertion":
        if name == "stick":
            c |= C_A
        if name.startswith("slot-"):
            c |= C_B
        if name == "pin-stick":
            c |= C_C
        if name == "pin-slot":
            c |= C_D
    elif task == "sew_needle":
        if name == "needle":
            c |= C_A
        if name.startswith("wall-"):
            c |= C_B
        if name == "pin-needle":
            c |= C_C
        if name == "pin-wall":
            c |= C_D
    elif task == "tube_transfer":
        if name.startswith("tube1-"):
            c |= C_A
        if name.startswith("tube2-"):
            c |= C_B
        if name == "ball":
            c |= C_C
        if name == "pin":
            c |= C_D
    elif task == "hook_package":
        if name.startswith("package-"):
            c |= C_A
        if name == "hook":
            c |= C_B
        if name == "pin-package":
            c |= C_C
        if name == "pin-hook":
            c |= C_D
    return c


def shape_inertia(g):
    """(volume, unit-density inertia diag about own COM in geom frame)."""
    t, s = g["type"], g["size"]
    if t == "box":
        a, b, c = s[:3]
        V = 8 * a * b * c
        I = V / 3.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    elif t == "sphere":
        r = s[0]
        V = 4.0 / 3.0 * np.pi * r ** 3
        I = 0.4 * V * r * r * np.ones(3)
    elif t == "cylinder":
        r, h = s[0], s[1]
        V = np.pi * r * r * 2 * h
        I = V * np.array([r * r / 4 + h * h / 3, r * r / 4 + h * h / 3, r * r / 2])
    else:
        raise NotImplementedError(f"inertia from geom type {t}")
    return V, I


def write_blob(path, arrays):
    """Format: b'AVSIMMDL' u32 version u32 count, then per entry
    {char name[32]; u32 dtype(0=f64,1=i32); u32 ndim; u32 dims[4]; u64 offset; u64 nbytes},
    then 64-byte aligned payloads.  Offsets are from file start."""
    names = list(arrays.keys())
    header = 16 + len(names) * (32 + 4 + 4 + 16 + 8 + 8)
    off = (header + 63) // 64 * 64
    entries, payload = [], []
    for n in names:
        a = arrays[n]
        if a.dtype.kind == "f":
            a = np.ascontiguousarray(a, dtype="<f8")
            dt = 0
        else:
            a = np.ascontiguousarray(a, dtype="<i4")
            dt = 1
        dims = list(a.shape) + [1] * (4 - a.ndim)
        if a.ndim == 0:
            dims = [1, 1, 1, 1]
        nb = a.nbytes
        entries.append(struct.pack("<32sII4IQQ", n.encode(), dt, max(a.ndim, 1), *dims, off, nb))
        payload.append((off, a.tobytes()))
        off = (off + nb + 63) // 64 * 64
    with open(path, "wb") as f:
        f.write(b"AVSIMMDL")
        f.write(struct.pack("<II", 1, len(names)))
        for e in entries:
            f.write(e)
        for o, b in payload:
            f.seek(o)
            f.write(b)
        f.truncate(off)


def read_blob(path):
    b = open(path, "rb").read()
    assert b[:8] == b"AVSIMMDL"
    ver, cnt = struct.unpack_from("<II", b, 8)
    out = {}
    p = 16
    for _ in range(cnt):
        name, dt, nd, d0, d1, d2, d3, off, nb = struct.unpack_from("<32sII4IQQ", b, p)
        p += 72
        name = name.rstrip(b"\0").decode()
        a = np.frombuffer(b, dtype="<f8" if dt == 0 else "<i4", count=nb // (8 if dt == 0 else 4), offset=off)
        out[name] = a.reshape([d0, d1, d2, d3][:nd]).copy()
    return out


def compile_task(assets, task, num_arms, kmax_default=20, kmax_finger=32, verbose=False, vis_ids=None, kmax_collision=128, xml_path=None):
    xml, task_id = TASKS[task]
    m = parse(xml_path if xml_path is not None else os.path.join(assets, xml))
    if num_arms == 2 and xml_path is None:
        # env.py:60-62, 394-395: hide_middle_arm() rewrites the base body position
        for b in m.bodies:
            if b["name"] == "middle_base_link":
                b["pos"] = np.array([0.0, -2.4, -0.4])
    nb = len(m.bodies)
    md = {"nbody": nb, "njnt": len(m.joints)}
    body_id = {b["name"]: i for i, b in enumerate(m.bodies)}
    jnt_id = {j["name"]: i for i, j in enumerate(m.joints)}

    # ---- joints / dofs ------------------------------------------------------------------
    nq = nv = 0
    jq, jd = [], []
    for j in m.joints:
        jq.append(nq)
        jd.append(nv)
        if j["type"] == "free":
            nq += 7
            nv += 6
        else:
            nq += 1
            nv += 1
    md["nq"], md["nv"] = nq, nv
    md["jnt_type"] = np.array([JTYPE[j["type"]] for j in m.joints], dtype=np.int32)
    md["jnt_body"] = np.array([j["body"] for j in m.joints], dtype=np.int32)
    md["jnt_qposadr"] = np.array(jq, dtype=np.int32)
    md["jnt_dofadr"] = np.array(jd, dtype=np.int32)
    md["jnt_pos"] = np.array([j["pos"] for j in m.joints])
    md["jnt_axis"] = np.array([j["axis"] for j in m.joints])
    md["jnt_limited"] = np.array([int(j["limited"]) for j in m.joints], dtype=np.int32)
    md["jnt_range"] = np.array([j["range"] for j in m.joints])
    md["jnt_actfrclimited"] = np.array([int(j["actfrclimited"]) for j in m.joints], dtype=np.int32)
    md["jnt_actfrcrange"] = np.array([j["actfrcrange"] for j in m.joints])
    md["jnt_solref"] = np.array([j["solreflimit"] for j in m.joints])
    md["jnt_solimp"] = np.array([j["solimplimit"] for j in m.joints])
    md["jnt_margin"] = np.array([j["margin"] for j in m.joints])

    md["body_parent"] = np.array([b["parent"] for b in m.bodies], dtype=np.int32)
    md["body_pos"] = np.array([b["pos"] for b in m.bodies])
    md["body_quat"] = np.array([b["quat"] for b in m.bodies])
    md["body_jntadr"] = np.array([b["joints"][0] if b["joints"] else -1 for b in m.bodies], dtype=np.int32)
    md["body_jntnum"] = np.array([len(b["joints"]) for b in m.bodies], dtype=np.int32)
    body_dofnum = np.array([sum(6 if m.joints[j]["type"] == "free" else 1 for j in b["joints"]) for b in m.bodies], dtype=np.int32)
    body_dofadr = np.array([jd[b["joints"][0]] if b["joints"] else -1 for b in m.bodies], dtype=np.int32)
    md["body_dofnum"], md["body_dofadr"] = body_dofnum, body_dofadr

    # weld ids (body a jointless body is rigidly attached to) and kinematic trees
    weld = np.zeros(nb, dtype=np.int32)
    tree = -np.ones(nb, dtype=np.int32)
    ntree = 0
    for i in range(1, nb):
        p = m.bodies[i]["parent"]
        weld[i] = i if m.bodies[i]["joints"] else weld[p]
        if m.bodies[i]["joints"]:
            if tree[p] >= 0:
                tree[i] = tree[p]
            else:
                tree[i] = ntree
                ntree += 1
        else:
            tree[i] = tree[p]
    md["body_weldid"], md["body_tree"], md["ntree"] = weld, tree, ntree

    dof_body = np.zeros(nv, dtype=np.int32)
    dof_jnt = np.zeros(nv, dtype=np.int32)
    dof_parent = -np.ones(nv, dtype=np.int32)
    arm = np.zeros(nv)
    damp = np.zeros(nv)
    floss = np.zeros(nv)
    dof_solref = np.zeros((nv, 2))
    dof_solimp = np.zeros((nv, 5))
    for ji, j in enumerate(m.joints):
        n = 6 if j["type"] == "free" else 1
        for k in range(n):
            d = jd[ji] + k
            dof_body[d], dof_jnt[d] = j["body"], ji
            arm[d], damp[d], floss[d] = j["armature"], j["damping"], j["frictionloss"]
            dof_solref[d], dof_solimp[d] = j["solreffriction"], j["solimpfriction"]
    # parent dof: previous dof on the same body, else last dof of nearest ancestor with dofs
    for d in range(nv):
        b = dof_body[d]
        if d > body_dofadr[b]:
            dof_parent[d] = d - 1
        else:
            p = m.bodies[b]["parent"]
            while p > 0 and body_dofnum[p] == 0:
                p = m.bodies[p]["parent"]
            dof_parent[d] = body_dofadr[p] + body_dofnum[p] - 1 if p > 0 else -1
    md.update(dof_body=dof_body, dof_jnt=dof_jnt, dof_parent=dof_parent, dof_armature=arm,
              dof_damping=damp, dof_frictionloss=floss, dof_solref=dof_solref, dof_solimp=dof_solimp)
    dof_tree = tree[dof_body]
    md["dof_tree"] = dof_tree.astype(np.int32)
    md["tree_dofadr"] = np.array([int(np.where(dof_tree == t)[0][0]) for t in range(ntree)], dtype=np.int32)
    md["tree_dofnum"] = np.array([int((dof_tree == t).sum()) for t in range(ntree)], dtype=np.int32)
    for t in range(ntree):  # dofs of a tree must be contiguous (block-diagonal M)
        idx = np.where(dof_tree == t)[0]
        assert idx[-1] - idx[0] + 1 == len(idx)

    # ---- body inertias ------------------------------------------------------------------
    mass = np.zeros(nb)
    ipos = np.zeros((nb, 3))
    inertia = np.zeros((nb, 6))
    for i, b in enumerate(m.bodies):
        if i == 0:
            continue
        if b["inertial"] is not None:
            it = b["inertial"]
            if "fullinertia" in it:
                f = it["fullinertia"]
                I3 = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
            else:
                R = quat_to_mat(it["quat"])
                I3 = R @ np.diag(it["diaginertia"]) @ R.T
            mass[i], ipos[i] = it["mass"], it["pos"]
        else:
            # inertiafromgeom (auto): combine all geoms of the body [EXT]
            ms, cs, Is = [], [], []
            for gi in b["geoms"]:
                g = m.geoms[gi]
                if g["type"] == "mesh":
                    if g["mass"] == 0:
                        continue
                    raise NotImplementedError("inertia from mesh geoms")
                V, Iu = shape_inertia(g)
                mg = g["mass"] if g["mass"] is not None else g["density"] * V
                Rg = quat_to_mat(g["quat"])
                Is.append(Rg @ np.diag(Iu * (mg / V)) @ Rg.T)
                ms.append(mg)
                cs.append(g["pos"])
            if not ms:
                continue
            ms, cs = np.array(ms), np.array(cs)
            mass[i] = ms.sum()
            ipos[i] = (ms[:, None] * cs).sum(0) / mass[i]
            I3 = np.zeros((3, 3))
            for mg, c, Ig in zip(ms, cs, Is):
                d = c - ipos[i]
                I3 += Ig + mg * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        inertia[i] = [I3[0, 0], I3[1, 1], I3[2, 2], I3[0, 1], I3[0, 2], I3[1, 2]]
    md.update(body_mass=mass, body_ipos=ipos, body_inertia=inertia)

    # ---- qpos0 / home pose ---------------------------------------------------------------
    qpos0 = np.zeros(nq)
    for ji, j in enumerate(m.joints):
        if j["type"] == "free":
            b = m.bodies[j["body"]]
            qpos0[jq[ji]:jq[ji] + 3] = b["pos"]
            qpos0[jq[ji] + 3:jq[ji] + 7] = b["quat"]
    md["qpos0"] = qpos0

    # ---- actuators -----------------------------------------------------------------------
    nu = len(m.actuators)
    md["nu"] = nu
    md["act_dof"] = np.array([jd[jnt_id[a["joint"]]] for a in m.actuators], dtype=np.int32)
    md["act_qposadr"] = np.array([jq[jnt_id[a["joint"]]] for a in m.actuators], dtype=np.int32)
    md["act_kp"] = np.array([a["kp"] for a in m.actuators])
    md["act_kv"] = np.array([a["kv"] for a in m.actuators])
    md["act_gear"] = np.array([a["gear"] for a in m.actuators])
    md["act_ctrllimited"] = np.array([int(a["ctrllimited"]) for a in m.actuators], dtype=np.int32)
    md["act_ctrlrange"] = np.array([a["ctrlrange"] for a in m.actuators])

    # ---- equalities ----------------------------------------------------------------------
    md["neq"] = len(m.equalities)
    md["eq_dof1"] = np.array([jd[jnt_id[e["joint1"]]] for e in m.equalities], dtype=np.int32)
    md["eq_dof2"] = np.array([jd[jnt_id[e["joint2"]]] for e in m.equalities], dtype=np.int32)
    md["eq_qpos1"] = np.array([jq[jnt_id[e["joint1"]]] for e in m.equalities], dtype=np.int32)
    md["eq_qpos2"] = np.array([jq[jnt_id[e["joint2"]]] for e in m.equalities], dtype=np.int32)
    md["eq_polycoef"] = np.array([e["polycoef"] for e in m.equalities]).reshape(-1, 5)
    md["eq_solref"] = np.array([e["solref"] for e in m.equalities]).reshape(-1, 2)
    md["eq_solimp"] = np.array([e["solimp"] for e in m.equalities]).reshape(-1, 5)

    # ---- invweight0 at qpos0 [EXT: engine_setconst set0] ---------------------------------
    kin = refdyn.kinematics(md, qpos0)
    M = refdyn.mass_matrix(md, kin)
    Minv = np.linalg.inv(M)
    dof_inv = np.diag(Minv).copy()
    for ji, j in enumerate(m.joints):
        if j["type"] == "free":
            d = jd[ji]
            dof_inv[d:d + 3] = dof_inv[d:d + 3].mean()
            dof_inv[d + 3:d + 6] = dof_inv[d + 3:d + 6].mean()
    body_inv = np.zeros((nb, 2))
    for b in range(1, nb):
        if weld[b] == 0:
            continue
        com = kin["xpos"][b] + kin["xmat"][b] @ ipos[b]
        J = refdyn.body_jacobian(md, kin, b, com)
        A = J @ Minv @ J.T
        body_inv[b, 0] = np.trace(A[:3, :3]) / 3
        body_inv[b, 1] = np.trace(A[3:, 3:]) / 3
    md["dof_invweight0"], md["body_invweight0"] = dof_inv, body_inv

    # ---- collision geoms -----------------------------------------------------------------
    cg = [i for i, g in enumerate(m.geoms) if (g["contype"] or g["conaffinity"])]
    mesh_cache = {}
    hull_verts = []
    hull_info = {}
    report = {}

    # Two hulls per collision mesh.  COLLISION (chull_*): up to kmax_collision = 128 vertices (outside error <= 0.3 mm, <= 0.14 mm for the
    # gripper parts, 0.02 mm for the fingers; MuJoCo collides the full hull [EXT] -- oracle faithful mode, profiles/r05_fidelity.json), behind
    # a support table (hull.support_table: cube-map cells -> candidate vertices), so that the narrow phase's support function costs two
    # round trips whatever the vertex count.  DEPTH IMAGES (hull_*): the 20 / 32-vertex polyhedra of rounds 1-4, whose faces, face
    # vertices and edges the rasteriser of convex polyhedra holds in registers and 32 / 64-bit masks (avsim_render.hip.h).
    chull_verts, chull_info, ctab = [], {}, {}

    def get_hull(name, finger):
        if name in hull_info:
            return hull_info[name]
        me = m.meshes[name]
        key = me["file"]
        if key is None:                      # inline vertex set (emit_mjcf.py's restatement of a compiled model)
            key = "inline:" + name
            mesh_cache[key] = me["vertex"]
        if key not in mesh_cache:
            mesh_cache[key] = hullmod.read_stl(key)
        pts = mesh_cache[key] * me["scale"]
        hv, err = hullmod.decimate_hull(pts, kmax_finger if finger else kmax_default)
        adr = sum(len(h) for h in hull_verts)
        hull_verts.append(hv)
        hull_info[name] = (adr, len(hv))
        cv, cerr = _collision_hull(key, tuple(np.atleast_1d(me["scale"]).tolist()), pts, kmax_collision)
        chull_info[name] = (sum(len(h) for h in chull_verts), len(cv))
        chull_verts.append(cv)
        ctab[name] = _support_table(key, tuple(np.atleast_1d(me["scale"]).tolist()), kmax_collision, cv)
        report[name] = {"nvert": int(len(hv)), "err_m": err, "collision_nvert": int(len(cv)), "collision_err_m": cerr, "support_table_R": int(ctab[name][0]),
                        "support_table_candidates_p50_p99_max": [float(x) for x in np.percentile([len(c) for c in ctab[name][1]], [50, 99, 100])]}
        return hull_info[name]

    ng = len(cg)
    g_type = np.zeros(ng, dtype=np.int32)
    g_body = np.zeros(ng, dtype=np.int32)
    g_pos = np.zeros((ng, 3))
    g_quat = np.zeros((ng, 4))
    g_size = np.zeros((ng, 3))
    g_hull = np.zeros((ng, 2), dtype=np.int32)
    g_chull = np.zeros((ng, 2), dtype=np.int32)
    g_condim = np.zeros(ng, dtype=np.int32)
    g_fric = np.zeros((ng, 3))
    g_solref = np.zeros((ng, 2))
    g_solimp = np.zeros((ng, 5))
    g_margin = np.zeros(ng)
    g_gap = np.zeros(ng)
    g_class = np.zeros(ng, dtype=np.int32)
    g_bcen = np.zeros((ng, 3))
    g_rb = np.zeros(ng)
    g_contype = np.zeros(ng, dtype=np.int32)
    g_conaff = np.zeros(ng, dtype=np.int32)
    g_prio = np.zeros(ng, dtype=np.int32)
    g_solmix = np.zeros(ng)
    names = []
    g_rgba = np.zeros((ng, 4))
    for k, gi in enumerate(cg):
        g = m.geoms[gi]
        g_rgba[k] = geom_colour(m, g)
        g_type[k] = GTYPE[g["type"]]
        g_body[k] = g["body"]
        g_pos[k], g_quat[k] = g["pos"], g["quat"]
        s = np.zeros(3)
        s[:len(g["size"])] = g["size"][:3]
        g_size[k] = s
        if g["type"] == "mesh":
            g_hull[k] = get_hull(g["mesh"], "finger" in g["mesh"])
            g_chull[k] = chull_info[g["mesh"]]
            hv = chull_verts[[i for i, n in enumerate(chull_info) if n == g["mesh"]][0]]      # bounding sphere of the COLLISION hull
            lo, hi = hv.min(0), hv.max(0)
            g_bcen[k] = 0.5 * (lo + hi)
            g_rb[k] = np.linalg.norm(hv - g_bcen[k], axis=1).max()
        elif g["type"] == "box":
            g_rb[k] = np.linalg.norm(s)
        elif g["type"] == "sphere":
            g_rb[k] = s[0]
        elif g["type"] == "cylinder":
            g_rb[k] = np.hypot(s[0], s[1])
        g_condim[k] = g["condim"]
        g_fric[k] = g["friction"]
        g_solref[k], g_solimp[k] = g["solref"], g["solimp"]
        g_margin[k], g_gap[k] = g["margin"], g["gap"]
        g_class[k] = geom_class(task, g["name"])
        g_contype[k], g_conaff[k] = g["contype"], g["conaffinity"]
        g_prio[k], g_solmix[k] = g["priority"], g["solmix"]
        names.append(g["name"])
    md.update(ngeom=ng, geom_type=g_type, geom_body=g_body, geom_pos=g_pos, geom_quat=g_quat,
              geom_size=g_size, geom_hull=g_hull, geom_condim=g_condim, geom_friction=g_fric,
              geom_solref=g_solref, geom_solimp=g_solimp, geom_margin=g_margin, geom_gap=g_gap,
              geom_class=g_class, geom_bcenter=g_bcen, geom_rbound=g_rb)
    md["hull_vert"] = np.concatenate(hull_verts) if hull_verts else np.zeros((0, 3))
    # collision hulls and their support tables: chull_cells[cell] = (first candidate << 8) | count, candidates = vertex indices local to the hull
    md["chull_vert"] = np.concatenate(chull_verts) if chull_verts else np.zeros((0, 3))
    md["geom_chull"] = g_chull
    cells_all, cand_all, tab_of = [], [], {}
    for name in chull_info:
        R, cells = ctab[name]
        tab_of[name] = (len(cells_all), R)
        for c in cells:
            assert len(c) < 256 and len(cand_all) < (1 << 23)
            cells_all.append((len(cand_all) << 8) | len(c))
            cand_all.extend(int(i) for i in c)
    md["chull_cells"] = np.array(cells_all, dtype=np.int32) if cells_all else np.zeros(1, dtype=np.int32)
    md["chull_cand"] = np.array(cand_all, dtype=np.int32) if cand_all else np.zeros(1, dtype=np.int32)
    g_ctab = np.zeros((ng, 2), dtype=np.int32)
    for k, gi in enumerate(cg):
        if m.geoms[gi]["type"] == "mesh":
            g_ctab[k] = tab_of[m.geoms[gi]["mesh"]]
    md["geom_ctab"] = g_ctab
    # half-space form of every hull for the depth ray-caster: rows [nx ny nz d], inside = {x : n.x <= d}, geom-local frame
    plane_of = {}
    planes = []
    # ... and what the rasteriser needs to outline a hull in an image: the vertices of every face and the edges with their two faces
    # (vertex indices local to the hull, face indices local to the hull's planes)
    edge_of = {}
    fv_adr, fv_num, fv_idx, hedges = [], [], [], []
    for name, (adr, cnt) in hull_info.items():
        pl, fverts, E = hullmod.hull_topology(md["hull_vert"][adr:adr + cnt])
        assert np.array_equal(pl, hullmod.hull_planes(md["hull_vert"][adr:adr + cnt]))
        plane_of[(adr, cnt)] = (sum(len(q) for q in planes), len(pl))
        planes.append(pl)
        for fv in fverts:
            fv_adr.append(len(fv_idx))
            fv_num.append(len(fv))
            fv_idx.extend(fv)
        edge_of[(adr, cnt)] = (sum(len(q) for q in hedges), len(E))
        hedges.append(E)
    g_hplane = np.zeros((ng, 2), dtype=np.int32)
    g_hedge = np.zeros((ng, 2), dtype=np.int32)
    for k in range(ng):
        if g_type[k] == 7:
            g_hplane[k] = plane_of[(int(g_hull[k][0]), int(g_hull[k][1]))]
            g_hedge[k] = edge_of[(int(g_hull[k][0]), int(g_hull[k][1]))]
    md["geom_hedge"] = g_hedge
    md["hull_face_vadr"] = np.array(fv_adr, dtype=np.int32)
    md["hull_face_vnum"] = np.array(fv_num, dtype=np.int32)
    md["hull_face_vidx"] = np.array(fv_idx, dtype=np.int32)
    md["hull_edge"] = np.concatenate(hedges).astype(np.int32) if hedges else np.zeros((0, 4), dtype=np.int32)
    md["geom_hplane"] = g_hplane
    # depth render proxies: the collision geoms stand in for the visual meshes; reward-only pins (group 3, gap=100) and
    # the 0.6 mm finger pad spheres (inside the finger hulls) are not drawn
    md["geom_visible"] = np.array([0 if (n.startswith("pin") or n[-3:] in ("_g0", "_g1", "_g2")) else 1 for n in names], dtype=np.int32)
    md["hull_plane"] = np.concatenate(planes) if planes else np.zeros((0, 4))
    md["geom_rgba"] = g_rgba
    # lights of the colour render (scene.xml:9 headlight, :48 directional light with MuJoCo's default diffuse 0.7) and the
    # skybox gradient (scene.xml:34): [headlight ambient, headlight diffuse, light diffuse, 0], light direction (world),
    # sky rgb at the zenith, sky rgb at the nadir
    # The spare fourth words hold the directional light's shadow box (round 5): [3] half extent = <statistic extent="0.6"> x MuJoCo's default
    # shadowclip 1 [EXT], [7] [11] [15] its centre = <statistic center="0 -0.1 0.2"> (scene.xml:6, the same in both asset sets)
    # [16] [17] the specular term of the directional light: light specular 0.3 (MJCF default; the headlight's is 0, scene.xml:9) x material specular 0.5
    # (MJCF default: no material of the assets sets one), exponent 128 x shininess 0.5 = 64 [EXT: fixed-function GL, viewer at infinity]
    md["render_light"] = np.array([0.3, 0.6, 0.7, 0.6,  0.0, 0.0, -1.0, 0.0,  0.3, 0.5, 0.7, -0.1,  0.0, 0.0, 0.0, 0.2,  0.15, 64.0, 0.0, 0.0])
    # instances of the visual mesh library (models/visual_meshes.avv, compiler/vismesh.py): what the colour renderer draws
    if vis_ids is not None:
        rows = vismesh.instance_table(m, lambda g: geom_colour(m, g))
        key = lambda r: r["mesh"] if r["mesh"].startswith("__") else os.path.basename(r["mesh"])
        md["vis_inst_mesh"] = np.array([vis_ids[key(r)] for r in rows], dtype=np.int32)
        md["vis_inst_body"] = np.array([r["body"] for r in rows], dtype=np.int32)
        md["vis_inst_pos"] = np.array([r["pos"] for r in rows])
        md["vis_inst_mat"] = np.array([r["mat"].reshape(9) for r in rows])
        md["vis_inst_scale"] = np.array([r["scale"] for r in rows])
        md["vis_inst_rgba"] = np.array([r["rgba"] for r in rows])
        md["vis_inst_tex"] = np.array([r["tex"] for r in rows], dtype=np.int32)

    # ---- candidate pair list -------------------------------------------------------------
    excl = set()
    for a, b in m.excludes:
        excl.add((body_id[a], body_id[b]))
        excl.add((body_id[b], body_id[a]))
    parent = md["body_parent"]

    def weldparent(b):
        return weld[parent[weld[b]]] if weld[b] > 0 else 0

    # reach bound: farthest a point of geom k can be from its tree's root body origin
    def reach(k):
        b = g_body[k]
        r = np.linalg.norm(g_pos[k] + quat_to_mat(g_quat[k]) @ g_bcen[k]) + g_rb[k]
        while b > 0 and weld[b] != 0:
            r += np.linalg.norm(m.bodies[b]["pos"])
            for j in m.bodies[b]["joints"]:
                jj = m.joints[j]
                if jj["type"] == "slide":
                    r += np.abs(jj["range"]).max()
                if jj["type"] == "free":
                    return np.inf
            b = parent[b]
        return r

    def root_origin(k):
        b = g_body[k]
        while weld[b] != 0:
            b = parent[b]
        # b is static (welded to world): world pose from qpos0 kinematics
        return kin["xpos"][b]

    def world_bsphere(k):
        b = g_body[k]
        R = kin["xmat"][b] @ quat_to_mat(g_quat[k])
        c = kin["xpos"][b] + kin["xmat"][b] @ g_pos[k] + R @ g_bcen[k]
        return c, g_rb[k]

    def static_aabb(k):
        b = g_body[k]
        R = kin["xmat"][b] @ quat_to_mat(g_quat[k])
        p = kin["xpos"][b] + kin["xmat"][b] @ g_pos[k]
        if g_type[k] == GEOM_MESH:
            adr, n = g_chull[k]
            v = md["chull_vert"][adr:adr + n] @ R.T + p
        elif g_type[k] == GEOM_BOX:
            s = g_size[k]
            corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]) * s
            v = corners @ R.T + p
        else:
            c, r = world_bsphere(k)
            return c - r, c + r
        return v.min(0), v.max(0)

    pairs = []
    n_pruned = 0
    reaches = [reach(k) for k in range(ng)]
    for a in range(ng):
        for b in range(a + 1, ng):
            b1, b2 = g_body[a], g_body[b]
            w1, w2 = weld[b1], weld[b2]
            if w1 == w2:
                continue
            if w1 != 0 and w2 != 0 and (w1 == weldparent(b2) or w2 == weldparent(b1)):
                continue
            if (b1, b2) in excl:
                continue
            if not ((g_contype[a] & g_conaff[b]) or (g_contype[b] & g_conaff[a])):
                continue
            # conservative reach pruning (build-side addition; never removes a feasible pair)
            ra, rb = reaches[a], reaches[b]
            if np.isfinite(ra) and w2 == 0:
                lo, hi = static_aabb(b)
                o = root_origin(a)
                d = np.linalg.norm(np.maximum(np.maximum(lo - o, o - hi), 0))
                if d > ra:
                    n_pruned += 1
                    continue
            elif np.isfinite(rb) and w1 == 0:
                lo, hi = static_aabb(a)
                o = root_origin(b)
                d = np.linalg.norm(np.maximum(np.maximum(lo - o, o - hi), 0))
                if d > rb:
                    n_pruned += 1
                    continue
            elif np.isfinite(ra) and np.isfinite(rb) and w1 != 0 and w2 != 0:
                if np.linalg.norm(root_origin(a) - root_origin(b)) > ra + rb:
                    n_pruned += 1
                    continue
            pairs.append((a, b))
    npair = len(pairs)
    p_geom = np.array(pairs, dtype=np.int32).reshape(-1, 2)
    p_condim = np.zeros(npair, dtype=np.int32)
    p_fric = np.zeros((npair, 5))
    p_solref = np.zeros((npair, 2))
    p_solimp = np.zeros((npair, 5))
    p_margin = np.zeros(npair)
    p_gap = np.zeros(npair)
    for i, (a, b) in enumerate(pairs):
        # contact parameter mixing [EXT: mj_contactParam]; priorities are all equal here
        assert g_prio[a] == g_prio[b]
        p_condim[i] = max(g_condim[a], g_condim[b])
        f = np.maximum(g_fric[a], g_fric[b])
        p_fric[i] = [f[0], f[0], f[1], f[2], f[2]]
        mix = g_solmix[a] / (g_solmix[a] + g_solmix[b])
        assert g_solref[a][0] > 0 and g_solref[b][0] > 0
        p_solref[i] = mix * g_solref[a] + (1 - mix) * g_solref[b]
        p_solimp[i] = mix * g_solimp[a] + (1 - mix) * g_solimp[b]
        p_margin[i] = max(g_margin[a], g_margin[b])
        p_gap[i] = max(g_gap[a], g_gap[b])
    md.update(npair=npair, pair_geom=p_geom, pair_condim=p_condim, pair_friction=p_fric,
              pair_solref=p_solref, pair_solimp=p_solimp, pair_margin=p_margin, pair_gap=p_gap)
    # what the pair list was made FROM (round 6): the contact filter's inputs, so that compiler/emit_mjcf.py can restate the model as a
    # self-contained MJCF for the MuJoCo pin (tests/test_mujoco_pin.py) without the reference's XML.  The library and the oracle read
    # the blob by name and never look at these.
    md["geom_contype"], md["geom_conaffinity"], md["geom_priority"], md["geom_solmix"] = g_contype, g_conaff, g_prio, g_solmix
    md["exclude_body"] = np.array([(body_id[a], body_id[b]) for a, b in m.excludes], dtype=np.int32).reshape(-1, 2)
    md["site_body"] = np.array([t["body"] for t in m.sites], dtype=np.int32)
    md["site_pos"] = np.array([t["pos"] for t in m.sites]).reshape(-1, 3)
    md["site_quat"] = np.array([t["quat"] for t in m.sites]).reshape(-1, 4)
    md["opt_multiccd"] = np.array([1 if m.option.get("flag_multiccd", "disable") == "enable" else 0], dtype=np.int32)

    # ---- options -------------------------------------------------------------------------
    md["opt"] = np.array([
        0.002,                                   # timestep, env.py:54 / constants.py:20
        0.0, 0.0, -9.81,                         # gravity [EXT default]
        float(m.option.get("impratio", 1)),      # aloha_sim.xml:4
        float(m.option.get("noslip_iterations", 0)),
        1.0 if m.option.get("cone", "pyramidal") == "elliptic" else 0.0,
        float(np.trace(M) / nv),                 # stat.meaninertia at qpos0 [EXT], scales the solver tolerance
    ])
    md["task_id"] = task_id
    md["num_arms"] = num_arms

    # ---- IK constants (kinematics.py:7-15, 28-33) at zero arm pose ------------------------
    from ..constants import (LEFT_JOINT_NAMES, RIGHT_JOINT_NAMES, MIDDLE_JOINT_NAMES,
                             LEFT_EEF_SITE, RIGHT_EEF_SITE, MIDDLE_EEF_SITE,
                             LEFT_ARM_POSE, RIGHT_ARM_POSE, MIDDLE_ARM_POSE,
                             LEFT_GRIPPER_JOINT_NAMES, RIGHT_GRIPPER_JOINT_NAMES)
    kin0 = refdyn.kinematics(md, qpos0)   # arm joints are all zero in qpos0
    site_id = {s["name"]: i for i, s in enumerate(m.sites)}
    ik_n = np.array([6, 6, 7], dtype=np.int32)
    ik_w0 = np.zeros((3, 7, 3))
    ik_p0 = np.zeros((3, 7, 3))
    ik_site0 = np.zeros((3, 4, 4))
    ik_range = np.zeros((3, 7, 2))
    ik_qadr = -np.ones((3, 7), dtype=np.int32)
    for a, (jn, sn) in enumerate([(LEFT_JOINT_NAMES[:6], LEFT_EEF_SITE), (RIGHT_JOINT_NAMES[:6], RIGHT_EEF_SITE),
                                  (MIDDLE_JOINT_NAMES, MIDDLE_EEF_SITE)]):
        for k, n in enumerate(jn):
            j = jnt_id[n]
            ik_w0[a, k] = kin0["xaxis"][j]
            ik_p0[a, k] = kin0["xanchor"][j]
            ik_range[a, k] = m.joints[j]["range"]
            ik_qadr[a, k] = jq[j]
        s = m.sites[site_id[sn]]
        Rb = kin0["xmat"][s["body"]]
        ik_site0[a] = np.eye(4)
        ik_site0[a, :3, :3] = Rb @ quat_to_mat(s["quat"])
        ik_site0[a, :3, 3] = kin0["xpos"][s["body"]] + Rb @ s["pos"]
    md.update(ik_n=ik_n, ik_w0=ik_w0, ik_p0=ik_p0, ik_site0=ik_site0, ik_range=ik_range, ik_qadr=ik_qadr)

    # ---- reset pose & observation gather (env.py:228-244, 168-178; constants.py:26-88) -----
    qhome = qpos0.copy()
    ctrl_home = np.zeros(nu)
    act_id = {a["name"]: i for i, a in enumerate(m.actuators)}
    from ..constants import LEFT_ACTUATOR_NAMES, RIGHT_ACTUATOR_NAMES, MIDDLE_ACTUATOR_NAMES
    grip_lo, grip_hi = m.actuators[act_id["left_gripper"]]["ctrlrange"]
    for jn, an, pose in [(LEFT_JOINT_NAMES, LEFT_ACTUATOR_NAMES, LEFT_ARM_POSE),
                         (RIGHT_JOINT_NAMES, RIGHT_ACTUATOR_NAMES, RIGHT_ARM_POSE),
                         (MIDDLE_JOINT_NAMES, MIDDLE_ACTUATOR_NAMES, MIDDLE_ARM_POSE)]:
        for n, v in zip(jn, pose):
            qhome[jq[jnt_id[n]]] = v
        for n, v in zip(an, pose):
            ctrl_home[act_id[n]] = v
    for n in LEFT_GRIPPER_JOINT_NAMES + RIGHT_GRIPPER_JOINT_NAMES:
        qhome[jq[jnt_id[n]]] = grip_hi           # unnorm(1)
    ctrl_home[act_id["left_gripper"]] = grip_hi
    ctrl_home[act_id["right_gripper"]] = grip_hi
    md["qpos_home"], md["ctrl_home"] = qhome, ctrl_home
    obs_names = LEFT_JOINT_NAMES + RIGHT_JOINT_NAMES + MIDDLE_JOINT_NAMES
    md["obs_qposadr"] = np.array([jq[jnt_id[n]] for n in obs_names], dtype=np.int32)
    md["obs_dofadr"] = np.array([jd[jnt_id[n]] for n in obs_names], dtype=np.int32)
    obs_off = np.zeros(21)
    obs_scale = np.ones(21)
    for k in (6, 13):
        obs_off[k], obs_scale[k] = grip_lo, 1.0 / (grip_hi - grip_lo)
    md["obs_offset"], md["obs_scale"] = obs_off, obs_scale
    md["grip_range"] = np.array([grip_lo, grip_hi])
    # action -> ctrl scatter (env.py:203-215): ctrl order is the actuator order
    md["objects_qposadr"] = np.array([jq[i] for i, j in enumerate(m.joints) if j["type"] == "free"], dtype=np.int32)

    # ---- cameras (for the later render rows) ----------------------------------------------
    md["cam_body"] = np.array([c["body"] for c in m.cameras], dtype=np.int32)
    md["cam_pos"] = np.array([c["pos"] for c in m.cameras])
    md["cam_quat"] = np.array([c["quat"] for c in m.cameras])
    md["cam_fovy"] = np.array([c["fovy"] for c in m.cameras])
    # clip planes in metres: <map znear="0.05"/> x <statistic extent="0.6"/> (scene.xml:6,13); zfar = MuJoCo default 50 x extent
    md["cam_clip"] = np.array([0.05 * 0.6, 50.0 * 0.6])

    manifest = {
        "task": task, "task_id": task_id, "num_arms": num_arms,
        "nq": nq, "nv": nv, "nu": nu, "nbody": nb, "njnt": len(m.joints), "ngeom_total": len(m.geoms),
        "ngeom_collision": ng, "npair": npair, "npair_pruned_by_reach": n_pruned, "ntree": ntree,
        "body_names": [b["name"] for b in m.bodies],
        "joint_names": [j["name"] for j in m.joints],
        "actuator_names": [a["name"] for a in m.actuators],
        "geom_names": names,
        "camera_names": [c["name"] for c in m.cameras],
        "site_names": [t["name"] for t in m.sites],
        "hulls": report,
        "total_mass": float(mass.sum()),
    }
    arrays = {}
    for k, v in md.items():
        arrays[k] = np.asarray(v)
        if arrays[k].dtype.kind in "iub":
            arrays[k] = arrays[k].astype(np.int32)
    return arrays, manifest


_CH_CACHE, _ST_CACHE = {}, {}


def _collision_hull(key, scale, pts, kmax):
    k = (key, scale, kmax)
    if k not in _CH_CACHE:
        _CH_CACHE[k] = hullmod.decimate_hull(pts, kmax)
    return _CH_CACHE[k]


def _support_table(key, scale, kmax, verts):
    k = (key, scale, kmax)
    if k not in _ST_CACHE:
        _ST_CACHE[k] = hullmod.support_table_for(verts, max_count=16)
    return _ST_CACHE[k]


def write_full_hulls(assets, out_dir):
    """`--oracle-hulls full`: the FULL convex hulls (scipy qhull over every STL vertex, scaled as the <mesh> asset says) of the collision
    meshes of all tasks -- what MuJoCo collides [EXT: mesh geoms collide as the convex hull of the mesh] -- for the CPU oracle's faithful
    mode (oracle/orc.h orc_model_set_hulls, tests/orc_ffi.py load_model(hulls="full")).  The device keeps the decimated hulls of the model
    blobs; the oracle has no LDS or register budget to respect.  -> models/oracle_full_hulls.avh (full_vert f64 [n, 3], full_adr, full_num)
    + .json (mesh names in the blob's order)."""
    from .mjcf import parse as _parse
    names, verts = [], []
    for t in TASKS:
        m = _parse(os.path.join(assets, TASKS[t][0]))
        for g in m.geoms:
            if (g["contype"] or g["conaffinity"]) and g.get("mesh") and g["mesh"] not in names:
                me = m.meshes[g["mesh"]]
                pts = hullmod.read_stl(me["file"]) * me["scale"]
                from scipy.spatial import ConvexHull
                names.append(g["mesh"])
                verts.append(pts[ConvexHull(pts).vertices])
    adr = np.cumsum([0] + [len(v) for v in verts[:-1]]).astype(np.int32)
    num = np.array([len(v) for v in verts], dtype=np.int32)
    write_blob(os.path.join(out_dir, "oracle_full_hulls.avh"), {"full_vert": np.concatenate(verts), "full_adr": adr, "full_num": num})
    with open(os.path.join(out_dir, "oracle_full_hulls.json"), "w") as f:
        json.dump({"mesh_names": names, "nvert": [int(x) for x in num],
                   "note": "full qhull vertex sets of the collision meshes (geom frame, metres), for the CPU oracle only"}, f, indent=1)
    print(f"oracle full hulls: {len(names)} meshes, {int(num.sum())} vertices")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--assets", default=None)
    ap.add_argument("--variant", choices=["gym", "data_collection"], default="gym",
                    help="gym: gym_guided_vision/gym_guided_vision/assets (the gym envs, env.py); data_collection: "
                         "data_collection_scripts/assets, the model sim_env.py / record_sim_episodes.py load (constants.py:5 XML_DIR): "
                         "needle and peg without the gym assets' solref=\"0.01 1\" (task_sew_needle.xml:17, task_insert_peg.xml:7), "
                         "ZED cameras with fovy 90 (aloha_sim.xml:357-358); written as models/dc_<task>_3arms.*")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "..", "models"))
    ap.add_argument("--tasks", nargs="*", default=list(TASKS))
    ap.add_argument("--vis-budget", type=int, default=20000, help="triangles of the biggest visual scene after decimation")
    ap.add_argument("--oracle-hulls", choices=["none", "full"], default="none",
                    help="full: also write models/oracle_full_hulls.* -- the undecimated convex hulls of the collision meshes for the CPU oracle's faithful mode")
    ap.add_argument("--only-oracle-hulls", action="store_true", help="write the oracle's full hulls and nothing else")
    args = ap.parse_args()
    if args.assets is None:
        args.assets = {"gym": "/root/reference/gym_guided_vision/gym_guided_vision/assets",
                       "data_collection": "/root/reference/data_collection_scripts/assets"}[args.variant]
    prefix = "dc_" if args.variant == "data_collection" else ""
    os.makedirs(args.out, exist_ok=True)
    if args.oracle_hulls == "full" or args.only_oracle_hulls:
        write_full_hulls(args.assets, args.out)
        if args.only_oracle_hulls:
            return
    # the visual mesh library, shared by every model (the gym and data-collection assets hold the same meshes): decimated so that the
    # biggest scene stays under the triangle budget
    lib_path = os.path.join(args.out, "visual_meshes.avv")
    scenes = [parse(os.path.join(args.assets, TASKS[t][0])) for t in TASKS]
    lib, vis_ids, info = vismesh.build_library(scenes, os.path.join(args.assets, "meshes", "small_meta_table_diffuse.png"), budget=args.vis_budget, verbose=True)
    if args.variant == "gym" or not os.path.exists(lib_path):
        write_blob(lib_path, lib)
        with open(os.path.join(args.out, "visual_meshes.json"), "w") as f:
            json.dump({"mesh_ids": vis_ids, **info}, f, indent=1)
        print(f"visual mesh library: cell {info['cell_m'] * 1e3:.2f} mm, scenes {info['scene_triangles']} triangles, {os.path.getsize(lib_path)} B")
    else:
        vis_ids = json.load(open(os.path.join(args.out, "visual_meshes.json")))["mesh_ids"]
    for t in args.tasks:
        for na in ((3,) if prefix else (2, 3)):          # sim_env.py always simulates the three arms
            arrays, man = compile_task(args.assets, t, na, vis_ids=vis_ids)
            man["variant"] = args.variant
            base = os.path.join(args.out, f"{prefix}{t}_{na}arms")
            write_blob(base + ".avm", arrays)
            with open(base + ".json", "w") as f:
                json.dump(man, f, indent=1)
            print(f"{t}-{na}arms: nq={man['nq']} nv={man['nv']} nu={man['nu']} nbody={man['nbody']} "
                  f"ngeom={man['ngeom_collision']} npair={man['npair']} (pruned {man['npair_pruned_by_reach']}) "
                  f"mass={man['total_mass']:.3f} blob={os.path.getsize(base + '.avm')} B")


if __name__ == "__main__":
    main()
