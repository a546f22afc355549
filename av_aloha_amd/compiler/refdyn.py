"""Small numpy rigid-body routines used by the model compiler (and by tests as
an independent check of the C oracle): forward kinematics, composite-rigid-body
mass matrix and body Jacobians, all in world coordinates with spatial vectors
taken about the world origin, ordered [angular(3); linear(3)].

Semantics follow MuJoCo's documented `mj_kinematics` / `mj_crb` [EXT]; the
reference calls them through `physics.forward()` (env.py:244).
"""
from __future__ import annotations

import numpy as np

from .mjcf import quat_mul, quat_to_mat

JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3


def kinematics(md, qpos):
    nb = md["nbody"]
    xpos = np.zeros((nb, 3))
    xquat = np.zeros((nb, 4))
    xquat[0, 0] = 1
    xmat = np.zeros((nb, 3, 3))
    xmat[0] = np.eye(3)
    nv = md["nv"]
    cdof = np.zeros((nv, 6))
    xanchor = np.zeros((md["njnt"], 3))
    xaxis = np.zeros((md["njnt"], 3))
    for b in range(1, nb):
        p = md["body_parent"][b]
        ja, jn = md["body_jntadr"][b], md["body_jntnum"][b]
        if jn == 1 and md["jnt_type"][ja] == JNT_FREE:
            qa = md["jnt_qposadr"][ja]
            pos = qpos[qa:qa + 3].copy()
            quat = qpos[qa + 3:qa + 7] / np.linalg.norm(qpos[qa + 3:qa + 7])
            R = quat_to_mat(quat)
            da = md["jnt_dofadr"][ja]
            for k in range(3):
                cdof[da + k, 3 + k] = 1.0
                w = R[:, k]
                cdof[da + 3 + k, :3] = w
                cdof[da + 3 + k, 3:] = np.cross(pos, w)
            xanchor[ja] = pos
            xaxis[ja] = R[:, 2]
        else:
            pos = xpos[p] + xmat[p] @ md["body_pos"][b]
            quat = quat_mul(xquat[p], md["body_quat"][b])
            for j in range(ja, ja + jn):
                R = quat_to_mat(quat)
                t = md["jnt_type"][j]
                axis_w = R @ md["jnt_axis"][j]
                anchor = pos + R @ md["jnt_pos"][j]
                q = qpos[md["jnt_qposadr"][j]]
                d = md["jnt_dofadr"][j]
                xanchor[j] = anchor
                xaxis[j] = axis_w
                if t == JNT_HINGE:
                    cdof[d, :3] = axis_w
                    cdof[d, 3:] = np.cross(anchor, axis_w)
                    a = md["jnt_axis"][j]
                    qr = np.concatenate([[np.cos(q / 2)], np.sin(q / 2) * a])
                    quat = quat_mul(quat, qr)
                    pos = anchor - quat_to_mat(quat) @ md["jnt_pos"][j]
                elif t == JNT_SLIDE:
                    cdof[d, 3:] = axis_w
                    pos = pos + axis_w * q
                else:
                    raise NotImplementedError
            quat = quat / np.linalg.norm(quat)
        xpos[b] = pos
        xquat[b] = quat
        xmat[b] = quat_to_mat(quat)
    return {"xpos": xpos, "xquat": xquat, "xmat": xmat, "cdof": cdof,
            "xanchor": xanchor, "xaxis": xaxis}


def body_spatial_inertia(md, kin):
    """Per body: (mass, h = m*c, I_origin 3x3) in world axes about world origin."""
    nb = md["nbody"]
    out = []
    for b in range(nb):
        m = md["body_mass"][b]
        R = kin["xmat"][b]
        c = kin["xpos"][b] + R @ md["body_ipos"][b]
        Ib = md["body_inertia"][b]
        I3 = np.array([[Ib[0], Ib[3], Ib[4]], [Ib[3], Ib[1], Ib[5]], [Ib[4], Ib[5], Ib[2]]])
        Ic = R @ I3 @ R.T
        Io = Ic + m * (np.dot(c, c) * np.eye(3) - np.outer(c, c))
        out.append((m, m * c, Io))
    return out


def inertia_mul(I, s):
    m, h, Io = I
    w, v = s[:3], s[3:]
    L = Io @ w + np.cross(h, v)
    p = m * v - np.cross(h, w)
    return np.concatenate([L, p])


def mass_matrix(md, kin):
    nb, nv = md["nbody"], md["nv"]
    I = body_spatial_inertia(md, kin)
    comp = [[i[0], i[1].copy(), i[2].copy()] for i in I]
    for b in range(nb - 1, 0, -1):
        p = md["body_parent"][b]
        comp[p][0] += comp[b][0]
        comp[p][1] += comp[b][1]
        comp[p][2] += comp[b][2]
    M = np.zeros((nv, nv))
    cdof = kin["cdof"]
    for i in range(nv):
        f = inertia_mul(comp[md["dof_body"][i]], cdof[i])
        j = i
        while j >= 0:
            M[i, j] = M[j, i] = np.dot(cdof[j], f)
            j = md["dof_parent"][j]
        M[i, i] += md["dof_armature"][i]
    return M


def body_jacobian(md, kin, b, point):
    """6 x nv: rows 0-2 translational at `point`, rows 3-5 rotational."""
    nv = md["nv"]
    J = np.zeros((6, nv))
    cdof = kin["cdof"]
    # walk up to the first body owning dofs
    while b > 0 and md["body_dofnum"][b] == 0:
        b = md["body_parent"][b]
    if b == 0:
        return J
    d = md["body_dofadr"][b] + md["body_dofnum"][b] - 1
    while d >= 0:
        w, v = cdof[d, :3], cdof[d, 3:]
        J[:3, d] = v + np.cross(w, point)
        J[3:, d] = w
        d = md["dof_parent"][d]
    return J
