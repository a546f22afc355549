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


# ------------------------------------------------------------------------------------------------------------------------------
# Constraint assembly and the primal problem, restated independently of oracle/orc_dyn.c / orc_newton.c (numpy; used by the
# tests as a second opinion on the C oracle).  Semantics: MuJoCo's documentation, chapter "Computation" [EXT]: solref = (timeconst,
# dampratio) -> stiffness / damping of the reference acceleration, solimp = (d0, dwidth, width, midpoint, power) -> impedance d(r),
# regulariser R = (1 - d) / d * (diagonal approximation of A from the invweight0 fields), elliptic friction cones with impratio.
# ------------------------------------------------------------------------------------------------------------------------------
MINVAL, MINIMP, MAXIMP = 1e-15, 1e-4, 0.9999
ROW_EQ, ROW_FLOSS, ROW_LIMIT, ROW_CONTACT = 0, 1, 2, 3


def impedance(solimp, r):
    """d(r): d0 at r = 0, dwidth at |r| >= width, a power-law sigmoid in between (MuJoCo "solimp")."""
    d0, dw, width, mid, power = (float(x) for x in solimp)
    d0, dw, mid = np.clip(d0, MINIMP, MAXIMP), np.clip(dw, MINIMP, MAXIMP), np.clip(mid, MINIMP, MAXIMP)
    width, power = max(width, MINVAL), max(power, 1.0)
    if d0 == dw:
        return 0.5 * (d0 + dw)
    x = abs(r) / width
    if x >= 1:
        return dw
    if x <= 0:
        return d0
    if power == 1:
        y = x
    elif x <= mid:
        y = (x / mid) ** power * mid
    else:
        y = 1 - ((1 - x) / (1 - mid)) ** power * (1 - mid)
    return d0 + y * (dw - d0)


def stiffness_damping(solref, solimp, timestep):
    """(k, b) of aref = -b v - k d(r) r for solref = (timeconst, dampratio) > 0; timeconst is kept >= 2 timestep."""
    tc, dr = max(float(solref[0]), 2 * timestep), float(solref[1])
    dmax = float(np.clip(solimp[1], MINIMP, MAXIMP))
    return 1.0 / max(MINVAL, dmax * dmax * tc * tc * dr * dr), 2.0 / max(MINVAL, dmax * tc)


def constraints(md, qpos, qvel, contacts, kin=None):
    """All constraint rows of the state in MuJoCo's order (equalities, dry friction, joint limits, contacts).
    contacts: dicts(dist, pos[3], frame[3][3] (rows: normal, tangent 1, tangent 2), geom1, geom2, dim, friction[5], solref[2],
    solimp[5], includemargin) as the collision stage lists them.
    -> dict(J [nefc, nv], aref, R, type, pos, margin, floss, con = [(first row, dim, friction)], ...)"""
    kin = kin or kinematics(md, qpos)
    nv, h = md["nv"], float(md["opt"][0])
    impratio = float(md["opt"][4])
    rows = []          # (type, J row, pos, margin, diag0, solref, solimp, floss)

    def add(t, J, pos, margin, diag0, solref, solimp, floss=0.0):
        rows.append([t, J, pos, margin, diag0, np.asarray(solref, float), np.asarray(solimp, float), floss])
    for e in range(int(md["neq"].reshape(-1)[0]) if hasattr(md["neq"], "reshape") else int(md["neq"])):
        c = md["eq_polycoef"].reshape(-1, 5)[e]
        a1, a2 = int(md["eq_qpos1"][e]), int(md["eq_qpos2"][e])
        q1, q2 = qpos[a1] - md["qpos0"][a1], qpos[a2] - md["qpos0"][a2]
        poly = np.polyval(c[::-1], q2)
        dpoly = np.polyval((c[1:] * np.arange(1, 5))[::-1], q2)
        d1, d2 = int(md["eq_dof1"][e]), int(md["eq_dof2"][e])
        J = np.zeros(nv)
        J[d1], J[d2] = 1.0, -dpoly
        add(ROW_EQ, J, q1 - poly, 0.0, md["dof_invweight0"][d1] + md["dof_invweight0"][d2], md["eq_solref"].reshape(-1, 2)[e], md["eq_solimp"].reshape(-1, 5)[e])
    for k in range(nv):
        if md["dof_frictionloss"][k] > 0:
            J = np.zeros(nv)
            J[k] = 1.0
            add(ROW_FLOSS, J, 0.0, 0.0, md["dof_invweight0"][k], md["dof_solref"].reshape(-1, 2)[k], md["dof_solimp"].reshape(-1, 5)[k], float(md["dof_frictionloss"][k]))
    for j in range(md["njnt"]):
        if not md["jnt_limited"][j]:
            continue
        q, k = qpos[md["jnt_qposadr"][j]], int(md["jnt_dofadr"][j])
        lo, hi = md["jnt_range"].reshape(-1, 2)[j]
        for side, dist in ((-1, q - lo), (1, hi - q)):
            if dist < md["jnt_margin"][j]:
                J = np.zeros(nv)
                J[k] = -side
                add(ROW_LIMIT, J, dist, float(md["jnt_margin"][j]), md["dof_invweight0"][k], md["jnt_solref"].reshape(-1, 2)[j], md["jnt_solimp"].reshape(-1, 5)[j])
    con = []
    biw = md["body_invweight0"].reshape(-1, 2)
    for c in contacts:
        if not c["dist"] < c["includemargin"]:
            con.append(None)
            continue
        b1, b2 = int(md["geom_body"][c["geom1"]]), int(md["geom_body"][c["geom2"]])
        Jd = body_jacobian(md, kin, b2, np.asarray(c["pos"])) - body_jacobian(md, kin, b1, np.asarray(c["pos"]))
        F = np.asarray(c["frame"], float).reshape(3, 3)
        first = len(rows)
        for r in range(c["dim"]):
            Jr = F[r % 3] @ (Jd[:3] if r < 3 else Jd[3:])
            add(ROW_CONTACT, Jr, c["dist"] if r == 0 else 0.0, c["includemargin"], biw[b1, 0] + biw[b2, 0], c["solref"], c["solimp"])
        con.append((first, c["dim"], np.asarray(c["friction"], float)))
    n = len(rows)
    out = {"J": np.array([r[1] for r in rows]).reshape(n, nv), "type": np.array([r[0] for r in rows], dtype=int),
           "pos": np.array([r[2] for r in rows]), "margin": np.array([r[3] for r in rows]), "floss": np.array([r[7] for r in rows]),
           "aref": np.zeros(n), "R": np.zeros(n), "con": con}
    imp = np.zeros(n)
    for i, (t, J, pos, margin, diag0, solref, solimp, floss) in enumerate(rows):
        imp[i] = impedance(solimp, pos - margin)
        out["R"][i] = max(MINVAL, (1 - imp[i]) / imp[i] * diag0)
    kd = [stiffness_damping(r[5], r[6], h) for r in rows]
    for cc in con:
        if cc is None:
            continue
        first, dim, mu = cc
        # elliptic cone: the friction rows carry no position term, take the normal row's impedance, and are regularised by
        # R_normal / impratio (first tangent) scaled with (mu_1 / mu_j)^2 for the others
        for r in range(1, dim):
            i = first + r
            kd[i] = (0.0, kd[i][1])
            imp[i] = imp[first]
            R1 = out["R"][first] / max(MINVAL, impratio)
            out["R"][i] = R1 if r == 1 else R1 * mu[0] ** 2 / max(MINVAL, mu[r - 1] ** 2)
    vel = out["J"] @ qvel
    for i in range(n):
        out["aref"][i] = -kd[i][1] * vel[i] - kd[i][0] * imp[i] * (rows[i][2] - rows[i][3])
    out["imp"] = imp
    return out


def primal_cost(C, M, a_smooth, a, grad=False):
    """MuJoCo's primal objective [EXT]: 1/2 (a - a_s)' M (a - a_s) + sum_i s_i(J_i a - aref_i): quadratic equalities, Huber-type dry
    friction, one-sided limits, three-zone elliptic cones (top: free, bottom: all rows quadratic, middle: 1/2 Dm (N - mu T)^2)."""
    jar = C["J"] @ a - C["aref"]
    D = 1.0 / C["R"]
    cost = 0.5 * (a - a_smooth) @ M @ (a - a_smooth)
    f = np.zeros(len(jar))          # constraint force = - d s / d jar
    in_con = np.zeros(len(jar), bool)
    for cc in C["con"]:
        if cc is not None:
            in_con[cc[0]:cc[0] + cc[1]] = True
    for i in np.nonzero(~in_con)[0]:
        t, z = C["type"][i], jar[i]
        if t == ROW_EQ:
            cost += 0.5 * D[i] * z * z
            f[i] = -D[i] * z
        elif t == ROW_FLOSS:
            eta, R = C["floss"][i], C["R"][i]
            if z <= -R * eta:
                cost += -eta * z - 0.5 * R * eta * eta
                f[i] = eta
            elif z >= R * eta:
                cost += eta * z - 0.5 * R * eta * eta
                f[i] = -eta
            else:
                cost += 0.5 * D[i] * z * z
                f[i] = -D[i] * z
        elif z < 0:
            cost += 0.5 * D[i] * z * z
            f[i] = -D[i] * z
    for cc in C["con"]:
        if cc is None:
            continue
        first, dim, fr = cc
        j = jar[first:first + dim]
        Dj = D[first:first + dim]
        if dim == 1:
            if j[0] < 0:
                cost += 0.5 * Dj[0] * j[0] ** 2
                f[first] = -Dj[0] * j[0]
            continue
        mu = fr[0] * np.sqrt(C["R"][first + 1] / C["R"][first])          # cone in the scaled space where the regulariser is isotropic
        S = np.concatenate([[mu], fr[:dim - 1]])
        U = j * S
        N, T = U[0], np.linalg.norm(U[1:])
        if N >= mu * T or (T <= 0 and N >= 0):
            continue
        if mu * N + T <= 0 or (T <= 0 and N < 0):
            cost += 0.5 * np.sum(Dj * j * j)
            f[first:first + dim] = -Dj * j
            continue
        Dm = Dj[0] / max(1e-15, mu * mu * (1 + mu * mu))
        NT = N - mu * T
        cost += 0.5 * Dm * NT * NT
        f[first] = -Dm * NT * mu
        f[first + 1:first + dim] = -f[first] * (U[1:] / T) * S[1:]
    if not grad:
        return cost
    return cost, M @ (a - a_smooth) - C["J"].T @ f, f
