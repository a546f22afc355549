/* oracle/orc_dyn.c -- rigid-body dynamics, constraint assembly, PGS solve and time stepping of the
 * CPU oracle.  TEST INFRASTRUCTURE ONLY (orc.h).
 *
 * Restates the stages of MuJoCo's mj_step [EXT] that the reference runs 20x per env step
 * (gym_guided_vision/gym_guided_vision/env.py:218) -- SURVEY.md section 8(a) rows P1..P9 -- following
 * the MuJoCo documentation ("Computation" chapter) and, as BASELINE.json's north_star asks, a projected
 * Gauss-Seidel solver on MuJoCo's soft-constraint model instead of MuJoCo's default Newton solver.
 * PARITY UNPINNED against MuJoCo itself (not importable/buildable here, no golden trajectories in the
 * reference); pinned instead by invariants (tests/test_oracle_physics.py) and by the independent numpy
 * routines of av_aloha_amd/compiler/refdyn.py.
 *
 * Spatial vectors are [angular(3); linear(3)] in world axes about the WORLD ORIGIN.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "orc.h"

#define MINVAL 1e-15
#define MINIMP 0.0001
#define MAXIMP 0.9999

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(const double* a, const double* b, double* c) {
    double t0 = a[1] * b[2] - a[2] * b[1], t1 = a[2] * b[0] - a[0] * b[2], t2 = a[0] * b[1] - a[1] * b[0];
    c[0] = t0; c[1] = t1; c[2] = t2;
}
static void quat2mat(const double* q, double* R) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
static void quatmul(const double* a, const double* b, double* c) {
    double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                   a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
    memcpy(c, t, sizeof t);
}
static void quatnorm(double* q) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
    for (int i = 0; i < 4; i++) q[i] /= n;
}
static void mulmat(const double* R, const double* v, double* o) {
    double t0 = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], t1 = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
           t2 = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}

static double* zalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

orc_data* orc_data_new(const orc_model* m) {
    orc_data* d = (orc_data*)calloc(1, sizeof(orc_data));
    d->m = m;
    int nv = m->nv, nb = m->nbody;
    d->qpos = zalloc(m->nq); d->qvel = zalloc(nv); d->ctrl = zalloc(m->nu); d->qacc_warmstart = zalloc(nv);
    d->xpos = zalloc(3 * nb); d->xquat = zalloc(4 * nb); d->xmat = zalloc(9 * nb); d->xipos = zalloc(3 * nb);
    d->ximat = zalloc(9 * nb); d->xanchor = zalloc(3 * m->njnt); d->xaxis = zalloc(3 * m->njnt); d->cdof = zalloc(6 * nv);
    d->geom_xpos = zalloc(3 * m->ngeom); d->geom_xmat = zalloc(9 * m->ngeom);
    d->M = zalloc((size_t)nv * nv); d->L = zalloc((size_t)nv * nv);
    d->qfrc_bias = zalloc(nv); d->qfrc_passive = zalloc(nv); d->qfrc_actuator = zalloc(nv); d->qfrc_smooth = zalloc(nv);
    d->qacc_smooth = zalloc(nv); d->qfrc_constraint = zalloc(nv); d->qacc = zalloc(nv);
    d->efc_J = zalloc((size_t)ORC_MAXEFC * nv); d->efc_B = zalloc((size_t)ORC_MAXEFC * nv);
    d->pgs_iters = 50;
    d->pgs_tol = 0;
    d->solver = 0;
    d->newton_iters = 100; /* MuJoCo default opt.iterations */
    d->newton_tol = 1e-8;
    d->ls_tol = 1e-10;
    d->ls_iters = 50;
    d->pgs_scale = 1.0 / (m->meaninertia * (m->nv > 1 ? m->nv : 1));
    memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
    return d;
}

int orc_capacity(int which) { return which == 0 ? ORC_MAXCON : (which == 1 ? ORC_MAXEFC : (int)sizeof(orc_data)); }

void orc_data_free(orc_data* d) {
    if (!d) return;
    double* p[] = {d->qpos, d->qvel, d->ctrl, d->qacc_warmstart, d->xpos, d->xquat, d->xmat, d->xipos, d->ximat, d->xanchor,
                   d->xaxis, d->cdof, d->geom_xpos, d->geom_xmat, d->M, d->L, d->qfrc_bias, d->qfrc_passive, d->qfrc_actuator,
                   d->qfrc_smooth, d->qacc_smooth, d->qfrc_constraint, d->qacc, d->efc_J, d->efc_B};
    for (size_t i = 0; i < sizeof p / sizeof p[0]; i++) free(p[i]);
    free(d);
}

/* P1: mj_kinematics [EXT] */
void orc_kinematics(orc_data* d) {
    const orc_model* m = d->m;
    d->xquat[0] = 1;
    d->xmat[0] = d->xmat[4] = d->xmat[8] = 1;
    memset(d->cdof, 0, sizeof(double) * 6 * m->nv);
    for (int b = 1; b < m->nbody; b++) {
        int p = m->body_parent[b], ja = m->body_jntadr[b], jn = m->body_jntnum[b];
        double pos[3], quat[4], R[9];
        if (jn == 1 && m->jnt_type[ja] == ORC_FREE) {
            int qa = m->jnt_qposadr[ja], da = m->jnt_dofadr[ja];
            memcpy(pos, d->qpos + qa, 24);
            memcpy(quat, d->qpos + qa + 3, 32);
            quatnorm(quat);
            quat2mat(quat, R);
            for (int k = 0; k < 3; k++) {
                d->cdof[6 * (da + k) + 3 + k] = 1.0;
                double w[3] = {R[k], R[3 + k], R[6 + k]}, c[3];
                cross3(pos, w, c);
                memcpy(d->cdof + 6 * (da + 3 + k), w, 24);
                memcpy(d->cdof + 6 * (da + 3 + k) + 3, c, 24);
            }
            memcpy(d->xanchor + 3 * ja, pos, 24);
            d->xaxis[3 * ja] = R[2]; d->xaxis[3 * ja + 1] = R[5]; d->xaxis[3 * ja + 2] = R[8];
        } else {
            double t[3];
            mulmat(d->xmat + 9 * p, m->body_pos + 3 * b, t);
            for (int k = 0; k < 3; k++) pos[k] = d->xpos[3 * p + k] + t[k];
            quatmul(d->xquat + 4 * p, m->body_quat + 4 * b, quat);
            for (int j = ja; j < ja + jn; j++) {
                quat2mat(quat, R);
                double axis[3], anchor[3];
                mulmat(R, m->jnt_axis + 3 * j, axis);
                mulmat(R, m->jnt_pos + 3 * j, anchor);
                for (int k = 0; k < 3; k++) anchor[k] += pos[k];
                memcpy(d->xanchor + 3 * j, anchor, 24);
                memcpy(d->xaxis + 3 * j, axis, 24);
                double q = d->qpos[m->jnt_qposadr[j]];
                int dof = m->jnt_dofadr[j];
                if (m->jnt_type[j] == ORC_HINGE) {
                    double c[3];
                    cross3(anchor, axis, c);
                    memcpy(d->cdof + 6 * dof, axis, 24);
                    memcpy(d->cdof + 6 * dof + 3, c, 24);
                    const double* a = m->jnt_axis + 3 * j;
                    double s = sin(q / 2), qr[4] = {cos(q / 2), s * a[0], s * a[1], s * a[2]};
                    quatmul(quat, qr, quat);
                    quat2mat(quat, R);
                    mulmat(R, m->jnt_pos + 3 * j, t);
                    for (int k = 0; k < 3; k++) pos[k] = anchor[k] - t[k];
                } else { /* slide */
                    memcpy(d->cdof + 6 * dof + 3, axis, 24);
                    for (int k = 0; k < 3; k++) pos[k] += axis[k] * q;
                }
            }
            quatnorm(quat);
            quat2mat(quat, R);
        }
        memcpy(d->xpos + 3 * b, pos, 24);
        memcpy(d->xquat + 4 * b, quat, 32);
        memcpy(d->xmat + 9 * b, R, 72);
        double t[3];
        mulmat(R, m->body_ipos + 3 * b, t);
        for (int k = 0; k < 3; k++) d->xipos[3 * b + k] = pos[k] + t[k];
    }
    for (int g = 0; g < m->ngeom; g++) {
        int b = m->geom_body[g];
        double t[3], Rg[9];
        mulmat(d->xmat + 9 * b, m->geom_pos + 3 * g, t);
        for (int k = 0; k < 3; k++) d->geom_xpos[3 * g + k] = d->xpos[3 * b + k] + t[k];
        quat2mat(m->geom_quat + 4 * g, Rg);
        const double* Rb = d->xmat + 9 * b;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                d->geom_xmat[9 * g + 3 * i + j] = Rb[3 * i] * Rg[j] + Rb[3 * i + 1] * Rg[3 + j] + Rb[3 * i + 2] * Rg[6 + j];
    }
}

/* spatial inertia about the world origin: {m, h = m c, Io(6: xx yy zz xy xz yz)} */
typedef struct { double m, h[3], I[6]; } sinert;

static void body_inertia(const orc_data* d, int b, sinert* s) {
    const orc_model* m = d->m;
    const double* R = d->xmat + 9 * b;
    const double* Ib = m->body_inertia + 6 * b;
    double I3[9] = {Ib[0], Ib[3], Ib[4], Ib[3], Ib[1], Ib[5], Ib[4], Ib[5], Ib[2]}, T[9], Ic[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[3 * i + j] = R[3 * i] * I3[j] + R[3 * i + 1] * I3[3 + j] + R[3 * i + 2] * I3[6 + j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Ic[3 * i + j] = T[3 * i] * R[3 * j] + T[3 * i + 1] * R[3 * j + 1] + T[3 * i + 2] * R[3 * j + 2];
    double ms = m->body_mass[b];
    const double* c = d->xipos + 3 * b;
    double cc = dot3(c, c);
    s->m = ms;
    for (int k = 0; k < 3; k++) s->h[k] = ms * c[k];
    s->I[0] = Ic[0] + ms * (cc - c[0] * c[0]); s->I[1] = Ic[4] + ms * (cc - c[1] * c[1]); s->I[2] = Ic[8] + ms * (cc - c[2] * c[2]);
    s->I[3] = Ic[1] - ms * c[0] * c[1]; s->I[4] = Ic[2] - ms * c[0] * c[2]; s->I[5] = Ic[5] - ms * c[1] * c[2];
}

/* f = I * s : momentum [angular about origin; linear] of motion s = [w; v] */
static void inert_mul(const sinert* s, const double* mv, double* f) {
    const double *w = mv, *v = mv + 3;
    double hv[3], hw[3];
    cross3(s->h, v, hv);
    cross3(s->h, w, hw);
    f[0] = s->I[0] * w[0] + s->I[3] * w[1] + s->I[4] * w[2] + hv[0];
    f[1] = s->I[3] * w[0] + s->I[1] * w[1] + s->I[5] * w[2] + hv[1];
    f[2] = s->I[4] * w[0] + s->I[5] * w[1] + s->I[2] * w[2] + hv[2];
    for (int k = 0; k < 3; k++) f[3 + k] = s->m * v[k] - hw[k];
}

/* dense Cholesky (lower) of an n x n SPD matrix; zero blocks stay zero */
static int cholesky(const double* A, double* L, int n) {
    memset(L, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; j++) {
        double dd = A[j * n + j];
        for (int k = 0; k < j; k++) dd -= L[j * n + k] * L[j * n + k];
        if (dd <= 0) return -1;
        dd = sqrt(dd);
        L[j * n + j] = dd;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = s / dd;
        }
    }
    return 0;
}
static void chol_solve(const double* L, double* x, int n) {
    for (int i = 0; i < n; i++) {
        double s = x[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
        x[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = x[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
}

/* P2: mj_crb + factorisation [EXT] */
void orc_crb(orc_data* d) {
    const orc_model* m = d->m;
    int nv = m->nv, nb = m->nbody;
    sinert* c = (sinert*)malloc(sizeof(sinert) * nb);
    for (int b = 0; b < nb; b++) body_inertia(d, b, &c[b]);
    for (int b = nb - 1; b > 0; b--) {
        int p = m->body_parent[b];
        c[p].m += c[b].m;
        for (int k = 0; k < 3; k++) c[p].h[k] += c[b].h[k];
        for (int k = 0; k < 6; k++) c[p].I[k] += c[b].I[k];
    }
    memset(d->M, 0, sizeof(double) * nv * nv);
    for (int i = 0; i < nv; i++) {
        double f[6];
        inert_mul(&c[m->dof_body[i]], d->cdof + 6 * i, f);
        for (int j = i; j >= 0; j = m->dof_parent[j]) {
            double v = 0;
            for (int k = 0; k < 6; k++) v += d->cdof[6 * j + k] * f[k];
            d->M[i * nv + j] = d->M[j * nv + i] = v;
        }
        d->M[i * nv + i] += m->dof_armature[i];
    }
    free(c);
    if (cholesky(d->M, d->L, nv)) fprintf(stderr, "orc_crb: mass matrix not positive definite\n");
}

/* spatial cross products for motion (x) and force (x*) vectors, [ang; lin] */
static void cross_motion(const double* v, const double* s, double* o) {
    double a[3], b[3], c[3];
    cross3(v, s, a);
    cross3(v, s + 3, b);
    cross3(v + 3, s, c);
    for (int k = 0; k < 3; k++) { o[k] = a[k]; o[3 + k] = b[k] + c[k]; }
}
static void cross_force(const double* v, const double* f, double* o) {
    double a[3], b[3], c[3];
    cross3(v, f, a);
    cross3(v + 3, f + 3, b);
    cross3(v, f + 3, c);
    for (int k = 0; k < 3; k++) { o[k] = a[k] + b[k]; o[3 + k] = c[k]; }
}

/* P5: mj_comVel + mj_rne(flg_acc = 0) [EXT]: Coriolis, centrifugal and gravity forces */
void orc_rne_bias(orc_data* d) {
    const orc_model* m = d->m;
    int nb = m->nbody, nv = m->nv;
    double* cvel = zalloc(6 * nb);
    double* cacc = zalloc(6 * nb);
    double* cfrc = zalloc(6 * nb);
    for (int k = 0; k < 3; k++) cacc[3 + k] = -m->gravity[k];
    for (int b = 1; b < nb; b++) {
        int p = m->body_parent[b];
        double* v = cvel + 6 * b;
        double* a = cacc + 6 * b;
        memcpy(v, cvel + 6 * p, 48);
        memcpy(a, cacc + 6 * p, 48);
        int da = m->body_dofadr[b], dn = m->body_dofnum[b];
        int j = 0;
        while (j < dn) {
            int dof = da + j;
            int jt = m->jnt_type[m->dof_jnt[dof]];
            if (jt == ORC_FREE) {
                /* translational dofs: world-fixed axes, cdof_dot = 0; then the three rotational dofs all use
                 * the velocity accumulated so far (parent + translation) */
                for (int k = 0; k < 3; k++)
                    for (int r = 0; r < 6; r++) v[r] += d->cdof[6 * (dof + k) + r] * d->qvel[dof + k];
                double dot[3][6];
                for (int k = 0; k < 3; k++) cross_motion(v, d->cdof + 6 * (dof + 3 + k), dot[k]);
                for (int k = 0; k < 3; k++)
                    for (int r = 0; r < 6; r++) {
                        a[r] += dot[k][r] * d->qvel[dof + 3 + k];
                        v[r] += d->cdof[6 * (dof + 3 + k) + r] * d->qvel[dof + 3 + k];
                    }
                j += 6;
            } else {
                double dot[6];
                cross_motion(v, d->cdof + 6 * dof, dot);
                for (int r = 0; r < 6; r++) {
                    a[r] += dot[r] * d->qvel[dof];
                    v[r] += d->cdof[6 * dof + r] * d->qvel[dof];
                }
                j += 1;
            }
        }
        sinert s;
        body_inertia(d, b, &s);
        double Ia[6], Iv[6], vIv[6];
        inert_mul(&s, a, Ia);
        inert_mul(&s, v, Iv);
        cross_force(v, Iv, vIv);
        for (int r = 0; r < 6; r++) cfrc[6 * b + r] = Ia[r] + vIv[r];
    }
    for (int b = nb - 1; b > 0; b--) {
        int p = m->body_parent[b];
        for (int r = 0; r < 6; r++) cfrc[6 * p + r] += cfrc[6 * b + r];
    }
    for (int i = 0; i < nv; i++) {
        double s = 0;
        for (int r = 0; r < 6; r++) s += d->cdof[6 * i + r] * cfrc[6 * m->dof_body[i] + r];
        d->qfrc_bias[i] = s;
    }
    free(cvel); free(cacc); free(cfrc);
}

/* P5 passive + P6 actuation + P7 smooth acceleration */
static void smooth_forces(orc_data* d) {
    const orc_model* m = d->m;
    int nv = m->nv;
    for (int i = 0; i < nv; i++) {
        d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
        d->qfrc_actuator[i] = 0;
    }
    for (int u = 0; u < m->nu; u++) {
        double c = d->ctrl[u];
        if (m->act_ctrllimited[u]) {
            if (c < m->act_ctrlrange[2 * u]) c = m->act_ctrlrange[2 * u];
            if (c > m->act_ctrlrange[2 * u + 1]) c = m->act_ctrlrange[2 * u + 1];
        }
        int dof = m->act_dof[u];
        /* position actuator: gain kp, bias (0, -kp, -kv) [EXT] */
        double f = m->act_kp[u] * c - m->act_kp[u] * d->qpos[m->act_qposadr[u]] - m->act_kv[u] * d->qvel[dof];
        d->qfrc_actuator[dof] += m->act_gear[u] * f;
    }
    for (int j = 0; j < m->njnt; j++)
        if (m->jnt_actfrclimited[j] && m->jnt_type[j] != ORC_FREE) {
            int dof = m->jnt_dofadr[j];
            double lo = m->jnt_actfrcrange[2 * j], hi = m->jnt_actfrcrange[2 * j + 1];
            if (d->qfrc_actuator[dof] < lo) d->qfrc_actuator[dof] = lo;
            if (d->qfrc_actuator[dof] > hi) d->qfrc_actuator[dof] = hi;
        }
    for (int i = 0; i < nv; i++) {
        d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
        d->qacc_smooth[i] = d->qfrc_smooth[i];
    }
    chol_solve(d->L, d->qacc_smooth, nv);
}

void orc_smooth(orc_data* d) { smooth_forces(d); }   /* exposed for the stage-by-stage tests */

/* translational (rows 0..2) and rotational (3..5) Jacobian of body b at world point p: 6 x nv */
static void body_jac(const orc_data* d, int b, const double* p, double* J) {
    const orc_model* m = d->m;
    int nv = m->nv;
    memset(J, 0, sizeof(double) * 6 * nv);
    while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parent[b];
    if (b == 0) return;
    for (int dof = m->body_dofadr[b] + m->body_dofnum[b] - 1; dof >= 0; dof = m->dof_parent[dof]) {
        const double *w = d->cdof + 6 * dof, *v = w + 3;
        double c[3];
        cross3(w, p, c);
        for (int k = 0; k < 3; k++) { J[k * nv + dof] = v[k] + c[k]; J[(3 + k) * nv + dof] = w[k]; }
    }
}

/* impedance from solimp and violation depth [EXT: getimpedance] */
static double impedance(const double* si, double pos, double margin) {
    double dmin = si[0], dmax = si[1], width = si[2], mid = si[3], power = si[4];
    if (dmin < MINIMP) dmin = MINIMP; if (dmin > MAXIMP) dmin = MAXIMP;
    if (dmax < MINIMP) dmax = MINIMP; if (dmax > MAXIMP) dmax = MAXIMP;
    if (width < MINVAL) width = MINVAL;
    if (mid < MINIMP) mid = MINIMP; if (mid > MAXIMP) mid = MAXIMP;
    if (power < 1) power = 1;
    if (dmin == dmax) return 0.5 * (dmin + dmax);
    double x = fabs(pos - margin) / width;
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    double y;
    if (power == 1) y = x;
    else if (x <= mid) y = pow(x / mid, power) * mid;   /* a*x^p with a = 1/mid^(p-1) */
    else y = 1 - pow((1 - x) / (1 - mid), power) * (1 - mid);
    return dmin + y * (dmax - dmin);
}

static int add_row(orc_data* d, int type, int id, double pos, double margin, double diag0, const double* solref,
                   const double* solimp, double floss) {
    if (d->nefc >= ORC_MAXEFC) { d->overflow = 1; return -1; }
    int i = d->nefc++;
    const orc_model* m = d->m;
    memset(d->efc_J + (size_t)i * m->nv, 0, sizeof(double) * m->nv);
    d->efc_type[i] = type; d->efc_id[i] = id; d->efc_pos[i] = pos; d->efc_margin[i] = margin; d->efc_floss[i] = floss;
    /* K, B from solref (timeconst, dampratio), clamped timeconst >= 2h (refsafe) [EXT: mj_makeImpedance] */
    double dmax = solimp[1];
    if (dmax < MINIMP) dmax = MINIMP; if (dmax > MAXIMP) dmax = MAXIMP;
    double tc = solref[0], dr = solref[1];
    if (tc < 2 * m->timestep) tc = 2 * m->timestep;
    double K = 1.0 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr), B = 2.0 / fmax(MINVAL, dmax * tc);
    double imp = impedance(solimp, pos, margin);
    d->efc_KBIP[4 * i] = K; d->efc_KBIP[4 * i + 1] = B; d->efc_KBIP[4 * i + 2] = imp; d->efc_KBIP[4 * i + 3] = 0;
    d->efc_R[i] = fmax(MINVAL, (1 - imp) * diag0 / imp);
    return i;
}

/* P4: mj_makeConstraint + mj_makeImpedance + reference acceleration [EXT] */
void orc_make_constraints(orc_data* d) {
    const orc_model* m = d->m;
    int nv = m->nv;
    d->nefc = 0;
    /* equality: joint coupling q1 - poly(q2) = 0 (aloha_sim.xml:376-379) */
    for (int e = 0; e < m->neq; e++) {
        const double* c = m->eq_polycoef + 5 * e;
        double q1 = d->qpos[m->eq_qpos1[e]] - m->qpos0[m->eq_qpos1[e]], q2 = d->qpos[m->eq_qpos2[e]] - m->qpos0[m->eq_qpos2[e]];
        double poly = c[0] + q2 * (c[1] + q2 * (c[2] + q2 * (c[3] + q2 * c[4])));
        double dpoly = c[1] + q2 * (2 * c[2] + q2 * (3 * c[3] + q2 * 4 * c[4]));
        int d1 = m->eq_dof1[e], d2 = m->eq_dof2[e];
        int i = add_row(d, ORC_EQ, e, q1 - poly, 0, m->dof_invweight0[d1] + m->dof_invweight0[d2], m->eq_solref + 2 * e, m->eq_solimp + 5 * e, 0);
        if (i < 0) return;
        d->efc_J[(size_t)i * nv + d1] = 1;
        d->efc_J[(size_t)i * nv + d2] = -dpoly;
    }
    /* dry joint friction */
    for (int k = 0; k < nv; k++)
        if (m->dof_frictionloss[k] > 0) {
            int i = add_row(d, ORC_FLOSS, k, 0, 0, m->dof_invweight0[k], m->dof_solref + 2 * k, m->dof_solimp + 5 * k, m->dof_frictionloss[k]);
            if (i < 0) return;
            d->efc_J[(size_t)i * nv + k] = 1;
        }
    /* joint limits (autolimits: every ranged hinge/slide) */
    for (int j = 0; j < m->njnt; j++) {
        if (!m->jnt_limited[j]) continue;
        double q = d->qpos[m->jnt_qposadr[j]];
        for (int side = -1; side <= 1; side += 2) {
            double dist = side < 0 ? q - m->jnt_range[2 * j] : m->jnt_range[2 * j + 1] - q;
            if (dist < m->jnt_margin[j]) {
                int k = m->jnt_dofadr[j];
                int i = add_row(d, ORC_LIMIT, j, dist, m->jnt_margin[j], m->dof_invweight0[k], m->jnt_solref + 2 * j, m->jnt_solimp + 5 * j, 0);
                if (i < 0) return;
                d->efc_J[(size_t)i * nv + k] = -side;
            }
        }
    }
    /* contacts, elliptic cones: condim rows each (normal, 2 tangents, torsion, 2 rolling) */
    double* J1 = zalloc(6 * nv);
    double* J2 = zalloc(6 * nv);
    for (int ci = 0; ci < d->ncon; ci++) {
        orc_contact* c = &d->contact[ci];
        c->efc_adr = -1;
        if (!(c->dist < c->includemargin)) continue; /* gap=100 "pin" geoms: listed, never active */
        if (d->nefc + c->dim > ORC_MAXEFC) { d->overflow = 1; break; }
        int b1 = m->geom_body[c->geom1], b2 = m->geom_body[c->geom2];
        body_jac(d, b1, c->pos, J1);
        body_jac(d, b2, c->pos, J2);
        double tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
        int first = -1;
        for (int r = 0; r < c->dim; r++) {
            int i = add_row(d, ORC_CONTACT, ci, r == 0 ? c->dist : 0, c->includemargin, tran, c->solref, c->solimp, 0);
            if (r == 0) { first = i; c->efc_adr = i; }
            const double* ax = c->frame + 3 * (r % 3);
            int off = r < 3 ? 0 : 3;
            for (int k = 0; k < nv; k++) {
                double s = 0;
                for (int q = 0; q < 3; q++) s += ax[q] * (J2[(off + q) * nv + k] - J1[(off + q) * nv + k]);
                d->efc_J[(size_t)i * nv + k] = s;
            }
            if (r > 0) {
                /* friction rows: no position term, impedance of the normal row, R scaled by impratio and by
                 * the friction-coefficient ratios [EXT: mj_makeImpedance, elliptic branch] */
                d->efc_KBIP[4 * i] = 0;
                d->efc_KBIP[4 * i + 2] = d->efc_KBIP[4 * first + 2];
                double R1 = d->efc_R[first] / fmax(MINVAL, m->impratio);
                double mu0 = c->friction[0], mur = c->friction[r - 1];
                d->efc_R[i] = (r == 1) ? R1 : R1 * mu0 * mu0 / fmax(MINVAL, mur * mur);
            }
        }
    }
    free(J1); free(J2);
    /* reference acceleration, D, and the rows of J M^-1 */
    for (int i = 0; i < d->nefc; i++) {
        const double* Jr = d->efc_J + (size_t)i * nv;
        double vel = 0;
        for (int k = 0; k < nv; k++) vel += Jr[k] * d->qvel[k];
        const double* kb = d->efc_KBIP + 4 * i;
        d->efc_aref[i] = -kb[1] * vel - kb[0] * kb[2] * (d->efc_pos[i] - d->efc_margin[i]);
        d->efc_D[i] = 1.0 / d->efc_R[i];
        double* Br = d->efc_B + (size_t)i * nv;
        memcpy(Br, Jr, sizeof(double) * nv);
        chol_solve(d->L, Br, nv);
        double dg = 0;
        for (int k = 0; k < nv; k++) dg += Jr[k] * Br[k];
        d->efc_diag[i] = dg;
    }
}

static double row_res(const orc_data* d, int i, int with_R) {
    int nv = d->m->nv;
    const double* Jr = d->efc_J + (size_t)i * nv;
    double s = 0;
    for (int k = 0; k < nv; k++) s += Jr[k] * d->qacc[k];
    s -= d->efc_aref[i];
    if (with_R) s += d->efc_R[i] * d->efc_force[i];
    return s;
}
static void apply_delta(orc_data* d, int i, double delta) {
    if (delta == 0) return;
    int nv = d->m->nv;
    const double* Br = d->efc_B + (size_t)i * nv;
    for (int k = 0; k < nv; k++) d->qacc[k] += Br[k] * delta;
}

/* one noslip row update: force -> newf, acceleration follows; *imp collects the decrease of the dual cost,
 * -(delta res + 1/2 A_ii delta^2) with the residual at the time of the update [EXT: costChange in mj_solNoSlip] */
static void noslip_set(orc_data* d, int i, double newf, double* imp) {
    double dl = newf - d->efc_force[i];
    if (dl == 0) return;
    *imp -= dl * row_res(d, i, 0) + 0.5 * d->efc_diag[i] * dl * dl;
    apply_delta(d, i, dl);
    d->efc_force[i] = newf;
}

/* scale the friction components of contact rows [i0+1, i0+dim) back onto the elliptic cone */
static void cone_project(orc_data* d, const orc_contact* c, int i0, double* imp) {
    double fn = d->efc_force[i0], s2 = 0;
    for (int r = 1; r < c->dim; r++) {
        double t = d->efc_force[i0 + r] / fmax(MINVAL, c->friction[r - 1]);
        s2 += t * t;
    }
    if (s2 > fn * fn) {
        double sc = fn / sqrt(s2), dummy = 0;
        for (int r = 1; r < c->dim; r++) noslip_set(d, i0 + r, d->efc_force[i0 + r] * sc, imp ? imp : &dummy);
    }
}


/* P8: projected Gauss-Seidel on the dual of MuJoCo's soft-constraint problem, carried in acceleration
 * space: qacc = qacc_smooth + M^-1 J^T f is kept current, row residual = J_i qacc - aref_i + R_i f_i. */
void orc_solve(orc_data* d) {
    const orc_model* m = d->m;
    int nv = m->nv, ne = d->nefc;
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    /* warm start: forces implied by last step's acceleration, f = -D (J qacc_ws - aref), made feasible */
    for (int i = 0; i < ne; i++) {
        const double* Jr = d->efc_J + (size_t)i * nv;
        double s = 0;
        for (int k = 0; k < nv; k++) s += Jr[k] * d->qacc_warmstart[k];
        double f = -d->efc_D[i] * (s - d->efc_aref[i]);
        switch (d->efc_type[i]) {
            case ORC_FLOSS: if (f > d->efc_floss[i]) f = d->efc_floss[i]; if (f < -d->efc_floss[i]) f = -d->efc_floss[i]; break;
            case ORC_LIMIT: if (f < 0) f = 0; break;
            case ORC_CONTACT: if (i == d->contact[d->efc_id[i]].efc_adr && f < 0) f = 0; break;
            default: break;
        }
        d->efc_force[i] = f;
    }
    for (int ci = 0; ci < d->ncon; ci++) {
        const orc_contact* c = &d->contact[ci];
        if (c->efc_adr < 0) continue;
        double fn = d->efc_force[c->efc_adr], s2 = 0;
        for (int r = 1; r < c->dim; r++) { double t = d->efc_force[c->efc_adr + r] / fmax(MINVAL, c->friction[r - 1]); s2 += t * t; }
        if (s2 > fn * fn) { double sc = fn / sqrt(s2); for (int r = 1; r < c->dim; r++) d->efc_force[c->efc_adr + r] *= sc; }
    }
    for (int i = 0; i < ne; i++) apply_delta(d, i, d->efc_force[i]);

    d->stat_sweeps = 0;
    for (int it = 0; it < d->pgs_iters; it++) {
        /* early termination as MuJoCo's PGS [EXT]: stop when the scaled decrease of the dual cost over one sweep,
         * sum_i 1/2 (A_ii + R_i) delta_i^2 / (meaninertia * nv), drops below the tolerance (0 disables the test) */
        double improvement = 0;
        d->stat_sweeps++;
        for (int i = 0; i < ne; i++) {
            double fprev = d->efc_force[i];
            double f = d->efc_force[i] - row_res(d, i, 1) / (d->efc_diag[i] + d->efc_R[i]);
            int t = d->efc_type[i];
            if (t == ORC_FLOSS) { if (f > d->efc_floss[i]) f = d->efc_floss[i]; if (f < -d->efc_floss[i]) f = -d->efc_floss[i]; }
            else if (t == ORC_LIMIT) { if (f < 0) f = 0; }
            else if (t == ORC_CONTACT) {
                const orc_contact* c = &d->contact[d->efc_id[i]];
                if (i == c->efc_adr && f < 0) f = 0;
                apply_delta(d, i, f - d->efc_force[i]);
                d->efc_force[i] = f;
                improvement += 0.5 * (d->efc_diag[i] + d->efc_R[i]) * (f - fprev) * (f - fprev);
                if (i == c->efc_adr + c->dim - 1 && c->dim > 1) cone_project(d, c, c->efc_adr, NULL);
                continue;
            }
            apply_delta(d, i, f - d->efc_force[i]);
            d->efc_force[i] = f;
            improvement += 0.5 * (d->efc_diag[i] + d->efc_R[i]) * (f - fprev) * (f - fprev);
        }
        if (d->pgs_tol > 0 && improvement * d->pgs_scale < d->pgs_tol) break;
    }
    orc_noslip(d);
}

/* mju_QCQP2 [EXT]: minimise 1/2 x'Ax + x'b subject to sum (x_i / d_i)^2 <= r^2, two unknowns.  The problem is scaled so that the
 * constraint becomes |y| <= r, then Newton's method on the multiplier la of the constraint: y(la) = -(A + la I)^-1 b, root of
 * |y|^2 - r^2.  Returns 1 when the constraint is active (la != 0). */
static int qcqp2(double* res, const double* Ain, const double* bin, const double* d, double r) {
    double b1 = bin[0] * d[0], b2 = bin[1] * d[1];
    double A11 = Ain[0] * d[0] * d[0], A22 = Ain[3] * d[1] * d[1], A12 = Ain[1] * d[0] * d[1];
    double la = 0, v1 = 0, v2 = 0;
    for (int iter = 0; iter < 20; iter++) {
        double det = (A11 + la) * (A22 + la) - A12 * A12;
        if (det < 1e-10) { res[0] = res[1] = 0; return 0; }
        double detinv = 1 / det, P11 = (A22 + la) * detinv, P22 = (A11 + la) * detinv, P12 = -A12 * detinv;
        v1 = -P11 * b1 - P12 * b2;
        v2 = -P12 * b1 - P22 * b2;
        double val = v1 * v1 + v2 * v2 - r * r;
        if (val < 1e-10) break;
        double deriv = -2 * (P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2);
        double delta = -val / deriv;
        if (delta < 1e-10) break;
        la += delta;
    }
    res[0] = v1 * d[0];
    res[1] = v2 * d[1];
    return la != 0;
}

/* mju_QCQP [EXT]: the same for n <= 5 unknowns, (A + la I) by Cholesky (pivots below 1e-10: singular, result 0) */
static int qcqpn(double* res, const double* Ain, const double* bin, const double* d, double r, int n) {
    double A[25], b[5], L[25], v[5], w[5], la = 0;
    for (int i = 0; i < n; i++) {
        b[i] = bin[i] * d[i];
        for (int j = 0; j < n; j++) A[n * i + j] = Ain[n * i + j] * d[i] * d[j];
    }
    for (int i = 0; i < n; i++) v[i] = 0;
    for (int iter = 0; iter < 20; iter++) {
        /* L L' = A + la I */
        for (int j = 0; j < n; j++) {
            double dd = A[n * j + j] + la;
            for (int k = 0; k < j; k++) dd -= L[n * j + k] * L[n * j + k];
            if (dd < 1e-10) { for (int i = 0; i < n; i++) res[i] = 0; return 0; }
            dd = sqrt(dd);
            L[n * j + j] = dd;
            for (int i = j + 1; i < n; i++) {
                double t = A[n * i + j];
                for (int k = 0; k < j; k++) t -= L[n * i + k] * L[n * j + k];
                L[n * i + j] = t / dd;
            }
        }
        /* v = -(A + la I)^-1 b */
        for (int i = 0; i < n; i++) { double t = -b[i]; for (int k = 0; k < i; k++) t -= L[n * i + k] * v[k]; v[i] = t / L[n * i + i]; }
        for (int i = n - 1; i >= 0; i--) { double t = v[i]; for (int k = i + 1; k < n; k++) t -= L[n * k + i] * v[k]; v[i] = t / L[n * i + i]; }
        double val = -r * r;
        for (int i = 0; i < n; i++) val += v[i] * v[i];
        if (val < 1e-10) break;
        /* deriv = -2 v' (A + la I)^-1 v */
        for (int i = 0; i < n; i++) { double t = v[i]; for (int k = 0; k < i; k++) t -= L[n * i + k] * w[k]; w[i] = t / L[n * i + i]; }
        for (int i = n - 1; i >= 0; i--) { double t = w[i]; for (int k = i + 1; k < n; k++) t -= L[n * k + i] * w[k]; w[i] = t / L[n * i + i]; }
        double deriv = 0;
        for (int i = 0; i < n; i++) deriv += v[i] * w[i];
        deriv *= -2;
        double delta = -val / deriv;
        if (delta < 1e-10) break;
        la += delta;
    }
    for (int i = 0; i < n; i++) res[i] = v[i] * d[i];
    return la != 0;
}

/* Inverse of the scaled friction block As = D A D (n x n, n <= 5) through its Cholesky factor; 0 when a pivot falls below 1e-10
 * (the singularity rule of mju_QCQP).  Computed once per contact and substep: at the multiplier 0 the QCQP's candidate is
 * y = -As^-1 (D b), and most resting contacts end there (inside the cone).  Same loops as the device (qc_inverse5). */
static int qc_inverse(const double* Ac, const double* dsc, int n, double* Inv) {
    double A[25], L[25];
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) A[5 * i + j] = (i < n && j < n) ? Ac[n * i + j] * dsc[i] * dsc[j] : (i == j ? 1.0 : 0.0);
    for (int j = 0; j < 5; j++) {
        double dd = A[5 * j + j];
        for (int k = 0; k < j; k++) dd -= L[5 * j + k] * L[5 * j + k];
        if (j < n && dd < 1e-10) return 0;
        dd = sqrt(dd);
        L[5 * j + j] = dd;
        for (int i = j + 1; i < 5; i++) {
            double t = A[5 * i + j];
            for (int k = 0; k < j; k++) t -= L[5 * i + k] * L[5 * j + k];
            L[5 * i + j] = t / dd;
        }
    }
    for (int c = 0; c < 5; c++) {
        double x[5];
        for (int i = 0; i < 5; i++) { double t = (i == c) ? 1.0 : 0.0; for (int k = 0; k < i; k++) t -= L[5 * i + k] * x[k]; x[i] = t / L[5 * i + i]; }
        for (int i = 4; i >= 0; i--) { double t = x[i]; for (int k = i + 1; k < 5; k++) t -= L[5 * k + i] * x[k]; x[i] = t / L[5 * i + i]; }
        for (int i = 0; i < 5; i++) Inv[5 * i + c] = x[i];
    }
    return 1;
}

void orc_noslip(orc_data* d) {
    const orc_model* m = d->m;
    int nv = m->nv, ne = d->nefc;
    /* mj_solNoSlip [EXT] (aloha_sim.xml:4 noslip_iterations=3): Gauss-Seidel sweeps over the dry-friction rows and the friction
     * blocks of the contacts with the regulariser R removed, normal forces held fixed.  A dry-friction row is a clamped scalar
     * update; the friction block of an elliptic contact (2 or 5 rows) is the exact minimiser of its quadratic over the cone
     * section sum (f_j / mu_j)^2 <= f_normal^2 (mju_QCQP2 / mju_QCQP), put exactly on the ellipsoid when the constraint is active.
     * An update that would increase the cost by more than 1e-10 is undone (costChange); a sweep whose scaled improvement falls
     * below noslip_tolerance (MuJoCo default 1e-6, not set by the reference's XML) ends the pass. */
    const double noslip_tolerance = 1e-6;
    for (int it = 0; it < m->noslip_iterations; it++) {
        double imp = 0;
        for (int i = 0; i < ne; i++) {
            int t = d->efc_type[i];
            if (t == ORC_FLOSS) {
                double res = row_res(d, i, 0), A = d->efc_diag[i];
                double f = d->efc_force[i] - res / fmax(MINVAL, A);
                if (f > d->efc_floss[i]) f = d->efc_floss[i];
                if (f < -d->efc_floss[i]) f = -d->efc_floss[i];
                double dl = f - d->efc_force[i], change = 0.5 * A * dl * dl + dl * res;
                if (change > 1e-10 || dl == 0) continue;
                apply_delta(d, i, dl);
                d->efc_force[i] = f;
                imp -= change;
            } else if (t == ORC_CONTACT) {
                const orc_contact* c = &d->contact[d->efc_id[i]];
                if (i != c->efc_adr || c->dim < 2) continue;
                const int n = c->dim - 1, i1 = i + 1;
                double res[5], old[5], Ac[25], bc[5], v[5], dl[5];
                for (int j = 0; j < n; j++) { res[j] = row_res(d, i1 + j, 0); old[j] = d->efc_force[i1 + j]; }
                for (int j = 0; j < n; j++)
                    for (int k = 0; k < n; k++) {
                        const double *Jr = d->efc_J + (size_t)(i1 + j) * nv, *Bk = d->efc_B + (size_t)(i1 + k) * nv;
                        double a = 0;
                        for (int q = 0; q < nv; q++) a += Jr[q] * Bk[q];
                        Ac[n * j + k] = a;
                    }
                for (int j = 0; j < n; j++)
                    for (int k = 0; k < j; k++) Ac[n * k + j] = Ac[n * j + k];      /* exactly symmetric: lower triangle rules */
                for (int j = 0; j < n; j++) { double a = 0; for (int k = 0; k < n; k++) a += Ac[n * j + k] * old[k]; bc[j] = res[j] - a; }
                const double fn = d->efc_force[i];
                if (fn < MINVAL) { for (int j = 0; j < n; j++) v[j] = 0; }
                else {
                    /* larger blocks first try the multiplier 0 through the block's inverse (the unconstrained minimiser): inside
                     * the cone section it is the answer; otherwise the multiplier iteration of mju_QCQP runs as it stands */
                    int active = -1;
                    if (n >= 3) {
                        double Inv[25], bs[5], y[5], val = -fn * fn;
                        if (qc_inverse(Ac, c->friction, n, Inv)) {
                            for (int j = 0; j < n; j++) bs[j] = bc[j] * c->friction[j];
                            for (int j = 0; j < n; j++) { double t = 0; for (int k = 0; k < n; k++) t += Inv[5 * j + k] * bs[k]; y[j] = -t; }
                            for (int j = 0; j < n; j++) val += y[j] * y[j];
                            if (val < 1e-10) { for (int j = 0; j < n; j++) v[j] = y[j] * c->friction[j]; active = 0; }
                        }
                    }
                    if (active < 0) active = n == 2 ? qcqp2(v, Ac, bc, c->friction, fn) : qcqpn(v, Ac, bc, c->friction, fn, n);
                    if (active) {
                        double s = 0;
                        for (int j = 0; j < n; j++) s += v[j] * v[j] / (c->friction[j] * c->friction[j]);
                        s = sqrt(fn * fn / fmax(MINVAL, s));
                        for (int j = 0; j < n; j++) v[j] *= s;
                    }
                }
                double change = 0;
                for (int j = 0; j < n; j++) {
                    dl[j] = v[j] - old[j];
                    double a = 0;
                    for (int k = 0; k < n; k++) a += Ac[n * j + k] * (v[k] - old[k]);
                    change += dl[j] * (0.5 * a + res[j]);
                }
                if (change > 1e-10) continue;
                for (int j = 0; j < n; j++) { apply_delta(d, i1 + j, dl[j]); d->efc_force[i1 + j] = v[j]; }
                imp -= change;
            }
        }
        d->stat_noslip = it + 1;
        if (imp * d->pgs_scale < noslip_tolerance) break;
    }
    for (int k = 0; k < nv; k++) {
        double s = 0;
        for (int i = 0; i < ne; i++) s += d->efc_J[(size_t)i * nv + k] * d->efc_force[i];
        d->qfrc_constraint[k] = s;
    }
}

/* mj_forward [EXT] */
void orc_forward(orc_data* d) {
    ORC_PHASE(ORC_PH_KINEMATICS); orc_kinematics(d);
    ORC_PHASE(ORC_PH_CRB); orc_crb(d);
    ORC_PHASE(ORC_PH_COLLIDE); orc_collide(d);
    ORC_PHASE(ORC_PH_RNE); orc_rne_bias(d);
    ORC_PHASE(ORC_PH_SMOOTH); smooth_forces(d);
    ORC_PHASE(ORC_PH_ROWS); orc_make_constraints(d);
    if (d->solver == 1) {
        ORC_PHASE(ORC_PH_NEWTON); orc_solve_newton(d);
        ORC_PHASE(ORC_PH_NOSLIP); orc_noslip(d);   /* noslip runs on the dual after the main solve [EXT]; qacc and forces stay consistent */
    } else {
        ORC_PHASE(ORC_PH_NEWTON); orc_solve(d);
    }
    ORC_PHASE(ORC_PH_OTHER);
}

/* P9: mj_Euler with implicit joint damping [EXT]: (M + h diag(b)) qacc_d = qfrc_smooth + qfrc_constraint */
static void euler(orc_data* d) {
    const orc_model* m = d->m;
    int nv = m->nv;
    double h = m->timestep;
    double* A = zalloc((size_t)nv * nv);
    double* La = zalloc((size_t)nv * nv);
    double* a = zalloc(nv);
    memcpy(A, d->M, sizeof(double) * nv * nv);
    for (int i = 0; i < nv; i++) {
        A[i * nv + i] += h * m->dof_damping[i];
        a[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    }
    cholesky(A, La, nv);
    chol_solve(La, a, nv);
    for (int i = 0; i < nv; i++) d->qvel[i] += h * a[i];
    /* positions with the NEW velocity (semi-implicit) */
    for (int j = 0; j < m->njnt; j++) {
        int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
        if (m->jnt_type[j] == ORC_FREE) {
            for (int k = 0; k < 3; k++) d->qpos[qa + k] += h * d->qvel[da + k];
            /* mju_quatIntegrate: q <- q * exp(h w/2), w in the body frame */
            double* q = d->qpos + qa + 3;
            const double* w = d->qvel + da + 3;
            double ang = h * sqrt(dot3(w, w));
            if (ang > 0) {
                double wn = sqrt(dot3(w, w)), s = sin(ang / 2), qr[4] = {cos(ang / 2), s * w[0] / wn, s * w[1] / wn, s * w[2] / wn};
                quatmul(q, qr, q);
            }
            quatnorm(q);
        } else {
            d->qpos[qa] += h * d->qvel[da];
        }
    }
    memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
    d->time += h;
    free(A); free(La); free(a);
}

void orc_step(orc_data* d, int nsub) {
    for (int s = 0; s < nsub; s++) {
        orc_forward(d);
        ORC_PHASE(ORC_PH_EULER); euler(d);
    }
    /* refresh position-dependent quantities of the final state (dm_control Physics.step's trailing
     * mj_step1): the contact list read by get_reward is that of the new state */
    ORC_PHASE(ORC_PH_KINEMATICS); orc_kinematics(d);
    ORC_PHASE(ORC_PH_COLLIDE); orc_collide(d);
    ORC_PHASE(ORC_PH_OTHER);
}

void orc_reset(orc_data* d, const double* obj_qpos) {
    const orc_model* m = d->m;
    memcpy(d->qpos, m->qpos_home, sizeof(double) * m->nq);
    for (int o = 0; o < m->nobj; o++) memcpy(d->qpos + m->objects_qposadr[o], obj_qpos + 7 * o, 56);
    memset(d->qvel, 0, sizeof(double) * m->nv);
    memset(d->qacc_warmstart, 0, sizeof(double) * m->nv);
    memcpy(d->ctrl, m->ctrl_home, sizeof(double) * m->nu);
    d->time = 0;
    d->threaded = 0;
    d->overflow = 0;
    orc_kinematics(d);
    orc_collide(d);
}

void orc_set_qpos(orc_data* d, const double* qpos) {
    memcpy(d->qpos, qpos, sizeof(double) * d->m->nq);
    orc_kinematics(d);
    orc_collide(d);
}

void orc_agent_pos(const orc_data* d, double* out) {
    const orc_model* m = d->m;
    int n = m->num_arms == 3 ? 21 : 14;
    for (int k = 0; k < n; k++) out[k] = (d->qpos[m->obs_qposadr[k]] - m->obs_offset[k]) * m->obs_scale[k];
}

int orc_reward(orc_data* d) {
    int gp[2 * ORC_MAXCON];
    for (int i = 0; i < d->ncon; i++) { gp[2 * i] = d->contact[i].geom1; gp[2 * i + 1] = d->contact[i].geom2; }
    return orc_reward_from_pairs(d->m, gp, d->ncon, &d->threaded);
}

/* env.py:203-226 */
void orc_env_step(orc_data* d, const double* action, int nsub, double* agent_pos, int* reward, int* success) {
    const orc_model* m = d->m;
    double lo = m->grip_range[0], hi = m->grip_range[1];
    for (int k = 0; k < 6; k++) { d->ctrl[k] = action[k]; d->ctrl[7 + k] = action[7 + k]; }
    d->ctrl[6] = action[6] * (hi - lo) + lo;
    d->ctrl[13] = action[13] * (hi - lo) + lo;
    if (m->num_arms == 3) for (int k = 0; k < 7; k++) d->ctrl[14 + k] = action[14 + k];
    orc_step(d, nsub);
    if (agent_pos) orc_agent_pos(d, agent_pos);
    int r = orc_reward(d);
    if (reward) *reward = r;
    if (success) *success = (r == orc_max_reward(m));
}

/* sim_env.py:277-301 */
void orc_cart_to_ctrl(const orc_data* d, const double* a, int mode, double* out21) {
    const orc_model* m = d->m;
    ORC_PHASE(ORC_PH_IK);
    const double kn[7] = {10.0, 10.0, 10.0, 10.0, 5.0, 5.0, 5.0};
    double lo = m->grip_range[0], hi = m->grip_range[1];
    for (int arm = 0; arm < 3; arm++) {
        const double* t = a + (arm == 0 ? 0 : (arm == 1 ? 8 : 16));
        int n = m->ik_n[arm], base = arm == 0 ? 0 : (arm == 1 ? 7 : 14);
        double q[7], q0[7], o[7];
        for (int k = 0; k < n; k++) { q[k] = d->qpos[m->ik_qadr[arm * 7 + k]]; q0[k] = m->qpos_home[m->ik_qadr[arm * 7 + k]]; }
        if (arm == 2 || mode == 1) orc_diffik(m, arm, q, t, t + 3, 0.9, 0.9, 1e-4, kn, q0, 3.14, 0.04, 10, o);
        else orc_gradik(m, arm, q, t, t + 3, o);
        for (int k = 0; k < n; k++) out21[base + k] = o[k];
        /* orc_env_step un-normalises the gripper entry, so hand back the NORMALISED opening 1 - trigger */
        if (arm < 2) out21[base + 6] = 1.0 - t[7];
    }
    (void)lo; (void)hi;
    ORC_PHASE(ORC_PH_OTHER);
}
