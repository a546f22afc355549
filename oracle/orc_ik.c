/* oracle/orc_ik.c -- f64 restatement of the reference's IK path.  TEST INFRASTRUCTURE ONLY (orc.h).
 * Each function cites the reference lines it follows (paths under /root/reference/data_collection_scripts/). */
#include <math.h>
#include <string.h>

#include "orc.h"

#define EPS4 (2.220446049250313e-16 * 4.0) /* transform_utils.py:7 */

static void mat3mul(const double* A, const double* B, double* C) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, t, sizeof t);
}
static void cross3(const double* a, const double* b, double* c) {
    double t0 = a[1] * b[2] - a[2] * b[1], t1 = a[2] * b[0] - a[0] * b[2], t2 = a[0] * b[1] - a[1] * b[0];
    c[0] = t0; c[1] = t1; c[2] = t2;
}
static double norm3(const double* a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

/* transform_utils.py:52-79.  The reference rounds the quaternion to float32 (:66); run as plain NumPy (the way the golden vectors
 * were produced) every product below stays float32.  Op for op: np.dot of two float32 vectors is OpenBLAS sdot, which rounds each
 * product to float and accumulates in double (kernel/x86_64/sdot.c tail loop); 2.0 / n is a float32 division, math.sqrt works in
 * double on that value and the scale goes back to float32 for the in-place multiply; outer products and the nine entries float32.
 * Reproduces tests/golden/so3_helpers.npz bit for bit. */
void orc_quat2mat(const double qx[4], double R[9]) {
    float q[4] = {(float)qx[3], (float)qx[0], (float)qx[1], (float)qx[2]}; /* w x y z */
    double acc = 0;
    for (int i = 0; i < 4; i++) { float p = q[i] * q[i]; acc += (double)p; }
    float n = (float)acc;
    if (n < (float)EPS4) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0);
        return;
    }
    float s = (float)sqrt((double)(2.0f / n));
    for (int i = 0; i < 4; i++) q[i] *= s;
    float q2[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) q2[i][j] = q[i] * q[j];
    R[0] = 1.0f - q2[2][2] - q2[3][3]; R[1] = q2[1][2] - q2[3][0]; R[2] = q2[1][3] + q2[2][0];
    R[3] = q2[1][2] + q2[3][0]; R[4] = 1.0f - q2[1][1] - q2[3][3]; R[5] = q2[2][3] - q2[1][0];
    R[6] = q2[1][3] - q2[2][0]; R[7] = q2[2][3] + q2[1][0]; R[8] = 1.0f - q2[1][1] - q2[2][2];
}

/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 6); V columns = eigenvectors */
static void jacobi_eig(double* A, int n, double* w, double* V) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}

/* transform_utils.py:9-49: quaternion = eigenvector of K for the largest eigenvalue (eigh uses the lower triangle) */
void orc_mat2quat(const double M[9], double q_xyzw[4]) {
    double m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[3], m11 = M[4], m12 = M[5], m20 = M[6], m21 = M[7], m22 = M[8];
    double K[16] = {m00 - m11 - m22, 0, 0, 0, m01 + m10, m11 - m00 - m22, 0, 0,
                    m02 + m20, m12 + m21, m22 - m00 - m11, 0, m21 - m12, m02 - m20, m10 - m01, m00 + m11 + m22};
    for (int i = 0; i < 4; i++)
        for (int j = i + 1; j < 4; j++) K[i * 4 + j] = K[j * 4 + i];
    for (int i = 0; i < 16; i++) K[i] /= 3.0;
    double w[4], V[16];
    jacobi_eig(K, 4, w, V);
    int best = 0;
    for (int i = 1; i < 4; i++)
        if (w[i] > w[best]) best = i;
    double q1[4] = {V[3 * 4 + best], V[0 * 4 + best], V[1 * 4 + best], V[2 * 4 + best]}; /* inds [3,0,1,2] -> w x y z */
    if (q1[0] < 0.0)
        for (int i = 0; i < 4; i++) q1[i] = -q1[i];
    q_xyzw[0] = q1[1]; q_xyzw[1] = q1[2]; q_xyzw[2] = q1[3]; q_xyzw[3] = q1[0];
}

/* transform_utils.py:82-106 */
void orc_quat2axisangle(const double qin[4], double aa[3]) {
    double w = qin[3];
    if (w > 1.0) w = 1.0;
    else if (w < -1.0) w = -1.0;
    double den = sqrt(1.0 - w * w);
    if (fabs(den) <= 1e-8) { aa[0] = aa[1] = aa[2] = 0; return; } /* np.isclose(den, 0.0) */
    double s = 2.0 * acos(w) / den;
    aa[0] = qin[0] * s; aa[1] = qin[1] * s; aa[2] = qin[2] * s;
}

/* transform_utils.py:108-133 */
void orc_axisangle2quat(const double v[3], double q[4]) {
    double a = norm3(v);
    if (fabs(a) <= 1e-8) { q[0] = q[1] = q[2] = 0; q[3] = 1; return; }
    double s = sin(a / 2.0);
    q[0] = v[0] / a * s; q[1] = v[1] / a * s; q[2] = v[2] / a * s; q[3] = cos(a / 2.0);
}

/* transform_utils.py:183-194 */
void orc_angular_error(const double D[9], const double C[9], double e[3]) {
    e[0] = e[1] = e[2] = 0;
    for (int k = 0; k < 3; k++) {
        double rc[3] = {C[k], C[3 + k], C[6 + k]}, rd[3] = {D[k], D[3 + k], D[6 + k]}, c[3];
        cross3(rc, rd, c);
        e[0] += c[0]; e[1] += c[1]; e[2] += c[2];
    }
    e[0] *= 0.5; e[1] *= 0.5; e[2] *= 0.5;
}

/* transform_utils.py:203-261 (skew_sym, exp2rot, exp2mat) */
void orc_exp2mat(const double w[3], const double v[3], double th, double T[16]) {
    double S[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}, S2[9];
    mat3mul(S, S, S2);
    double s = sin(th), c = cos(th);
    double R[9];
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) + s * S[i] + (1 - c) * S2[i];
    double p[3];
    if (fabs(norm3(w)) <= 1e-8) { /* prismatic branch :232-240 */
        for (int i = 0; i < 3; i++) p[i] = v[i] * th;
    } else {
        for (int i = 0; i < 3; i++) {
            p[i] = 0;
            for (int j = 0; j < 3; j++) p[i] += ((i == j) * th + (1 - c) * S[3 * i + j] + (th - s) * S2[3 * i + j]) * v[j];
        }
    }
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[4 * i + j] = R[3 * i + j];
        T[4 * i + 3] = p[i];
    }
    T[12] = T[13] = T[14] = 0; T[15] = 1;
}

/* transform_utils.py:289-301 */
void orc_adjoint(const double T[16], double A[36]) {
    double R[9], p[3] = {T[3], T[7], T[11]};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[3 * i + j] = T[4 * i + j];
    double S[9] = {0, -p[2], p[1], p[2], 0, -p[0], -p[1], p[0], 0}, pR[9];
    mat3mul(S, R, pR);
    memset(A, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            A[6 * i + j] = R[3 * i + j];
            A[6 * (i + 3) + j + 3] = R[3 * i + j];
            A[6 * (i + 3) + j] = pR[3 * i + j];
        }
}

static void inv3(const double* A, double* B) {
    double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    double det = A[0] * c00 + A[1] * c01 + A[2] * c02, id = 1.0 / det;
    B[0] = c00 * id; B[1] = (A[2] * A[7] - A[1] * A[8]) * id; B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    B[3] = c01 * id; B[4] = (A[0] * A[8] - A[2] * A[6]) * id; B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    B[6] = c02 * id; B[7] = (A[1] * A[6] - A[0] * A[7]) * id; B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

/* transform_utils.py:263-287 */
void orc_limit_pose(const double cp[3], const double cR[9], const double tp[3], const double tR[9], double maxp,
                    double maxr, double op[3], double oR[9]) {
    double d[3] = {tp[0] - cp[0], tp[1] - cp[1], tp[2] - cp[2]};
    double n = norm3(d);
    if (n > maxp)
        for (int i = 0; i < 3; i++) d[i] = d[i] / n * maxp;
    for (int i = 0; i < 3; i++) op[i] = cp[i] + d[i];
    double ci[9], rel[9], q[4], aa[3];
    inv3(cR, ci);
    mat3mul(tR, ci, rel);
    orc_mat2quat(rel, q);
    orc_quat2axisangle(q, aa);
    double ang = norm3(aa);
    if (ang > maxr) {
        double v[3] = {aa[0] * (maxr / ang), aa[1] * (maxr / ang), aa[2] * (maxr / ang)}, q2[4], lim[9];
        orc_axisangle2quat(v, q2);
        orc_quat2mat(q2, lim);
        mat3mul(lim, cR, oR);
    } else {
        memcpy(oR, tR, 9 * sizeof(double));
    }
}

static void mat4mul(const double* A, const double* B, double* C) {
    double t[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += A[4 * i + k] * B[4 * k + j];
            t[4 * i + j] = s;
        }
    memcpy(C, t, sizeof t);
}

static void screw(const orc_model* m, int arm, int i, double w[3], double v[3]) {
    const double* w0 = m->ik_w0 + (arm * 7 + i) * 3;
    const double* p0 = m->ik_p0 + (arm * 7 + i) * 3;
    double c[3];
    cross3(w0, p0, c);
    for (int k = 0; k < 3; k++) { w[k] = w0[k]; v[k] = -c[k]; } /* kinematics.py:12 v0 = -cross(w0, p0) */
}

/* kinematics.py:17-24 */
void orc_fk(const orc_model* m, int arm, const double* q, double T[16]) {
    int n = m->ik_n[arm];
    memcpy(T, m->ik_site0 + arm * 16, 16 * sizeof(double));
    for (int i = n - 1; i >= 0; i--) {
        double w[3], v[3], E[16];
        screw(m, arm, i, w, v);
        orc_exp2mat(w, v, q[i], E);
        mat4mul(E, T, T);
    }
}

/* kinematics.py:35-50: space Jacobian columns Ad(T_1..T_{i-1}) S_i, then rows swapped to [linear; angular] */
void orc_jac(const orc_model* m, int arm, const double* q, double* J) {
    int n = m->ik_n[arm];
    double Ts[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < n; i++) {
        double w[3], v[3], A[36], S[6], col[6], E[16];
        screw(m, arm, i, w, v);
        for (int k = 0; k < 3; k++) { S[k] = w[k]; S[3 + k] = v[k]; }
        orc_adjoint(Ts, A);
        for (int r = 0; r < 6; r++) {
            col[r] = 0;
            for (int c = 0; c < 6; c++) col[r] += A[6 * r + c] * S[c];
        }
        for (int r = 0; r < 3; r++) { J[r * n + i] = col[3 + r]; J[(3 + r) * n + i] = col[r]; }
        orc_exp2mat(w, v, q[i], E);
        mat4mul(Ts, E, Ts);
    }
}

/* np.linalg.solve: LU with partial pivoting */
static void solve_n(double* A, double* b, int n) {
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int i = k + 1; i < n; i++)
            if (fabs(A[i * n + k]) > fabs(A[p * n + k])) p = i;
        if (p != k) {
            for (int j = 0; j < n; j++) { double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
            double t = b[k]; b[k] = b[p]; b[p] = t;
        }
        for (int i = k + 1; i < n; i++) {
            double f = A[i * n + k] / A[k * n + k];
            for (int j = k; j < n; j++) A[i * n + j] -= f * A[k * n + j];
            b[i] -= f * b[k];
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int j = i + 1; j < n; j++) s -= A[i * n + j] * b[j];
        b[i] = s / A[i * n + i];
    }
}

/* diff_ik.py:51-85 */
void orc_diffik(const orc_model* m, int arm, const double* qin, const double pos[3], const double quat_wxyz[4],
                double k_pos, double k_ori, double damping, const double* k_null, const double* q0, double max_angvel,
                double dt, int iterations, double* q) {
    double qx[4] = {quat_wxyz[1], quat_wxyz[2], quat_wxyz[3], quat_wxyz[0]}, Rt[9]; /* wxyz_to_xyzw, diff_ik.py:59 */
    orc_quat2mat(qx, Rt);
    orc_diffik_R(m, arm, qin, pos, Rt, k_pos, k_ori, damping, k_null, q0, max_angvel, dt, iterations, q);
}

/* same with the target rotation given as a matrix (lets tests feed the reference's own float32-rounded matrix) */
void orc_diffik_R(const orc_model* m, int arm, const double* qin, const double pos[3], const double Rt[9],
                  double k_pos, double k_ori, double damping, const double* k_null, const double* q0,
                  double max_angvel, double dt, int iterations, double* q) {
    int n = m->ik_n[arm];
    const double* range = m->ik_range + arm * 14;
    for (int i = 0; i < n; i++) q[i] = qin[i];
    for (int it = 0; it < iterations; it++) {
        double T[16], Rc[9], tw[6], dr[3], J[42];
        orc_fk(m, arm, q, T);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rc[3 * i + j] = T[4 * i + j];
        for (int i = 0; i < 3; i++) tw[i] = k_pos * (pos[i] - T[4 * i + 3]) / dt;
        orc_angular_error(Rt, Rc, dr);
        for (int i = 0; i < 3; i++) tw[3 + i] = k_ori * dr[i] / dt;
        orc_jac(m, arm, q, J);
        double JJt[36], A[36], x[6];
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) {
                double s = 0;
                for (int k = 0; k < n; k++) s += J[i * n + k] * J[j * n + k];
                JJt[6 * i + j] = s;
                A[6 * i + j] = s + (i == j ? damping : 0.0);
            }
        memcpy(x, tw, sizeof x);
        solve_n(A, x, 6);
        double dq[7];
        for (int k = 0; k < n; k++) {
            dq[k] = 0;
            for (int i = 0; i < 6; i++) dq[k] += J[i * n + k] * x[i];
        }
        /* null-space term (I - pinv(J) J) (k_null * (q0 - q)); pinv(J) = J^T U S^-2 U^T from the
         * eigen-decomposition J J^T = U S^2 U^T, singular values below 1e-15*smax dropped (np.linalg.pinv) */
        double w[6], U[36], B[36];
        memcpy(B, JJt, sizeof B);
        jacobi_eig(B, 6, w, U);
        double wmax = 0;
        for (int i = 0; i < 6; i++)
            if (w[i] > wmax) wmax = w[i];
        double z[7], Jz[6], y[6];
        for (int k = 0; k < n; k++) z[k] = k_null[k] * (q0[k] - q[k]);
        for (int i = 0; i < 6; i++) {
            Jz[i] = 0;
            for (int k = 0; k < n; k++) Jz[i] += J[i * n + k] * z[k];
        }
        for (int i = 0; i < 6; i++) y[i] = 0;
        for (int e = 0; e < 6; e++) {
            if (!(w[e] > 0) || sqrt(w[e]) <= 1e-15 * sqrt(wmax)) continue;
            double c = 0;
            for (int i = 0; i < 6; i++) c += U[i * 6 + e] * Jz[i];
            c /= w[e];
            for (int i = 0; i < 6; i++) y[i] += U[i * 6 + e] * c;
        }
        for (int k = 0; k < n; k++) {
            double pj = 0;
            for (int i = 0; i < 6; i++) pj += J[i * n + k] * y[i];
            dq[k] += z[k] - pj;
        }
        for (int k = 0; k < n; k++) {
            if (dq[k] > max_angvel) dq[k] = max_angvel;
            if (dq[k] < -max_angvel) dq[k] = -max_angvel;
            q[k] += dq[k] * dt;
            if (q[k] < range[2 * k]) q[k] = range[2 * k];
            if (q[k] > range[2 * k + 1]) q[k] = range[2 * k + 1];
        }
    }
}

/* grad_ik.py:168-198 with the parameters of sim_env.py:89-122 */
typedef struct {
    const orc_model* m;
    int arm, n;
    double pw, rw, jcw[6], centers[6], jdw[6];
} gik;

static double gik_cost(const gik* g, const double* q, const double* qs, const double* tp, const double* tR) {
    double T[16], Rc[9], e[3];
    orc_fk(g->m, g->arm, q, T);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rc[3 * i + j] = T[4 * i + j];
    double d[3] = {tp[0] - T[3], tp[1] - T[7], tp[2] - T[11]};
    double c = 0, t;
    t = g->pw * norm3(d); c += t * t;
    orc_angular_error(tR, Rc, e);
    t = g->rw * norm3(e); c += t * t;
    double s = 0;
    for (int i = 0; i < g->n; i++) { t = g->jcw[i] * (q[i] - g->centers[i]); s += t * t; }
    c += s;
    s = 0;
    for (int i = 0; i < g->n; i++) { t = g->jdw[i] * (q[i] - qs[i]); s += t * t; }
    c += s;
    return c;
}

int orc_gradik_max_it = 50; /* sim_env.py:95 max_iterations (exported so tests can truncate the descent) */

/* grad_ik.py:8-99 */
void orc_gradik(const orc_model* m, int arm, const double* qs, const double pos[3], const double quat_wxyz[4],
                double* qout) {
    double qx[4] = {quat_wxyz[1], quat_wxyz[2], quat_wxyz[3], quat_wxyz[0]}, tR0[9];
    orc_quat2mat(qx, tR0);
    orc_gradik_R(m, arm, qs, pos, tR0, qout);
}

void orc_gradik_R(const orc_model* m, int arm, const double* qs, const double pos[3], const double tR0[9],
                  double* qout) {
    const double step = 1e-4, min_delta = 1e-12, joint_p = 0.9, maxp = 0.1, maxr = 0.3, pthr = 1e-3, rthr = 1e-3;
    const int max_it = orc_gradik_max_it;
    const double jc[6] = {10.0, 10.0, 1.0, 50.0, 1.0, 1.0};
    gik g;
    g.m = m; g.arm = arm; g.n = m->ik_n[arm]; g.pw = 500.0; g.rw = 100.0;
    const double* range = m->ik_range + arm * 14;
    int n = g.n;
    for (int i = 0; i < n; i++) {
        double lo = range[2 * i], hi = range[2 * i + 1];
        g.centers[i] = 0.5 * (lo + hi);
        g.jcw[i] = jc[i] / (0.5 * (hi - lo));
        g.jdw[i] = 50.0;
    }
    double T[16], cR[9], cp[3], tp[3], tR[9];
    orc_fk(m, arm, qs, T);
    for (int i = 0; i < 3; i++) {
        cp[i] = T[4 * i + 3];
        for (int j = 0; j < 3; j++) cR[3 * i + j] = T[4 * i + j];
    }
    orc_limit_pose(cp, cR, pos, tR0, maxp, maxr, tp, tR);
    double init = gik_cost(&g, qs, qs, tp, tR);
    double grad[6] = {0}, work[6], local[6], best[6];
    for (int i = 0; i < n; i++) work[i] = local[i] = best[i] = qs[i];
    double local_cost = init, best_cost = init, prev = 0.0;
    for (int it = 0; it < max_it; it++) {
        for (int i = 0; i < n; i++) {
            work[i] = local[i] - step;
            double p1 = gik_cost(&g, work, qs, tp, tR);
            work[i] = local[i] + step;
            double p3 = gik_cost(&g, work, qs, tp, tR);
            work[i] = local[i];
            grad[i] = p3 - p1;
        }
        double sum = 0;
        for (int i = 0; i < n; i++) sum += fabs(grad[i]);
        sum += step;
        double f = step / sum;
        for (int i = 0; i < n; i++) grad[i] *= f;
        for (int i = 0; i < n; i++) work[i] = local[i] - grad[i];
        double p1 = gik_cost(&g, work, qs, tp, tR);
        for (int i = 0; i < n; i++) work[i] = local[i] + grad[i];
        double p3 = gik_cost(&g, work, qs, tp, tR);
        double p2 = 0.5 * (p1 + p3), cd = 0.5 * (p3 - p1);
        double jd = (isfinite(cd) && cd != 0.0) ? p2 / cd : 0.0;
        for (int i = 0; i < n; i++) {
            double x = local[i] - grad[i] * jd;
            if (x < range[2 * i]) x = range[2 * i];
            if (x > range[2 * i + 1]) x = range[2 * i + 1];
            work[i] = x;
        }
        for (int i = 0; i < n; i++) local[i] = work[i];
        local_cost = gik_cost(&g, local, qs, tp, tR);
        if (local_cost < best_cost) {
            for (int i = 0; i < n; i++) best[i] = local[i];
            best_cost = local_cost;
        }
        /* solution_fn (grad_ik.py:200-220) -> within_pose_threshold (transform_utils.py:196-201) */
        double Tl[16], Rl[9], e[3];
        orc_fk(m, arm, local, Tl);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rl[3 * i + j] = Tl[4 * i + j];
        double d[3] = {tp[0] - Tl[3], tp[1] - Tl[7], tp[2] - Tl[11]};
        orc_angular_error(tR, Rl, e);
        if (norm3(d) < pthr && norm3(e) < rthr) break;
        if (fabs(local_cost - prev) <= min_delta) break;
        prev = local_cost;
    }
    for (int i = 0; i < n; i++) qout[i] = qs[i] + joint_p * (best[i] - qs[i]);
}
