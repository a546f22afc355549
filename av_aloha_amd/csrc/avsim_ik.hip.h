// avsim_ik.hip.h -- device IK: one (env, arm) problem per lane, everything in registers.
// Mirrors data_collection_scripts/{kinematics.py:17-50, diff_ik.py:51-85, grad_ik.py:8-99,
// transform_utils.py:52-79,183-301}.  NJ is a template constant so every loop unrolls.
#pragma once
#include <hip/hip_runtime.h>

#include "avsim_model.h"

namespace avs {

template <typename T>
struct M3 {
    T m[9];
};

template <typename T>
__device__ __forceinline__ void mat3mul(const T* A, const T* B, T* C) {
    T t[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) C[i] = t[i];
}

// transform_utils.py:52-79 incl. the float32 cast at :66, op for op as NumPy evaluates it (oracle/orc_ik.c orc_quat2mat): float32
// products accumulated in double for the norm (OpenBLAS sdot), a correctly rounded float32 division, sqrt in double, float32 outer
// products and entries, nothing fused -- bit-identical to the reference's matrices (tests/golden/so3_helpers.npz)
template <typename T>
__device__ __forceinline__ void quat2mat_xyzw(const T qx[4], T R[9]) {
#pragma clang fp contract(off)
    float q[4] = {(float)qx[3], (float)qx[0], (float)qx[1], (float)qx[2]};
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { const float p = q[i] * q[i]; acc += (double)p; }
    const float n = (float)acc;
    if (n < 8.8817841970012523e-16f) {
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? T(1) : T(0);
        return;
    }
    float s = (float)sqrt((double)__fdiv_rn(2.0f, n));
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] *= s;
    float q2[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) q2[i][j] = q[i] * q[j];
    R[0] = 1.0f - q2[2][2] - q2[3][3]; R[1] = q2[1][2] - q2[3][0]; R[2] = q2[1][3] + q2[2][0];
    R[3] = q2[1][2] + q2[3][0]; R[4] = 1.0f - q2[1][1] - q2[3][3]; R[5] = q2[2][3] - q2[1][0];
    R[6] = q2[1][3] - q2[2][0]; R[7] = q2[2][3] + q2[1][0]; R[8] = 1.0f - q2[1][1] - q2[2][2];
}

// transform_utils.py:183-194
template <typename T>
__device__ __forceinline__ void angular_error(const T* D, const T* C, T e[3]) {
    e[0] = e[1] = e[2] = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        T c0 = C[k], c1 = C[3 + k], c2 = C[6 + k], d0 = D[k], d1 = D[3 + k], d2 = D[6 + k];
        e[0] += c1 * d2 - c2 * d1;
        e[1] += c2 * d0 - c0 * d2;
        e[2] += c0 * d1 - c1 * d0;
    }
    e[0] *= T(0.5); e[1] *= T(0.5); e[2] *= T(0.5);
}

// exp([S] th) for a unit revolute screw (transform_utils.py:212-261): R (row-major 9) and p
template <typename T>
__device__ __forceinline__ void exp_screw(const double w_[3], const double v_[3], T th, T R[9], T p[3]) {
    T w[3] = {(T)w_[0], (T)w_[1], (T)w_[2]}, v[3] = {(T)v_[0], (T)v_[1], (T)v_[2]};
    T S[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}, S2[9];
    mat3mul(S, S, S2);
    T s, c;
    sincos(th, &s, &c);      // (one argument reduction for both)
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = ((i % 4 == 0) ? T(1) : T(0)) + s * S[i] + (1 - c) * S2[i];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        T a = 0;
#pragma unroll
        for (int j = 0; j < 3; j++) a += (((i == j) ? th : T(0)) + (1 - c) * S[3 * i + j] + (th - s) * S2[3 * i + j]) * v[j];
        p[i] = a;
    }
}

// kinematics.py:17-24: T(theta) = exp(S1 th1) ... exp(Sn thn) site0, built from the last joint to the first
template <typename T, int NJ>
__device__ __forceinline__ void fk(const IkArm& A, const T* q, T R[9], T p[3]) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) R[3 * i + j] = (T)A.site0[4 * i + j];
        p[i] = (T)A.site0[4 * i + 3];
    }
#pragma unroll
    for (int i = NJ - 1; i >= 0; i--) {
        T Re[9], pe[3], np[3];
        exp_screw<T>(A.w[i], A.v[i], q[i], Re, pe);
#pragma unroll
        for (int r = 0; r < 3; r++) np[r] = Re[3 * r] * p[0] + Re[3 * r + 1] * p[1] + Re[3 * r + 2] * p[2] + pe[r];
        mat3mul(Re, R, R);
        p[0] = np[0]; p[1] = np[1]; p[2] = np[2];
    }
}

// kinematics.py:35-50: space Jacobian, rows [linear(3); angular(3)], J[r][i]
template <typename T, int NJ>
__device__ __forceinline__ void jac(const IkArm& A, const T* q, T J[6][NJ]) {
    T R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < NJ; i++) {
        T w[3] = {(T)A.w[i][0], (T)A.w[i][1], (T)A.w[i][2]}, v[3] = {(T)A.v[i][0], (T)A.v[i][1], (T)A.v[i][2]};
        T Rw[3], Rv[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            Rw[r] = R[3 * r] * w[0] + R[3 * r + 1] * w[1] + R[3 * r + 2] * w[2];
            Rv[r] = R[3 * r] * v[0] + R[3 * r + 1] * v[1] + R[3 * r + 2] * v[2];
        }
        // Ad(T) [w; v] = [R w; p x (R w) + R v]; rows swapped to linear first (kinematics.py:48)
        J[3][i] = Rw[0]; J[4][i] = Rw[1]; J[5][i] = Rw[2];
        J[0][i] = p[1] * Rw[2] - p[2] * Rw[1] + Rv[0];
        J[1][i] = p[2] * Rw[0] - p[0] * Rw[2] + Rv[1];
        J[2][i] = p[0] * Rw[1] - p[1] * Rw[0] + Rv[2];
        T Re[9], pe[3], np[3];
        exp_screw<T>(A.w[i], A.v[i], q[i], Re, pe);
#pragma unroll
        for (int r = 0; r < 3; r++) np[r] = R[3 * r] * pe[0] + R[3 * r + 1] * pe[1] + R[3 * r + 2] * pe[2] + p[r];
        mat3mul(R, Re, R);
        p[0] = np[0]; p[1] = np[1]; p[2] = np[2];
    }
}

// The same two with the joints' exponentials exp([S_i] q_i) made once and shared (DiffIK evaluates both at the same q every
// iteration; an exponential is a sine, a cosine and two 3 x 3 products in double): operation for operation the results of fk / jac.
template <typename T, int NJ>
__device__ __forceinline__ void joint_exps(const IkArm& A, const T* q, T Re[NJ][9], T pe[NJ][3]) {
#pragma unroll
    for (int i = 0; i < NJ; i++) exp_screw<T>(A.w[i], A.v[i], q[i], Re[i], pe[i]);
}
template <typename T, int NJ>
__device__ __forceinline__ void fk_from(const IkArm& A, const T Re[NJ][9], const T pe[NJ][3], T R[9], T p[3]) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) R[3 * i + j] = (T)A.site0[4 * i + j];
        p[i] = (T)A.site0[4 * i + 3];
    }
#pragma unroll
    for (int i = NJ - 1; i >= 0; i--) {
        T np[3];
#pragma unroll
        for (int r = 0; r < 3; r++) np[r] = Re[i][3 * r] * p[0] + Re[i][3 * r + 1] * p[1] + Re[i][3 * r + 2] * p[2] + pe[i][r];
        mat3mul(Re[i], R, R);
        p[0] = np[0]; p[1] = np[1]; p[2] = np[2];
    }
}
template <typename T, int NJ>
__device__ __forceinline__ void jac_from(const IkArm& A, const T Re[NJ][9], const T pe[NJ][3], T J[6][NJ]) {
    T R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < NJ; i++) {
        T w[3] = {(T)A.w[i][0], (T)A.w[i][1], (T)A.w[i][2]}, v[3] = {(T)A.v[i][0], (T)A.v[i][1], (T)A.v[i][2]};
        T Rw[3], Rv[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            Rw[r] = R[3 * r] * w[0] + R[3 * r + 1] * w[1] + R[3 * r + 2] * w[2];
            Rv[r] = R[3 * r] * v[0] + R[3 * r + 1] * v[1] + R[3 * r + 2] * v[2];
        }
        J[3][i] = Rw[0]; J[4][i] = Rw[1]; J[5][i] = Rw[2];
        J[0][i] = p[1] * Rw[2] - p[2] * Rw[1] + Rv[0];
        J[1][i] = p[2] * Rw[0] - p[0] * Rw[2] + Rv[1];
        J[2][i] = p[0] * Rw[1] - p[1] * Rw[0] + Rv[2];
        T np[3];
#pragma unroll
        for (int r = 0; r < 3; r++) np[r] = R[3 * r] * pe[i][0] + R[3 * r + 1] * pe[i][1] + R[3 * r + 2] * pe[i][2] + p[r];
        mat3mul(R, Re[i], R);
        p[0] = np[0]; p[1] = np[1]; p[2] = np[2];
    }
}

// in-register Cholesky solve of a 6x6 SPD system A x = b (lower triangle of A used). `guard`: pivots
// below guard*max_diag are treated as singular directions (their solution component is dropped).
template <typename T>
__device__ __forceinline__ void chol6_solve(T A[6][6], T b[6], T guard) {
    T dmax = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) dmax = A[i][i] > dmax ? A[i][i] : dmax;
    T inv[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        T d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= A[j][k] * A[j][k];
        bool ok = d > guard * dmax;
        T l = ok ? sqrt(d) : T(1);
        inv[j] = ok ? T(1) / l : T(0);
        A[j][j] = l;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            T s = A[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= A[i][k] * A[j][k];
            A[i][j] = s * inv[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
        T s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= A[i][k] * b[k];
        b[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        T s = b[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) s -= A[k][i] * b[k];
        b[i] = s * inv[i];
    }
}

// diff_ik.py:51-85.  The 6x6 solves use Cholesky (J J^T + lambda I is SPD); the null-space projector
// I - pinv(J) J is evaluated as z - J^T (J J^T)^-1 J z, equal to the reference's SVD pinv whenever J has
// full row rank with a pivot guard for rank-deficient poses.
template <typename T, int NJ>
__device__ void diffik(const IkParams& P, int arm, const T* qin, const T pos[3], const T Rt[9], int iters, T* q) {
    const IkArm& A = P.arm[arm];
#pragma unroll
    for (int i = 0; i < NJ; i++) q[i] = qin[i];
    const T k_pos = (T)P.k_pos, k_ori = (T)P.k_ori, dt = (T)P.dt, vmax = (T)P.max_angvel;
    for (int it = 0; it < iters; it++) {
        T Rc[9], pc[3], tw[6], dr[3];
        T Ee[NJ][9], Ep[NJ][3];
        joint_exps<T, NJ>(A, q, Ee, Ep);
        fk_from<T, NJ>(A, Ee, Ep, Rc, pc);
#pragma unroll
        for (int i = 0; i < 3; i++) tw[i] = k_pos * (pos[i] - pc[i]) / dt;
        angular_error(Rt, Rc, dr);
#pragma unroll
        for (int i = 0; i < 3; i++) tw[3 + i] = k_ori * dr[i] / dt;
        T J[6][NJ];
        jac_from<T, NJ>(A, Ee, Ep, J);
        T JJt[6][6], Ad[6][6];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) {
                T s = 0;
#pragma unroll
                for (int k = 0; k < NJ; k++) s += J[i][k] * J[j][k];
                JJt[i][j] = s;
                Ad[i][j] = s + (i == j ? (T)P.damping : T(0));
            }
        chol6_solve(Ad, tw, T(0));
        T dq[NJ], z[NJ], Jz[6];
#pragma unroll
        for (int k = 0; k < NJ; k++) {
            T s = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) s += J[i][k] * tw[i];
            dq[k] = s;
            z[k] = (T)P.k_null[arm][k] * ((T)P.q0[arm][k] - q[k]);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
            T s = 0;
#pragma unroll
            for (int k = 0; k < NJ; k++) s += J[i][k] * z[k];
            Jz[i] = s;
        }
        chol6_solve(JJt, Jz, T(1e-26));
#pragma unroll
        for (int k = 0; k < NJ; k++) {
            T pj = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) pj += J[i][k] * Jz[i];
            T d = dq[k] + (z[k] - pj);
            d = d > vmax ? vmax : (d < -vmax ? -vmax : d);
            T x = q[k] + d * dt;
            x = x < (T)A.lo[k] ? (T)A.lo[k] : (x > (T)A.hi[k] ? (T)A.hi[k] : x);
            q[k] = x;
        }
    }
}

// transform_utils.py:9-49 mat2quat followed by :82-106 quat2axisangle, as limit_pose uses them (:276-278): the quaternion is the
// eigenvector of the 4 x 4 matrix K for its largest eigenvalue (numpy.linalg.eigh in the reference, cyclic Jacobi here as in
// oracle/orc_ik.c).  The relative rotation is the product of a float32-rounded target matrix and a transpose, i.e. orthonormal
// only to 1e-7, and the eigenvector is the least-squares quaternion of such a matrix: a closed-form extraction from single
// entries differs from it by 1e-8, which the float32 quat2mat of the clamped target then turns into a different matrix.
template <typename T>
__device__ void rotvec_from_mat(const T* M, T aa[3]) {
    T A[16], V[16];
    {
        const T m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[3], m11 = M[4], m12 = M[5], m20 = M[6], m21 = M[7], m22 = M[8];
        const T K[16] = {m00 - m11 - m22, 0, 0, 0, m01 + m10, m11 - m00 - m22, 0, 0, m02 + m20, m12 + m21, m22 - m00 - m11, 0,
                         m21 - m12, m02 - m20, m10 - m01, m00 + m11 + m22};
#pragma unroll
        for (int i = 0; i < 16; i++) A[i] = K[i];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = i + 1; j < 4; j++) A[i * 4 + j] = A[j * 4 + i];
#pragma unroll
        for (int i = 0; i < 16; i++) { A[i] /= T(3); V[i] = (i % 5 == 0) ? T(1) : T(0); }
    }
    for (int sweep = 0; sweep < 64; sweep++) {
        T off = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = i + 1; j < 4; j++) off += A[i * 4 + j] * A[i * 4 + j];
        if (off < T(1e-300)) break;
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                const T apq = A[p * 4 + q];
                if (fabs(apq) < T(1e-300)) continue;
                const T theta = (A[q * 4 + q] - A[p * 4 + p]) / (2 * apq);
                const T t = (theta >= 0 ? T(1) : T(-1)) / (fabs(theta) + sqrt(theta * theta + 1));
                const T c = 1 / sqrt(t * t + 1), s = t * c;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const T akp = A[k * 4 + p], akq = A[k * 4 + q];
                    A[k * 4 + p] = c * akp - s * akq;
                    A[k * 4 + q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const T apk = A[p * 4 + k], aqk = A[q * 4 + k];
                    A[p * 4 + k] = c * apk - s * aqk;
                    A[q * 4 + k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const T vkp = V[k * 4 + p], vkq = V[k * 4 + q];
                    V[k * 4 + p] = c * vkp - s * vkq;
                    V[k * 4 + q] = s * vkp + c * vkq;
                }
            }
    }
    T wb = A[0], qv[4] = {V[0], V[4], V[8], V[12]};     // (x, y, z, w) of eigenvector 0
#pragma unroll
    for (int b = 1; b < 4; b++)
        if (A[b * 5] > wb) { wb = A[b * 5]; qv[0] = V[b]; qv[1] = V[4 + b]; qv[2] = V[8 + b]; qv[3] = V[12 + b]; }
    if (qv[3] < 0) { qv[0] = -qv[0]; qv[1] = -qv[1]; qv[2] = -qv[2]; qv[3] = -qv[3]; }
    T w = qv[3] > 1 ? T(1) : qv[3];
    const T den = sqrt(1 - w * w);
    if (fabs(den) <= T(1e-8)) { aa[0] = aa[1] = aa[2] = 0; return; }
    const T s = 2 * acos(w) / den;
    aa[0] = qv[0] * s; aa[1] = qv[1] * s; aa[2] = qv[2] * s;
}

// transform_utils.py:263-287
template <typename T>
__device__ __forceinline__ void limit_pose(const T cp[3], const T cR[9], const T tp[3], const T tR[9], T maxp, T maxr,
                                           T op[3], T oR[9]) {
    T d[3] = {tp[0] - cp[0], tp[1] - cp[1], tp[2] - cp[2]};
    T n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n > maxp) {
#pragma unroll
        for (int i = 0; i < 3; i++) d[i] = d[i] / n * maxp;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) op[i] = cp[i] + d[i];
    T ct[9] = {cR[0], cR[3], cR[6], cR[1], cR[4], cR[7], cR[2], cR[5], cR[8]}, rel[9], aa[3];
    mat3mul(tR, ct, rel);  // inv of a rotation = transpose
    rotvec_from_mat(rel, aa);
    T ang = sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
    if (ang > maxr) {
        // axisangle2quat (transform_utils.py:108-133) of aa * (maxr / ang), same expressions as oracle/orc_ik.c
        const T k = maxr / ang, v[3] = {aa[0] * k, aa[1] * k, aa[2] * k};
        const T a = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), sn = sin(a / 2);
        T q[4] = {v[0] / a * sn, v[1] / a * sn, v[2] / a * sn, cos(a / 2)}, lim[9];
        quat2mat_xyzw(q, lim);
        mat3mul(lim, cR, oR);
    } else {
#pragma unroll
        for (int i = 0; i < 9; i++) oR[i] = tR[i];
    }
}

// cost of grad_ik.py:168-198 at q; also hands back the position / rotation error norms of the pose (solution_fn, :200-220)
template <typename T>
__device__ __forceinline__ T gik_cost_at(const IkParams& P, const IkArm& A, const T* Rc, const T* pc, const T* q, const T* qs, const T* tp, const T* tR, T* perr, T* rerr) {
    T e[3];
    T d0 = tp[0] - pc[0], d1 = tp[1] - pc[1], d2 = tp[2] - pc[2];
    const T pn = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    T t = (T)P.g_pw * pn;
    T c = t * t;
    angular_error(tR, Rc, e);
    const T rn = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    t = (T)P.g_rw * rn;
    c += t * t;
    T s = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        T ctr = T(0.5) * ((T)A.lo[i] + (T)A.hi[i]), hr = T(0.5) * ((T)A.hi[i] - (T)A.lo[i]);
        t = ((T)P.g_jcw[i] / hr) * (q[i] - ctr);
        s += t * t;
    }
    c += s;
    s = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        t = (T)P.g_jdw[i] * (q[i] - qs[i]);
        s += t * t;
    }
    *perr = pn;
    *rerr = rn;
    return c + s;
}
template <typename T>
__device__ __forceinline__ T gik_cost(const IkParams& P, const IkArm& A, const T* q, const T* qs, const T* tp, const T* tR, T* perr, T* rerr) {
    T Rc[9], pc[3];
    fk<T, 6>(A, q, Rc, pc);
    return gik_cost_at(P, A, Rc, pc, q, qs, tp, tR, perr, rerr);
}

// Forward kinematics with the six joints' exponentials spread over the lanes of a row: `joint(i, Re, pe)` hands this lane joint i's
// exp([S_i] q_i) (from the lane that made it, or its own); the products are fk()'s, in fk()'s order, so the pose is fk()'s bit for bit.
template <typename T, typename F>
__device__ __forceinline__ void fk_spread(const IkArm& A, F&& joint, T R[9], T p[3]) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) R[3 * i + j] = (T)A.site0[4 * i + j];
        p[i] = (T)A.site0[4 * i + 3];
    }
    // (one joint's exponential in flight while the one before it is multiplied in: fetching all six at once costs 144 registers)
    T Rn[9], pn[3];
    joint(5, Rn, pn);
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        T Re[9], pe[3], np[3];
#pragma unroll
        for (int k = 0; k < 9; k++) Re[k] = Rn[k];
#pragma unroll
        for (int k = 0; k < 3; k++) pe[k] = pn[k];
        if (i > 0) joint(i - 1, Rn, pn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 3; r++) np[r] = Re[3 * r] * p[0] + Re[3 * r + 1] * p[1] + Re[3 * r + 2] * p[2] + pe[r];
        mat3mul(Re, R, R);
        p[0] = np[0]; p[1] = np[1]; p[2] = np[2];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// grad_ik.py:8-99 (6-DoF manipulators), one problem per 16-lane row: the reference's loop evaluates its cost function 15 times per
// iteration one after the other -- 12 central-difference points, the two secant points, the new iterate (+ a forward kinematics for
// the solution test, which the last evaluation already contains).  The 12 difference points are independent: lane t < 12 of the row
// takes joint t / 2, sign t & 1; the secant points go to the even / odd lanes; so an iteration is three evaluations deep instead of
// sixteen.  An evaluation's forward kinematics is six exponentials exp([S_i] q_i) (a sine, a cosine and two 3 x 3 products each, in
// double) and six pose products; the exponentials are made ONE per lane and handed round the row (fk_spread): a difference point
// differs from the iterate in one joint, so its lane makes that joint's exponential and takes the other five from the lanes 0..5 that
// keep the iterate's; the two secant points' twelve are made by lanes 0..11; the new iterate's six by lanes 0..5, which keep them for
// the next iteration.  Every exponential and every product is the arithmetic of the one-lane form on the same operands and the gradient
// is gathered in joint order, so the result is bit-identical to it.  All 16 lanes of a row must call this with the same arguments;
// every lane returns the answer.
template <typename T>
// (inlined into its kernels: an out-of-line device function is compiled for the full 512-register budget, which caps its callers
// at one wave per SIMD whatever they ask for)
__device__ __forceinline__ void gradik(const IkParams& P, int arm, const T* qs, const T pos[3], const T tR0[9], int max_it, T* qout) {
    const IkArm& A = P.arm[arm];
    const int lane = threadIdx.x & 63, t = lane & 15, row0 = lane & ~15;
    const T step = (T)P.g_step;
    const int jt = t < 6 ? t : (t < 12 ? t - 6 : t - 12);      // the joint whose exponential this lane makes for a point shared by the row
    const int gi = t < 12 ? (t >> 1) : 0;                       // joint of this lane's difference point (t < 12)
    const T gs = (t & 1) ? step : -step;                // odd lanes +step (p3), even lanes -step (p1)
    T BR[9], Bp[3], ER[9], Ep[3];                       // exponentials: of the iterate (lanes 0..5: joint t), of this lane's point
    auto sel = [](const T* v, int k) -> T { T x = v[0];
#pragma unroll
        for (int i = 1; i < 6; i++) x = (k == i) ? v[i] : x;
        return x; };
    // the row's own point: lane i < 6 makes joint i's exponential
    auto from_row = [&](int i, T* Re, T* pe) {
#pragma unroll
        for (int k = 0; k < 9; k++) Re[k] = __shfl(ER[k], row0 + i, 64);
#pragma unroll
        for (int k = 0; k < 3; k++) pe[k] = __shfl(Ep[k], row0 + i, 64);
    };
    T cR[9], cp[3], tp[3], tR[9], pe, re;
    exp_screw<T>(A.w[jt], A.v[jt], sel(qs, jt), ER, Ep);
    fk_spread<T>(A, from_row, cR, cp);
#pragma unroll
    for (int k = 0; k < 9; k++) BR[k] = ER[k];
#pragma unroll
    for (int k = 0; k < 3; k++) Bp[k] = Ep[k];
    limit_pose(cp, cR, pos, tR0, (T)P.g_maxp, (T)P.g_maxr, tp, tR);
    const T init = gik_cost_at(P, A, cR, cp, qs, qs, tp, tR, &pe, &re);
    T grad[6], work[6], local[6], best[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { work[i] = local[i] = best[i] = qs[i]; grad[i] = 0; }
    T best_cost = init, prev = 0;
    bool done = false;
    for (int it = 0; it < max_it; it++) {
        if (!__any(!done)) break;
        T Rc[9], pc[3];
        // ---- 12 difference points ----
#pragma unroll
        for (int i = 0; i < 6; i++) work[i] = local[i] + ((t < 12 && i == gi) ? gs : T(0));
        // (local[i] - step and local[i] + step exactly as the reference forms them: x + (-step) == x - step)
        exp_screw<T>(A.w[gi], A.v[gi], sel(work, gi), ER, Ep);
        fk_spread<T>(A, [&](int i, T* Re, T* pe_) {
            const bool own = t < 12 && i == gi;
#pragma unroll
            for (int k = 0; k < 9; k++) { const T b = __shfl(BR[k], row0 + i, 64); Re[k] = own ? ER[k] : b; }
#pragma unroll
            for (int k = 0; k < 3; k++) { const T b = __shfl(Bp[k], row0 + i, 64); pe_[k] = own ? Ep[k] : b; }
        }, Rc, pc);
        const T cd1 = gik_cost_at(P, A, Rc, pc, work, qs, tp, tR, &pe, &re);
#pragma unroll
        for (int i = 0; i < 6; i++) grad[i] = __shfl(cd1, row0 + 2 * i + 1, 64) - __shfl(cd1, row0 + 2 * i, 64);
        T sum = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) sum += fabs(grad[i]);
        sum += step;
        const T f = step / sum;
        // ---- secant points: even lanes local - g, odd lanes local + g; lane 2 i + parity makes joint i's exponential of that point ----
#pragma unroll
        for (int i = 0; i < 6; i++) { grad[i] *= f; work[i] = (t & 1) ? local[i] + grad[i] : local[i] - grad[i]; }
        exp_screw<T>(A.w[gi], A.v[gi], sel(work, gi), ER, Ep);
        fk_spread<T>(A, [&](int i, T* Re, T* pe_) {
            const int src = row0 + 2 * i + (t & 1);
#pragma unroll
            for (int k = 0; k < 9; k++) Re[k] = __shfl(ER[k], src, 64);
#pragma unroll
            for (int k = 0; k < 3; k++) pe_[k] = __shfl(Ep[k], src, 64);
        }, Rc, pc);
        const T cd2 = gik_cost_at(P, A, Rc, pc, work, qs, tp, tR, &pe, &re);
        const T p1 = __shfl(cd2, row0, 64), p3 = __shfl(cd2, row0 + 1, 64);
        const T p2 = T(0.5) * (p1 + p3), cd = T(0.5) * (p3 - p1);
        const T jd = (isfinite(cd) && cd != T(0)) ? p2 / cd : T(0);
        T nl[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            T x = local[i] - grad[i] * jd;
            x = x < (T)A.lo[i] ? (T)A.lo[i] : (x > (T)A.hi[i] ? (T)A.hi[i] : x);
            nl[i] = x;
        }
        // ---- the new iterate: cost and pose errors in one evaluation ----
        exp_screw<T>(A.w[jt], A.v[jt], sel(nl, jt), ER, Ep);
        fk_spread<T>(A, from_row, Rc, pc);
        const T lc = gik_cost_at(P, A, Rc, pc, nl, qs, tp, tR, &pe, &re);
        if (!done) {
#pragma unroll
            for (int i = 0; i < 6; i++) local[i] = nl[i];
#pragma unroll
            for (int k = 0; k < 9; k++) BR[k] = ER[k];
#pragma unroll
            for (int k = 0; k < 3; k++) Bp[k] = Ep[k];
            if (lc < best_cost) {
#pragma unroll
                for (int i = 0; i < 6; i++) best[i] = local[i];
                best_cost = lc;
            }
            if (pe < (T)P.g_pthr && re < (T)P.g_rthr) done = true;
            else if (fabs(lc - prev) <= (T)P.g_min_delta) done = true;
            prev = lc;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) qout[i] = qs[i] + (T)P.g_joint_p * (best[i] - qs[i]);
}

}  // namespace avs
