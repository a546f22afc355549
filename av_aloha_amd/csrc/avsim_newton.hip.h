// avsim_newton.hip.h -- primal Newton solver of the soft-constraint problem, one env per wavefront, state in LDS.
//
// The reference leaves MuJoCo's solver at its default (Newton; gym_guided_vision/.../assets/aloha_sim.xml:4 sets only
// noslip_iterations / cone / impratio), so this is the solver whose fixed point the reference actually integrates;
// BASELINE.json's north_star asks for PGS, which stays available (avsim_set_option "solver" 0).  Both minimise the
// same convex problem: PGS on the dual (forces), Newton on the primal (accelerations)
//     min_a  1/2 (a - a_s)^T M (a - a_s) + sum_i s_i(J_i a - aref_i)
// with MuJoCo's per-row costs [EXT]: quadratic equalities, Huber-type dry friction, one-sided limits, three-zone
// elliptic-cone contacts.  oracle/orc_newton.c is the f64 restatement it is tested against.
//
// Mapping: constraint rows are spread over lanes (row i -> lane i % 64); a contact is evaluated by the lane of its first
// row.  H = M + J^T D J (+ cone blocks) is assembled into a packed lower triangle in LDS with returnless LDS atomics,
// factorised in place (one lane per matrix row), the Newton direction comes from two register-resident triangular
// solves (v_readlane broadcasts), and the exact line search is a safeguarded 1-D Newton iteration whose phi', phi''
// are wave-wide DPP sums.
#pragma once
#include "avsim_math.hip.h"

namespace avs {

template <typename real>
struct NewtonArgs {
    // LDS views of one env
    LDS_PTR(real) rowS;          // 8 reals per row: aref, R, 1/(diag+R), 1/diag|0, lo, hi, force, 1/friction
    LDS_PTR(const int) rowI;     // dof windows of the row: (adr 6 | n 4 | tree 3) x 2
    LDS_PTR(const int) rmeta;    // type 2 | id 10 | sub 8 | tree ids
    LDS_PTR(const real) rJ;      // 16 reals per row
    LDS_PTR(const real) M;       // per-tree dense blocks
    LDS_PTR(real) a;             // qacc (in: start point, out: solution)
    LDS_PTR(const real) as;      // qacc_smooth
    LDS_PTR(real) H;             // packed lower triangle nv(nv+1)/2
    LDS_PTR(real) g;             // gradient / scratch vector nv
    LDS_PTR(real) dl;            // search direction nv
    LDS_PTR(real) x;             // trial point nv
    LDS_PTR(int) czone;          // per contact-head row: zone of the contact (indexed by row)
    LDS_PTR(const int) tree_dofadr;
    LDS_PTR(const int) tree_dofnum;
    LDS_PTR(const int) tree_madr;
    LDS_PTR(const int) dof_tree;
    int nv, nefc, ntree, iters;
    real tol, scale, ls_tol;
};

enum { NR_EQ = 0, NR_FLOSS = 1, NR_LIMIT = 2, NR_CONTACT = 3 };

// J_i . v for row i (v in LDS, dof indexed)
template <typename real>
AVS_DEV real nrow_dot(const NewtonArgs<real>& A, int i, LDS_PTR(const real) v) {
    const int ra = A.rowI[i];
    const int a0 = ra & 63, nA = (ra >> 6) & 15, b0 = (ra >> 13) & 63, nB = (ra >> 19) & 15;
    LDS_PTR(const real) J = A.rJ + ROW_W * i;
    real s = 0;
#pragma unroll
    for (int k = 0; k < TREE_W; k++) {
        if (k < nA) s += J[k] * v[a0 + k];
        if (k < nB) s += J[TREE_W + k] * v[b0 + k];
    }
    return s;
}

// scalar rows: force and curvature at constraint-space residual z
template <typename real>
AVS_DEV void nrow_scalar(int type, real z, real R, real eta, real* f, real* h) {
    const real D = real(1) / R;
    if (type == NR_EQ) { *f = -D * z; *h = D; }
    else if (type == NR_FLOSS) {
        if (z <= -R * eta) { *f = eta; *h = 0; }
        else if (z >= R * eta) { *f = -eta; *h = 0; }
        else { *f = -D * z; *h = D; }
    } else {   // limit, frictionless contact
        if (z < 0) { *f = -D * z; *h = D; } else { *f = 0; *h = 0; }
    }
}

template <typename real>
AVS_DEV real nrow_scalar_cost(int type, real z, real R, real eta) {
    const real D = real(1) / R;
    if (type == NR_EQ) return real(0.5) * D * z * z;
    if (type == NR_FLOSS) {
        if (z <= -R * eta) return -eta * z - real(0.5) * R * eta * eta;
        if (z >= R * eta) return eta * z - real(0.5) * R * eta * eta;
        return real(0.5) * D * z * z;
    }
    return z < 0 ? real(0.5) * D * z * z : real(0);
}

// elliptic contact with rows i..i+dim-1 at residuals jar[]: zone, forces, and (middle zone) the dim x dim curvature block
template <typename real>
AVS_DEV int ncontact(const NewtonArgs<real>& A, int i, int dim, const real* jar, real* f, real* C, bool want_C, real* cost = nullptr) {
    const real R0 = A.rowS[8 * i + 1];
    if (cost) *cost = 0;
    if (dim == 1) {
        real h;
        nrow_scalar<real>(NR_LIMIT, jar[0], R0, real(0), f, &h);
        if (cost) *cost = nrow_scalar_cost<real>(NR_LIMIT, jar[0], R0, real(0));
        return jar[0] < 0 ? 1 : 0;
    }
    const real R1 = A.rowS[8 * (i + 1) + 1];
    real fr[6];   // friction coefficient of row j (j >= 1)
    fr[0] = 0;
#pragma unroll
    for (int j = 1; j < 6; j++) fr[j] = j < dim ? real(1) / A.rowS[8 * (i + j) + 7] : real(0);
    const real mu = fr[1] * sqrt(R1 / R0);
    real U[6], t2 = 0;
    U[0] = jar[0] * mu;
#pragma unroll
    for (int j = 1; j < 6; j++) { U[j] = j < dim ? jar[j] * fr[j] : real(0); t2 += U[j] * U[j]; }
    const real N = U[0], T = sqrt(t2);
    if (N >= mu * T || (T <= 0 && N >= 0)) {
#pragma unroll
        for (int j = 0; j < 6; j++) f[j] = 0;
        return 0;
    }
    if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const real Dj = real(1) / A.rowS[8 * (i + (j < dim ? j : 0)) + 1];
            f[j] = j < dim ? -jar[j] * Dj : real(0);
            if (cost && j < dim) *cost += real(0.5) * Dj * jar[j] * jar[j];
        }
        return 1;
    }
    const real Dm = (real(1) / R0) / tmax(real(1e-15), mu * mu * (1 + mu * mu)), NT = N - mu * T;
    f[0] = -Dm * NT * mu;
    if (cost) *cost = real(0.5) * Dm * NT * NT;
#pragma unroll
    for (int j = 1; j < 6; j++) f[j] = j < dim ? -f[0] / T * U[j] * fr[j] : real(0);
    if (want_C) {
        real S[6];
        S[0] = mu;
#pragma unroll
        for (int j = 1; j < 6; j++) S[j] = fr[j];
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int q = 0; q < 6; q++) {
                real h;
                if (p == 0 && q == 0) h = Dm;
                else if (p == 0 || q == 0) { const real u = p == 0 ? U[q] : U[p]; h = -Dm * mu * u / T; }
                else h = Dm * mu * mu * U[p] * U[q] / (T * T) - Dm * NT * mu * ((p == q ? real(1) / T : real(0)) - U[p] * U[q] / (T * T * T));
                C[6 * p + q] = (p < dim && q < dim) ? h * S[p] * S[q] : real(0);
            }
    }
    return 2;
}

// global dof of window slot s (0..15) of a row, or -1
AVS_DEV int nslot_dof(int ra, int s) {
    const int sh = s < TREE_W ? 0 : 13, k = s & (TREE_W - 1);
    return k < ((ra >> (sh + 6)) & 15) ? ((ra >> sh) & 63) + k : -1;
}

// H[p,q] += w * (Jp_row slot outer Jq_row slot) over the two rows' windows (rp, rq may be the same row)
template <typename real>
AVS_DEV void nouter(const NewtonArgs<real>& A, int rp, int rq, real w, bool sym_same) {
    const int rap = A.rowI[rp], raq = A.rowI[rq];
    LDS_PTR(const real) Jp = A.rJ + ROW_W * rp;
    LDS_PTR(const real) Jq = A.rJ + ROW_W * rq;
    for (int s = 0; s < ROW_W; s++) {
        const int gp = nslot_dof(rap, s);
        if (gp < 0) continue;
        const real jp = Jp[s] * w;
        for (int t = 0; t < ROW_W; t++) {
            const int gq = nslot_dof(raq, t);
            if (gq < 0 || gq > gp) continue;                    // lower triangle only
            real v = jp * Jq[t];
            if (!sym_same && gq == gp) { /* diagonal entry of an off-diagonal block pair is added once per ordered pair */ }
            __hip_atomic_fetch_add(A.H + gp * (gp + 1) / 2 + gq, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

#define NSYNC()                                              \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                     \
    } while (0)

template <typename real>
__device__ __attribute__((noinline)) int newton_solve(NewtonArgs<real> A) {
    const int lane = threadIdx.x & 63;
    const int nv = A.nv, ne = A.nefc;
    int used = 0;
    // ---- start from the warm start (already in a) or from the smooth acceleration, whichever costs less ----
    {
        real c[2];
        for (int trial = 0; trial < 2; trial++) {
            LDS_PTR(const real) v = trial == 0 ? (LDS_PTR(const real))A.a : A.as;
            real cs = 0;
            for (int i = lane; i < ne; i += 64) {
                const int meta = A.rmeta[i], type = meta & 3, sub = (meta >> 12) & 255;
                if (type != NR_CONTACT) cs += nrow_scalar_cost<real>(type, nrow_dot(A, i, v) - A.rowS[8 * i], A.rowS[8 * i + 1], A.rowS[8 * i + 5]);
                else if (sub == 0) {
                    const int dim = A.czone[i] >> 8;
                    real jar[6], f[6], C[1], cc;
#pragma unroll
                    for (int j = 0; j < 6; j++) jar[j] = j < dim ? nrow_dot(A, i + j, v) - A.rowS[8 * (i + j)] : real(0);
                    ncontact(A, i, dim, jar, f, C, false, &cc);
                    cs += cc;
                }
            }
            for (int k = lane; k < nv; k += 64) {
                const int t = A.dof_tree[k], a0 = A.tree_dofadr[t], n = A.tree_dofnum[t], kk = k - a0;
                real sacc = 0;
                for (int j = 0; j < n; j++) sacc += A.M[A.tree_madr[t] + kk * n + j] * (v[a0 + j] - A.as[a0 + j]);
                cs += real(0.5) * sacc * (v[k] - A.as[k]);
            }
            c[trial] = wave_sum(cs);
        }
        if (!(c[0] < c[1])) {
            for (int k = lane; k < nv; k += 64) A.a[k] = A.as[k];
            NSYNC();
        }
    }
    for (int it = 0; it < A.iters; it++) {
        used++;
        // ---- gradient g = M (a - a_s) - J^T f(a), forces written to rowS.f ----
        for (int i = lane; i < ne; i += 64) {
            const int meta = A.rmeta[i], type = meta & 3, sub = (meta >> 12) & 255;
            if (type != NR_CONTACT) {
                real f, h;
                nrow_scalar<real>(type, nrow_dot(A, i, (LDS_PTR(const real))A.a) - A.rowS[8 * i], A.rowS[8 * i + 1], A.rowS[8 * i + 5], &f, &h);
                A.rowS[8 * i + 6] = f;
            } else if (sub == 0) {
                const int dim = A.czone[i] >> 8;
                real jar[6], f[6], C[1];
#pragma unroll
                for (int j = 0; j < 6; j++) jar[j] = j < dim ? nrow_dot(A, i + j, (LDS_PTR(const real))A.a) - A.rowS[8 * (i + j)] : real(0);
                const int zn = ncontact(A, i, dim, jar, f, C, false);
                A.czone[i] = (dim << 8) | zn;
#pragma unroll
                for (int j = 0; j < 6; j++) if (j < dim) A.rowS[8 * (i + j) + 6] = f[j];
            }
        }
        NSYNC();
        real gn2 = 0;
        for (int k = lane; k < nv; k += 64) {
            const int t = A.dof_tree[k], a0 = A.tree_dofadr[t], n = A.tree_dofnum[t], kk = k - a0;
            real s = 0;
            for (int j = 0; j < n; j++) s += A.M[A.tree_madr[t] + kk * n + j] * (A.a[a0 + j] - A.as[a0 + j]);
            for (int i = 0; i < ne; i++) {
                const int ra = A.rowI[i];
                const int da = k - (ra & 63), db = k - ((ra >> 13) & 63);
                if ((unsigned)da < (unsigned)((ra >> 6) & 15)) s -= A.rJ[ROW_W * i + da] * A.rowS[8 * i + 6];
                else if ((unsigned)db < (unsigned)((ra >> 19) & 15)) s -= A.rJ[ROW_W * i + TREE_W + db] * A.rowS[8 * i + 6];
            }
            A.g[k] = s;
            gn2 += s * s;
        }
        gn2 = wave_sum(gn2);
        if (sqrt(gn2) * A.scale < A.tol) break;
        // ---- Hessian: packed lower triangle ----
        for (int e = lane; e < nv * (nv + 1) / 2; e += 64) A.H[e] = 0;
        NSYNC();
        for (int e = lane; e < nv * TREE_W; e += 64) {           // M blocks
            const int k = e >> 3, j8 = e & 7, t = A.dof_tree[k], a0 = A.tree_dofadr[t], n = A.tree_dofnum[t], kk = k - a0;
            if (j8 < n && j8 <= kk) A.H[k * (k + 1) / 2 + a0 + j8] = A.M[A.tree_madr[t] + kk * n + j8];
        }
        NSYNC();
        for (int i = lane; i < ne; i += 64) {
            const int meta = A.rmeta[i], type = meta & 3, sub = (meta >> 12) & 255;
            if (type != NR_CONTACT) {
                real f, h;
                nrow_scalar<real>(type, nrow_dot(A, i, (LDS_PTR(const real))A.a) - A.rowS[8 * i], A.rowS[8 * i + 1], A.rowS[8 * i + 5], &f, &h);
                if (h != 0) nouter(A, i, i, h, true);
            } else {
                const int head = i - sub, zn = A.czone[head] & 255, dim = A.czone[head] >> 8;
                if (zn == 1) nouter(A, i, i, real(1) / A.rowS[8 * i + 1], true);
                else if (zn == 2 && sub == 0) {
                    real jar[6], f[6], C[36];
#pragma unroll
                    for (int j = 0; j < 6; j++) jar[j] = j < dim ? nrow_dot(A, i + j, (LDS_PTR(const real))A.a) - A.rowS[8 * (i + j)] : real(0);
                    ncontact(A, i, dim, jar, f, C, true);
                    // J_c^T C J_c, lower triangle: ordered pairs (p,q) contribute their lower part; symmetric C
                    for (int p = 0; p < dim; p++)
                        for (int q = 0; q < dim; q++) {
                            real w = 0;
#pragma unroll
                            for (int u = 0; u < 36; u++) w = (u == 6 * p + q) ? C[u] : w;
                            nouter(A, i + p, i + q, w, false);
                        }
                }
            }
        }
        NSYNC();
        // ---- Cholesky in place, one lane per matrix row ----
        for (int j = 0; j < nv; j++) {
            const real dj = sqrt(tmax(A.H[j * (j + 1) / 2 + j], real(1e-30)));
            real lij = 0;
            if (lane > j && lane < nv) lij = A.H[lane * (lane + 1) / 2 + j] / dj;
            NSYNC();
            if (lane == j) A.H[j * (j + 1) / 2 + j] = dj;
            if (lane > j && lane < nv) A.H[lane * (lane + 1) / 2 + j] = lij;
            NSYNC();
            if (lane > j && lane < nv)
                for (int k = j + 1; k <= lane; k++) A.H[lane * (lane + 1) / 2 + k] -= lij * A.H[k * (k + 1) / 2 + j];
            NSYNC();
        }
        // ---- dl = -H^-1 g : register-resident triangular solves (lane i holds component i) ----
        real xi = lane < nv ? -A.g[lane] : real(0);
        const real dinv = lane < nv ? real(1) / A.H[lane * (lane + 1) / 2 + lane] : real(0);
        for (int j = 0; j < nv; j++) {
            const real yj = lane_get(xi * dinv, j);
            if (lane == j) xi = yj;
            if (lane > j && lane < nv) xi -= A.H[lane * (lane + 1) / 2 + j] * yj;
        }
        for (int j = nv - 1; j >= 0; j--) {
            const real yj = lane_get(xi * dinv, j);
            if (lane == j) xi = yj;
            if (lane < j) xi -= A.H[j * (j + 1) / 2 + lane] * yj;
        }
        if (lane < nv) A.dl[lane] = xi;
        NSYNC();
        // ---- exact line search along dl ----
        real q1 = 0, q2 = 0;
        for (int k = lane; k < nv; k += 64) {
            const int t = A.dof_tree[k], a0 = A.tree_dofadr[t], n = A.tree_dofnum[t], kk = k - a0;
            real s = 0;
            for (int j = 0; j < n; j++) s += A.M[A.tree_madr[t] + kk * n + j] * A.dl[a0 + j];
            q2 += s * A.dl[k];
            q1 += s * (A.a[k] - A.as[k]);
        }
        q1 = wave_sum(q1);
        q2 = wave_sum(q2);
        real alpha = 0, lo = 0, hi = -1, dphi0 = 0;
        for (int ls = 0; ls < 40; ls++) {
            // trial point, then phi'(alpha), phi''(alpha)
            for (int k = lane; k < nv; k += 64) A.x[k] = A.a[k] + alpha * A.dl[k];
            NSYNC();
            real gsum = 0, hsum = 0;
            for (int i = lane; i < ne; i += 64) {
                const int meta = A.rmeta[i], type = meta & 3, sub = (meta >> 12) & 255;
                if (type != NR_CONTACT) {
                    real f, h;
                    const real jvi = nrow_dot(A, i, (LDS_PTR(const real))A.dl);
                    nrow_scalar<real>(type, nrow_dot(A, i, (LDS_PTR(const real))A.x) - A.rowS[8 * i], A.rowS[8 * i + 1], A.rowS[8 * i + 5], &f, &h);
                    gsum -= f * jvi;
                    hsum += h * jvi * jvi;
                } else if (sub == 0) {
                    const int dim = A.czone[i] >> 8;
                    real jar[6], jv[6], f[6], C[36];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        jar[j] = j < dim ? nrow_dot(A, i + j, (LDS_PTR(const real))A.x) - A.rowS[8 * (i + j)] : real(0);
                        jv[j] = j < dim ? nrow_dot(A, i + j, (LDS_PTR(const real))A.dl) : real(0);
                    }
                    const int zn = ncontact(A, i, dim, jar, f, C, true);
#pragma unroll
                    for (int j = 0; j < 6; j++) gsum -= f[j] * jv[j];
                    if (zn == 1) {
#pragma unroll
                        for (int j = 0; j < 6; j++) if (j < dim) hsum += jv[j] * jv[j] / A.rowS[8 * (i + j) + 1];
                    } else if (zn == 2) {
#pragma unroll
                        for (int p = 0; p < 6; p++)
#pragma unroll
                            for (int q = 0; q < 6; q++) hsum += jv[p] * C[6 * p + q] * jv[q];
                    }
                }
            }
            const real dphi = q1 + alpha * q2 + wave_sum(gsum), ddphi = q2 + wave_sum(hsum);
            if (ls == 0) {
                dphi0 = dphi;
                if (!(dphi0 < 0)) break;
                alpha = -dphi0 / ddphi;
                continue;
            }
            if (fabs(dphi) < A.ls_tol * fabs(dphi0)) break;
            if (dphi < 0) lo = alpha; else hi = alpha;
            real nx = alpha - dphi / ddphi;
            if (hi < 0) { if (!(nx > lo)) nx = 2 * alpha + real(1e-12); }
            else if (!(nx > lo && nx < hi)) nx = real(0.5) * (lo + hi);
            if (fabs(nx - alpha) < real(1e-7) * A.ls_tol * (1 + fabs(alpha))) { alpha = nx; break; }
            alpha = nx;
        }
        if (!(dphi0 < 0)) break;
        real st2 = 0;
        for (int k = lane; k < nv; k += 64) { const real s = alpha * A.dl[k]; A.a[k] += s; st2 += s * s; }
        st2 = wave_sum(st2);
        NSYNC();
        if (sqrt(st2) * A.scale < real(1e-2) * A.tol) break;
    }
    // ---- forces at the solution ----
    for (int i = lane; i < ne; i += 64) {
        const int meta = A.rmeta[i], type = meta & 3, sub = (meta >> 12) & 255;
        if (type != NR_CONTACT) {
            real f, h;
            nrow_scalar<real>(type, nrow_dot(A, i, (LDS_PTR(const real))A.a) - A.rowS[8 * i], A.rowS[8 * i + 1], A.rowS[8 * i + 5], &f, &h);
            A.rowS[8 * i + 6] = f;
        } else if (sub == 0) {
            const int dim = A.czone[i] >> 8;
            real jar[6], f[6], C[1];
#pragma unroll
            for (int j = 0; j < 6; j++) jar[j] = j < dim ? nrow_dot(A, i + j, (LDS_PTR(const real))A.a) - A.rowS[8 * (i + j)] : real(0);
            ncontact(A, i, dim, jar, f, C, false);
#pragma unroll
            for (int j = 0; j < 6; j++) if (j < dim) A.rowS[8 * (i + j) + 6] = f[j];
        }
    }
    NSYNC();
    return used;
}

}  // namespace avs
