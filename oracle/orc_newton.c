/* oracle/orc_newton.c -- primal Newton solver on MuJoCo's soft-constraint cost (the reference's actual solver:
 * aloha_sim.xml:4 leaves `solver` at MuJoCo's default, Newton [EXT]).  TEST INFRASTRUCTURE ONLY (orc.h).
 *
 *   minimise over qacc:  1/2 (a - a_s)^T M (a - a_s) + sum_i s_i(J_i a - aref_i)
 *
 * with the per-row costs of the MuJoCo documentation: quadratic equality rows, Huber-type dry friction,
 * one-sided quadratic limits, and the three-zone elliptic-cone contact cost (top: free, bottom: quadratic in all
 * rows, middle: 1/2 Dm (N - mu T)^2 in the scaled variables U = (mu jar_0, f_j jar_j)).  Its optimum is the optimum
 * of the dual problem orc_solve()'s PGS iterates on (same R, same cone), which tests use as a cross-check.
 * Exact Hessian + dense Cholesky, safeguarded 1-D Newton line search on the piecewise-quadratic cost.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "orc.h"

#define MINVAL 1e-15

typedef struct {
    double cost, dphi, ddphi; /* line-search accumulators */
} lsacc;

/* cost / force / (optionally) Hessian block of one contact in the middle zone; jar, out arrays are per row of the contact */
static int cone_zone(const orc_contact* c, const double* D, double mu, const double* jar, double* U, double* N, double* T) {
    U[0] = jar[0] * mu;
    double t2 = 0;
    for (int j = 1; j < c->dim; j++) { U[j] = jar[j] * c->friction[j - 1]; t2 += U[j] * U[j]; }
    *N = U[0];
    *T = sqrt(t2);
    if (*N >= mu * (*T) || (*T <= 0 && *N >= 0)) return 0;           /* top: no force */
    if (mu * (*N) + (*T) <= 0 || (*T <= 0 && *N < 0)) return 1;      /* bottom: quadratic */
    return 2;                                                        /* middle: cone surface */
}

/* evaluate all rows at jar; fills force (if non-NULL), returns the constraint cost; hdiag[i] = second derivative for
 * scalar rows (0 when inactive/linear); zone[ci] for contacts */
static double eval_rows(const orc_data* d, const double* jar, double* force, double* hdiag, int* zone) {
    double cost = 0;
    for (int i = 0; i < d->nefc; i++) {
        double D = d->efc_D[i], R = d->efc_R[i], z = jar[i];
        switch (d->efc_type[i]) {
            case ORC_EQ:
                cost += 0.5 * D * z * z;
                if (force) force[i] = -D * z;
                if (hdiag) hdiag[i] = D;
                break;
            case ORC_FLOSS: {
                double eta = d->efc_floss[i];
                if (z <= -R * eta) { cost += -eta * z - 0.5 * R * eta * eta; if (force) force[i] = eta; if (hdiag) hdiag[i] = 0; }
                else if (z >= R * eta) { cost += eta * z - 0.5 * R * eta * eta; if (force) force[i] = -eta; if (hdiag) hdiag[i] = 0; }
                else { cost += 0.5 * D * z * z; if (force) force[i] = -D * z; if (hdiag) hdiag[i] = D; }
                break;
            }
            case ORC_LIMIT:
                if (z < 0) { cost += 0.5 * D * z * z; if (force) force[i] = -D * z; if (hdiag) hdiag[i] = D; }
                else { if (force) force[i] = 0; if (hdiag) hdiag[i] = 0; }
                break;
            case ORC_CONTACT: {
                const orc_contact* c = &d->contact[d->efc_id[i]];
                if (i != c->efc_adr) break; /* handled at the first row */
                if (c->dim == 1) {
                    if (z < 0) { cost += 0.5 * D * z * z; if (force) force[i] = -D * z; if (hdiag) hdiag[i] = D; }
                    else { if (force) force[i] = 0; if (hdiag) hdiag[i] = 0; }
                    if (zone) zone[d->efc_id[i]] = z < 0 ? 1 : 0;
                    break;
                }
                double mu = c->friction[0] * sqrt(d->efc_R[i + 1] / d->efc_R[i]);
                double U[6], N, T;
                int zn = cone_zone(c, d->efc_D + i, mu, jar + i, U, &N, &T);
                if (zone) zone[d->efc_id[i]] = zn;
                if (zn == 0) {
                    for (int j = 0; j < c->dim; j++) { if (force) force[i + j] = 0; if (hdiag) hdiag[i + j] = 0; }
                } else if (zn == 1) {
                    for (int j = 0; j < c->dim; j++) {
                        double Dj = d->efc_D[i + j];
                        cost += 0.5 * Dj * jar[i + j] * jar[i + j];
                        if (force) force[i + j] = -Dj * jar[i + j];
                        if (hdiag) hdiag[i + j] = Dj;
                    }
                } else {
                    double Dm = d->efc_D[i] / fmax(MINVAL, mu * mu * (1 + mu * mu)), NT = N - mu * T;
                    cost += 0.5 * Dm * NT * NT;
                    if (force) {
                        force[i] = -Dm * NT * mu;
                        for (int j = 1; j < c->dim; j++) force[i + j] = -force[i] / T * U[j] * c->friction[j - 1];
                    }
                    if (hdiag) for (int j = 0; j < c->dim; j++) hdiag[i + j] = 0; /* dense block, see cone_hessian */
                }
                break;
            }
        }
    }
    return cost;
}

/* dim x dim Hessian of the middle-zone cost with respect to the contact's jar rows */
static void cone_hessian(const orc_data* d, int i, const double* jar, double* C) {
    const orc_contact* c = &d->contact[d->efc_id[i]];
    int dim = c->dim;
    double mu = c->friction[0] * sqrt(d->efc_R[i + 1] / d->efc_R[i]);
    double U[6], N, T, S[6];
    cone_zone(c, d->efc_D + i, mu, jar + i, U, &N, &T);
    double Dm = d->efc_D[i] / fmax(MINVAL, mu * mu * (1 + mu * mu)), NT = N - mu * T;
    S[0] = mu;
    for (int j = 1; j < dim; j++) S[j] = c->friction[j - 1];
    for (int a = 0; a < dim; a++)
        for (int b = 0; b < dim; b++) {
            double h;
            if (a == 0 && b == 0) h = Dm;
            else if (a == 0 || b == 0) { int j = a == 0 ? b : a; h = -Dm * mu * U[j] / T; }
            else h = Dm * mu * mu * U[a] * U[b] / (T * T) - Dm * NT * mu * ((a == b ? 1.0 / T : 0.0) - U[a] * U[b] / (T * T * T));
            C[a * dim + b] = h * S[a] * S[b];
        }
}

static int chol_dense(double* A, int n) { /* in place, lower */
    for (int j = 0; j < n; j++) {
        double dd = A[j * n + j];
        for (int k = 0; k < j; k++) dd -= A[j * n + k] * A[j * n + k];
        if (dd <= 0) return -1;
        dd = sqrt(dd);
        A[j * n + j] = dd;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / dd;
        }
    }
    return 0;
}
static void chol_dense_solve(const double* L, double* x, int n) {
    for (int i = 0; i < n; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k]; x[i] = s / L[i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
}

/* line-search derivatives at step alpha: phi'(alpha), phi''(alpha) */
static void ls_eval(const orc_data* d, const double* jar0, const double* jv, double alpha, double q1, double q2, double* dphi, double* ddphi,
                    double* jar, double* force, double* hdiag, int* zone) {
    int ne = d->nefc;
    for (int i = 0; i < ne; i++) jar[i] = jar0[i] + alpha * jv[i];
    eval_rows(d, jar, force, hdiag, zone);
    double g = q1 + alpha * q2, h = q2;
    for (int i = 0; i < ne; i++) { g -= force[i] * jv[i]; h += hdiag[i] * jv[i] * jv[i]; }
    for (int ci = 0; ci < d->ncon; ci++) {
        const orc_contact* c = &d->contact[ci];
        if (c->efc_adr < 0 || c->dim == 1 || zone[ci] != 2) continue;
        double C[36];
        cone_hessian(d, c->efc_adr, jar, C);
        for (int a = 0; a < c->dim; a++)
            for (int b = 0; b < c->dim; b++) h += jv[c->efc_adr + a] * C[a * c->dim + b] * jv[c->efc_adr + b];
    }
    *dphi = g;
    *ddphi = h;
}

void orc_solve_newton(orc_data* d) {
    const orc_model* m = d->m;
    int nv = m->nv, ne = d->nefc;
    double* a = d->qacc;
    double *jar = (double*)calloc(ne + 1, 8), *jar2 = (double*)calloc(ne + 1, 8), *force = (double*)calloc(ne + 1, 8), *hd = (double*)calloc(ne + 1, 8),
           *jv = (double*)calloc(ne + 1, 8), *g = (double*)calloc(nv, 8), *dl = (double*)calloc(nv, 8), *Md = (double*)calloc(nv, 8),
           *H = (double*)calloc((size_t)nv * nv, 8), *tmp = (double*)calloc(nv, 8);
    int* zone = (int*)calloc(ORC_MAXCON, sizeof(int));
    const int dbg = getenv("ORC_DEBUG_NEWTON") != NULL;      /* per-iteration trace on stderr (tools/dbg_newton_options.py) */
    /* start: warm start or smooth acceleration, whichever costs less */
    double best = 1e300;
    for (int trial = 0; trial < 2; trial++) {
        const double* x = trial == 0 ? d->qacc_warmstart : d->qacc_smooth;
        for (int i = 0; i < ne; i++) { double s = -d->efc_aref[i]; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * x[k]; jar[i] = s; }
        double c = eval_rows(d, jar, NULL, NULL, NULL);
        for (int i = 0; i < nv; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->M[i * nv + k] * (x[k] - d->qacc_smooth[k]); c += 0.5 * s * (x[i] - d->qacc_smooth[i]); }
        if (c < best) { best = c; memcpy(a, x, sizeof(double) * nv); }
    }
    d->stat_sweeps = 0;
    for (int it = 0; it < d->newton_iters; it++) {
        d->stat_sweeps++;
        for (int i = 0; i < ne; i++) { double s = -d->efc_aref[i]; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * a[k]; jar[i] = s; }
        eval_rows(d, jar, force, hd, zone);
        /* gradient M (a - a_s) - J^T f */
        double gnorm = 0;
        for (int i = 0; i < nv; i++) {
            double s = 0;
            for (int k = 0; k < nv; k++) s += d->M[i * nv + k] * (a[k] - d->qacc_smooth[k]);
            for (int r = 0; r < ne; r++) s -= d->efc_J[(size_t)r * nv + i] * force[r];
            g[i] = s;
            gnorm += s * s;
        }
        if (dbg) fprintf(stderr, "  orc newton it %d: |g| scaled %.6e\n", it, sqrt(gnorm) * d->pgs_scale);
        if (sqrt(gnorm) * d->pgs_scale < d->newton_tol) break;
        /* Hessian */
        memcpy(H, d->M, sizeof(double) * nv * nv);
        for (int r = 0; r < ne; r++) {
            if (hd[r] == 0) continue;
            const double* Jr = d->efc_J + (size_t)r * nv;
            for (int i = 0; i < nv; i++) { if (Jr[i] == 0) continue; for (int k = 0; k < nv; k++) H[i * nv + k] += hd[r] * Jr[i] * Jr[k]; }
        }
        for (int ci = 0; ci < d->ncon; ci++) {
            const orc_contact* c = &d->contact[ci];
            if (c->efc_adr < 0 || c->dim == 1 || zone[ci] != 2) continue;
            double C[36];
            cone_hessian(d, c->efc_adr, jar, C);
            for (int p = 0; p < c->dim; p++)
                for (int q = 0; q < c->dim; q++) {
                    const double *Jp = d->efc_J + (size_t)(c->efc_adr + p) * nv, *Jq = d->efc_J + (size_t)(c->efc_adr + q) * nv;
                    double w = C[p * c->dim + q];
                    for (int i = 0; i < nv; i++) { if (Jp[i] == 0) continue; for (int k = 0; k < nv; k++) H[i * nv + k] += w * Jp[i] * Jq[k]; }
                }
        }
        if (chol_dense(H, nv)) break;
        for (int i = 0; i < nv; i++) dl[i] = -g[i];
        chol_dense_solve(H, dl, nv);
        /* line search on phi(alpha) = cost(a + alpha dl) */
        double q1 = 0, q2 = 0;
        for (int i = 0; i < nv; i++) { double s = 0; for (int k = 0; k < nv; k++) s += d->M[i * nv + k] * dl[k]; Md[i] = s; q2 += s * dl[i]; q1 += s * (a[i] - d->qacc_smooth[i]); }
        for (int r = 0; r < ne; r++) { double s = 0; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)r * nv + k] * dl[k]; jv[r] = s; }
        double dphi0, ddphi0, dphi, ddphi;
        ls_eval(d, jar, jv, 0.0, q1, q2, &dphi0, &ddphi0, jar2, force, hd, zone);
        if (!(dphi0 < 0)) break;
        if (dbg && it == 2) {       /* the cost and its two derivatives along the line, sampled: finite differences must agree with them */
            double prev_c = 0, prev_g = 0;
            for (int k = 0; k <= 20; k++) {
                double al = 0.05 * k, g1, h1;
                ls_eval(d, jar, jv, al, q1, q2, &g1, &h1, jar2, force, hd, zone);
                double cst = eval_rows(d, jar2, NULL, NULL, NULL);
                /* smooth part: 1/2 (a + al dl - as)' M (a + al dl - as) = const + al q1 + 1/2 al^2 q2 */
                cst += al * q1 + 0.5 * al * al * q2;
                int nm = 0, nb = 0, nt = 0;
                for (int ci = 0; ci < d->ncon; ci++) if (d->contact[ci].efc_adr >= 0) { nm += zone[ci] == 2; nb += zone[ci] == 1; nt += zone[ci] == 0; }
                fprintf(stderr, "         sample alpha %.2f: cost %.9e dphi %.6e ddphi %.6e | FD dphi %.6e FD ddphi %.6e | zones top %d middle %d bottom %d\n", al, cst, g1, h1,
                        k ? (cst - prev_c) / 0.05 : 0.0, k ? (g1 - prev_g) / 0.05 : 0.0, nt, nm, nb);
                prev_c = cst; prev_g = g1;
            }
        }
        /* Newton on phi' with a bracket [lo, hi] (phi'(lo) < 0 <= phi'(hi)), safeguarded as rtsafe (Numerical Recipes 9.4): a Newton step
         * that leaves the bracket OR is longer than half the step before last is replaced by the bracket's midpoint.  The second rule
         * matters: the cone's middle-zone cost is not quadratic, phi'' along a line can be four times larger in the middle than at the ends
         * (a sigmoid-shaped phi'), and plain Newton then cycles between the two flat ends of the bracket for ever -- it did, on the two-arm
         * grasp of HookPackage (tools/dbg_newton_options.py, tools/dbg_state_hook7_183.npz): 100 stalled Newton iterations on the device,
         * a lucky parity of the evaluation count in this oracle.  (Rounds 1-4 had the first rule only.)  Same rule, same numbers in
         * avsim_newton.hip.h. */
        double lo = 0, hi = -1, alpha = -dphi0 / ddphi0, glo = dphi0, dxold = alpha, dx = alpha;
        for (int ls = 0; ls < d->ls_iters; ls++) {
            ls_eval(d, jar, jv, alpha, q1, q2, &dphi, &ddphi, jar2, force, hd, zone);
            if (dbg && it == 2) fprintf(stderr, "         orc ls %d: alpha %.12e dphi %.6e ddphi %.6e lo %.6e hi %.6e (ddphi0 %.6e)\n", ls + 1, alpha, dphi, ddphi, lo, hi, ddphi0);
            if (fabs(dphi) < d->ls_tol * fabs(dphi0) + 1e-300) break;
            if (dphi < 0) { lo = alpha; glo = dphi; } else hi = alpha;
            double nx = alpha - dphi / ddphi;
            if (hi < 0) { if (!(nx > lo)) nx = 2 * alpha + 1e-12; }
            else if (!(nx > lo && nx < hi) || fabs(nx - alpha) > 0.5 * fabs(dxold)) nx = 0.5 * (lo + hi);
            dxold = dx;
            dx = nx - alpha;
            if (fabs(nx - alpha) < 1e-14 * (1 + fabs(alpha))) { alpha = nx; break; }
            alpha = nx;
        }
        (void)glo;
        if (dbg) {
            int nmid = 0, nbot = 0;
            for (int ci = 0; ci < d->ncon; ci++) { nmid += zone[ci] == 2; nbot += zone[ci] == 1; }
            fprintf(stderr, "      orc line search: dphi0 %.6e alpha %.6e q1 %.6e q2 %.6e lo %.6e hi %.6e  zones at the new point: middle %d bottom %d\n", dphi0, alpha, q1, q2, lo, hi, nmid, nbot);
        }
        double step2 = 0;
        for (int i = 0; i < nv; i++) { a[i] += alpha * dl[i]; step2 += alpha * dl[i] * alpha * dl[i]; }
        if (sqrt(step2) * d->pgs_scale < 1e-2 * d->newton_tol) break;
        /* MuJoCo's improvement test [EXT]: the cost decrease of this iteration, -alpha phi'(0) / 2 to second order, scaled */
        if (-0.5 * alpha * dphi0 * d->pgs_scale < d->newton_tol) break;
    }
    /* forces and qfrc_constraint at the solution */
    for (int i = 0; i < ne; i++) { double s = -d->efc_aref[i]; for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * a[k]; jar[i] = s; }
    eval_rows(d, jar, d->efc_force, hd, zone);
    for (int k = 0; k < nv; k++) { double s = 0; for (int i = 0; i < ne; i++) s += d->efc_J[(size_t)i * nv + k] * d->efc_force[i]; d->qfrc_constraint[k] = s; }
    free(jar); free(jar2); free(force); free(hd); free(jv); free(g); free(dl); free(Md); free(H); free(tmp); free(zone);
}
