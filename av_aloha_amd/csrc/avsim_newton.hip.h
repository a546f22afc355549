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
// Mapping on the wavefront
//   rows      row i -> lane i % 64 (residuals J_i a - aref_i, J_i dl; gradient scatter with returnless LDS atomics)
//   contacts  contact c -> lane c % 64: zone, cone forces and the line-search derivatives of its <= 6 rows in registers
//   Hessian   H = M + sum_blocks J_b^T C_b J_b in a packed lower triangle in LDS.  The wave walks the blocks (a scalar
//             row or a contact) one at a time; the cone block is never formed: in the middle zone
//                 C = S (Dm n n^T + kappa (I_t - u u^T)) S      n = (1, -mu u),  u = U_t / |U_t|
//             so J^T C J = sum_p w_p J_p^T J_p + Dm y1 y1^T - kappa y2 y2^T with two 16-vectors y1, y2; four contacts per pass
//             (one per 16-lane group, lane t = column t of the 16 x 16 dof window), returnless LDS atomics; scalar rows
//             one per lane.
//   Cholesky  lane i holds row i of H in registers; the column loop is rolled, the register row is rotated by one
//             entry per column so that the pivot column is always element 0 (static register indices, no scratch);
//             multipliers travel by v_readlane.  Lane nv carries -g as an extra row, so the forward substitution
//             comes out of the factorisation; the backward substitution reads L's columns back from LDS.  When no row
//             couples two kinematic trees H is block diagonal: lane 8 t + i then factors / solves tree t's block inside
//             its octet (nblock_chol / nblock_solve), eight steps instead of nv columns.  When rows do couple trees (a needle in a
//             gripper) the dense loops take the dofs of the coupled trees only, one per lane, and the other trees keep their octets
//             (ncomponent / ndense_chol / nblock_chol_fwd / ncomp_subst; the solver instance for such scenes is a function of its
//             own, newton_solve_coupled, so that Env::solve carries only the uncoupled one).
//   search    exact line search: safeguarded 1-D Newton on phi'(alpha); J a and J dl are fixed per iteration, so an
//             evaluation is a handful of FMAs per lane and two wave-wide DPP sums.
#pragma once
#include "avsim_math.hip.h"

namespace avs {

// A whole 16-entry row of the global row scratch in four (eight) vector loads issued back to back, then pinned by one empty asm:
// without it the compiler sinks each element's load into the branch that consumes it (load - wait - use, sixteen times)
typedef float nv4f __attribute__((ext_vector_type(4)));
typedef double nv2d __attribute__((ext_vector_type(2)));
AVS_DEV void load_row16(GLB_PTR(const float) src, float* v) {
    GLB_PTR(const nv4f) p = (GLB_PTR(const nv4f))src;
    nv4f a = p[0], b = p[1], c = p[2], d = p[3];
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w; v[12] = d.x; v[13] = d.y; v[14] = d.z; v[15] = d.w;
}
AVS_DEV void load_row16(GLB_PTR(const double) src, double* v) {
    GLB_PTR(const nv2d) p = (GLB_PTR(const nv2d))src;
    nv2d t[8];
#pragma unroll
    for (int q = 0; q < 8; q++) t[q] = p[q];
    asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
#pragma unroll
    for (int q = 0; q < 8; q++) { v[2 * q] = t[q].x; v[2 * q + 1] = t[q].y; }
}

template <typename real>
struct NewtonArgs {
    // LDS views of one env
    LDS_PTR(real) rowS;          // 8 reals per row: aref, R, [Newton: J a - aref], 1/diag|0, lo, hi, force, 1/friction
    LDS_PTR(const int) rowI;     // dof windows of the row: (adr 6 | n 4 | tree 3) x 2
    LDS_PTR(const int) rmeta;    // type 2 | id 10 | sub 8 | tree ids
    GLB_PTR(const real) rJ;      // 16 reals per row, global memory (L2-resident scratch of this env)
    LDS_PTR(const real) M;       // per-tree dense blocks
    LDS_PTR(real) a;             // qacc (in: start point, out: solution)
    LDS_PTR(const real) as;      // qacc_smooth
    LDS_PTR(real) H;             // packed lower triangle nv(nv+1)/2
    LDS_PTR(real) g;             // gradient nv; MUST be H + nv(nv+1)/2 (the factorisation reads it as row nv)
    LDS_PTR(real) dl;            // search direction nv
    LDS_PTR(real) jv;            // per row: J dl (line search) / curvature of the scalar rows (Hessian phase)
    LDS_PTR(const int) cefc;     // contact c: first row | rows << 16, or -1
    LDS_PTR(const int) tree_dofadr;
    LDS_PTR(const int) tree_dofnum;
    LDS_PTR(const int) tree_madr;
    LDS_PTR(const int) dof_tree;
    LDS_PTR(int) prof;           // optional cycle counters (8 ints) or null
    int nv, nefc, ncon, nlead, ntree, iters;
    real tol, scale, ls_tol;
    int ls_iters;                // option "ls_iterations": evaluations of phi' after the one at alpha = 0
    int early_exit;              // option "newton_early_exit": leave without the confirming gradient after an exact step of a quadratic piece
    // this lane's dof (lane < nv <= 64): first dof and size of its tree, offset of its row in the tree's block of M
    int k_a0, k_n, k_mb;
};

enum { NR_EQ = 0, NR_FLOSS = 1, NR_LIMIT = 2, NR_CONTACT = 3 };
constexpr int NVMAX = 48;    // register row of the factorisation

// J_i . v for row i (v in LDS, dof indexed)
template <typename real>
AVS_DEV real nrow_dot(const NewtonArgs<real>& A, int i, LDS_PTR(const real) v) {
    const int ra = A.rowI[i];
    const int a0 = ra & 63, nA = (ra >> 6) & 15, b0 = (ra >> 13) & 63, nB = (ra >> 19) & 15;
    GLB_PTR(const real) J = A.rJ + ROW_S * i;
    real s = 0;
#pragma unroll
    for (int k = 0; k < TREE_W; k++) {
        const real ja = J[k], va = v[k < nA ? a0 + k : 0], jb = J[TREE_W + k], vb = v[k < nB ? b0 + k : 0];
        s += (k < nA ? ja * va : real(0)) + (k < nB ? jb * vb : real(0));
    }
    return s;
}

// the same for a leading row (see lead_d1 below): two products, no global memory
template <typename real>
AVS_DEV real nrow_dot_lead(const NewtonArgs<real>& A, int i, LDS_PTR(const real) v) {
    const int ra = A.rowI[i];
    const int d1 = (ra & 63) + ((ra >> 26) & 7), d2 = (ra & 63) + ((ra >> 29) & 7);
    const real v1 = ((ra >> 13) & 1) ? real(-1) : real(1), v2 = A.rowS[RS_S * i + 7];
    return v1 * v[d1] + v2 * v[d2];
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

// the alpha-independent part of one contact (held by its lane for the whole solve)
template <typename real>
struct NCon {
    int head, dim;        // first row (-1: none), rows
    real S[6];            // S[0] = mu (cone scaling of the normal), S[j] = friction of row j
    real D[6];            // 1 / R_j
    real mu, Dm;          // middle-zone curvature scale
};

// zone (0 top / 1 bottom / 2 middle), forces, cost, and the Hessian coefficients of the block:
//   J^T C J = sum_p w_p J_p^T J_p + s1 y1 y1^T - s2 y2 y2^T,  y1 = sum_p c1_p J_p,  y2 = sum_p c2_p J_p
template <typename real>
AVS_DEV int ncone(const NCon<real>& c, const real* jar, real* f, real* cost, real* w, real* c1, real* c2, real* s1, real* s2) {
#pragma unroll
    for (int j = 0; j < 6; j++) { f[j] = 0; w[j] = 0; c1[j] = 0; c2[j] = 0; }
    *cost = 0; *s1 = 0; *s2 = 0;
    if (c.dim == 1) {
        if (!(jar[0] < 0)) return 0;
        f[0] = -c.D[0] * jar[0]; w[0] = c.D[0];
        *cost = real(0.5) * c.D[0] * jar[0] * jar[0];
        return 1;
    }
    real U[6], t2 = 0;
    U[0] = jar[0] * c.mu;
#pragma unroll
    for (int j = 1; j < 6; j++) { U[j] = j < c.dim ? jar[j] * c.S[j] : real(0); t2 += U[j] * U[j]; }
    const real N = U[0], T = sqrt(t2), mu = c.mu;
    if (N >= mu * T || (T <= 0 && N >= 0)) return 0;
    if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
#pragma unroll
        for (int j = 0; j < 6; j++)
            if (j < c.dim) { f[j] = -jar[j] * c.D[j]; w[j] = c.D[j]; *cost += real(0.5) * c.D[j] * jar[j] * jar[j]; }
        return 1;
    }
    const real NT = N - mu * T, Ti = real(1) / T, kap = -c.Dm * NT * mu * Ti;
    f[0] = -c.Dm * NT * mu;
    *cost = real(0.5) * c.Dm * NT * NT;
    c1[0] = mu; *s1 = c.Dm; *s2 = kap;
#pragma unroll
    for (int j = 1; j < 6; j++)
        if (j < c.dim) {
            const real us = U[j] * Ti * c.S[j];     // u_j S_j
            f[j] = -f[0] * us;
            w[j] = kap * c.S[j] * c.S[j];
            c1[j] = -mu * us;
            c2[j] = us;
        }
    return 2;
}

// phi'(alpha), phi''(alpha) of one contact at residuals jar (= jar0 + alpha jv)
template <typename real>
AVS_DEV void ncone_ls(const NCon<real>& c, const real* jar, const real* jv, real* d1, real* d2) {
    *d1 = 0; *d2 = 0;
    if (c.dim == 1) {
        if (jar[0] < 0) { *d1 = c.D[0] * jar[0] * jv[0]; *d2 = c.D[0] * jv[0] * jv[0]; }
        return;
    }
    real U[6], V[6], t2 = 0, uv = 0, v2 = 0;
    U[0] = jar[0] * c.mu; V[0] = jv[0] * c.mu;
#pragma unroll
    for (int j = 1; j < 6; j++) {
        U[j] = j < c.dim ? jar[j] * c.S[j] : real(0);
        V[j] = j < c.dim ? jv[j] * c.S[j] : real(0);
        t2 += U[j] * U[j]; uv += U[j] * V[j]; v2 += V[j] * V[j];
    }
    const real N = U[0], T = sqrt(t2), mu = c.mu;
    if (N >= mu * T || (T <= 0 && N >= 0)) return;
    if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
#pragma unroll
        for (int j = 0; j < 6; j++)
            if (j < c.dim) { *d1 += c.D[j] * jar[j] * jv[j]; *d2 += c.D[j] * jv[j] * jv[j]; }
        return;
    }
    const real Ti = real(1) / T, T1 = uv * Ti, T2 = (v2 - T1 * T1) * Ti, NT = N - mu * T, N1 = V[0] - mu * T1;
    *d1 = c.Dm * NT * N1;
    *d2 = c.Dm * N1 * N1 - c.Dm * NT * mu * T2;
}

// global dof of window slot s (0..15) of a row, or -1
AVS_DEV int nslot_dof(int ra, int s) {
    const int sh = s < TREE_W ? 0 : 13, k = s & (TREE_W - 1);
    return k < ((ra >> (sh + 6)) & 15) ? ((ra >> sh) & 63) + k : -1;
}

// A leading (non-contact) row has at most two non-zero entries, both in its first dof window: the row word carries their slots
// (bits 26-28, 29-31; equal when there is one entry) and the sign of the first (bit 13; the value is +-1), word 7 of the row record
// the second value (0: none).  Nothing that walks these rows needs the 16-word record in global memory.
AVS_DEV int lead_d1(int ra) { return (ra & 63) + ((ra >> 26) & 7); }
AVS_DEV int lead_d2(int ra) { return (ra & 63) + ((ra >> 29) & 7); }
template <typename real> AVS_DEV real lead_v1(int ra) { return ((ra >> 13) & 1) ? real(-1) : real(1); }

// out[dof] += sgn * sum_rows J[row][dof] f[row] (f = word 6 of the row records).  Leading rows: one row per lane, one or two LDS
// atomics.  Contacts: four at a time, 16-lane group q takes a contact and lane t of the group column t of its 16-slot dof window,
// sums the contact's <= 6 rows in registers (coalesced 64-byte row reads) and adds once: an atomic instruction then carries 16
// distinct addresses per contact instead of one address per tree from every row (the rows of a tree all share their slots).
template <typename real>
AVS_DEV void rows_jt_force(LDS_PTR(const real) rowS, LDS_PTR(const int) rowI, LDS_PTR(const int) cefc, GLB_PTR(const real) rJ, LDS_PTR(real) out,
                           int nlead, int ncon, int lane, real sgn) {
    for (int i = lane; i < nlead; i += 64) {
        const real f = rowS[RS_S * i + 6];
        if (f == 0) continue;
        const int ra = rowI[i];
        const real v2 = rowS[RS_S * i + 7];
        __hip_atomic_fetch_add(out + lead_d1(ra), sgn * lead_v1<real>(ra) * f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v2 != 0) __hip_atomic_fetch_add(out + lead_d2(ra), sgn * v2 * f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    const int t = lane & 15;
    for (int c0 = 0; c0 < ncon; c0 += 4) {
        const int c = c0 + (lane >> 4);
        const int ce = c < ncon ? cefc[c] : -1;
        const bool on = ce >= 0;
        const int head = on ? (ce & 0xffff) : 0, dim = on ? (ce >> 16) : 0;
        const int ra = rowI[head];
        real acc = 0;
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const int row = head + (p < dim ? p : 0);          // (a row of this contact in any case: always inside the env's rows)
            const real Jp = rJ[ROW_S * row + t], fp = rowS[RS_S * row + 6];
            acc += p < dim ? Jp * fp : real(0);
        }
        const int dof = on ? nslot_dof(ra, t) : -1;
        if (dof >= 0 && acc != 0) __hip_atomic_fetch_add(out + dof, sgn * acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// H += J_b^T C_b J_b for the block of rows r0 .. r0+dim-1 of a contact: C_b = diag(w) + s1 c1 c1^T - s2 c2 c2^T in the cone's
// middle zone (`full`), diag(w) otherwise.  Lane t of a 16-lane group owns column t of the block's 16 x 16 dof window (two
// tree windows of 8) and walks the rows s; lower-triangle entries only, added to the packed H by LDS atomics.
// Four contacts at once: 16-lane group q of the wave takes the contact whose arguments its lanes carry (group-uniform
// values); the groups' row loads are in flight together, one memory round trip per four contacts.
// column t of the block's rows: issued one pass ahead of nblock4 (rows past the block re-read its first row and are masked at the use)
template <typename real>
AVS_DEV void nblock4_load(const NewtonArgs<real>& A, int lane, int r0, int dim, bool on, real* Jraw) {
    const int t = lane & 15;
    GLB_PTR(const real) J = A.rJ + ROW_S * (on ? r0 : 0);
#pragma unroll
    for (int p = 0; p < 6; p++) Jraw[p] = J[ROW_S * (p < dim ? p : 0) + t];
}
template <typename real, bool BATCH>
AVS_DEV void nblock4(const NewtonArgs<real>& A, int lane, int r0, int dim, bool on, bool full, const real* w, int c, const real* cc1, const real* cc2, real cs1, real cs2,
                     const real* Jraw) {
    const int ra = A.rowI[on ? r0 : 0], t = lane & 15, gq = on ? nslot_dof(ra, t) : -1;
    real Jt[6];
#pragma unroll
    for (int p = 0; p < 6; p++) Jt[p] = (on && p < dim) ? Jraw[p] : real(0);
    const bool two = __any(on && ((ra >> 19) & 15) > 0);               // does some contact of this pass have a second window?
    const int smax = two ? 16 : 8;
    if (!__any(full)) {
        // top / bottom zone only: C = diag(w)
        if (!two) {
            // one-tree contacts: the block is 8 x 8, so the two halves of the 16-lane group share the rows (lane t and lane
            // t + 8 both hold column t & 7; the upper half takes rows 4..7)
            // The entries of the other columns come straight from the rows in memory (the look-ahead read has just brought the
            // lines in): 30 four-byte reads in flight together instead of 30 ds_bpermute round trips of 50 - 75 cycles each.
            const int tc = lane & 7, s0 = (lane & 8) >> 1, gqc = on ? nslot_dof(ra, tc) : -1;
            GLB_PTR(const real) J = A.rJ + ROW_S * (on ? r0 : 0);
            real Jc[6], Js[4][6];
#pragma unroll
            for (int p = 0; p < 6; p++) {
                GLB_PTR(const real) Jp = J + ROW_S * (p < dim ? p : 0);
                Jc[p] = Jp[tc];
#pragma unroll
                for (int u = 0; u < 4; u++) Js[u][p] = Jp[s0 + u];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int s = s0 + u, gp = nslot_dof(ra, s);
                real acc = 0;
#pragma unroll
                for (int p = 0; p < 6; p++) acc += (on && p < dim) ? w[p] * Js[u][p] * Jc[p] : real(0);
                if (on && gp >= 0 && gqc >= 0 && gqc <= gp) __hip_atomic_fetch_add(A.H + gp * (gp + 1) / 2 + gqc, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            return;
        }
        if constexpr (BATCH) {
            // (the coupled instance: eight rows' entries fetched from the other lanes at once instead of one row per round trip)
            for (int s0 = 0; s0 < smax; s0 += 8) {
                real js[8][6];
#pragma unroll
                for (int u = 0; u < 8; u++)
#pragma unroll
                    for (int p = 0; p < 6; p++) js[u][p] = __shfl(Jt[p], (lane & 48) | (s0 + u), 64);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int gp = nslot_dof(ra, s0 + u);
                    real acc = 0;
#pragma unroll
                    for (int p = 0; p < 6; p++) acc += w[p] * js[u][p] * Jt[p];
                    if (on && gp >= 0 && gq >= 0 && gq <= gp) __hip_atomic_fetch_add(A.H + gp * (gp + 1) / 2 + gq, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            return;
        }
        for (int s = 0; s < smax; s++) {
            const int gp = nslot_dof(ra, s);
            real acc = 0;
#pragma unroll
            for (int p = 0; p < 6; p++) acc += w[p] * __shfl(Jt[p], (lane & 48) | s, 64) * Jt[p];
            if (on && gp >= 0 && gq >= 0 && gq <= gp) __hip_atomic_fetch_add(A.H + gp * (gp + 1) / 2 + gq, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    // some contact of this pass is in the middle zone: its rank-two terms come from the owner lane's registers (c1, c2, s1, s2
    // of contact c live in lane c), fetched only here so that they are not live in the common path
    real c1[6], c2[6], y1t = 0, y2t = 0;
#pragma unroll
    for (int p = 0; p < 6; p++) { c1[p] = __shfl(cc1[p], c & 63, 64); c2[p] = __shfl(cc2[p], c & 63, 64); }
    const real s1 = __shfl(cs1, c & 63, 64), s2 = __shfl(cs2, c & 63, 64);
#pragma unroll
    for (int p = 0; p < 6; p++) { y1t += c1[p] * Jt[p]; y2t += c2[p] * Jt[p]; }
    if constexpr (BATCH) {
        for (int s0 = 0; s0 < smax; s0 += 8) {
            real jsb[8][6];
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int p = 0; p < 6; p++) jsb[u][p] = __shfl(Jt[p], (lane & 48) | (s0 + u), 64);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int gp = nslot_dof(ra, s0 + u);
                real acc = 0, y1s = 0, y2s = 0;
#pragma unroll
                for (int p = 0; p < 6; p++) {
                    const real js = jsb[u][p];
                    acc += w[p] * js * Jt[p];
                    y1s += c1[p] * js;
                    y2s += c2[p] * js;
                }
                if (full) acc += s1 * y1s * y1t - s2 * y2s * y2t;
                if (on && gp >= 0 && gq >= 0 && gq <= gp) __hip_atomic_fetch_add(A.H + gp * (gp + 1) / 2 + gq, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        return;
    }
    for (int s = 0; s < smax; s++) {
        const int gp = nslot_dof(ra, s);
        real acc = 0, y1s = 0, y2s = 0;
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const real js = __shfl(Jt[p], (lane & 48) | s, 64);
            acc += w[p] * js * Jt[p];
            y1s += c1[p] * js;
            y2s += c2[p] * js;
        }
        if (full) acc += s1 * y1s * y1t - s2 * y2s * y2t;
        if (on && gp >= 0 && gq >= 0 && gq <= gp) __hip_atomic_fetch_add(A.H + gp * (gp + 1) / 2 + gq, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
// scalar rows (equality, dry friction, limits): one row per lane, H += w J^T J over the row's single dof window
template <typename real>
AVS_DEV void nlead_rows(const NewtonArgs<real>& A, int lane) {
    for (int i = lane; i < A.nlead; i += 64) {
        const real w = A.jv[i];
        if (w == 0) continue;
        const int ra = A.rowI[i], d1 = lead_d1(ra), d2 = lead_d2(ra);
        const real v1 = lead_v1<real>(ra), v2 = A.rowS[RS_S * i + 7];
        __hip_atomic_fetch_add(A.H + d1 * (d1 + 1) / 2 + d1, w * v1 * v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v2 != 0) {
            const int hi = d1 > d2 ? d1 : d2, lo = d1 > d2 ? d2 : d1;
            const real vh = d1 > d2 ? v1 : v2, vl = d1 > d2 ? v2 : v1;
            __hip_atomic_fetch_add(A.H + d2 * (d2 + 1) / 2 + d2, w * v2 * v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(A.H + hi * (hi + 1) / 2 + lo, w * vh * vl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

#define NSYNC()                                              \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, AVS_SYNC_SCOPE); \
        __builtin_amdgcn_wave_barrier();                     \
    } while (0)
#define NPROF(k)                                                                           \
    do {                                                                                   \
        if (A.prof) { const long long t_ = __builtin_readcyclecounter(); if (lane == 0) A.prof[k] += (int)(t_ - tp0); tp0 = t_; } \
    } while (0)

// cost of the point v (LDS, dof indexed): rows + contacts + 1/2 (v - a_s)^T M (v - a_s).  The row residuals J v - aref are read
// from word `slot` of the row records: make_constraints leaves them there for the two start candidates (2: warm start, 8: a_s)
template <typename real, int NCH>
AVS_DEV real ncost(const NewtonArgs<real>& A, int lane, const NCon<real>* con, LDS_PTR(const real) v, int slot, bool inertial = true) {
    real cs = 0;
    for (int i = lane; i < A.nlead; i += 64) cs += nrow_scalar_cost<real>(A.rmeta[i] & 3, A.rowS[RS_S * i + slot], A.rowS[RS_S * i + 1], A.rowS[RS_S * i + 5]);
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        if (ch * 64 >= A.ncon) break;
        const NCon<real>& c = con[ch];
        if (c.head >= 0) {
            real jar[6], f[6], w[6], c1[6], c2[6], cc, s1, s2;
#pragma unroll
            for (int j = 0; j < 6; j++) jar[j] = j < c.dim ? A.rowS[RS_S * (c.head + j) + slot] : real(0);
            ncone(c, jar, f, &cc, w, c1, c2, &s1, &s2);
            cs += cc;
        }
    }
    if (inertial)        // (v = a_s: the Gauss term is exactly zero)
    for (int k = lane; k < A.nv; k += 64) {
        const int a0 = A.k_a0, n = A.k_n, mb = A.k_mb;
        real sacc = 0;
#pragma unroll
        for (int j = 0; j < TREE_W; j++) { const int jj = j < n ? j : 0; const real m = A.M[mb + jj], d = v[a0 + jj] - A.as[a0 + jj]; sacc += j < n ? m * d : real(0); }
        cs += real(0.5) * sacc * (v[k] - A.as[k]);
    }
    return wave_sum(cs);
}

// ---- block-diagonal Hessian: no active row couples two kinematic trees, so H = diag(H_t) and every tree factors on its own.
// Lane 8 t + i owns row i of tree t (<= 8 trees of <= 8 dofs): eight elimination steps with the multipliers shuffled inside the
// octet instead of nv serial columns over the whole wave.  L is written back into the packed triangle (the entries a dense
// factorisation would produce; off-block entries stay zero), so the stored factor serves the later iterations as well.
template <typename real>
AVS_DEV void nblock_chol(const NewtonArgs<real>& A, int lane) {
    const int t = lane >> 3, i = lane & 7, gb = lane & ~7;
    const bool act = t < A.ntree;
    const int n = act ? A.tree_dofnum[t] : 0, a0 = act ? A.tree_dofadr[t] : 0;
    const int rbase = (a0 + i) * (a0 + i + 1) / 2 + a0;
    real row[TREE_W];
#pragma unroll
    for (int k = 0; k < TREE_W; k++) {
        const bool in = i < n && k <= i;
        const real v = A.H[in ? rbase + k : 0];
        row[k] = in ? v : (k == i ? real(1) : real(0));
    }
#pragma unroll
    for (int j = 0; j < TREE_W; j++) {
        const real piv = tmax(oct_bcast_n(row[j], j), real(1e-30));
        real d, rinv;
        if (sizeof(real) == 4) { rinv = (real)__builtin_amdgcn_rsqf((float)piv); d = piv * rinv; }
        else { d = sqrt(piv); rinv = real(1) / d; }
        const real lij = i == j ? d : row[j] * rinv;
        row[j] = lij;
        const real mul = i > j ? lij : real(0);
#pragma unroll
        for (int k = j + 1; k < TREE_W; k++) row[k] -= mul * oct_bcast_n(lij, k);
    }
    if (i < n) {
#pragma unroll
        for (int k = 0; k < TREE_W; k++)
            if (k <= i) A.H[rbase + k] = row[k];
    }
}
// The trees that no row couples to another one, in a scene where some rows do (`tmask`: the trees of the coupled component): their
// blocks of H factor as above, inside their octets, while the dense factorisation below takes the coupled dofs only.  The forward
// substitution comes with it in the dense path's arithmetic -- there lane nv carries -g through the columns as one more row:
// y_j = g'_j rsq(pivot_j), g'_k -= y_j L_kj -- so L and y are entry for entry what ndense_chol over all nv columns leaves in H and g.
template <typename real>
AVS_DEV void nblock_chol_fwd(LDS_PTR(real) H, LDS_PTR(real) g, int a0_, int n_, bool act, int lane) {
    // (a0_, n_: first dof and size of the tree of this lane's octet; act: the octet has a tree and it lies outside the component)
    struct { LDS_PTR(real) H; LDS_PTR(real) g; } A = {H, g};
    const int i = lane & 7;
    const int n = act ? n_ : 0, a0 = act ? a0_ : 0;
    const int rbase = (a0 + i) * (a0 + i + 1) / 2 + a0;
    real row[TREE_W];
#pragma unroll
    for (int k = 0; k < TREE_W; k++) {
        const bool in = i < n && k <= i;
        const real v = A.H[in ? rbase + k : 0];
        row[k] = in ? v : (k == i ? real(1) : real(0));
    }
    const real gi = A.g[i < n ? a0 + i : 0];
    real x = i < n ? -gi : real(0), y = 0;
#pragma unroll
    for (int j = 0; j < TREE_W; j++) {
        const real piv = tmax(oct_bcast_n(row[j], j), real(1e-30));
        real d, rinv;
        if (sizeof(real) == 4) { rinv = (real)__builtin_amdgcn_rsqf((float)piv); d = piv * rinv; }
        else { d = sqrt(piv); rinv = real(1) / d; }
        const real lij = i == j ? d : row[j] * rinv;
        row[j] = lij;
        const real mul = i > j ? lij : real(0);
#pragma unroll
        for (int k = j + 1; k < TREE_W; k++) row[k] -= mul * oct_bcast_n(lij, k);
        const real yj = oct_bcast_n(x, j) * rinv;
        y = i == j ? yj : y;
        x = x - yj * mul;
    }
    if (i < n) {
#pragma unroll
        for (int k = 0; k < TREE_W; k++)
            if (k <= i) A.H[rbase + k] = row[k];
        A.g[a0 + i] = y;
    }
}
// dl = (L L^T)^-1 (-g) with the block factor stored in the packed triangle: both substitutions inside the octets
template <typename real>
AVS_DEV void nblock_solve(const NewtonArgs<real>& A, int lane) {
    const int t = lane >> 3, i = lane & 7, gb = lane & ~7;
    const bool act = t < A.ntree;
    const int n = act ? A.tree_dofnum[t] : 0, a0 = act ? A.tree_dofadr[t] : 0;
    const bool mine = i < n;
    const int rbase = (a0 + i) * (a0 + i + 1) / 2 + a0;
    real row[TREE_W], col[TREE_W];
#pragma unroll
    for (int k = 0; k < TREE_W; k++) {
        const bool lo = mine && k < i, up = mine && k < n && k > i;
        const real a = A.H[lo ? rbase + k : 0], b = A.H[up ? (a0 + k) * (a0 + k + 1) / 2 + a0 + i : 0];
        row[k] = lo ? a : real(0);
        col[k] = up ? b : real(0);
    }
    const real dg = A.H[mine ? rbase + i : 0];
    const real dinv = mine ? real(1) / dg : real(1);
    real x = mine ? -A.g[a0 + i] : real(0);
#pragma unroll
    for (int j = 0; j < TREE_W; j++) {
        const real yj = oct_bcast_n(x * dinv, j);
        x = i == j ? yj : x - row[j] * yj;
    }
#pragma unroll
    for (int j = TREE_W - 1; j >= 0; j--) {
        const real xj = oct_bcast_n(x * dinv, j);
        x = i == j ? xj : x - col[j] * xj;
    }
    if (mine) A.dl[a0 + i] = x;
}

// Dense factorisation for scenes where a contact couples two trees.  Out of line on purpose: its 48-entry register row would
// otherwise push the Newton loop's per-contact state out to scratch memory in every solve, coupled or not.
// It runs over the `nc` dofs of the coupled component only (lane p < nc holds the row of dof `mydof`, ascending; lane nc the row
// "nv" = -g; the trees outside the component go through nblock_chol_fwd): a column is a serial step of ~570 cycles whatever it
// holds, and the entries between the component and the other trees are zeros that stay zeros -- the same L and y, entry for entry,
// as the factorisation over all nv columns.
template <typename real>
__device__ AVS_OUTLINE_4 void ndense_chol(LDS_PTR(real) H_, int nv_, int mydof, int nc_, int oa0, int on, bool oact) {
    const int lane = threadIdx.x & 63;
    LDS_PTR(real) H = uni_lds(H_);
    const int nv = __builtin_amdgcn_readfirstlane(nv_), nc = __builtin_amdgcn_readfirstlane(nc_);
    if (nc < nv) nblock_chol_fwd<real>(H, H + nv * (nv + 1) / 2, oa0, on, oact, lane);
    // g sits right behind the packed triangle (NewtonArgs contract), i.e. it is "row nv" of the same array
    real row[NVMAX];
    const int rbase = mydof * (mydof + 1) / 2;
    const real sgn = lane == nc ? real(-1) : real(1);
#pragma unroll
    for (int k = 0; k < NVMAX; k++) {
        const bool ok = lane <= nc && k <= lane && k < nc;
        const int ck = __builtin_amdgcn_readlane(mydof, k);
        const real v = H[ok ? rbase + ck : 0];
        row[k] = ok ? sgn * v : real(0);
    }
    for (int j = 0; j < nc; j++) {
        // pivot: d = sqrt(H_jj), column j = row[0] / d.  f32 takes the hardware rsq (1 ulp), f64 the exact pair
        const real piv = tmax(lane_get(row[0], j), real(1e-30));
        real d, rinv;
        if (sizeof(real) == 4) { rinv = (real)__builtin_amdgcn_rsqf((float)piv); d = piv * rinv; }
        else { d = sqrt(piv); rinv = real(1) / d; }
        const real lij = row[0] * rinv;
        // L_ij for the rows below, d on the diagonal, y_j = (L^-1 (-g))_j from lane nc ("row nv" is g's storage)
        const int cj = __builtin_amdgcn_readlane(mydof, j);
        if (lane >= j && lane <= nc) H[rbase + cj] = lane == j ? d : lij;
#pragma unroll
        for (int mb = 1; mb < NVMAX; mb += 8) {
            if (j + mb <= nc - 1) {          // wave-uniform: the rest of the row is past the matrix
#pragma unroll
                for (int mm = 0; mm < 8; mm++) {
                    const int m = mb + mm;
                    if (m < NVMAX) row[m - 1] = row[m] - lij * lane_get(lij, (j + m) & 63);
                }
            }
        }
    }
}

// The coupled component of a solve: the trees that some row couples to another tree (by the rows present, active or not), its
// dofs in ascending order one per lane (lane p < nc: dof `mydof`; nv on the other lanes); everything when there are more than eight
// trees.  The trees outside it keep their octets: lane 8 t + i = dof i of tree t (oa0, on; oact: the octet has such a tree).
struct NComp { int mydof, nc, oa0, on; bool oact; };
template <typename real>
AVS_DEV NComp ncomponent(const NewtonArgs<real>& A, int lane, int ne, LDS_PTR(int) tmp, bool whole) {
    const int nv = A.nv;
    NComp c;
    c.mydof = lane < nv ? lane : nv; c.nc = nv; c.oa0 = 0; c.on = 0; c.oact = false;
    if (A.ntree > 8 || whole) return c;
    unsigned mine = 0;
    for (int i = lane; i < ne; i += 64) {
        const int ra = A.rowI[i];
        if (((ra >> 19) & 15) != 0) mine |= (1u << ((ra >> 10) & 7)) | (1u << ((ra >> 23) & 7));
    }
    unsigned tmask = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) tmask |= __any((mine >> t) & 1u) ? (1u << t) : 0u;
    const bool isc = lane < nv && ((tmask >> A.dof_tree[lane < nv ? lane : 0]) & 1u);
    const unsigned long long cm = __ballot(isc);
    c.nc = __popcll(cm);
    if (c.nc == nv) return c;
    if (isc) tmp[__popcll(cm & ((1ull << lane) - 1ull))] = lane;
    NSYNC();
    c.mydof = lane < c.nc ? tmp[lane] : nv;
    NSYNC();
    const int t = lane >> 3;
    c.oact = t < A.ntree && !((tmask >> t) & 1u);
    c.oa0 = c.oact ? A.tree_dofadr[t] : 0;
    c.on = c.oact ? A.tree_dofnum[t] : 0;
    return c;
}

// Substitutions with the stored factor in a coupled scene, the same split: the trees outside the component inside their octets
// (nblock_solve's steps), the component's dofs one per lane over its nc columns (the dense loops' steps) -- the products with the
// zeros between the two are left out, every other operation is the one the loop over all nv columns did.
//   fwd: L y = -g (y -> g), for an iteration that keeps the last factor;  bwd: L^T x = y (g -> dl)
template <typename real, bool FWD>
AVS_DEV void ncomp_subst(const NewtonArgs<real>& A, const NComp& c, int lane) {
    const int nv = A.nv, nc = c.nc, i8 = lane & 7;
    // ---- octets ----
    const bool omine = c.oact && i8 < c.on;
    const int a0 = c.oa0, obase = (a0 + i8) * (a0 + i8 + 1) / 2 + a0;
    real ocf[TREE_W];
#pragma unroll
    for (int k = 0; k < TREE_W; k++) {
        const bool in = FWD ? (omine && k < i8) : (omine && k < c.on && k > i8);
        const real v = A.H[in ? (FWD ? obase + k : (a0 + k) * (a0 + k + 1) / 2 + a0 + i8) : 0];
        ocf[k] = in ? v : real(0);
    }
    const real odg = A.H[omine ? obase + i8 : 0];
    const real odinv = omine ? real(1) / odg : real(1);
    const real og = A.g[omine ? a0 + i8 : 0];
    real ox = omine ? (FWD ? -og : og) : real(0);
    // ---- the component ----
    const int md = c.mydof, rb = md * (md + 1) / 2;
    const bool cm = lane < nc;
    const real cg = A.g[cm ? md : 0];
    real x = cm ? (FWD ? -cg : cg) : real(0);
    const real dinv = cm ? real(1) / A.H[rb + md] : real(0);
    if (nc < nv) {
#pragma unroll
        for (int jj = 0; jj < TREE_W; jj++) {
            const int j = FWD ? jj : TREE_W - 1 - jj;
            const real xj = oct_bcast_n(ox * odinv, j);
            ox = i8 == j ? xj : ox - ocf[j] * xj;
        }
    }
    if (FWD) {
        int cn = __builtin_amdgcn_readlane(md, 0);
        real Lnext = (lane > 0 && cm) ? A.H[rb + cn] : real(0);
        for (int j = 0; j < nc; j++) {
            const real Lc = Lnext;
            if (j + 1 < nc) { cn = __builtin_amdgcn_readlane(md, j + 1); Lnext = (lane > j + 1 && cm) ? A.H[rb + cn] : real(0); }
            const real yj = lane_get(x * dinv, j);
            if (lane == j) x = yj;
            else if (lane > j) x -= Lc * yj;
        }
    } else {
        int cn = __builtin_amdgcn_readlane(md, nc - 1);
        real Lnext = lane < nc - 1 ? A.H[cn * (cn + 1) / 2 + md] : real(0);
        for (int i = nc - 1; i >= 0; i--) {
            const real Lc = Lnext;
            if (i > 0) { cn = __builtin_amdgcn_readlane(md, i - 1); Lnext = lane < i - 1 ? A.H[cn * (cn + 1) / 2 + md] : real(0); }
            const real xi = lane_get(x * dinv, i);
            if (lane == i) x = xi;
            else if (lane < i) x -= Lc * xi;
        }
    }
    NSYNC();
    LDS_PTR(real) out = FWD ? A.g : A.dl;
    if (cm) out[md] = x;
    if (nc < nv && omine) out[a0 + i8] = ox;
}

// r / ii: the env's real and int LDS regions, li: the block's hot-table image; everything else comes from the layout
// NCH = contact chunks of 64 (one contact per lane and chunk): 1 when the model's contact cap is <= 64
// COUPLED: some row reaches into two kinematic trees (the caller looks: solve_i) -- two instances, so that the solves of scenes
// without such rows carry nothing of the dense path (its call alone costs the Newton loop SGPRs: 1 % of the headline configuration)
template <typename real, int NCH, bool COUPLED>
__device__ __attribute__((always_inline)) int newton_solve(KPtr<real> ka, GLB_PTR(const real) rows, LDS_PTR(real) r_, LDS_PTR(int) ii_, LDS_PTR(const int) li_, int nefc, int ncon, int nlead,
                                                      int iters, real tol, real scale, int profiling) {
    const int lane = threadIdx.x & 63;
    NewtonArgs<real> A;
    {
        // arguments of a non-kernel function arrive in VGPRs: make the wave-uniform ones scalar again
        LDS_PTR(real) r = (LDS_PTR(real))(unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)r_);
        LDS_PTR(int) ii = (LDS_PTR(int))(unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)ii_);
        LDS_PTR(const int) li = (LDS_PTR(const int))(unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)li_);
        tol = lane_get(tol, 0); scale = lane_get(scale, 0);
        const Layout __attribute__((address_space(4)))* L = &ka->lay;
        const MOff __attribute__((address_space(4)))* O = &ka->mo;
        A.rowS = r + L->rowS; A.rowI = ii + L->rowI; A.rmeta = ii + L->rmeta;
        {   // the row pointer is wave-uniform: back to SGPRs
            const unsigned long long p_ = (unsigned long long)rows;
            A.rJ = (GLB_PTR(const real))(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p_ >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p_));
        }
        A.M = r + L->M; A.a = r + L->qacc; A.as = r + L->asm_;
        A.H = r + L->nH; A.g = r + L->ng; A.dl = r + L->ndl; A.jv = r + L->njv;
        A.cefc = ii + L->cefc;
        A.prof = profiling ? ii + L->nprof : (LDS_PTR(int))nullptr;
        A.tree_dofadr = li + O->tree_dofadr; A.tree_dofnum = li + O->tree_dofnum; A.tree_madr = li + O->tree_madr; A.dof_tree = li + O->dof_tree;
        A.nv = __builtin_amdgcn_readfirstlane(ka->m.nv); A.ntree = __builtin_amdgcn_readfirstlane(ka->m.ntree);
        A.nefc = __builtin_amdgcn_readfirstlane(nefc); A.ncon = __builtin_amdgcn_readfirstlane(ncon);
        A.nlead = __builtin_amdgcn_readfirstlane(nlead); A.iters = __builtin_amdgcn_readfirstlane(iters);
        A.tol = tol; A.scale = scale; A.ls_tol = ka->m.ls_tolerance; A.ls_iters = __builtin_amdgcn_readfirstlane(ka->m.ls_iterations);
        A.early_exit = __builtin_amdgcn_readfirstlane(ka->m.newton_early_exit);
    }
    const int nv = A.nv, ne = A.nefc;
    {   // nv <= 64 (one dof per lane; the register row of the dense factorisation holds 48)
        const int k = lane < nv ? lane : 0, t = A.dof_tree[k];
        A.k_a0 = A.tree_dofadr[t]; A.k_n = A.tree_dofnum[t]; A.k_mb = A.tree_madr[t] + (k - A.k_a0) * A.k_n;
    }
    int used = 0;
    constexpr bool coupled = COUPLED;
    NComp comp;
    if constexpr (coupled) comp = ncomponent<real>(A, lane, ne, (LDS_PTR(int))A.dl, __builtin_amdgcn_readfirstlane(ka->m.newton_component) == 0);      // (the search direction's words are idle until the first back substitution)
    long long tp0 = A.prof ? __builtin_readcyclecounter() : 0;
    // ---- per-contact constants ----
    NCon<real> con[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        NCon<real>& c = con[ch];
        const int ci = ch * 64 + lane;
        const int ce = ci < A.ncon ? A.cefc[ci] : -1;
        c.head = ce >= 0 ? (ce & 0xffff) : -1;
        c.dim = ce >= 0 ? (ce >> 16) : 0;
        c.mu = 0; c.Dm = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) { c.S[j] = 0; c.D[j] = 0; }
        if (c.head >= 0) {
#pragma unroll
            for (int j = 0; j < 6; j++)
                if (j < c.dim) { c.D[j] = real(1) / A.rowS[RS_S * (c.head + j) + 1]; if (j > 0) c.S[j] = real(1) / A.rowS[RS_S * (c.head + j) + 7]; }
            if (c.dim > 1) {
                const real R0 = A.rowS[RS_S * c.head + 1], R1 = A.rowS[RS_S * (c.head + 1) + 1];
                c.mu = c.S[1] * sqrt(R1 / R0);
                c.S[0] = c.mu;
                c.Dm = (real(1) / R0) / tmax(real(1e-15), c.mu * c.mu * (1 + c.mu * c.mu));
            }
        }
    }
    // ---- start from the warm start (already in a) or from the smooth acceleration, whichever costs less ----
    // (the residuals of both candidates come with the row records; the winner's end up in word 2 for the first iteration)
    const bool jar_ready = true;
    {
        const real c1 = ncost<real, NCH>(A, lane, con, A.as, 8, false);
        const real c0 = ncost<real, NCH>(A, lane, con, (LDS_PTR(const real))A.a, 2);
        if (!(c0 < c1)) {
            for (int k = lane; k < nv; k += 64) A.a[k] = A.as[k];
            for (int i = lane; i < ne; i += 64) A.rowS[RS_S * i + 2] = A.rowS[RS_S * i + 8];
        }
        NSYNC();
    }
    NPROF(0);
    int zone[NCH];
    real cw[NCH][6], cc1[NCH][6], cc2[NCH][6], cs1[NCH], cs2[NCH];
    // the Hessian depends only on the active set while no contact sits in the cone's middle zone: remember the set the
    // last factorisation was made for and skip assembly + factorisation when an iteration finds it unchanged
    bool have_L = false;
    unsigned long long sig_lead = 0, sig_z1[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) sig_z1[ch] = 0;
    bool forces_current = false, quad_step = false;
    unsigned long long prev_lead = 0, prev_sgn = 0, prev_z1[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) prev_z1[ch] = 0;
    real alpha_prev = 0;
    for (int it = 0; it < A.iters; it++) {
        used++;
        // ---- residuals, forces (-> rowS.f), curvature of the scalar rows (-> jv) ----
        if (it > 0) {
            // a moved by alpha dl and the line search left J dl in jv: the residuals follow without touching the rows [EXT: the
            // same incremental update as MuJoCo's primal solvers]
            for (int i = lane; i < ne; i += 64) A.rowS[RS_S * i + 2] += alpha_prev * A.jv[i];
            NSYNC();
        } else if (!jar_ready) {
            for (int i = lane; i < ne; i += 64) A.rowS[RS_S * i + 2] = nrow_dot(A, i, (LDS_PTR(const real))A.a) - A.rowS[RS_S * i];
            NSYNC();
        }
        for (int i = lane; i < A.nlead; i += 64) {
            real f, h;
            nrow_scalar<real>(A.rmeta[i] & 3, A.rowS[RS_S * i + 2], A.rowS[RS_S * i + 1], A.rowS[RS_S * i + 5], &f, &h);
            A.rowS[RS_S * i + 6] = f;
            A.jv[i] = h;
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            zone[ch] = 0;
            if (ch * 64 >= A.ncon) break;
            const NCon<real>& c = con[ch];
            if (c.head >= 0) {
                real jar[6], f[6], cc;
#pragma unroll
                for (int j = 0; j < 6; j++) jar[j] = j < c.dim ? A.rowS[RS_S * (c.head + j) + 2] : real(0);
                zone[ch] = ncone(c, jar, f, &cc, cw[ch], cc1[ch], cc2[ch], &cs1[ch], &cs2[ch]);
#pragma unroll
                for (int j = 0; j < 6; j++) if (j < c.dim) A.rowS[RS_S * (c.head + j) + 6] = f[j];
            }
        }
        // ---- active set now: scalar rows with curvature, contacts in the bottom / middle zone ----
        NSYNC();
        unsigned long long cur_lead = __ballot(lane < A.nlead && A.jv[lane < A.nlead ? lane : 0] != 0), cur_z1[NCH];
        bool middle = false, same = have_L && A.nlead <= 64 && cur_lead == sig_lead;
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            cur_z1[ch] = 0;
            if (ch * 64 >= A.ncon) break;
            cur_z1[ch] = __ballot(zone[ch] == 1);
            middle = middle || __ballot(zone[ch] == 2) != 0;
            same = same && cur_z1[ch] == sig_z1[ch];
        }
        // The step just taken ended in the zones it started from -- the same scalar rows quadratic, the saturated dry-friction rows on the
        // same side, the same contacts in the bottom zone -- and no contact of either end sits in the cone's middle zone: every row's cost
        // is ONE quadratic on the whole segment (the zones are convex sets, a segment with both ends inside stays inside), the factor was
        // that quadratic's Hessian and the line search exact -- this point is the minimiser, its gradient is zero to rounding.  The
        // gradient (a third of an iteration) would only confirm it: the forces just computed are the solution's.
        const unsigned long long cur_sgn = __ballot(lane < A.nlead && A.rowS[RS_S * (lane < A.nlead ? lane : 0) + 6] > 0);
        {
            bool unchanged = it > 0 && A.nlead <= 64 && cur_lead == prev_lead && cur_sgn == prev_sgn;
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) unchanged = unchanged && cur_z1[ch] == prev_z1[ch];
            if (unchanged && !middle && quad_step && A.early_exit) { forces_current = true; break; }
            prev_lead = cur_lead; prev_sgn = cur_sgn;
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) prev_z1[ch] = cur_z1[ch];
        }
        // ---- gradient g = M (a - a_s) - J^T f ----
        for (int k = lane; k < nv; k += 64) {
            const int a0 = A.k_a0, n = A.k_n, mb = A.k_mb;
            real s = 0;
#pragma unroll
            for (int j = 0; j < TREE_W; j++) { const int jj = j < n ? j : 0; const real m = A.M[mb + jj], d = A.a[a0 + jj] - A.as[a0 + jj]; s += j < n ? m * d : real(0); }
            A.g[k] = s;
        }
        NSYNC();
        rows_jt_force<real>((LDS_PTR(const real))A.rowS, A.rowI, A.cefc, A.rJ, A.g, A.nlead, A.ncon, lane, real(-1));
        NSYNC();
        real gn2 = 0;
        for (int k = lane; k < nv; k += 64) { const real s = A.g[k]; gn2 += s * s; }
        gn2 = wave_sum(gn2);
        NPROF(1);
#ifdef AVSIM_DEBUG_NEWTON
        if (lane == 0 && A.iters == 99) printf("  newton it %d: |g| scaled %.6e  (ne %d ncon %d nlead %d coupled %d)\n", it, (double)(sqrt(gn2) * A.scale), ne, A.ncon, A.nlead, (int)coupled);
#endif
        if (sqrt(gn2) * A.scale < A.tol) { forces_current = true; break; }     // residuals and forces were just computed at this a
        quad_step = !middle;       // (this iteration's step starts from a point without middle-zone contacts)
        if (!(same && !middle)) {
            // ---- Hessian: packed lower triangle ----
            // (block-diagonal case: only entries inside the tree blocks are read, and the M blocks below overwrite all of them)
            if constexpr (coupled) {
                for (int e = lane; e < nv * (nv + 1) / 2; e += 64) A.H[e] = 0;
                NSYNC();
            }
            if (lane < nv) {                                         // M blocks: lane = dof copies the lower part of its row
                const int k = lane, a0 = A.k_a0, kk = k - a0, hb = k * (k + 1) / 2 + a0;
#pragma unroll
                for (int j8 = 0; j8 < TREE_W; j8++)
                    if (j8 <= kk) A.H[hb + j8] = A.M[A.k_mb + j8];
            }
            NSYNC();
            nlead_rows<real>(A, lane);
    #pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
                if (ch * 64 >= A.ncon) break;
                const int nc = A.ncon - ch * 64 < 64 ? A.ncon - ch * 64 : 64;
                // one pass = four contacts (one per 16-lane group); the next pass's rows are requested before this pass is worked on
                int c = lane >> 4;                                 // this 16-lane group's contact
                int zn = c < nc ? __shfl(zone[ch], c & 63, 64) : 0; // (every lane takes part in the shuffles: any lane can be a source)
                int head = __shfl(con[ch].head, c & 63, 64), dim = __shfl(con[ch].dim, c & 63, 64);
                real Jraw[6];
                nblock4_load<real>(A, lane, head, dim, zn != 0, Jraw);
                for (int c0 = 0; c0 < nc; c0 += 4) {
                    const int cn = c0 + 4 + (lane >> 4);
                    const int zsn = __shfl(zone[ch], cn & 63, 64);
                    const int znn = (c0 + 4 < nc && cn < nc) ? zsn : 0;
                    const int headn = __shfl(con[ch].head, cn & 63, 64), dimn = __shfl(con[ch].dim, cn & 63, 64);
                    real Jnext[6];
                    nblock4_load<real>(A, lane, headn, dimn, znn != 0, Jnext);
                    if (__any(zn != 0)) {
                        real w[6];
    #pragma unroll
                        for (int p = 0; p < 6; p++) w[p] = __shfl(cw[ch][p], c & 63, 64);
                        nblock4<real, COUPLED>(A, lane, head, dim, zn != 0, zn == 2, w, c, cc1[ch], cc2[ch], cs1[ch], cs2[ch], Jraw);
                    }
                    c = cn; zn = znn; head = headn; dim = dimn;
    #pragma unroll
                    for (int p = 0; p < 6; p++) Jraw[p] = Jnext[p];
                }
            }
            NSYNC();
            NPROF(2);
            // ---- Cholesky + forward substitution in registers: lane i = row i, lane nv = -g ----
            if constexpr (!coupled) nblock_chol<real>(A, lane);
            else ndense_chol<real>(A.H, nv, comp.mydof, comp.nc, comp.oa0, comp.on, comp.oact);
            have_L = !middle && A.nlead <= 64;
            sig_lead = cur_lead;
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) sig_z1[ch] = cur_z1[ch];
        } else if constexpr (coupled) {
            // same Hessian as last time: forward substitution L y = -g with the stored factor
            ncomp_subst<real, true>(A, comp, lane);
        }
        NSYNC();
        NPROF(3);
        // ---- backward substitution L^T x = y, lane j holds x_j; column entries prefetched one step ahead ----
        if constexpr (!coupled) {
            nblock_solve<real>(A, lane);
        } else {
            ncomp_subst<real, false>(A, comp, lane);
        }
        NSYNC();
        NPROF(7);
        // ---- exact line search along dl ----
        real q1 = 0, q2 = 0;
        for (int k = lane; k < nv; k += 64) {
            const int a0 = A.k_a0, n = A.k_n, mb = A.k_mb;
            real s = 0;
#pragma unroll
            for (int j = 0; j < TREE_W; j++) { const int jj = j < n ? j : 0; const real m = A.M[mb + jj], d = A.dl[a0 + jj]; s += j < n ? m * d : real(0); }
            q2 += s * A.dl[k];
            q1 += s * (A.a[k] - A.as[k]);
        }
        q1 = wave_sum(q1);
        q2 = wave_sum(q2);
        for (int i = lane; i < ne; i += 64) A.jv[i] = i < A.nlead ? nrow_dot_lead(A, i, (LDS_PTR(const real))A.dl) : nrow_dot(A, i, (LDS_PTR(const real))A.dl);
        NSYNC();
        real cj0[NCH][6], cjv[NCH][6];
#pragma unroll
        for (int ch = 0; ch < NCH; ch++)
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const bool on = con[ch].head >= 0 && j < con[ch].dim;
                cj0[ch][j] = on ? A.rowS[RS_S * (con[ch].head + j) + 2] : real(0);
                cjv[ch][j] = on ? A.jv[con[ch].head + j] : real(0);
            }
        // Newton on phi' with a bracket [lo, hi], safeguarded as rtsafe (Numerical Recipes 9.4): a step that leaves the bracket OR is longer
        // than half the step before last becomes the bracket's midpoint.  The cone's middle-zone cost is not quadratic: phi'' along a line can
        // be four times larger in the middle than at its ends, and plain Newton then cycles between the two flat ends of the bracket (the
        // two-arm grasp of HookPackage: 100 stalled Newton iterations; oracle/orc_newton.c has the story and the same rule)
        real alpha = 0, lo = 0, hi = -1, dphi0 = 0, dxold = 0, dx = 0;
        for (int ls = 0; ls <= A.ls_iters; ls++) {       // (the cap as a run-time option costs nothing: 1 029 k against 1 029 k with the constant 51 of round 5, one box)
            real gsum = 0, hsum = 0;
            for (int i = lane; i < A.nlead; i += 64) {
                real f, h;
                const real jvi = A.jv[i];
                nrow_scalar<real>(A.rmeta[i] & 3, A.rowS[RS_S * i + 2] + alpha * jvi, A.rowS[RS_S * i + 1], A.rowS[RS_S * i + 5], &f, &h);
                gsum -= f * jvi;
                hsum += h * jvi * jvi;
            }
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
                if (ch * 64 >= A.ncon) break;
                if (con[ch].head >= 0) {
                    real jar[6], d1, d2;
#pragma unroll
                    for (int j = 0; j < 6; j++) jar[j] = cj0[ch][j] + alpha * cjv[ch][j];
                    ncone_ls(con[ch], jar, cjv[ch], &d1, &d2);
                    gsum += d1;
                    hsum += d2;
                }
            }
            const real dphi = q1 + alpha * q2 + wave_sum(gsum), ddphi = q2 + wave_sum(hsum);
#ifdef AVSIM_DEBUG_NEWTON
            if (lane == 0 && A.iters == 99 && it == 2) printf("         ls %d: alpha %.12e dphi %.6e ddphi %.6e lo %.6e hi %.6e\n", ls, (double)alpha, (double)dphi, (double)ddphi, (double)lo, (double)hi);
#endif
            if (ls == 0) {
                dphi0 = dphi;
                if (!(dphi0 < 0)) break;
                alpha = -dphi0 / ddphi;
                dxold = alpha; dx = alpha;
                continue;
            }
            if (fabs(dphi) < A.ls_tol * fabs(dphi0)) break;
            if (dphi < 0) lo = alpha; else hi = alpha;
            real nx = alpha - dphi / ddphi;
            if (hi < 0) { if (!(nx > lo)) nx = 2 * alpha + real(1e-12); }
            else if (!(nx > lo && nx < hi) || fabs(nx - alpha) > real(0.5) * fabs(dxold)) nx = real(0.5) * (lo + hi);
            dxold = dx;
            dx = nx - alpha;
            if (fabs(nx - alpha) < (sizeof(real) == 8 ? real(1e-14) : real(1e-7)) * (1 + fabs(alpha))) { alpha = nx; break; }
            alpha = nx;
        }
        NPROF(4);
#ifdef AVSIM_DEBUG_NEWTON
        if (lane == 0 && A.iters == 99) printf("      line search: dphi0 %.6e alpha %.6e q1 %.6e q2 %.6e lo %.6e hi %.6e  same-H %d middle %d\n", (double)dphi0, (double)alpha, (double)q1, (double)q2, (double)lo, (double)hi, (int)(same && !middle), (int)middle);
#endif
        if (!(dphi0 < 0)) break;
        real st2 = 0;
        for (int k = lane; k < nv; k += 64) { const real s = alpha * A.dl[k]; A.a[k] += s; st2 += s * s; }
        alpha_prev = alpha;
        st2 = wave_sum(st2);
        NSYNC();
        if (sqrt(st2) * A.scale < real(1e-2) * A.tol) break;
        // MuJoCo's improvement test [EXT]: the cost decrease of this iteration, -alpha phi'(0) / 2 to second order, scaled
        if (real(-0.5) * alpha * dphi0 * A.scale < A.tol) break;
    }
    // ---- forces at the solution (already there when the loop ended on the gradient test) ----
    if (!forces_current) {
        for (int i = lane; i < ne; i += 64) A.rowS[RS_S * i + 2] = (i < A.nlead ? nrow_dot_lead(A, i, (LDS_PTR(const real))A.a) : nrow_dot(A, i, (LDS_PTR(const real))A.a)) - A.rowS[RS_S * i];
        NSYNC();
        for (int i = lane; i < A.nlead; i += 64) {
            real f, h;
            nrow_scalar<real>(A.rmeta[i] & 3, A.rowS[RS_S * i + 2], A.rowS[RS_S * i + 1], A.rowS[RS_S * i + 5], &f, &h);
            A.rowS[RS_S * i + 6] = f;
        }
    #pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            if (ch * 64 >= A.ncon) break;
            const NCon<real>& c = con[ch];
            if (c.head >= 0) {
                real jar[6], f[6], w[6], c1[6], c2[6], cc, s1, s2;
    #pragma unroll
                for (int j = 0; j < 6; j++) jar[j] = j < c.dim ? A.rowS[RS_S * (c.head + j) + 2] : real(0);
                ncone(c, jar, f, &cc, w, c1, c2, &s1, &s2);
    #pragma unroll
                for (int j = 0; j < 6; j++) if (j < c.dim) A.rowS[RS_S * (c.head + j) + 6] = f[j];
            }
        }
    }
    NSYNC();
    NPROF(5);
    return used;
}

// The coupled instances as functions of their own: what they keep in registers (the component, the batched Hessian rows) then
// does not weigh on the register allocation of Env::solve, where the uncoupled instance of the headline scene is inlined.
template <typename real, int NCH>
__device__ AVS_OUTLINE_5 int newton_solve_coupled(KPtr<real> ka, GLB_PTR(const real) rows, LDS_PTR(real) r_, LDS_PTR(int) ii_, LDS_PTR(const int) li_, int nefc, int ncon, int nlead,
                                                              int iters, real tol, real scale, int profiling) {
    return newton_solve<real, NCH, true>(ka, rows, r_, ii_, li_, nefc, ncon, nlead, iters, tol, scale, profiling);
}

}  // namespace avs
