/* oracle/orc_collide.c -- collision detection of the CPU oracle.  TEST INFRASTRUCTURE ONLY (orc.h).
 *
 * Restates the geometry MuJoCo's mj_collision [EXT] applies to the reference scenes
 * (aloha_sim.xml:103-111 collision class, scene.xml:55 table box, task_*.xml boxes/cylinders):
 *   sphere-sphere, sphere-box: closed form;  box-box: separating axes + reference-face clipping,
 *   every clipped vertex behind the reference face is a contact (<= 8: a quadrilateral clipped by a
 *   rectangle; MuJoCo's mjc_BoxBox returns up to 8 [EXT]) -- orc_set_boxbox_maxpoints(4) restores the
 *   rounds 1-4 reduction to four extremal points;  everything involving a convex mesh or a cylinder:
 *   Minkowski Portal Refinement on support functions, the scheme of libccd's ccdMPRPenetration that
 *   MuJoCo 3.2 calls for these pairs (tolerance 1e-6, 50 iterations) + the multiccd perturbation passes.
 *   Mesh geoms: the hull vertices of the model blob (decimated to 20 / 64 vertices for the device), or,
 *   after orc_model_set_hulls, the FULL convex hulls of the STL files as MuJoCo collides them [EXT]
 *   (tests/orc_ffi.py load_model(hulls="full"); models/oracle_full_hulls.*).
 * Contact convention (MuJoCo): normal points from geom1 to geom2, pos is midway between the
 * surfaces, dist < 0 is penetration.
 */
#include <math.h>
#include <string.h>

#include "orc.h"

typedef struct {
    int type;
    const double* size;
    const double* pos; /* world position of the geom frame */
    const double* mat; /* 3x3 row-major rotation geom->world */
    const double* hull; /* hull vertices, geom frame */
    int nh;
    double center[3]; /* an interior point, world */
    double rbound;    /* bounding radius about the interior point (tolerance of the multiccd distinctness test) */
} shape;

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(const double* a, const double* b, double* c) {
    double t0 = a[1] * b[2] - a[2] * b[1], t1 = a[2] * b[0] - a[0] * b[2], t2 = a[0] * b[1] - a[1] * b[0];
    c[0] = t0; c[1] = t1; c[2] = t2;
}
static void sub3(const double* a, const double* b, double* c) { c[0] = a[0] - b[0]; c[1] = a[1] - b[1]; c[2] = a[2] - b[2]; }
static double normalize3(double* a) {
    double n = sqrt(dot3(a, a));
    if (n > 0) { double i = 1.0 / n; a[0] *= i; a[1] *= i; a[2] *= i; } /* one reciprocal, as the device kernel does */
    return n;
}
/* Margins of the discrete choices among candidates that are equal in exact arithmetic (support vertices of a face the direction
 * is normal to, vertices of a clipped polygon on an edge parallel to the base line, equal depths of a flat contact): a later
 * candidate replaces the incumbent only when it is better by more than rounding noise.  Same rule and constants as the device
 * (avsim_collide.hip.h TieTol<double>), so that both sides choose alike although their inputs differ in the last bits. */
#define TIE_REL 1e-9
#define TIE_LEN 1e-12
static void mulmatT(const double* R, const double* v, double* o) { /* o = R^T v */
    double t0 = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], t1 = R[1] * v[0] + R[4] * v[1] + R[7] * v[2],
           t2 = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
static void mulmat(const double* R, const double* v, double* o) { /* o = R v */
    double t0 = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], t1 = R[3] * v[0] + R[4] * v[1] + R[5] * v[2],
           t2 = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}

/* support point of a shape in world direction d (need not be unit) */
static void support(const shape* s, const double* d, double* out) {
    double l[3], p[3] = {0, 0, 0};
    mulmatT(s->mat, d, l);
    /* a component that is zero up to rounding (direction normal to a box face / a cylinder cap: every point of that face is a
       support point) resolves to the + corner / the cap centre instead of following the sign of the noise */
    const double lz = TIE_REL * (fabs(l[0]) + fabs(l[1]) + fabs(l[2]));
    switch (s->type) {
        case ORC_SPHERE: {
            double n = sqrt(dot3(l, l));
            if (n > 0) { double k = s->size[0] / n; p[0] = k * l[0]; p[1] = k * l[1]; p[2] = k * l[2]; }
            break;
        }
        case ORC_BOX:
            for (int i = 0; i < 3; i++) p[i] = l[i] >= -lz ? s->size[i] : -s->size[i];
            break;
        case ORC_CYLINDER: {
            double n = sqrt(l[0] * l[0] + l[1] * l[1]);
            if (n > lz) { double k = s->size[0] / n; p[0] = k * l[0]; p[1] = k * l[1]; }
            p[2] = l[2] >= -lz ? s->size[1] : -s->size[1];
            break;
        }
        case ORC_MESH: {
            /* the vertex with the largest projection; among vertices within the tie margin of it (the vertices of a face the direction is
               normal to: equal projections up to rounding) the LOWEST INDEX, whatever order they are looked at in -- the device looks at the
               candidates of the direction's support-table cell only (avsim_collide.hip.h support) */
            int best = 0;
            double bd = -1e30;
            const double tie = TIE_REL * 0.1 * (fabs(l[0]) + fabs(l[1]) + fabs(l[2]));
            for (int i = 0; i < s->nh; i++) {
                double v = dot3(s->hull + 3 * i, l);
                if (v > bd) bd = v;
            }
            for (int i = 0; i < s->nh; i++)
                if (dot3(s->hull + 3 * i, l) >= bd - tie) { best = i; break; }
            memcpy(p, s->hull + 3 * best, sizeof p);
            break;
        }
    }
    mulmat(s->mat, p, out);
    for (int i = 0; i < 3; i++) out[i] += s->pos[i];
}

/* ---------------------------------------------------------------------------------------------
 * Minkowski Portal Refinement on A - B.  Returns 1 with depth (>0), dir (unit, from A towards B)
 * and pos when the shapes overlap, else 0.
 * ------------------------------------------------------------------------------------------- */
typedef struct { double v[3], a[3], b[3]; } mpt; /* point of A-B with its witnesses */

static void msupport(const shape* A, const shape* B, const double* d, mpt* o) {
    double nd[3] = {-d[0], -d[1], -d[2]};
    support(A, d, o->a);
    support(B, nd, o->b);
    sub3(o->a, o->b, o->v);
}

static double point_tri_dist2(const double* p0, const double* p1, const double* p2, double* witness) {
    /* closest point to the origin on triangle (p0,p1,p2) */
    double e0[3], e1[3], d[3] = {p0[0], p0[1], p0[2]};
    sub3(p1, p0, e0);
    sub3(p2, p0, e1);
    double a = dot3(e0, e0), b = dot3(e0, e1), c = dot3(e1, e1), dd = dot3(e0, d), e = dot3(e1, d);
    double det = a * c - b * b, s = b * e - c * dd, t = b * dd - a * e;
    if (s + t <= det) {
        if (s < 0) {
            if (t < 0) { /* region 4 */
                if (dd < 0) { t = 0; s = (-dd >= a ? 1 : -dd / a); }
                else { s = 0; t = (e >= 0 ? 0 : (-e >= c ? 1 : -e / c)); }
            } else { s = 0; t = (e >= 0 ? 0 : (-e >= c ? 1 : -e / c)); }
        } else if (t < 0) { t = 0; s = (dd >= 0 ? 0 : (-dd >= a ? 1 : -dd / a)); }
        else { double inv = det > 0 ? 1 / det : 0; s *= inv; t *= inv; }
    } else {
        if (s < 0) {
            double t0 = b + dd, t1 = c + e;
            if (t1 > t0) { double num = t1 - t0, den = a - 2 * b + c; s = (num >= den ? 1 : num / den); t = 1 - s; }
            else { s = 0; t = (t1 <= 0 ? 1 : (e >= 0 ? 0 : -e / c)); }
        } else if (t < 0) {
            double t0 = b + e, t1 = a + dd;
            if (t1 > t0) { double num = t1 - t0, den = a - 2 * b + c; t = (num >= den ? 1 : num / den); s = 1 - t; }
            else { t = 0; s = (t1 <= 0 ? 1 : (dd >= 0 ? 0 : -dd / a)); }
        } else {
            double num = (c + e) - (b + dd), den = a - 2 * b + c;
            s = num <= 0 ? 0 : (num >= den ? 1 : num / den);
            t = 1 - s;
        }
    }
    for (int i = 0; i < 3; i++) witness[i] = p0[i] + s * e0[i] + t * e1[i];
    return dot3(witness, witness);
}

static int mpr_penetration(const shape* A, const shape* B, double* depth, double* dir, double* pos) {
    const double tol = 1e-6;
    const int maxit = 50;
    mpt v0, v1, v2, v3, v4;
    double d[3], t[3], t2[3];
    sub3(A->center, B->center, v0.v);
    memcpy(v0.a, A->center, sizeof v0.a);
    memcpy(v0.b, B->center, sizeof v0.b);
    if (dot3(v0.v, v0.v) < 1e-20) { v0.v[0] = 1e-5; v0.a[0] += 1e-5; }
    d[0] = -v0.v[0]; d[1] = -v0.v[1]; d[2] = -v0.v[2];
    normalize3(d);
    msupport(A, B, d, &v1);
    if (dot3(v1.v, d) <= 0) return 0;
    cross3(v0.v, v1.v, d);
    if (dot3(d, d) < 1e-24) {
        /* origin on the ray v0->v1: the contact direction is that ray */
        double n = sqrt(dot3(v1.v, v1.v));
        *depth = n;
        for (int i = 0; i < 3; i++) { dir[i] = v1.v[i] / n; pos[i] = 0.5 * (v1.a[i] + v1.b[i]); }
        return 1;
    }
    normalize3(d);
    msupport(A, B, d, &v2);
    if (dot3(v2.v, d) <= 0) return 0;
    sub3(v1.v, v0.v, t);
    sub3(v2.v, v0.v, t2);
    cross3(t, t2, d);
    normalize3(d);
    if (dot3(d, v0.v) > 0) { mpt s = v1; v1 = v2; v2 = s; d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }
    /* portal discovery */
    for (int it = 0;; it++) {
        if (it > maxit) return 0;
        msupport(A, B, d, &v3);
        if (dot3(v3.v, d) <= 0) return 0;
        cross3(v1.v, v3.v, t);
        if (dot3(t, v0.v) < -1e-14) {
            v2 = v3;
            sub3(v1.v, v0.v, t); sub3(v3.v, v0.v, t2); cross3(t, t2, d); normalize3(d);
            continue;
        }
        cross3(v3.v, v2.v, t);
        if (dot3(t, v0.v) < -1e-14) {
            v1 = v3;
            sub3(v3.v, v0.v, t); sub3(v2.v, v0.v, t2); cross3(t, t2, d); normalize3(d);
            continue;
        }
        break;
    }
    /* portal refinement */
    for (int it = 0;; it++) {
        sub3(v2.v, v1.v, t);
        sub3(v3.v, v1.v, t2);
        cross3(t, t2, d);
        if (normalize3(d) == 0) return 0;
        msupport(A, B, d, &v4);
        double dv1 = dot3(v1.v, d), dv2 = dot3(v2.v, d), dv3 = dot3(v3.v, d), dv4 = dot3(v4.v, d);
        double m = dv4 - dv1;
        if (dv4 - dv2 < m) m = dv4 - dv2;
        if (dv4 - dv3 < m) m = dv4 - dv3;
        if (dv4 <= 0) return 0; /* origin is outside the support plane: no overlap */
        if (m <= tol || it >= maxit) {
            if (dv1 < 0) return 0; /* converged with the origin beyond the surface: no overlap */
            double w[3];
            double d2 = point_tri_dist2(v1.v, v2.v, v3.v, w);
            *depth = sqrt(d2);
            if (*depth > 1e-12) { for (int i = 0; i < 3; i++) dir[i] = w[i] / *depth; }
            else memcpy(dir, d, 3 * sizeof(double));
            /* position from barycentric weights of the origin in the portal tetrahedron */
            double b0, b1, b2, b3, c[3];
            cross3(v1.v, v2.v, c); b0 = dot3(c, v3.v);
            cross3(v3.v, v2.v, c); b1 = dot3(c, v0.v);
            cross3(v0.v, v1.v, c); b2 = dot3(c, v3.v);
            cross3(v2.v, v1.v, c); b3 = dot3(c, v0.v);
            double sum = b0 + b1 + b2 + b3;
            if (sum <= 0) {
                b0 = 0;
                cross3(v2.v, v3.v, c); b1 = dot3(c, d);
                cross3(v3.v, v1.v, c); b2 = dot3(c, d);
                cross3(v1.v, v2.v, c); b3 = dot3(c, d);
                sum = b1 + b2 + b3;
            }
            double inv = 1.0 / sum;
            for (int i = 0; i < 3; i++) {
                double p1 = (b0 * v0.a[i] + b1 * v1.a[i] + b2 * v2.a[i] + b3 * v3.a[i]) * inv;
                double p2 = (b0 * v0.b[i] + b1 * v1.b[i] + b2 * v2.b[i] + b3 * v3.b[i]) * inv;
                pos[i] = 0.5 * (p1 + p2);
            }
            return 1;
        }
        /* expand the portal with v4 */
        double v4v0[3];
        cross3(v4.v, v0.v, v4v0);
        if (dot3(v1.v, v4v0) > 0) {
            if (dot3(v2.v, v4v0) > 0) v1 = v4; else v3 = v4;
        } else {
            if (dot3(v3.v, v4v0) > 0) v2 = v4; else v1 = v4;
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * closed forms
 * ------------------------------------------------------------------------------------------- */
static int sphere_sphere(const shape* a, const shape* b, double* dist, double* pos, double* nrm) {
    double d[3];
    sub3(b->pos, a->pos, d);
    double n = sqrt(dot3(d, d)), r = a->size[0] + b->size[0];
    if (n - r >= 0) return 0;
    if (n < 1e-12) { d[0] = 0; d[1] = 0; d[2] = 1; } else { d[0] /= n; d[1] /= n; d[2] /= n; }
    *dist = n - r;
    for (int i = 0; i < 3; i++) { nrm[i] = d[i]; pos[i] = a->pos[i] + d[i] * (a->size[0] + 0.5 * (*dist)); }
    return 1;
}

/* sphere a vs box b; normal from sphere to box */
static int sphere_box(const shape* a, const shape* b, double* dist, double* pos, double* nrm) {
    double rel[3], c[3], cl[3];
    sub3(a->pos, b->pos, rel);
    mulmatT(b->mat, rel, c); /* sphere centre in box frame */
    int inside = 1;
    for (int i = 0; i < 3; i++) {
        cl[i] = c[i];
        if (cl[i] > b->size[i]) { cl[i] = b->size[i]; inside = 0; }
        if (cl[i] < -b->size[i]) { cl[i] = -b->size[i]; inside = 0; }
    }
    double r = a->size[0], nl[3], dl;
    if (!inside) {
        double d[3] = {cl[0] - c[0], cl[1] - c[1], cl[2] - c[2]}; /* from sphere centre to the box */
        dl = sqrt(dot3(d, d));
        if (dl - r >= 0) return 0;
        for (int i = 0; i < 3; i++) nl[i] = d[i] / dl;
        *dist = dl - r;
    } else {
        int k = 0;
        double best = 1e300;
        for (int i = 0; i < 3; i++) {
            double m = b->size[i] - fabs(c[i]);
            if (m < best) { best = m; k = i; }
        }
        nl[0] = nl[1] = nl[2] = 0;
        nl[k] = c[k] >= 0 ? -1.0 : 1.0; /* towards the box interior */
        *dist = -best - r;
        memcpy(cl, c, sizeof cl);
        cl[k] = c[k] >= 0 ? b->size[k] : -b->size[k];
    }
    mulmat(b->mat, nl, nrm);
    /* sphere surface point along the normal and the box surface point; pos midway */
    for (int i = 0; i < 3; i++) pos[i] = a->pos[i] + nrm[i] * (r + 0.5 * (*dist));
    return 1;
}

static int boxbox_maxpoints = 8;
void orc_set_boxbox_maxpoints(int n) { boxbox_maxpoints = n >= 8 ? 8 : 4; }
int orc_get_boxbox_maxpoints(void) { return boxbox_maxpoints; }

/* box-box: 15-axis SAT, then reference-face clipping (face contact) or closest edge points */
static int box_box(const shape* a, const shape* b, double* dist, double* pos, double* nrm) {
    const double *Ra = a->mat, *Rb = b->mat;
    double p[3], pa[3];
    sub3(b->pos, a->pos, p);
    mulmatT(Ra, p, pa); /* centre offset in a's frame */
    double R[3][3], Q[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            R[i][j] = Ra[0 + i] * Rb[0 + j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j]; /* a_i . b_j */
            Q[i][j] = fabs(R[i][j]) + 1e-12;
        }
    double best = -1e300;
    int code = -1;
    double bn[3] = {0, 0, 0};
    int flip = 0;
    /* face axes of a */
    for (int i = 0; i < 3; i++) {
        double s = fabs(pa[i]) - (a->size[i] + b->size[0] * Q[i][0] + b->size[1] * Q[i][1] + b->size[2] * Q[i][2]);
        if (s > 0) return 0;
        if (s > best) { best = s; code = i; flip = pa[i] < 0; }
    }
    /* face axes of b */
    double pb[3];
    mulmatT(Rb, p, pb);
    for (int j = 0; j < 3; j++) {
        double s = fabs(pb[j]) - (b->size[j] + a->size[0] * Q[0][j] + a->size[1] * Q[1][j] + a->size[2] * Q[2][j]);
        if (s > 0) return 0;
        if (s > best) { best = s; code = 3 + j; flip = pb[j] < 0; }
    }
    /* edge-edge axes a_i x b_j, normalised; preferred only when clearly better (fudge as in ODE's dBoxBox) */
    const double fudge = 1.05;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            double ax[3]; /* axis in a's frame: e_i x (R col j) */
            double c[3] = {R[0][j], R[1][j], R[2][j]};
            double e[3] = {0, 0, 0};
            e[i] = 1;
            cross3(e, c, ax);
            double l = sqrt(dot3(ax, ax));
            if (l < 1e-8) continue;
            double sep = fabs(dot3(pa, ax)) - (a->size[i1] * Q[i2][j] + a->size[i2] * Q[i1][j] + b->size[j1] * Q[i][j2] + b->size[j2] * Q[i][j1]);
            sep /= l;
            if (sep > 0) return 0;
            if (sep * fudge > best) {
                /* separations are negative here: an edge axis must beat the best face axis by the fudge factor */
                best = sep; code = 6 + 3 * i + j;
                double axn[3] = {ax[0] / l, ax[1] / l, ax[2] / l};
                flip = dot3(pa, axn) < 0;
                mulmat(Ra, axn, bn);
            }
        }
    double depth = -best;
    double n[3]; /* world normal from a to b */
    if (code < 3) { n[0] = Ra[code]; n[1] = Ra[3 + code]; n[2] = Ra[6 + code]; }
    else if (code < 6) { n[0] = Rb[code - 3]; n[1] = Rb[3 + code - 3]; n[2] = Rb[6 + code - 3]; }
    else memcpy(n, bn, sizeof n);
    if (flip) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }

    if (code >= 6) {
        /* edge-edge: closest points of the two supporting edges */
        int i = (code - 6) / 3, j = (code - 6) % 3;
        double pA[3], pB[3], la[3], lb[3];
        mulmatT(Ra, n, la);
        mulmatT(Rb, n, lb);
        for (int k = 0; k < 3; k++) {
            la[k] = (k == i) ? 0 : (la[k] > 0 ? a->size[k] : -a->size[k]);
            lb[k] = (k == j) ? 0 : (lb[k] > 0 ? -b->size[k] : b->size[k]);
        }
        mulmat(Ra, la, pA);
        mulmat(Rb, lb, pB);
        for (int k = 0; k < 3; k++) { pA[k] += a->pos[k]; pB[k] += b->pos[k]; }
        double ua[3] = {Ra[i], Ra[3 + i], Ra[6 + i]}, ub[3] = {Rb[j], Rb[3 + j], Rb[6 + j]}, w[3];
        sub3(pB, pA, w);
        double uaub = dot3(ua, ub), q1 = dot3(ua, w), q2 = -dot3(ub, w), den = 1 - uaub * uaub;
        double alpha = 0, beta = 0;
        if (den > 1e-10) { alpha = (q1 + uaub * q2) / den; beta = (uaub * q1 + q2) / den; }
        for (int k = 0; k < 3; k++) {
            pA[k] += ua[k] * alpha;
            pB[k] += ub[k] * beta;
            pos[k] = 0.5 * (pA[k] + pB[k]);
            nrm[k] = n[k];
        }
        dist[0] = -depth;
        return 1;
    }

    /* face contact: reference box = owner of the axis, incident box = the other */
    const shape *ref = code < 3 ? a : b, *inc = code < 3 ? b : a;
    double nr[3] = {n[0], n[1], n[2]}; /* outward normal of the reference face */
    if (code >= 3) { nr[0] = -n[0]; nr[1] = -n[1]; nr[2] = -n[2]; }
    int ax = code % 3;
    /* incident face: the face of `inc` most anti-parallel to nr */
    double li[3];
    mulmatT(inc->mat, nr, li);
    int k = 0;
    for (int q = 1; q < 3; q++)
        if (fabs(li[q]) > fabs(li[k])) k = q;
    double sgn = li[k] > 0 ? -1.0 : 1.0;
    int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
    double poly[16][3], tmp[16][3];
    int np = 4;
    const double cs[4][2] = {{1, 1}, {-1, 1}, {-1, -1}, {1, -1}};
    for (int q = 0; q < 4; q++) {
        double l[3];
        l[k] = sgn * inc->size[k];
        l[k1] = cs[q][0] * inc->size[k1];
        l[k2] = cs[q][1] * inc->size[k2];
        double wv[3], rel[3];
        mulmat(inc->mat, l, wv);
        for (int c = 0; c < 3; c++) rel[c] = wv[c] + inc->pos[c] - ref->pos[c];
        mulmatT(ref->mat, rel, poly[q]); /* vertex in the reference box frame */
    }
    int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
    /* Sutherland-Hodgman against the 4 side planes of the reference face */
    for (int side = 0; side < 4; side++) {
        int axis = side < 2 ? a1 : a2;
        double s = (side & 1) ? -1.0 : 1.0, lim = ref->size[axis];
        int m = 0;
        for (int q = 0; q < np; q++) {
            double* P = poly[q];
            double* Qp = poly[(q + 1) % np];
            double dp = s * P[axis] - lim, dq = s * Qp[axis] - lim;
            if (dp <= 0) { memcpy(tmp[m++], P, 24); }
            if ((dp < 0 && dq > 0) || (dp > 0 && dq < 0)) {
                double t = dp / (dp - dq);
                for (int c = 0; c < 3; c++) tmp[m][c] = P[c] + t * (Qp[c] - P[c]);
                m++;
            }
            if (m >= 15) break;
        }
        np = m;
        memcpy(poly, tmp, sizeof(double) * 3 * np);
        if (np == 0) return 0;
    }
    /* keep points below the reference face */
    const double refax[3] = {ref->mat[ax], ref->mat[3 + ax], ref->mat[6 + ax]};
    double face = (dot3(nr, refax) > 0 ? 1.0 : -1.0);
    double dep[16];
    int m = 0;
    for (int q = 0; q < np; q++) {
        double dq = ref->size[ax] - face * poly[q][ax];
        if (dq >= 0) { memcpy(tmp[m], poly[q], 24); dep[m] = dq; m++; }
    }
    if (m == 0) return 0;
    /* every clipped vertex is a contact (<= 8); boxbox_maxpoints == 4: reduce to at most 4 -- deepest, farthest from it, farthest from
       that segment on either side (the selection of rounds 1-4, kept for comparison) */
    int keep[8], nk = 0;
    if (m > 8) m = 8;
    if (m <= boxbox_maxpoints) { for (int q = 0; q < m; q++) keep[nk++] = q; }
    else {
        int i0 = 0;
        for (int q = 1; q < m; q++) if (dep[q] > dep[i0] + TIE_LEN) i0 = q;
        int i1 = i0; double bd = -1;
        for (int q = 0; q < m; q++) {
            double dx = tmp[q][a1] - tmp[i0][a1], dy = tmp[q][a2] - tmp[i0][a2], dd = dx * dx + dy * dy;
            if (dd > bd + TIE_REL * fabs(bd)) { bd = dd; i1 = q; }
        }
        double ex = tmp[i1][a1] - tmp[i0][a1], ey = tmp[i1][a2] - tmp[i0][a2];
        int i2 = -1, i3 = -1; double mx = 1e-18, mn = -1e-18;
        for (int q = 0; q < m; q++) {
            double cr = ex * (tmp[q][a2] - tmp[i0][a2]) - ey * (tmp[q][a1] - tmp[i0][a1]);
            if (cr > mx + TIE_REL * fabs(mx)) { mx = cr; i2 = q; }
            if (cr < mn - TIE_REL * fabs(mn)) { mn = cr; i3 = q; }
        }
        keep[nk++] = i0; keep[nk++] = i1;
        if (i2 >= 0) keep[nk++] = i2;
        if (i3 >= 0) keep[nk++] = i3;
    }
    /* keep input order for determinism */
    for (int x = 0; x < nk; x++)
        for (int y = x + 1; y < nk; y++)
            if (keep[y] < keep[x]) { int t = keep[x]; keep[x] = keep[y]; keep[y] = t; }
    for (int x = 0; x < nk; x++) {
        int q = keep[x];
        double l[3] = {tmp[q][0], tmp[q][1], tmp[q][2]};
        l[ax] += 0.5 * dep[q] * face; /* midway between incident vertex and reference face */
        double wv[3];
        mulmat(ref->mat, l, wv);
        for (int c = 0; c < 3; c++) { pos[3 * x + c] = wv[c] + ref->pos[c]; nrm[3 * x + c] = n[c]; }
        dist[x] = -dep[q];
    }
    return nk;
}

static void make_frame(double* f);

/* ---- multiccd (aloha_sim.xml:5 <flag multiccd="enable"/>) ----
 * MuJoCo 3.2's mjc_Convex [EXT] looks for further contacts of a convex pair that its penetration routine found in contact (not
 * for spheres / ellipsoids): both geoms are turned by a small angle in opposite senses about the first contact point, about each
 * of the two tangent axes of the contact frame and in both directions; the penetration routine runs again in each of the four
 * perturbed configurations and a contact found there is kept if it lies farther than 1e-3 x the smaller bounding radius from
 * every contact kept so far (at most 1 + 4).  The kept contacts share the first contact's frame.  Same constants and expression
 * order as the device (avsim_collide.hip.h MultiCcd, rotate_shape, mpr_perturbed). */
#define MCCD_COS 0.9999995000000417
#define MCCD_SIN 0.0009999998333333417
#define MCCD_RELTOL 1e-3

typedef struct { shape s; double pos[3], mat[9]; } rshape; /* a shape with its own (turned) pose */

static void rotate_shape(rshape* r, const shape* src, const double* ax, double c, double s, const double* o) {
    const double oc = 1 - c;
    const double R[9] = {c + ax[0] * ax[0] * oc, ax[0] * ax[1] * oc - ax[2] * s, ax[0] * ax[2] * oc + ax[1] * s,
                         ax[1] * ax[0] * oc + ax[2] * s, c + ax[1] * ax[1] * oc, ax[1] * ax[2] * oc - ax[0] * s,
                         ax[2] * ax[0] * oc - ax[1] * s, ax[2] * ax[1] * oc + ax[0] * s, c + ax[2] * ax[2] * oc};
    r->s = *src;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r->mat[3 * i + j] = R[3 * i] * src->mat[j] + R[3 * i + 1] * src->mat[3 + j] + R[3 * i + 2] * src->mat[6 + j];
    double d[3], t[3];
    sub3(src->pos, o, d);
    mulmat(R, d, t);
    for (int k = 0; k < 3; k++) r->pos[k] = o[k] + t[k];
    sub3(src->center, o, d);
    mulmat(R, d, t);
    for (int k = 0; k < 3; k++) r->s.center[k] = o[k] + t[k];
    r->s.pos = r->pos;
    r->s.mat = r->mat;
}

static int mpr_perturbed(const shape* A, const shape* B, const double* p0, const double* n0, int pert, double* dist, double* pos) {
    double f[9] = {n0[0], n0[1], n0[2], 0, 0, 0, 0, 0, 0};
    make_frame(f);
    const double* ax = pert < 2 ? f + 3 : f + 6;
    const double s = (pert & 1) ? -MCCD_SIN : MCCD_SIN;
    rshape ra, rb;
    rotate_shape(&ra, A, ax, MCCD_COS, s, p0);
    rotate_shape(&rb, B, ax, MCCD_COS, -s, p0);
    double depth, dir[3];
    if (!mpr_penetration(&ra.s, &rb.s, &depth, dir, pos)) return 0;
    *dist = -depth;
    return 1;
}

static int multiccd(const shape* a, const shape* b, double* dist, double* pos, double* nrm) {
    const double tol = MCCD_RELTOL * (a->rbound < b->rbound ? a->rbound : b->rbound);
    int n = 1;
    for (int pert = 0; pert < 4; pert++) {
        double d, p[3];
        if (!mpr_perturbed(a, b, pos, nrm, pert, &d, p)) continue;
        int ok = 1;
        for (int k = 0; k < n; k++) {
            const double e[3] = {p[0] - pos[3 * k], p[1] - pos[3 * k + 1], p[2] - pos[3 * k + 2]};
            ok = ok && dot3(e, e) > tol * tol;
        }
        if (!ok) continue;
        dist[n] = d;
        memcpy(pos + 3 * n, p, 24);
        memcpy(nrm + 3 * n, nrm, 24);
        n++;
    }
    return n;
}

static int narrow(const shape* a, const shape* b, double* dist, double* pos, double* nrm) {
    int ta = a->type, tb = b->type;
    if (ta == ORC_SPHERE && tb == ORC_SPHERE) return sphere_sphere(a, b, dist, pos, nrm);
    if (ta == ORC_SPHERE && tb == ORC_BOX) return sphere_box(a, b, dist, pos, nrm);
    if (ta == ORC_BOX && tb == ORC_SPHERE) {
        int n = sphere_box(b, a, dist, pos, nrm);
        for (int i = 0; i < 3 * n; i++) nrm[i] = -nrm[i];
        return n;
    }
    if (ta == ORC_BOX && tb == ORC_BOX) return box_box(a, b, dist, pos, nrm);
    double depth;
    if (!mpr_penetration(a, b, &depth, nrm, pos)) return 0;
    dist[0] = -depth;
    if (ta == ORC_SPHERE || tb == ORC_SPHERE) return 1;
    return multiccd(a, b, dist, pos, nrm);
}

static void shape_center(shape* s, const double* bcenter) {
    double c[3];
    mulmat(s->mat, bcenter, c);
    for (int i = 0; i < 3; i++) s->center[i] = s->pos[i] + c[i];
}

int orc_narrow(int t1, const double* size1, const double* pos1, const double* mat1, const double* hull1, int nh1,
               int t2, const double* size2, const double* pos2, const double* mat2, const double* hull2, int nh2,
               double* dist, double* pos, double* normal) {
    shape a = {t1, size1, pos1, mat1, hull1, nh1, {0, 0, 0}, 0.05}, b = {t2, size2, pos2, mat2, hull2, nh2, {0, 0, 0}, 0.05};
    double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
    for (int i = 0; i < nh1; i++) for (int k = 0; k < 3; k++) ca[k] += hull1[3 * i + k] / nh1;
    for (int i = 0; i < nh2; i++) for (int k = 0; k < 3; k++) cb[k] += hull2[3 * i + k] / nh2;
    shape_center(&a, ca);
    shape_center(&b, cb);
    return narrow(&a, &b, dist, pos, normal);
}

/* mju_makeFrame [EXT]: complete the contact frame from its normal */
static void make_frame(double* f) {
    double* y = f + 3;
    y[0] = y[1] = y[2] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) y[1] = 1; else y[2] = 1;
    double t = dot3(f, y);
    for (int i = 0; i < 3; i++) y[i] -= t * f[i];
    normalize3(y);
    cross3(f, y, f + 6);
}

void orc_collide(orc_data* d) {
    const orc_model* m = d->m;
    d->ncon = 0;
    for (int p = 0; p < m->npair; p++) {
        int g1 = m->pair_geom[2 * p], g2 = m->pair_geom[2 * p + 1];
        shape a = {m->geom_type[g1], m->geom_size + 3 * g1, d->geom_xpos + 3 * g1, d->geom_xmat + 9 * g1,
                   m->hull_vert + 3 * m->geom_hull[2 * g1], m->geom_hull[2 * g1 + 1], {0, 0, 0}, m->geom_rbound[g1]};
        shape b = {m->geom_type[g2], m->geom_size + 3 * g2, d->geom_xpos + 3 * g2, d->geom_xmat + 9 * g2,
                   m->hull_vert + 3 * m->geom_hull[2 * g2], m->geom_hull[2 * g2 + 1], {0, 0, 0}, m->geom_rbound[g2]};
        shape_center(&a, m->geom_bcenter + 3 * g1);
        shape_center(&b, m->geom_bcenter + 3 * g2);
        /* broad phase: bounding spheres about the interior points */
        double dc[3];
        sub3(b.center, a.center, dc);
        double rr = m->geom_rbound[g1] + m->geom_rbound[g2] + m->pair_margin[p];
        if (dot3(dc, dc) > rr * rr) continue;
        double dist[8], pos[24], nrm[24];
        d->stat_narrow++;
        int n = narrow(&a, &b, dist, pos, nrm);
        for (int k = 0; k < n; k++) {
            if (dist[k] >= m->pair_margin[p]) continue;
            if (d->ncon >= ORC_MAXCON) { d->overflow = 1; break; }
            orc_contact* c = &d->contact[d->ncon++];
            c->dist = dist[k];
            memcpy(c->pos, pos + 3 * k, 24);
            memcpy(c->frame, nrm + 3 * k, 24);
            make_frame(c->frame);
            c->geom1 = g1; c->geom2 = g2; c->pair = p;
            c->dim = m->pair_condim[p];
            memcpy(c->friction, m->pair_friction + 5 * p, 40);
            memcpy(c->solref, m->pair_solref + 2 * p, 16);
            memcpy(c->solimp, m->pair_solimp + 5 * p, 40);
            c->includemargin = m->pair_margin[p] - m->pair_gap[p];
            c->efc_adr = -1;
        }
    }
}
