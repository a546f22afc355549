// avsim_collide.hip.h -- narrow-phase geometry, one candidate pair per lane.
// Replaces the part of MuJoCo's mj_collision [EXT] the reference scenes exercise:
// sphere-sphere / sphere-box closed forms, box-box by separating axes + reference-face clipping (<=4
// points), and Minkowski Portal Refinement on support functions for anything involving a convex mesh
// hull or a cylinder (the libccd scheme MuJoCo 3.2 uses for those pairs; tolerance 1e-6, 50 iterations).
// Contact convention: normal from geom1 to geom2, pos midway between the surfaces, dist<0 = penetration.
#pragma once
#include "avsim_math.hip.h"

namespace avs {

enum { G_SPHERE = 2, G_CYLINDER = 5, G_BOX = 6, G_MESH = 7 };
// narrow-phase result slot of one candidate pair: up to 5 contacts (box-box keeps <= 4, multiccd <= 1 + 4) that share a normal
constexpr int SLOT_W = 24, SLOT_P = 5, SLOT_N = 20, SLOT_MAXC = 5;
// box-box manifolds hold up to 8 points (every vertex of the clipped incident face that lies behind the reference face, as MuJoCo's
// mjc_BoxBox returns [EXT]): points 0..3 go to the pair's LDS result slot, points 4..7 -- a quadrilateral cut by the reference rectangle
// into a polygon of five to eight vertices: rare -- to a per-pair overflow record of BOX_OVF_W words in global scratch (point 4 + j at
// words 4 j .. 4 j + 3: dist, pos), so that the eight-point manifold costs no LDS
constexpr int BOX_MAXC = 8, BOX_SLOTC = 4, BOX_OVF_W = 16;

template <typename T>
struct Shape {
    int type;
    T size[3];
    T pos[3];      // world position of the geom frame
    T mat[9];      // geom->world rotation, row-major
    // mesh geoms: the collision hull (<= 128 vertices, compile.py) behind its support table -- the unit sphere of directions cut into the
    // 6 R^2 cells of a cube map, every cell listing the vertices that support SOME direction of the (padded) cell (compiler/hull.py
    // support_table).  `hull` = the model's table: one record of eight entries per cell (four words each: a candidate vertex in the geom
    // frame; the fourth word of entry 0 holds the cell's candidate count, that of entry 1 the first of its further candidates in the
    // overflow part -- cells with more than eight: the normal of a face with many vertices), lists sorted by vertex index, short lists
    // padded with their last vertex: a support call is ONE round trip of eight 16-byte loads (a 128-byte line in f32) for most directions.
    // hbase = the hull's first cell record, hR = its cube-map resolution, hovf = entry index at which the overflow part starts
    GLB_PTR(const T) hull;
    int hbase, hR, hovf;
    int nh;
    T center[3];   // an interior point (world)
    T lc[3], lh[3]; // local bounding box: centre and half extents in the geom frame
};

// Margins of the discrete choices among candidates that are equal in exact arithmetic (support vertices of a face, the vertices of a
// clipped polygon on an edge parallel to the base line, ...): a later candidate replaces the incumbent only when it is better by
// more than rounding noise, so both sides of a parity comparison (f64 device, f64 oracle, and mostly the f32 product kernel too)
// make the same choice although their inputs differ in the last bits (kinematic chains are multiplied out in different orders).
template <typename T> struct TieTol;
template <> struct TieTol<double> { static constexpr double rel = 1e-9, len = 1e-12; };
template <> struct TieTol<float> { static constexpr float rel = 1e-5f, len = 1e-7f; };

template <typename T> AVS_DEV T sel3c(const T* v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }

template <typename T>
AVS_DEV void support(const Shape<T>& s, const T* d, T* out) {
    T l[3], p[3] = {0, 0, 0};
    mulmatT(s.mat, d, l);
    // a direction component that is zero up to rounding (the direction is normal to a box face / a cylinder cap: every point of
    // that face supports it) resolves to the + corner / the cap centre instead of following the sign of the noise
    const T lz = TieTol<T>::rel * (fabs(l[0]) + fabs(l[1]) + fabs(l[2]));
    switch (s.type) {
        case G_SPHERE: {
            T n = sqrt(dot3(l, l));
            if (n > T(0)) { T k = s.size[0] / n; p[0] = k * l[0]; p[1] = k * l[1]; p[2] = k * l[2]; }
            break;
        }
        case G_BOX:
            p[0] = l[0] >= -lz ? s.size[0] : -s.size[0];
            p[1] = l[1] >= -lz ? s.size[1] : -s.size[1];
            p[2] = l[2] >= -lz ? s.size[2] : -s.size[2];
            break;
        case G_CYLINDER: {
            T n = sqrt(l[0] * l[0] + l[1] * l[1]);
            if (n > lz) { T k = s.size[0] / n; p[0] = k * l[0]; p[1] = k * l[1]; }
            p[2] = l[2] >= -lz ? s.size[1] : -s.size[1];
            break;
        }
        default: {
            // the vertex with the largest projection, and among the vertices within the tie margin of it (TieTol: a rounding-level fraction
            // of |l| x 0.1 m -- the vertices of a face the direction is normal to have equal projections up to rounding) the one with the
            // LOWEST INDEX, on the device and in the oracle alike.  Only the candidates of the direction's cube-map cell are looked at:
            // typically two to eight of the hull's <= 128 vertices, one round trip for the cell record and one for eight candidates
            const T ax = fabs(l[0]), ay = fabs(l[1]), az = fabs(l[2]);
            const int a = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
            const T lm = sel3c(l, a), am = fabs(lm), inv = am > T(0) ? T(1) / am : T(0);
            const T u = sel3c(l, a == 2 ? 0 : a + 1) * inv, v = sel3c(l, a == 0 ? 2 : a - 1) * inv;
            const int R = s.hR;
            int iu = (int)((u + T(1)) * T(0.5) * T(R)), iv = (int)((v + T(1)) * T(0.5) * T(R));
            iu = iu > 0 ? (iu < R - 1 ? iu : R - 1) : 0;
            iv = iv > 0 ? (iv < R - 1 ? iv : R - 1) : 0;
            GLB_PTR(const T) C = s.hull + 32 * (size_t)(s.hbase + ((2 * a + (lm < T(0) ? 1 : 0)) * R + iu) * R + iv);
            const T tie = TieTol<T>::rel * T(0.1) * (ax + ay + az);
            T vx[8], vy[8], vz[8], pr[8], w0, w1;
#pragma unroll
            for (int k = 0; k < 8; k++) {           // the cell's record: eight candidates, one round trip
                vx[k] = C[4 * k]; vy[k] = C[4 * k + 1]; vz[k] = C[4 * k + 2];
            }
            w0 = C[3]; w1 = C[7];
            const int cnt = (int)w0;
            T bd = T(-1e30);
#pragma unroll
            for (int k = 0; k < 8; k++) { pr[k] = vx[k] * l[0] + vy[k] * l[1] + vz[k] * l[2]; bd = pr[k] > bd ? pr[k] : bd; }
            GLB_PTR(const T) O = s.hull + 4 * ((size_t)s.hovf + (size_t)(cnt > 8 ? (int)w1 : 0));
            const int more = cnt - 8, lastm = more - 1;
            for (int i = 0; i < more; i += 8) {     // (cells at the normal of a face with many vertices: the further candidates)
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int ik = i + k < lastm ? i + k : lastm;
                    const T q = O[4 * ik] * l[0] + O[4 * ik + 1] * l[1] + O[4 * ik + 2] * l[2];
                    bd = q > bd ? q : bd;
                }
            }
            const T thr = bd - tie;
            bool found = false;
            T bx = vx[0], by = vy[0], bz = vz[0];
#pragma unroll
            for (int k = 7; k >= 0; k--)            // (descending: the lowest index is assigned last)
                if (pr[k] >= thr) { bx = vx[k]; by = vy[k]; bz = vz[k]; found = true; }
            for (int i = 0; i < more && !found; i += 8) {
#pragma unroll
                for (int k = 7; k >= 0; k--) {
                    const int ik = i + k < lastm ? i + k : lastm;
                    const T qx = O[4 * ik], qy = O[4 * ik + 1], qz = O[4 * ik + 2];
                    if (qx * l[0] + qy * l[1] + qz * l[2] >= thr) { bx = qx; by = qy; bz = qz; found = true; }
                }
            }
            p[0] = bx; p[1] = by; p[2] = bz;
        }
    }
    mulmat(s.mat, p, out);
    out[0] += s.pos[0]; out[1] += s.pos[1]; out[2] += s.pos[2];
}

template <typename T>
struct MPt {
    T v[3], a[3], b[3];
};

template <typename T>
AVS_DEV void msupport(const Shape<T>& A, const Shape<T>& B, const T* d, MPt<T>& o) {
    T nd[3] = {-d[0], -d[1], -d[2]};
    support(A, d, o.a);
    support(B, nd, o.b);
    sub3(o.a, o.b, o.v);
}

// closest point to the origin on a triangle (Eberly's region decomposition)
template <typename T>
AVS_DEV T point_tri_dist2(const T* p0, const T* p1, const T* p2, T* w) {
    T e0[3], e1[3];
    sub3(p1, p0, e0);
    sub3(p2, p0, e1);
    T a = dot3(e0, e0), b = dot3(e0, e1), c = dot3(e1, e1), dd = dot3(e0, p0), e = dot3(e1, p0);
    T det = a * c - b * b, s = b * e - c * dd, t = b * dd - a * e;
    if (s + t <= det) {
        if (s < 0) {
            if (t < 0) {
                if (dd < 0) { t = 0; s = (-dd >= a ? T(1) : -dd / a); }
                else { s = 0; t = (e >= 0 ? T(0) : (-e >= c ? T(1) : -e / c)); }
            } else { s = 0; t = (e >= 0 ? T(0) : (-e >= c ? T(1) : -e / c)); }
        } else if (t < 0) { t = 0; s = (dd >= 0 ? T(0) : (-dd >= a ? T(1) : -dd / a)); }
        else { T inv = det > 0 ? T(1) / det : T(0); s *= inv; t *= inv; }
    } else {
        if (s < 0) {
            T t0 = b + dd, t1 = c + e;
            if (t1 > t0) { T num = t1 - t0, den = a - 2 * b + c; s = (num >= den ? T(1) : num / den); t = 1 - s; }
            else { s = 0; t = (t1 <= 0 ? T(1) : (e >= 0 ? T(0) : -e / c)); }
        } else if (t < 0) {
            T t0 = b + e, t1 = a + dd;
            if (t1 > t0) { T num = t1 - t0, den = a - 2 * b + c; t = (num >= den ? T(1) : num / den); s = 1 - t; }
            else { t = 0; s = (t1 <= 0 ? T(1) : (dd >= 0 ? T(0) : -dd / a)); }
        } else {
            T num = (c + e) - (b + dd), den = a - 2 * b + c;
            s = num <= 0 ? T(0) : (num >= den ? T(1) : num / den);
            t = 1 - s;
        }
    }
    for (int i = 0; i < 3; i++) w[i] = p0[i] + s * e0[i] + t * e1[i];
    return dot3(w, w);
}

template <typename T> AVS_DEV void make_frame(const T* n, T* t1, T* t2);
template <typename T> struct MprTol;
template <> struct MprTol<double> { static constexpr double tol = 1e-6, tiny2 = 1e-24, eps = 1e-14; };
template <> struct MprTol<float> { static constexpr float tol = 1e-6f, tiny2 = 1e-16f, eps = 1e-9f; };

// MPR on A - B: returns 1 with depth>0, dir (unit, A -> B) and pos when the shapes overlap
template <typename T>
__device__ int mpr_penetration(const Shape<T>& A, const Shape<T>& B, T* depth, T* dir, T* pos) {
    const T tol = MprTol<T>::tol;
    const int maxit = 50;
    MPt<T> v0, v1, v2, v3, v4;
    T d[3], t[3], t2[3];
    sub3(A.center, B.center, v0.v);
    for (int i = 0; i < 3; i++) { v0.a[i] = A.center[i]; v0.b[i] = B.center[i]; }
    if (dot3(v0.v, v0.v) < T(1e-20)) { v0.v[0] = T(1e-5); v0.a[0] += T(1e-5); }
    d[0] = -v0.v[0]; d[1] = -v0.v[1]; d[2] = -v0.v[2];
    normalize3(d);
    msupport(A, B, d, v1);
    if (dot3(v1.v, d) <= 0) return 0;
    cross3(v0.v, v1.v, d);
    if (dot3(d, d) < MprTol<T>::tiny2) {
        T n = sqrt(dot3(v1.v, v1.v));
        *depth = n;
        for (int i = 0; i < 3; i++) { dir[i] = v1.v[i] / n; pos[i] = T(0.5) * (v1.a[i] + v1.b[i]); }
        return 1;
    }
    normalize3(d);
    msupport(A, B, d, v2);
    if (dot3(v2.v, d) <= 0) return 0;
    sub3(v1.v, v0.v, t);
    sub3(v2.v, v0.v, t2);
    cross3(t, t2, d);
    normalize3(d);
    if (dot3(d, v0.v) > 0) { MPt<T> s = v1; v1 = v2; v2 = s; d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }
    for (int it = 0;; it++) {
        if (it > maxit) return 0;
        msupport(A, B, d, v3);
        if (dot3(v3.v, d) <= 0) return 0;
        cross3(v1.v, v3.v, t);
        if (dot3(t, v0.v) < -MprTol<T>::eps) {
            v2 = v3;
            sub3(v1.v, v0.v, t); sub3(v3.v, v0.v, t2); cross3(t, t2, d); normalize3(d);
            continue;
        }
        cross3(v3.v, v2.v, t);
        if (dot3(t, v0.v) < -MprTol<T>::eps) {
            v1 = v3;
            sub3(v3.v, v0.v, t); sub3(v2.v, v0.v, t2); cross3(t, t2, d); normalize3(d);
            continue;
        }
        break;
    }
    for (int it = 0;; it++) {
        sub3(v2.v, v1.v, t);
        sub3(v3.v, v1.v, t2);
        cross3(t, t2, d);
        if (normalize3(d) == T(0)) return 0;
        msupport(A, B, d, v4);
        T dv1 = dot3(v1.v, d), dv2 = dot3(v2.v, d), dv3 = dot3(v3.v, d), dv4 = dot3(v4.v, d);
        T m = dv4 - dv1;
        if (dv4 - dv2 < m) m = dv4 - dv2;
        if (dv4 - dv3 < m) m = dv4 - dv3;
        if (dv4 <= 0) return 0;
        if (m <= tol || it >= maxit) {
            if (dv1 < 0) return 0;
            T w[3];
            T d2 = point_tri_dist2(v1.v, v2.v, v3.v, w);
            *depth = sqrt(d2);
            if (*depth > T(1e-12)) { for (int i = 0; i < 3; i++) dir[i] = w[i] / *depth; }
            else { dir[0] = d[0]; dir[1] = d[1]; dir[2] = d[2]; }
            T b0, b1, b2, b3, c[3];
            cross3(v1.v, v2.v, c); b0 = dot3(c, v3.v);
            cross3(v3.v, v2.v, c); b1 = dot3(c, v0.v);
            cross3(v0.v, v1.v, c); b2 = dot3(c, v3.v);
            cross3(v2.v, v1.v, c); b3 = dot3(c, v0.v);
            T sum = b0 + b1 + b2 + b3;
            if (sum <= 0) {
                b0 = 0;
                cross3(v2.v, v3.v, c); b1 = dot3(c, d);
                cross3(v3.v, v1.v, c); b2 = dot3(c, d);
                cross3(v1.v, v2.v, c); b3 = dot3(c, d);
                sum = b1 + b2 + b3;
            }
            T inv = T(1) / sum;
            for (int i = 0; i < 3; i++) {
                T p1 = (b0 * v0.a[i] + b1 * v1.a[i] + b2 * v2.a[i] + b3 * v3.a[i]) * inv;
                T p2 = (b0 * v0.b[i] + b1 * v1.b[i] + b2 * v2.b[i] + b3 * v3.b[i]) * inv;
                pos[i] = T(0.5) * (p1 + p2);
            }
            return 1;
        }
        T v4v0[3];
        cross3(v4.v, v0.v, v4v0);
        if (dot3(v1.v, v4v0) > 0) {
            if (dot3(v2.v, v4v0) > 0) v1 = v4; else v3 = v4;
        } else {
            if (dot3(v3.v, v4v0) > 0) v2 = v4; else v1 = v4;
        }
    }
}

// ---- multiccd (aloha_sim.xml:5 <flag multiccd="enable"/>) -----------------------------------------------------------------
// MuJoCo 3.2's mjc_Convex [EXT] looks for further contacts of a convex pair that its penetration routine found in contact (not for
// spheres / ellipsoids): both geoms are turned by a small angle in opposite senses about the first contact point, about each of
// the two tangent axes of the contact frame and in both directions, the penetration routine runs again in each of the four
// perturbed configurations, and a contact found there is kept if it lies farther than 1e-3 x the smaller bounding radius from
// every contact kept so far (at most 1 + 4).  A flat finger pad on a flat face thus gets the corners of the touching patch
// instead of one point somewhere inside it.  The kept contacts share the first contact's frame.
template <typename T> struct MultiCcd {
    static constexpr T cs = T(0.9999995000000417), sn = T(0.0009999998333333417);   // cos / sin of the 1e-3 rad perturbation
    static constexpr T reltol = T(1e-3);
};

// turn the shape by the angle with cosine c / sine s about the unit axis through the point o
template <typename T>
AVS_DEV void rotate_shape(Shape<T>& sh, const T* ax, T c, T s, const T* o) {
    const T oc = 1 - c;
    const T R[9] = {c + ax[0] * ax[0] * oc, ax[0] * ax[1] * oc - ax[2] * s, ax[0] * ax[2] * oc + ax[1] * s,
                    ax[1] * ax[0] * oc + ax[2] * s, c + ax[1] * ax[1] * oc, ax[1] * ax[2] * oc - ax[0] * s,
                    ax[2] * ax[0] * oc - ax[1] * s, ax[2] * ax[1] * oc + ax[0] * s, c + ax[2] * ax[2] * oc};
    T M[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) M[3 * i + j] = R[3 * i] * sh.mat[j] + R[3 * i + 1] * sh.mat[3 + j] + R[3 * i + 2] * sh.mat[6 + j];
#pragma unroll
    for (int k = 0; k < 9; k++) sh.mat[k] = M[k];
    T d[3], t[3];
    sub3(sh.pos, o, d);
    mulmat(R, d, t);
    for (int k = 0; k < 3; k++) sh.pos[k] = o[k] + t[k];
    sub3(sh.center, o, d);
    mulmat(R, d, t);
    for (int k = 0; k < 3; k++) sh.center[k] = o[k] + t[k];
}

// perturbation `pert` (0..3: tangent 1 +, tangent 1 -, tangent 2 +, tangent 2 -) of the pair about the contact point p0 with
// normal n0: 1 with the contact's distance and position, or 0
// (inlined into its caller: as an out-of-line function its two Shapes -- 79 dwords per lane -- are passed through the wave's private segment)
template <typename T>
AVS_DEV int mpr_perturbed(Shape<T> A, Shape<T> B, const T* p0, const T* n0, int pert, T* dist, T* pos) {
    T t1[3], t2[3];
    make_frame(n0, t1, t2);
    T ax[3];
#pragma unroll
    for (int k = 0; k < 3; k++) ax[k] = pert < 2 ? t1[k] : t2[k];
    const T s = (pert & 1) ? -MultiCcd<T>::sn : MultiCcd<T>::sn;
    rotate_shape(A, ax, MultiCcd<T>::cs, s, p0);
    rotate_shape(B, ax, MultiCcd<T>::cs, -s, p0);
    T depth, dir[3];
    if (!mpr_penetration(A, B, &depth, dir, pos)) return 0;
    *dist = -depth;
    return 1;
}

// is the point p farther than tol from the first n kept positions?
template <typename T, typename P>
AVS_DEV bool multiccd_distinct(const T* p, const P kept, int n, T tol) {
    bool ok = true;
    for (int k = 0; k < n; k++) {
        const T d[3] = {p[0] - kept[3 * k], p[1] - kept[3 * k + 1], p[2] - kept[3 * k + 2]};
        ok = ok && dot3(d, d) > tol * tol;
    }
    return ok;
}

// the four perturbations one after the other (host builds and tests; the kernel spreads them over lanes, same order of
// acceptance): dist[0], pos[0..2] hold the first contact, nrm its normal; returns the number of contacts kept (1..5)
template <typename T>
__device__ int multiccd_serial(const Shape<T>& a, const Shape<T>& b, T tol, T* dist, T* pos, const T* nrm) {
    int n = 1;
    for (int pert = 0; pert < 4; pert++) {
        T d, p[3];
        if (mpr_perturbed(a, b, pos, nrm, pert, &d, p) && multiccd_distinct(p, pos, n, tol)) {
            dist[n] = d;
            pos[3 * n] = p[0]; pos[3 * n + 1] = p[1]; pos[3 * n + 2] = p[2];
            n++;
        }
    }
    return n;
}

template <typename T>
AVS_DEV int sphere_sphere(const Shape<T>& a, const Shape<T>& b, T* dist, T* pos, T* nrm) {
    T d[3];
    sub3(b.pos, a.pos, d);
    T n = sqrt(dot3(d, d)), r = a.size[0] + b.size[0];
    if (n - r >= 0) return 0;
    if (n < T(1e-12)) { d[0] = 0; d[1] = 0; d[2] = 1; } else { d[0] /= n; d[1] /= n; d[2] /= n; }
    *dist = n - r;
    for (int i = 0; i < 3; i++) { nrm[i] = d[i]; pos[i] = a.pos[i] + d[i] * (a.size[0] + T(0.5) * (*dist)); }
    return 1;
}

// sphere a vs box b; normal from the sphere towards the box
template <typename T>
AVS_DEV int sphere_box(const Shape<T>& a, const Shape<T>& b, T* dist, T* pos, T* nrm) {
    T rel[3], c[3], cl[3];
    sub3(a.pos, b.pos, rel);
    mulmatT(b.mat, rel, c);
    bool inside = true;
    for (int i = 0; i < 3; i++) {
        cl[i] = c[i];
        if (cl[i] > b.size[i]) { cl[i] = b.size[i]; inside = false; }
        if (cl[i] < -b.size[i]) { cl[i] = -b.size[i]; inside = false; }
    }
    T r = a.size[0], nl[3];
    if (!inside) {
        T d[3] = {cl[0] - c[0], cl[1] - c[1], cl[2] - c[2]};
        T dl = sqrt(dot3(d, d));
        if (dl - r >= 0) return 0;
        for (int i = 0; i < 3; i++) nl[i] = d[i] / dl;
        *dist = dl - r;
    } else {
        int k = 0;
        T best = T(1e30);
        for (int i = 0; i < 3; i++) {
            T m = b.size[i] - fabs(c[i]);
            if (m < best) { best = m; k = i; }
        }
        nl[0] = nl[1] = nl[2] = 0;
        T sg = c[k] >= 0 ? T(-1) : T(1);
        if (k == 0) nl[0] = sg; else if (k == 1) nl[1] = sg; else nl[2] = sg;
        *dist = -best - r;
    }
    mulmat(b.mat, nl, nrm);
    for (int i = 0; i < 3; i++) pos[i] = a.pos[i] + nrm[i] * (r + T(0.5) * (*dist));
    return 1;
}

#ifndef AVS_LDS
#define AVS_LDS(T) __attribute__((address_space(3))) T*
#endif

// select component k of a 3-vector / column k of a row-major 3x3 without dynamic indexing (keeps data in registers)
template <typename T> AVS_DEV T sel3(const T* v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }
template <typename T> AVS_DEV void col3(const T* M, int k, T* o) { o[0] = sel3(M, k); o[1] = sel3(M + 3, k); o[2] = sel3(M + 6, k); }

// box-box: 15-axis SAT then reference-face clipping (face contact, <= 8 points: BOX_MAXC) or closest edge points.
// `work` = 56 words of LDS scratch for the clipped polygon (dynamic indexing would otherwise spill it to scratch memory),
// `scr` = the lane's result slot (SLOT_W words): dist [0,5), pos [SLOT_P, SLOT_P + 15), common normal [SLOT_N, SLOT_N + 3);
// `ovf` = the pair's overflow record for points 4..7 (BOX_OVF_W words, any pointer type).
template <typename T, typename OVF>
__device__ int box_box(const Shape<T>& a, const Shape<T>& b, AVS_LDS(T) scr, AVS_LDS(T) work, OVF ovf) {
    const T *Ra = a.mat, *Rb = b.mat;
    T p[3], pa[3], pb[3];
    sub3(b.pos, a.pos, p);
    mulmatT(Ra, p, pa);
    mulmatT(Rb, p, pb);
    T R[3][3], Q[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            R[i][j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j];
            Q[i][j] = fabs(R[i][j]) + T(1e-12);
        }
    T best = T(-1e30);
    int code = -1;
    T bn[3] = {0, 0, 0};
    bool flip = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        T s = fabs(pa[i]) - (a.size[i] + b.size[0] * Q[i][0] + b.size[1] * Q[i][1] + b.size[2] * Q[i][2]);
        if (s > 0) return 0;
        if (s > best) { best = s; code = i; flip = pa[i] < 0; }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        T s = fabs(pb[j]) - (b.size[j] + a.size[0] * Q[0][j] + a.size[1] * Q[1][j] + a.size[2] * Q[2][j]);
        if (s > 0) return 0;
        if (s > best) { best = s; code = 3 + j; flip = pb[j] < 0; }
    }
    const T fudge = T(1.05);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            T c[3] = {R[0][j], R[1][j], R[2][j]}, e[3] = {0, 0, 0}, ax[3];
            e[i] = 1;
            cross3(e, c, ax);
            T l = sqrt(dot3(ax, ax));
            if (l < T(1e-8)) continue;
            T sep = fabs(dot3(pa, ax)) - (a.size[i1] * Q[i2][j] + a.size[i2] * Q[i1][j] + b.size[j1] * Q[i][j2] + b.size[j2] * Q[i][j1]);
            sep /= l;
            if (sep > 0) return 0;
            if (sep * fudge > best) {
                best = sep;
                code = 6 + 3 * i + j;
                T axn[3] = {ax[0] / l, ax[1] / l, ax[2] / l};
                flip = dot3(pa, axn) < 0;
                mulmat(Ra, axn, bn);
            }
        }
    T depth = -best;
    T n[3];
    if (code < 3) col3(Ra, code, n);
    else if (code < 6) col3(Rb, code - 3, n);
    else { n[0] = bn[0]; n[1] = bn[1]; n[2] = bn[2]; }
    if (flip) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }

    if (code >= 6) {
        int i = (code - 6) / 3, j = (code - 6) % 3;
        T pA[3], pB[3], la[3], lb[3];
        mulmatT(Ra, n, la);
        mulmatT(Rb, n, lb);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            la[k] = (k == i) ? T(0) : (la[k] > 0 ? a.size[k] : -a.size[k]);
            lb[k] = (k == j) ? T(0) : (lb[k] > 0 ? -b.size[k] : b.size[k]);
        }
        mulmat(Ra, la, pA);
        mulmat(Rb, lb, pB);
        for (int k = 0; k < 3; k++) { pA[k] += a.pos[k]; pB[k] += b.pos[k]; }
        T ua[3], ub[3], w[3];
        col3(Ra, i, ua);
        col3(Rb, j, ub);
        sub3(pB, pA, w);
        T uaub = dot3(ua, ub), q1 = dot3(ua, w), q2 = -dot3(ub, w), den = 1 - uaub * uaub;
        T alpha = 0, beta = 0;
        if (den > T(1e-10)) { alpha = (q1 + uaub * q2) / den; beta = (uaub * q1 + q2) / den; }
        for (int k = 0; k < 3; k++) {
            pA[k] += ua[k] * alpha;
            pB[k] += ub[k] * beta;
            scr[SLOT_P + k] = T(0.5) * (pA[k] + pB[k]);
            scr[SLOT_N + k] = n[k];
        }
        scr[0] = -depth;
        return 1;
    }

    // reference / incident box chosen by component-wise selects (a dynamic reference would force both shapes to memory)
    const bool ra = code < 3;
    T rpos[3], rmat[9], rsize[3], ipos[3], imat[9], isize[3];
#pragma unroll
    for (int q = 0; q < 3; q++) { rpos[q] = ra ? a.pos[q] : b.pos[q]; ipos[q] = ra ? b.pos[q] : a.pos[q]; rsize[q] = ra ? a.size[q] : b.size[q]; isize[q] = ra ? b.size[q] : a.size[q]; }
#pragma unroll
    for (int q = 0; q < 9; q++) { rmat[q] = ra ? a.mat[q] : b.mat[q]; imat[q] = ra ? b.mat[q] : a.mat[q]; }
    T nr[3] = {n[0], n[1], n[2]};
    if (code >= 3) { nr[0] = -n[0]; nr[1] = -n[1]; nr[2] = -n[2]; }
    int ax = code % 3;
    T li[3];
    mulmatT(imat, nr, li);
    int k = 0;
    if (fabs(li[1]) > fabs(li[0])) k = 1;
    if (fabs(li[2]) > fabs(sel3(li, k))) k = 2;
    T sgn = sel3(li, k) > 0 ? T(-1) : T(1);
    int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
    AVS_LDS(T) poly = work;          // [8][3]
    AVS_LDS(T) tmp = work + 24;      // [8][3]
    AVS_LDS(T) dep = work + 48;      // [8]
    int np = 4;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        T cs0 = (q == 0 || q == 3) ? T(1) : T(-1), cs1 = (q < 2) ? T(1) : T(-1);
        T l[3];
#pragma unroll
        for (int j = 0; j < 3; j++) l[j] = (j == k ? sgn : (j == k1 ? cs0 : cs1)) * isize[j];
        T wv[3], rel[3];
        mulmat(imat, l, wv);
        for (int cc = 0; cc < 3; cc++) rel[cc] = wv[cc] + ipos[cc] - rpos[cc];
        { T pv[3]; mulmatT(rmat, rel, pv); poly[3 * q] = pv[0]; poly[3 * q + 1] = pv[1]; poly[3 * q + 2] = pv[2]; }
    }
    int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
    // Sutherland-Hodgman against the four side planes of the reference face: the two LDS polygons swap roles instead of being
    // copied back, and the edge's end point becomes the next edge's start point in registers (one LDS read per edge)
    AVS_LDS(T) src = poly;
    AVS_LDS(T) dst = tmp;
    for (int side = 0; side < 4; side++) {
        int axis = side < 2 ? a1 : a2;
        T s = (side & 1) ? T(-1) : T(1), lim = sel3(rsize, axis);
        int m = 0;
        const T P0[3] = {src[0], src[1], src[2]};
        T P[3] = {P0[0], P0[1], P0[2]};
        for (int q = 0; q < np; q++) {
            const bool last = q + 1 >= np;
            const int qn = last ? 0 : q + 1;
            T Qp[3] = {src[3 * qn], src[3 * qn + 1], src[3 * qn + 2]};
            if (last) { Qp[0] = P0[0]; Qp[1] = P0[1]; Qp[2] = P0[2]; }
            T dp = s * sel3(P, axis) - lim, dq = s * sel3(Qp, axis) - lim;
            if (dp <= 0 && m < 8) { dst[3 * m] = P[0]; dst[3 * m + 1] = P[1]; dst[3 * m + 2] = P[2]; m++; }
            if (((dp < 0 && dq > 0) || (dp > 0 && dq < 0)) && m < 8) {
                T t = dp / (dp - dq);
                for (int c = 0; c < 3; c++) dst[3 * m + c] = P[c] + t * (Qp[c] - P[c]);
                m++;
            }
            P[0] = Qp[0]; P[1] = Qp[1]; P[2] = Qp[2];
        }
        np = m;
        if (np == 0) return 0;
        AVS_LDS(T) t_ = src; src = dst; dst = t_;
    }
    // four swaps: the clipped polygon is back in `poly`, `tmp` is free for the points that lie behind the reference face
    T refax[3];
    col3(rmat, ax, refax);
    T face = dot3(nr, refax) > 0 ? T(1) : T(-1);
    int m = 0;
    for (int q = 0; q < np; q++) {
        const T px = src[3 * q], py = src[3 * q + 1], pz = src[3 * q + 2];
        const T pc[3] = {px, py, pz};
        T dq = sel3(rsize, ax) - face * sel3(pc, ax);
        if (dq >= 0) { dst[3 * m] = px; dst[3 * m + 1] = py; dst[3 * m + 2] = pz; dep[m] = dq; m++; }
    }
    if (m == 0) return 0;
    // every clipped vertex behind the reference face is a contact, in polygon order (at most 8)
    const int nk = m < BOX_MAXC ? m : BOX_MAXC;
    for (int x = 0; x < nk; x++) {
        const int q = x;
        T l[3] = {tmp[3 * q], tmp[3 * q + 1], tmp[3 * q + 2]};
#pragma unroll
        for (int j = 0; j < 3; j++) l[j] += (j == ax) ? T(0.5) * dep[q] * face : T(0);
        T wv[3];
        mulmat(rmat, l, wv);
        T dq_ = dep[q];
        if (x < BOX_SLOTC) {
            for (int c = 0; c < 3; c++) scr[SLOT_P + 3 * x + c] = wv[c] + rpos[c];
            scr[x] = -dq_;
        } else {
            ovf[4 * (x - BOX_SLOTC)] = -dq_;
            for (int c = 0; c < 3; c++) ovf[4 * (x - BOX_SLOTC) + 1 + c] = wv[c] + rpos[c];
        }
    }
    for (int c = 0; c < 3; c++) scr[SLOT_N + c] = n[c];
    return nk;
}

// ---- box-box on 16 lanes ------------------------------------------------------------------------------------------------
// The same algorithm as box_box (same expressions, same tie-breaks, bit-identical results), spread over the 16 lanes of a DPP
// row that all hold the same pair: lane t evaluates separating axis t (15 axes), the winner is chosen by replaying the serial
// comparison chain on the gathered separations; for a face contact lane q carries polygon vertex q through the four clips
// (neighbour by shuffle, compaction by ballot + LDS), the depth filter and the output.  Every lane of the row must call it
// with the same arguments; `g16` = lane & ~15 (first lane of the row), `on` = the row holds a real pair.  Returns the number
// of contacts (row-uniform); results go to `scr` as in box_box.
template <typename T> AVS_DEV T sel33(const T (*M)[3], int i, int j) { return sel3(i == 0 ? M[0] : (i == 1 ? M[1] : M[2]), j); }
template <typename T> AVS_DEV T row_get(T x, int g16, int l) { return __shfl(x, g16 | l, 64); }

template <typename T, typename OVF>
__device__ int box_box16(const Shape<T>& a, const Shape<T>& b, AVS_LDS(T) scr, AVS_LDS(T) work, OVF ovf, int lane, bool on) {
    const int t = lane & 15, g16 = lane & ~15;
    const unsigned rowmask_sh = g16;
    const T *Ra = a.mat, *Rb = b.mat;
    T p[3], pa[3], pb[3];
    sub3(b.pos, a.pos, p);
    mulmatT(Ra, p, pa);
    mulmatT(Rb, p, pb);
    T R[3][3], Q[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            R[i][j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j];
            Q[i][j] = fabs(R[i][j]) + T(1e-12);
        }
    // ---- this lane's axis ----
    T sv = T(-1e30);          // separation along axis t
    bool skip = t >= 15;      // degenerate edge pair (or the idle 16th lane)
    bool flipv = false;
    T bnv[3] = {0, 0, 0};
    if (t < 3) {
        const int i = t;
        sv = fabs(sel3(pa, i)) - (sel3(a.size, i) + b.size[0] * sel33(Q, i, 0) + b.size[1] * sel33(Q, i, 1) + b.size[2] * sel33(Q, i, 2));
        flipv = sel3(pa, i) < 0;
    } else if (t < 6) {
        const int j = t - 3;
        sv = fabs(sel3(pb, j)) - (sel3(b.size, j) + a.size[0] * sel33(Q, 0, j) + a.size[1] * sel33(Q, 1, j) + a.size[2] * sel33(Q, 2, j));
        flipv = sel3(pb, j) < 0;
    } else if (t < 15) {
        const int i = (t - 6) / 3, j = (t - 6) % 3;
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        T c[3] = {sel3(R[0], j), sel3(R[1], j), sel3(R[2], j)}, e[3] = {i == 0 ? T(1) : T(0), i == 1 ? T(1) : T(0), i == 2 ? T(1) : T(0)}, ax[3];
        cross3(e, c, ax);
        const T l = sqrt(dot3(ax, ax));
        if (l < T(1e-8)) skip = true;
        else {
            T sep = fabs(dot3(pa, ax)) - (sel3(a.size, i1) * sel33(Q, i2, j) + sel3(a.size, i2) * sel33(Q, i1, j) + sel3(b.size, j1) * sel33(Q, i, j2) + sel3(b.size, j2) * sel33(Q, i, j1));
            sep /= l;
            sv = sep;
            T axn[3] = {ax[0] / l, ax[1] / l, ax[2] / l};
            flipv = dot3(pa, axn) < 0;
            mulmat(Ra, axn, bnv);
        }
    }
    // a separating axis anywhere in the row: no contact
    {
        const unsigned long long sepm = __ballot(!skip && sv > 0);
        if ((sepm >> g16) & 0xffffull) return 0;
    }
    // ---- the serial comparison chain on the gathered values ----
    T best = T(-1e30);
    int code = -1;
    const T fudge = T(1.05);
    const unsigned skipm = (unsigned)((__ballot(skip) >> g16) & 0xffffull);
#pragma unroll
    for (int k = 0; k < 15; k++) {
        const T sk = row_get(sv, g16, k);
        if (k < 6) { if (sk > best) { best = sk; code = k; } }
        else if (!((skipm >> k) & 1u)) { if (sk * fudge > best) { best = sk; code = k; } }
    }
    const bool flip = row_get(flipv ? 1 : 0, g16, code) != 0;
    T bn[3] = {row_get(bnv[0], g16, code), row_get(bnv[1], g16, code), row_get(bnv[2], g16, code)};
    const T depth = -best;
    T n[3];
    if (code < 3) col3(Ra, code, n);
    else if (code < 6) col3(Rb, code - 3, n);
    else { n[0] = bn[0]; n[1] = bn[1]; n[2] = bn[2]; }
    if (flip) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }

    if (code >= 6) {     // edge-edge: short, every lane computes it, the first lane of the row writes
        int i = (code - 6) / 3, j = (code - 6) % 3;
        T pA[3], pB[3], la[3], lb[3];
        mulmatT(Ra, n, la);
        mulmatT(Rb, n, lb);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            la[k] = (k == i) ? T(0) : (la[k] > 0 ? a.size[k] : -a.size[k]);
            lb[k] = (k == j) ? T(0) : (lb[k] > 0 ? -b.size[k] : b.size[k]);
        }
        mulmat(Ra, la, pA);
        mulmat(Rb, lb, pB);
        for (int k = 0; k < 3; k++) { pA[k] += a.pos[k]; pB[k] += b.pos[k]; }
        T ua[3], ub[3], w[3];
        col3(Ra, i, ua);
        col3(Rb, j, ub);
        sub3(pB, pA, w);
        T uaub = dot3(ua, ub), q1 = dot3(ua, w), q2 = -dot3(ub, w), den = 1 - uaub * uaub;
        T alpha = 0, beta = 0;
        if (den > T(1e-10)) { alpha = (q1 + uaub * q2) / den; beta = (uaub * q1 + q2) / den; }
        if (on && t == 0) {
            for (int k = 0; k < 3; k++) {
                pA[k] += ua[k] * alpha;
                pB[k] += ub[k] * beta;
                scr[SLOT_P + k] = T(0.5) * (pA[k] + pB[k]);
                scr[SLOT_N + k] = n[k];
            }
            scr[0] = -depth;
        }
        return 1;
    }

    // ---- face contact ----
    const bool ra = code < 3;
    T rpos[3], rmat[9], rsize[3], ipos[3], imat[9], isize[3];
#pragma unroll
    for (int q = 0; q < 3; q++) { rpos[q] = ra ? a.pos[q] : b.pos[q]; ipos[q] = ra ? b.pos[q] : a.pos[q]; rsize[q] = ra ? a.size[q] : b.size[q]; isize[q] = ra ? b.size[q] : a.size[q]; }
#pragma unroll
    for (int q = 0; q < 9; q++) { rmat[q] = ra ? a.mat[q] : b.mat[q]; imat[q] = ra ? b.mat[q] : a.mat[q]; }
    T nr[3] = {n[0], n[1], n[2]};
    if (code >= 3) { nr[0] = -n[0]; nr[1] = -n[1]; nr[2] = -n[2]; }
    const int ax = code % 3;
    T li[3];
    mulmatT(imat, nr, li);
    int k = 0;
    if (fabs(li[1]) > fabs(li[0])) k = 1;
    if (fabs(li[2]) > fabs(sel3(li, k))) k = 2;
    const T sgn = sel3(li, k) > 0 ? T(-1) : T(1);
    const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
    (void)k2;
    AVS_LDS(T) poly = work;          // [8][3]
    AVS_LDS(T) tmp = work + 24;      // [8][3]
    AVS_LDS(T) dep = work + 48;      // [8]
    // vertex q of the incident face, in the reference box's frame (lanes 0..3; the other lanes compute vertex t & 3, unused)
    T P[3];
    {
        const int q = t & 3;
        const T cs0 = (q == 0 || q == 3) ? T(1) : T(-1), cs1 = (q < 2) ? T(1) : T(-1);
        T l[3];
#pragma unroll
        for (int j = 0; j < 3; j++) l[j] = (j == k ? sgn : (j == k1 ? cs0 : cs1)) * isize[j];
        T wv[3], rel[3];
        mulmat(imat, l, wv);
        for (int cc = 0; cc < 3; cc++) rel[cc] = wv[cc] + ipos[cc] - rpos[cc];
        mulmatT(rmat, rel, P);
    }
    int np = 4;
    const int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
    AVS_LDS(T) dst = tmp;
    // The incident face entirely inside the reference face's rectangle (a box resting on a larger one): every clip would keep the
    // four vertices as they are, in their order (dp <= 0 for all of them, no crossing edge), so the four rounds of neighbour
    // shuffles, ballots and LDS compaction are skipped -- the result is the same polygon, bit for bit
    const bool inside = t >= 4 || (fabs(sel3(P, a1)) <= sel3(rsize, a1) && fabs(sel3(P, a2)) <= sel3(rsize, a2));
    const bool noclip = ((__ballot(inside) >> g16) & 0xffffull) == 0xffffull;
    if (!noclip)
    for (int side = 0; side < 4; side++) {
        const int axis = side < 2 ? a1 : a2;
        const T s = (side & 1) ? T(-1) : T(1), lim = sel3(rsize, axis);
        const int qn = t + 1 >= np ? 0 : t + 1;
        const T Qp[3] = {row_get(P[0], g16, qn), row_get(P[1], g16, qn), row_get(P[2], g16, qn)};
        const T dp = s * sel3(P, axis) - lim, dq = s * sel3(Qp, axis) - lim;
        const bool live = t < np;
        const bool e1 = live && dp <= 0, e2 = live && ((dp < 0 && dq > 0) || (dp > 0 && dq < 0));
        const unsigned m1 = (unsigned)((__ballot(e1) >> g16) & 0xffffull), m2 = (unsigned)((__ballot(e2) >> g16) & 0xffffull);
        const unsigned below = (1u << t) - 1u;
        const int off = __popc(m1 & below) + __popc(m2 & below);
        if (e1 && off < 8) { dst[3 * off] = P[0]; dst[3 * off + 1] = P[1]; dst[3 * off + 2] = P[2]; }
        if (e2 && off + (e1 ? 1 : 0) < 8) {
            const int o2 = off + (e1 ? 1 : 0);
            const T tt = dp / (dp - dq);
            for (int c = 0; c < 3; c++) dst[3 * o2 + c] = P[c] + tt * (Qp[c] - P[c]);
        }
        const int total = __popc(m1) + __popc(m2);
        np = total < 8 ? total : 8;
        if (np == 0) return 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int rq = t < np ? t : 0;
        P[0] = dst[3 * rq]; P[1] = dst[3 * rq + 1]; P[2] = dst[3 * rq + 2];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        dst = dst == tmp ? poly : tmp;
    }
    // after four sides `dst` is `tmp` again: points behind the reference face go there, with their depths
    T refax[3];
    col3(rmat, ax, refax);
    const T face = dot3(nr, refax) > 0 ? T(1) : T(-1);
    int m;
    {
        const T dqv = sel3(rsize, ax) - face * sel3(P, ax);
        const bool kp = t < np && dqv >= 0;
        const unsigned km = (unsigned)((__ballot(kp) >> g16) & 0xffffull);
        const int off = __popc(km & ((1u << t) - 1u));
        if (kp) { tmp[3 * off] = P[0]; tmp[3 * off + 1] = P[1]; tmp[3 * off + 2] = P[2]; dep[off] = dqv; }
        m = __popc(km);
        if (m == 0) return 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // every clipped vertex behind the reference face is a contact, in polygon order (at most 8): lane x writes contact x
    const int nk = m < BOX_MAXC ? m : BOX_MAXC;
    if (on && t < nk) {
        const int q = t;
        T l[3] = {tmp[3 * q], tmp[3 * q + 1], tmp[3 * q + 2]};
        const T dq_ = dep[q];
#pragma unroll
        for (int j = 0; j < 3; j++) l[j] += (j == ax) ? T(0.5) * dq_ * face : T(0);
        T wv[3];
        mulmat(rmat, l, wv);
        if (t < BOX_SLOTC) {
            for (int c = 0; c < 3; c++) scr[SLOT_P + 3 * t + c] = wv[c] + rpos[c];
            scr[t] = -dq_;
        } else {
            ovf[4 * (t - BOX_SLOTC)] = -dq_;
            for (int c = 0; c < 3; c++) ovf[4 * (t - BOX_SLOTC) + 1 + c] = wv[c] + rpos[c];
        }
    }
    if (on && t == 0) { for (int c = 0; c < 3; c++) scr[SLOT_N + c] = n[c]; }
    (void)rowmask_sh;
    return nk;
}

// conservative cull: separating-axis test of the two local bounding boxes on their 6 face axes
template <typename T>
AVS_DEV bool boxes_separated(const Shape<T>& a, const Shape<T>& b) {
    T ca[3], cb[3], t[3];
    mulmat(a.mat, a.lc, ca);
    mulmat(b.mat, b.lc, cb);
    for (int k = 0; k < 3; k++) t[k] = (b.pos[k] + cb[k]) - (a.pos[k] + ca[k]);
    T R[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i][j] = fabs(a.mat[i] * b.mat[j] + a.mat[3 + i] * b.mat[3 + j] + a.mat[6 + i] * b.mat[6 + j]) + T(1e-6);
    for (int i = 0; i < 3; i++) {
        T ta = fabs(t[0] * a.mat[i] + t[1] * a.mat[3 + i] + t[2] * a.mat[6 + i]);
        if (ta > a.lh[i] + b.lh[0] * R[i][0] + b.lh[1] * R[i][1] + b.lh[2] * R[i][2]) return true;
    }
    for (int j = 0; j < 3; j++) {
        T tb = fabs(t[0] * b.mat[j] + t[1] * b.mat[3 + j] + t[2] * b.mat[6 + j]);
        if (tb > b.lh[j] + a.lh[0] * R[0][j] + a.lh[1] * R[1][j] + a.lh[2] * R[2][j]) return true;
    }
    return false;
}

// dispatch for every pair that is not box-box (those go through box_box with their own work area); at most one contact,
// written to the lane's result slot: dist [0], pos [SLOT_P, +3), normal [SLOT_N, +3)
template <typename T>
__device__ int narrow(const Shape<T>& a, const Shape<T>& b, AVS_LDS(T) scr) {
    int ta = a.type, tb = b.type;
    if (boxes_separated(a, b)) return 0;
    T dist[1], pos[3], nrm[3];
    int n;
    if (ta == G_SPHERE && tb == G_SPHERE) n = sphere_sphere(a, b, dist, pos, nrm);
    else if (ta == G_SPHERE && tb == G_BOX) n = sphere_box(a, b, dist, pos, nrm);
    else if (ta == G_BOX && tb == G_SPHERE) {
        n = sphere_box(b, a, dist, pos, nrm);
        nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2];
    } else {
        T depth;
        n = mpr_penetration(a, b, &depth, nrm, pos);
        dist[0] = -depth;
    }
    if (n) {
        scr[0] = dist[0];
        for (int k = 0; k < 3; k++) { scr[SLOT_P + k] = pos[k]; scr[SLOT_N + k] = nrm[k]; }
    }
    return n;
}

// mju_makeFrame [EXT]: tangents from the normal
template <typename T>
AVS_DEV void make_frame(const T* n, T* t1, T* t2) {
    t1[0] = t1[1] = t1[2] = 0;
    if (n[1] < T(0.5) && n[1] > T(-0.5)) t1[1] = 1; else t1[2] = 1;
    T d = dot3(n, t1);
    t1[0] -= d * n[0]; t1[1] -= d * n[1]; t1[2] -= d * n[2];
    normalize3(t1);
    cross3(n, t1, t2);
}

}  // namespace avs
