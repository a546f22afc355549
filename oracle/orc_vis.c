/* orc_vis.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see orc.h) of the colour image of the visual meshes.
 *
 * Stands where the reference draws its cameras through MuJoCo's OpenGL renderer (gym_guided_vision/gym_guided_vision/env.py:180-188
 * get_obs "pixels", :195-200 render [EXT]); like the device's rasteriser (av_aloha_amd/csrc/avsim_vis.hip.h) it draws the decimated
 * visual scene of compiler/vismesh.py with flat Lambert shading, and PARITY with the reference's OpenGL pixels is UNPINNED.
 * What it checks is the device's projection / clipping / binning / depth test: here every pixel's ray is intersected with every
 * triangle that faces the camera (Moeller-Trumbore, f64, depth along the optical axis >= znear), no projection and no tiles.
 *
 * The caller passes the expanded scene (tests: vismesh.expand_instances) and the body poses of an orc_data. */
#include <math.h>
#include <stdlib.h>

#include "orc.h"

static void vquat2mat(const double* q, double* R) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}

static unsigned char vq8(double x) { x = x < 0 ? 0 : (x > 1 ? 1 : x); return (unsigned char)(x * 255.0 + 0.5); }

/* Light-space frame of the directional light: e1, e2 span the plane normal to the (unit) light direction lw.  The device builds its shadow
 * map in the same frame (avsim_vis.hip.h vis_light_frame). */
static void light_frame(const double* lw, double* e1, double* e2) {
    const double ax[3] = {fabs(lw[0]) < 0.9 ? 1.0 : 0.0, fabs(lw[0]) < 0.9 ? 0.0 : 1.0, 0.0};
    double d = ax[0] * lw[0] + ax[1] * lw[1] + ax[2] * lw[2];
    for (int k = 0; k < 3; k++) e1[k] = ax[k] - d * lw[k];
    d = sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    for (int k = 0; k < 3; k++) e1[k] /= d;
    e2[0] = lw[1] * e1[2] - lw[2] * e1[1]; e2[1] = lw[2] * e1[0] - lw[0] * e1[2]; e2[2] = lw[0] * e1[1] - lw[1] * e1[0];
}

/* out u8[H][W][3]; tri_out (optional) int[H][W]: index of the triangle seen, -1 for the sky; depth_out (optional) double[H][W].
 * orc_vis_render_ex: ss = 1 | 2 samples per pixel and axis (2: the four samples at +-1/4 pixel of the centre are resolved and averaged; the
 * samples of a pixel that see the same triangle share the colour of the first of them -- MuJoCo's offscreen buffer is multisampled,
 * <quality offsamples> default 4 [EXT]); shadows != 0: a surface point that faces the scene's
 * directional light (scene.xml:48, castshadow default true [EXT]) and lies inside the light's shadow box (centre and half extent from
 * <statistic center extent>, scene.xml:6: render_light[7], [11], [15] and [3]) loses the light's diffuse term when another triangle lies
 * between it and the light -- one exact ray per sample where the device looks up a depth map rendered from the light.  tri_out /
 * depth_out hold the pixel-centre ray's answers. */
/* tnorm (orc_vis_render_sm; NULL = flat): per triangle the three corners' lighting normals in the body frame (compiler/vismesh.py corner_normals).  With
 * them every corner is lit with its own normal -- headlight term along the ray to the corner, the directional light's diffuse and specular terms, each
 * clamped as fixed-function GL clamps a vertex colour [EXT] -- and the three terms (shade, shade without the light's part, specular) are interpolated with
 * the hit point's barycentrics, which are perspective-correct by construction here. */
static int vis_render(const orc_data* d, int cam, int nvert, const double* vert, const int* vbody, int ntri, const int* tri, const double* rgb,
                      const double* uv, const int* tex, const int* texel, int texn, int H, int W, int ss, int shadows, const double* tnorm, unsigned char* out, int* tri_out, double* depth_out) {
    const orc_model* m = d->m;
    if (cam < 0 || cam >= m->ncam || !m->render_light || (ss != 1 && ss != 2)) return -1;
    int b = m->cam_body[cam];
    double Rl[9], Rc[9], pc[3];
    vquat2mat(m->cam_quat + 4 * cam, Rl);
    const double *Rb = d->xmat + 9 * b, *pb = d->xpos + 3 * b, *cp = m->cam_pos + 3 * cam;
    for (int i = 0; i < 3; i++) {
        pc[i] = pb[i] + Rb[3 * i] * cp[0] + Rb[3 * i + 1] * cp[1] + Rb[3 * i + 2] * cp[2];
        for (int j = 0; j < 3; j++) Rc[3 * i + j] = Rb[3 * i] * Rl[j] + Rb[3 * i + 1] * Rl[3 + j] + Rb[3 * i + 2] * Rl[6 + j];
    }
    const double znear = m->cam_clip[0];
    const double scale = 2.0 * tan(0.5 * m->cam_fovy[cam] * 3.14159265358979323846 / 180.0) / H;
    const double *L = m->render_light, amb = L[0], hd = L[1], ld = L[2];
    double lw[3] = {L[4], L[5], L[6]}, ln = sqrt(lw[0] * lw[0] + lw[1] * lw[1] + lw[2] * lw[2]), lc[3], up[3], e1[3], e2[3];
    for (int j = 0; j < 3; j++) lw[j] /= ln;
    for (int j = 0; j < 3; j++) {
        lc[j] = Rc[j] * lw[0] + Rc[3 + j] * lw[1] + Rc[6 + j] * lw[2];     /* light direction, world up: camera frame */
        up[j] = Rc[6 + j];
    }
    light_frame(lw, e1, e2);
    const double sh_half = shadows ? L[3] : 0.0, sh_c[3] = {L[7], L[11], L[15]};
    const double sh_s = sh_c[0] * e1[0] + sh_c[1] * e1[1] + sh_c[2] * e1[2], sh_t = sh_c[0] * e2[0] + sh_c[1] * e2[1] + sh_c[2] * e2[2];
    /* vertices into the camera frame (and the world frame, for the shadow rays) */
    double* vc = (double*)malloc(sizeof(double) * 6 * (size_t)nvert);
    if (!vc) return -2;
    double* vw = vc + 3 * (size_t)nvert;
    for (int v = 0; v < nvert; v++) {
        const double *R = d->xmat + 9 * vbody[v], *p = d->xpos + 3 * vbody[v], *x = vert + 3 * v;
        double w[3];
        for (int i = 0; i < 3; i++) { vw[3 * v + i] = R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2] + p[i]; w[i] = vw[3 * v + i] - pc[i]; }
        for (int j = 0; j < 3; j++) vc[3 * v + j] = Rc[j] * w[0] + Rc[3 + j] * w[1] + Rc[6 + j] * w[2];
    }
    /* a triangle faces the camera when its outward normal points against the direction from the eye to its centroid */
    unsigned char* front = (unsigned char*)malloc((size_t)ntri + 1);
    if (!front) { free(vc); return -2; }
    for (int t = 0; t < ntri; t++) {
        const double *a = vc + 3 * tri[3 * t], *bb = vc + 3 * tri[3 * t + 1], *c = vc + 3 * tri[3 * t + 2];
        const double n[3] = {(bb[1] - a[1]) * (c[2] - a[2]) - (bb[2] - a[2]) * (c[1] - a[1]), (bb[2] - a[2]) * (c[0] - a[0]) - (bb[0] - a[0]) * (c[2] - a[2]),
                             (bb[0] - a[0]) * (c[1] - a[1]) - (bb[1] - a[1]) * (c[0] - a[0])};
        front[t] = n[0] * (a[0] + bb[0] + c[0]) + n[1] * (a[1] + bb[1] + c[1]) + n[2] * (a[2] + bb[2] + c[2]) < 0;
    }
    int hits = 0;
    const int ns = ss * ss;
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            double acc[3] = {0, 0, 0};
            int sid[4] = {-2, -2, -2, -2};            /* what the samples shaded so far saw (-1: the sky) and their quantised colours */
            double scol[4][3];
            for (int smp = -1; smp < ns; smp++) {
                /* smp = -1: the pixel centre (tri_out / depth_out; the colour too when ss = 1); 0 .. 3: the samples at +-1/4 pixel */
                if ((ss == 1) != (smp == -1)) { if (!(smp == -1 && (tri_out || depth_out))) continue; }
                const double ox = smp < 0 ? 0.0 : ((smp & 1) ? 0.25 : -0.25), oy = smp < 0 ? 0.0 : ((smp & 2) ? 0.25 : -0.25);
                const double dir[3] = {(j + 0.5 + ox - 0.5 * W) * scale, -(i + 0.5 + oy - 0.5 * H) * scale, -1.0};
                double best = 1e300, bu = 0, bv = 0;
                int bt = -1;
                for (int t = 0; t < ntri; t++) {
                    const double *a = vc + 3 * tri[3 * t], *bb = vc + 3 * tri[3 * t + 1], *c = vc + 3 * tri[3 * t + 2];
                    const double e1_[3] = {bb[0] - a[0], bb[1] - a[1], bb[2] - a[2]}, e2_[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
                    const double h[3] = {dir[1] * e2_[2] - dir[2] * e2_[1], dir[2] * e2_[0] - dir[0] * e2_[2], dir[0] * e2_[1] - dir[1] * e2_[0]};
                    const double det = e1_[0] * h[0] + e1_[1] * h[1] + e1_[2] * h[2];
                    if (fabs(det) < 1e-300) continue;
                    const double s_[3] = {-a[0], -a[1], -a[2]};
                    const double u = (s_[0] * h[0] + s_[1] * h[1] + s_[2] * h[2]) / det;
                    if (u < 0 || u > 1) continue;
                    const double q[3] = {s_[1] * e1_[2] - s_[2] * e1_[1], s_[2] * e1_[0] - s_[0] * e1_[2], s_[0] * e1_[1] - s_[1] * e1_[0]};
                    const double v = (dir[0] * q[0] + dir[1] * q[1] + dir[2] * q[2]) / det;
                    if (v < 0 || u + v > 1) continue;
                    const double tt = (e2_[0] * q[0] + e2_[1] * q[1] + e2_[2] * q[2]) / det;      /* = depth along the optical axis (dir z = -1) */
                    if (!front[t]) continue;                                                     /* back faces are culled (as MuJoCo's renderer does [EXT]) */
                    if (tt >= znear && tt < best) { best = tt; bt = t; bu = u; bv = v; }
                }
                if (smp == -1) {
                    if (tri_out) tri_out[(size_t)i * W + j] = bt;
                    if (depth_out) depth_out[(size_t)i * W + j] = bt >= 0 ? best : 0.0;
                    if (bt >= 0) hits++;
                    if (ss != 1) continue;
                }
                double col[3];
                /* a fragment -- the samples of the pixel that one triangle wins, or that see the sky -- is shaded once, at its first sample, as a
                 * multisampled GL buffer does [EXT]: a later sample of the same fragment takes that colour */
                int prev = -1;
                for (int s2 = 0; s2 < smp; s2++)
                    if (sid[s2] == bt && prev < 0) prev = s2;
                if (smp >= 0 && prev >= 0) {
                    sid[smp] = bt;
                    for (int k = 0; k < 3; k++) { scol[smp][k] = scol[prev][k]; acc[k] += scol[prev][k]; }
                    continue;
                }
                if (bt >= 0) {
                    const double *a = vc + 3 * tri[3 * bt], *bb = vc + 3 * tri[3 * bt + 1], *c = vc + 3 * tri[3 * bt + 2];
                    double n[3] = {(bb[1] - a[1]) * (c[2] - a[2]) - (bb[2] - a[2]) * (c[1] - a[1]), (bb[2] - a[2]) * (c[0] - a[0]) - (bb[0] - a[0]) * (c[2] - a[2]),
                                   (bb[0] - a[0]) * (c[1] - a[1]) - (bb[1] - a[1]) * (c[0] - a[0])};
                    const double g[3] = {(a[0] + bb[0] + c[0]) / 3, (a[1] + bb[1] + c[1]) / 3, (a[2] + bb[2] + c[2]) / 3};
                    const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), gg = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
                    const double ch = -(n[0] * g[0] + n[1] * g[1] + n[2] * g[2]) / (nn * gg);     /* headlight term at the centroid: flat per triangle */
                    double cl = -(n[0] * lc[0] + n[1] * lc[1] + n[2] * lc[2]) / nn;
                    if (cl > 0 && sh_half > 0) {
                        /* the sample's surface point in the world, a ray from it towards the light */
                        const double P[3] = {dir[0] * best, dir[1] * best, dir[2] * best};
                        double pw[3];
                        for (int k = 0; k < 3; k++) pw[k] = pc[k] + Rc[3 * k] * P[0] + Rc[3 * k + 1] * P[1] + Rc[3 * k + 2] * P[2];
                        const double ps = pw[0] * e1[0] + pw[1] * e1[1] + pw[2] * e1[2], pt = pw[0] * e2[0] + pw[1] * e2[1] + pw[2] * e2[2];
                        if (fabs(ps - sh_s) < sh_half && fabs(pt - sh_t) < sh_half) {
                            const double rd[3] = {-lw[0], -lw[1], -lw[2]};
                            for (int t = 0; t < ntri; t++) {
                                if (t == bt) continue;
                                const double *A = vw + 3 * tri[3 * t], *B = vw + 3 * tri[3 * t + 1], *C = vw + 3 * tri[3 * t + 2];
                                const double f1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, f2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
                                const double h[3] = {rd[1] * f2[2] - rd[2] * f2[1], rd[2] * f2[0] - rd[0] * f2[2], rd[0] * f2[1] - rd[1] * f2[0]};
                                const double det = f1[0] * h[0] + f1[1] * h[1] + f1[2] * h[2];
                                if (fabs(det) < 1e-300) continue;
                                const double s_[3] = {pw[0] - A[0], pw[1] - A[1], pw[2] - A[2]};
                                const double u = (s_[0] * h[0] + s_[1] * h[1] + s_[2] * h[2]) / det;
                                if (u < 0 || u > 1) continue;
                                const double q[3] = {s_[1] * f1[2] - s_[2] * f1[1], s_[2] * f1[0] - s_[0] * f1[2], s_[0] * f1[1] - s_[1] * f1[0]};
                                const double v = (rd[0] * q[0] + rd[1] * q[1] + rd[2] * q[2]) / det;
                                if (v < 0 || u + v > 1) continue;
                                const double tt = (f2[0] * q[0] + f2[1] * q[1] + f2[2] * q[2]) / det;
                                if (tt > 1e-4) { cl = 0; break; }
                            }
                        }
                    }
                    const int in_shadow = cl == 0;      /* (set by the shadow ray above; a face turned away from the light has cl < 0) */
                    double lum = amb + hd * ch + ld * (cl > 0 ? cl : 0);
                    if (lum > 1) lum = 1;
                    /* the light's specular term (render_light[16] = light specular x material specular, [17] the exponent): Blinn's half vector with
                     * the viewer at infinity along the optical axis, as fixed-function GL has it [EXT]; white, added to every channel; gone in shadow */
                    double spec = 0;
                    if (cl > 0 && L[16] > 0) {
                        double h[3] = {-lc[0], -lc[1], -lc[2] + 1.0};
                        const double hn = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
                        if (hn > 1e-12) {
                            double sgn = ch >= 0 ? 1.0 : -1.0;          /* the normal turned towards the camera */
                            const double nh = sgn * (n[0] * h[0] + n[1] * h[1] + n[2] * h[2]) / (nn * hn);
                            if (nh > 0) spec = L[16] * pow(nh, L[17]);
                        }
                    }
                    if (tnorm) {
                        /* smooth shading: the corners' own shades, interpolated at the hit point (1 - u - v, u, v) */
                        const double* Rbd = d->xmat + 9 * vbody[tri[3 * bt]];
                        const double* P3[3] = {a, bb, c};
                        double at[3][3];
                        for (int k = 0; k < 3; k++) {
                            const double* tn = tnorm + 9 * (size_t)bt + 3 * k;
                            double nw[3], nc[3];
                            for (int i = 0; i < 3; i++) nw[i] = Rbd[3 * i] * tn[0] + Rbd[3 * i + 1] * tn[1] + Rbd[3 * i + 2] * tn[2];
                            for (int j = 0; j < 3; j++) nc[j] = Rc[j] * nw[0] + Rc[3 + j] * nw[1] + Rc[6 + j] * nw[2];
                            const double* p = P3[k];
                            const double pl = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
                            double chv = -(nc[0] * p[0] + nc[1] * p[1] + nc[2] * p[2]) / (pl > 1e-300 ? pl : 1e-300);
                            if (chv < 0) chv = 0;
                            const double clv = -(nc[0] * lc[0] + nc[1] * lc[1] + nc[2] * lc[2]);
                            at[k][0] = amb + hd * chv + ld * (clv > 0 ? clv : 0); if (at[k][0] > 1) at[k][0] = 1;
                            at[k][1] = amb + hd * chv; if (at[k][1] > 1) at[k][1] = 1;
                            at[k][2] = 0;
                            if (clv > 0 && L[16] > 0) {
                                double h[3] = {-lc[0], -lc[1], -lc[2] + 1.0};
                                const double hn = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
                                const double nh = hn > 1e-12 ? (nc[0] * h[0] + nc[1] * h[1] + nc[2] * h[2]) / hn : 0;
                                if (nh > 0) at[k][2] = L[16] * pow(nh, L[17]);
                            }
                        }
                        const double b0 = 1.0 - bu - bv;
                        lum = b0 * at[0][in_shadow ? 1 : 0] + bu * at[1][in_shadow ? 1 : 0] + bv * at[2][in_shadow ? 1 : 0];
                        spec = in_shadow ? 0.0 : b0 * at[0][2] + bu * at[1][2] + bv * at[2][2];
                        if (lum < 0) lum = 0;
                        if (lum > 1) lum = 1;
                        if (spec < 0) spec = 0;
                    }
                    if (tex[bt]) {
                        const double* w = uv + 6 * bt;
                        const double tu = w[0] + bu * (w[2] - w[0]) + bv * (w[4] - w[0]), tv = w[1] + bu * (w[3] - w[1]) + bv * (w[5] - w[1]);
                        const double fu = tu - floor(tu), fv = tv - floor(tv);
                        int xi = (int)(fu * texn), yi = (int)((1.0 - fv) * texn);
                        xi = xi > texn - 1 ? texn - 1 : xi; yi = yi > texn - 1 ? texn - 1 : yi;
                        const unsigned px = (unsigned)texel[yi * texn + xi];
                        col[0] = (px & 255u) / 255.0 * lum + spec; col[1] = ((px >> 8) & 255u) / 255.0 * lum + spec; col[2] = ((px >> 16) & 255u) / 255.0 * lum + spec;
                    } else
                        for (int k = 0; k < 3; k++) col[k] = rgb[3 * bt + k] * lum + spec;
                } else {
                    const double idn = 1.0 / sqrt(dir[0] * dir[0] + dir[1] * dir[1] + 1.0);
                    const double w = 0.5 + 0.5 * (up[0] * dir[0] + up[1] * dir[1] - up[2]) * idn;
                    for (int k = 0; k < 3; k++) col[k] = L[12 + k] + (L[8 + k] - L[12 + k]) * w;
                }
                /* (the device shades a triangle once, in rgb8, and averages those: quantise before averaging) */
                for (int k = 0; k < 3; k++) {
                    const double qv = (double)vq8(col[k]);
                    acc[k] += qv;
                    if (smp >= 0) scol[smp][k] = qv;
                }
                if (smp >= 0) sid[smp] = bt;
            }
            for (int k = 0; k < 3; k++) out[((size_t)i * W + j) * 3 + k] = (unsigned char)(acc[k] / (ss == 1 ? 1 : ns) + 0.5);
        }
    free(vc);
    free(front);
    return hits;
}

int orc_vis_render_ex(const orc_data* d, int cam, int nvert, const double* vert, const int* vbody, int ntri, const int* tri, const double* rgb,
                      const double* uv, const int* tex, const int* texel, int texn, int H, int W, int ss, int shadows, unsigned char* out, int* tri_out, double* depth_out) {
    return vis_render(d, cam, nvert, vert, vbody, ntri, tri, rgb, uv, tex, texel, texn, H, W, ss, shadows, NULL, out, tri_out, depth_out);
}

int orc_vis_render_sm(const orc_data* d, int cam, int nvert, const double* vert, const int* vbody, int ntri, const int* tri, const double* rgb,
                      const double* uv, const int* tex, const int* texel, int texn, const double* tnorm, int H, int W, int ss, int shadows, unsigned char* out, int* tri_out, double* depth_out) {
    return vis_render(d, cam, nvert, vert, vbody, ntri, tri, rgb, uv, tex, texel, texn, H, W, ss, shadows, tnorm, out, tri_out, depth_out);
}

int orc_vis_render(const orc_data* d, int cam, int nvert, const double* vert, const int* vbody, int ntri, const int* tri, const double* rgb,
                   const double* uv, const int* tex, const int* texel, int texn, int H, int W, unsigned char* out, int* tri_out, double* depth_out) {
    return orc_vis_render_ex(d, cam, nvert, vert, vbody, ntri, tri, rgb, uv, tex, texel, texn, H, W, 1, 0, out, tri_out, depth_out);
}
