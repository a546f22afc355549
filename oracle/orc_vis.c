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

/* out u8[H][W][3]; tri_out (optional) int[H][W]: index of the triangle seen, -1 for the sky; depth_out (optional) double[H][W] */
int orc_vis_render(const orc_data* d, int cam, int nvert, const double* vert, const int* vbody, int ntri, const int* tri, const double* rgb,
                   const double* uv, const int* tex, const int* texel, int texn, int H, int W, unsigned char* out, int* tri_out, double* depth_out) {
    const orc_model* m = d->m;
    if (cam < 0 || cam >= m->ncam || !m->render_light) return -1;
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
    double lw[3] = {L[4], L[5], L[6]}, ln = sqrt(lw[0] * lw[0] + lw[1] * lw[1] + lw[2] * lw[2]), lc[3], up[3];
    for (int j = 0; j < 3; j++) {
        lc[j] = (Rc[j] * lw[0] + Rc[3 + j] * lw[1] + Rc[6 + j] * lw[2]) / ln;     /* light direction, world up: camera frame */
        up[j] = Rc[6 + j];
    }
    /* vertices into the camera frame */
    double* vc = (double*)malloc(sizeof(double) * 3 * (size_t)nvert);
    if (!vc) return -2;
    for (int v = 0; v < nvert; v++) {
        const double *R = d->xmat + 9 * vbody[v], *p = d->xpos + 3 * vbody[v], *x = vert + 3 * v;
        double w[3];
        for (int i = 0; i < 3; i++) w[i] = R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2] + p[i] - pc[i];
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
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const double dir[3] = {(j + 0.5 - 0.5 * W) * scale, -(i + 0.5 - 0.5 * H) * scale, -1.0};
            double best = 1e300, bu = 0, bv = 0;
            int bt = -1;
            for (int t = 0; t < ntri; t++) {
                const double *a = vc + 3 * tri[3 * t], *bb = vc + 3 * tri[3 * t + 1], *c = vc + 3 * tri[3 * t + 2];
                const double e1[3] = {bb[0] - a[0], bb[1] - a[1], bb[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
                const double h[3] = {dir[1] * e2[2] - dir[2] * e2[1], dir[2] * e2[0] - dir[0] * e2[2], dir[0] * e2[1] - dir[1] * e2[0]};
                const double det = e1[0] * h[0] + e1[1] * h[1] + e1[2] * h[2];
                if (fabs(det) < 1e-300) continue;
                const double s[3] = {-a[0], -a[1], -a[2]};
                const double u = (s[0] * h[0] + s[1] * h[1] + s[2] * h[2]) / det;
                if (u < 0 || u > 1) continue;
                const double q[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
                const double v = (dir[0] * q[0] + dir[1] * q[1] + dir[2] * q[2]) / det;
                if (v < 0 || u + v > 1) continue;
                const double tt = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) / det;      /* = depth along the optical axis (dir z = -1) */
                if (!front[t]) continue;                                                     /* back faces are culled (as MuJoCo's renderer does [EXT]) */
                if (tt >= znear && tt < best) { best = tt; bt = t; bu = u; bv = v; }
            }
            double col[3];
            if (bt >= 0) {
                hits++;
                const double *a = vc + 3 * tri[3 * bt], *bb = vc + 3 * tri[3 * bt + 1], *c = vc + 3 * tri[3 * bt + 2];
                double n[3] = {(bb[1] - a[1]) * (c[2] - a[2]) - (bb[2] - a[2]) * (c[1] - a[1]), (bb[2] - a[2]) * (c[0] - a[0]) - (bb[0] - a[0]) * (c[2] - a[2]),
                               (bb[0] - a[0]) * (c[1] - a[1]) - (bb[1] - a[1]) * (c[0] - a[0])};
                const double g[3] = {(a[0] + bb[0] + c[0]) / 3, (a[1] + bb[1] + c[1]) / 3, (a[2] + bb[2] + c[2]) / 3};
                const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), gg = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
                const double ch = -(n[0] * g[0] + n[1] * g[1] + n[2] * g[2]) / (nn * gg);     /* headlight term at the centroid: flat per triangle */
                const double cl = -(n[0] * lc[0] + n[1] * lc[1] + n[2] * lc[2]) / nn;
                double lum = amb + hd * ch + ld * (cl > 0 ? cl : 0);
                if (lum > 1) lum = 1;
                if (tex[bt]) {
                    const double* w = uv + 6 * bt;
                    const double tu = w[0] + bu * (w[2] - w[0]) + bv * (w[4] - w[0]), tv = w[1] + bu * (w[3] - w[1]) + bv * (w[5] - w[1]);
                    const double fu = tu - floor(tu), fv = tv - floor(tv);
                    int xi = (int)(fu * texn), yi = (int)((1.0 - fv) * texn);
                    xi = xi > texn - 1 ? texn - 1 : xi; yi = yi > texn - 1 ? texn - 1 : yi;
                    const unsigned px = (unsigned)texel[yi * texn + xi];
                    col[0] = (px & 255u) / 255.0 * lum; col[1] = ((px >> 8) & 255u) / 255.0 * lum; col[2] = ((px >> 16) & 255u) / 255.0 * lum;
                } else
                    for (int k = 0; k < 3; k++) col[k] = rgb[3 * bt + k] * lum;
            } else {
                const double idn = 1.0 / sqrt(dir[0] * dir[0] + dir[1] * dir[1] + 1.0);
                const double w = 0.5 + 0.5 * (up[0] * dir[0] + up[1] * dir[1] - up[2]) * idn;
                for (int k = 0; k < 3; k++) col[k] = L[12 + k] + (L[8 + k] - L[12 + k]) * w;
            }
            for (int k = 0; k < 3; k++) out[((size_t)i * W + j) * 3 + k] = vq8(col[k]);
            if (tri_out) tri_out[(size_t)i * W + j] = bt;
            if (depth_out) depth_out[(size_t)i * W + j] = bt >= 0 ? best : 0.0;
        }
    free(vc);
    free(front);
    return hits;
}
