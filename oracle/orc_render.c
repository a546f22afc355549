/* oracle/orc_render.c -- depth ray-caster over the model's visible geoms (f64 arithmetic, brute force over all geoms).
 * TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * Stands where the reference renders its cameras (gym_guided_vision/gym_guided_vision/env.py:180-188 get_obs pixels,
 * :195-200 render; MuJoCo OpenGL renderer [EXT]).  BASELINE config 5 / SURVEY 8d replace the RGB render by a depth image
 * of the same cameras: float32 metres along the optical axis.  Conventions follow MuJoCo's camera model [EXT]: the camera
 * looks along its -z, +y is up, fovy is the vertical field of view, near/far planes znear*extent / zfar*extent
 * (scene.xml:6,13), back faces are culled (a camera inside a convex geom does not see it).  What is drawn are the
 * collision proxies (boxes, spheres, cylinders, decimated convex hulls), not the visual meshes: PARITY UNPINNED against the
 * reference's pixels; pinned by analytic known answers (tests/test_oracle_render.py). */
#include <math.h>
#include <string.h>

#include "orc.h"

static void quat2mat(const double* q, double* R) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}

/* parametric interval [*t0, *t1] of the ray o + t v inside convex geom g (local frame); returns 0 when empty */
static int ray_interval(const orc_model* m, int g, const double* o, const double* v, double* t0, double* t1) {
    const double* sz = m->geom_size + 3 * g;
    double lo = -1e30, hi = 1e30;
    switch (m->geom_type[g]) {
    case ORC_SPHERE: {
        double a = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], b = o[0] * v[0] + o[1] * v[1] + o[2] * v[2];
        double c = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] - sz[0] * sz[0], disc = b * b - a * c;
        if (disc < 0) return 0;
        double s = sqrt(disc);
        lo = (-b - s) / a; hi = (-b + s) / a;
        break;
    }
    case ORC_BOX:
        for (int k = 0; k < 3; k++) {
            if (v[k] == 0) { if (fabs(o[k]) > sz[k]) return 0; continue; }
            double ta = (-sz[k] - o[k]) / v[k], tb = (sz[k] - o[k]) / v[k];
            if (ta > tb) { double t = ta; ta = tb; tb = t; }
            if (ta > lo) lo = ta;
            if (tb < hi) hi = tb;
        }
        break;
    case ORC_CYLINDER: { /* axis z, radius sz[0], half height sz[1] */
        double a = v[0] * v[0] + v[1] * v[1], b = o[0] * v[0] + o[1] * v[1], c = o[0] * o[0] + o[1] * o[1] - sz[0] * sz[0];
        if (a > 0) {
            double disc = b * b - a * c;
            if (disc < 0) return 0;
            double s = sqrt(disc);
            lo = (-b - s) / a; hi = (-b + s) / a;
        } else if (c > 0) return 0;
        if (v[2] == 0) { if (fabs(o[2]) > sz[1]) return 0; }
        else {
            double ta = (-sz[1] - o[2]) / v[2], tb = (sz[1] - o[2]) / v[2];
            if (ta > tb) { double t = ta; ta = tb; tb = t; }
            if (ta > lo) lo = ta;
            if (tb < hi) hi = tb;
        }
        break;
    }
    case ORC_MESH: {
        const double* P = m->hull_plane + 4 * m->geom_hplane[2 * g];
        int np = m->geom_hplane[2 * g + 1];
        for (int k = 0; k < np; k++) {
            const double* n = P + 4 * k;
            double nv = n[0] * v[0] + n[1] * v[1] + n[2] * v[2], no = n[3] - (n[0] * o[0] + n[1] * o[1] + n[2] * o[2]);
            if (nv == 0) { if (no < 0) return 0; continue; }
            double t = no / nv;
            if (nv < 0) { if (t > lo) lo = t; } else { if (t < hi) hi = t; }
        }
        break;
    }
    default: return 0;
    }
    if (lo > hi) return 0;
    *t0 = lo; *t1 = hi;
    return 1;
}

/* rows row0, row0 + row_step, ... (nrows of them) of the H x W depth image of camera `cam`: out = float[nrows][W].  The pixel rays do
 * not depend on which rows are asked for, so a strided subset of a 480 x 640 image is checked at a fraction of the cost. */
int orc_render_depth_rows(const orc_data* d, int cam, int H, int W, int row0, int row_step, int nrows, float* out) {
    const orc_model* m = d->m;
    if (cam < 0 || cam >= m->ncam) return -1;
    int b = m->cam_body[cam];
    double Rl[9], Rc[9], pc[3];
    quat2mat(m->cam_quat + 4 * cam, Rl);
    const double *Rb = d->xmat + 9 * b, *pb = d->xpos + 3 * b, *cp = m->cam_pos + 3 * cam;
    for (int i = 0; i < 3; i++) {
        pc[i] = pb[i] + Rb[3 * i] * cp[0] + Rb[3 * i + 1] * cp[1] + Rb[3 * i + 2] * cp[2];
        for (int j = 0; j < 3; j++) Rc[3 * i + j] = Rb[3 * i] * Rl[j] + Rb[3 * i + 1] * Rl[3 + j] + Rb[3 * i + 2] * Rl[6 + j];
    }
    const double znear = m->cam_clip[0], zfar = m->cam_clip[1];
    const double scale = 2.0 * tan(0.5 * m->cam_fovy[cam] * 3.14159265358979323846 / 180.0) / H;
    /* per geom: camera position and camera axes in the geom's frame */
    int hits = 0;
    if (row_step < 1 || row0 < 0 || nrows < 0 || (nrows > 0 && row0 + (nrows - 1) * row_step >= H)) return -1;
    for (int ri = 0; ri < nrows; ri++)
        for (int j = 0; j < W; j++) {
            const int i = row0 + ri * row_step;
            double dc[3] = {(j + 0.5 - 0.5 * W) * scale, -(i + 0.5 - 0.5 * H) * scale, -1.0}, dw[3];
            for (int k = 0; k < 3; k++) dw[k] = Rc[3 * k] * dc[0] + Rc[3 * k + 1] * dc[1] + Rc[3 * k + 2] * dc[2];
            double best = zfar;
            for (int g = 0; g < m->ngeom; g++) {
                if (!m->geom_visible[g]) continue;
                const double *Rg = d->geom_xmat + 9 * g, *pg = d->geom_xpos + 3 * g;
                double o[3], v[3], rel[3] = {pc[0] - pg[0], pc[1] - pg[1], pc[2] - pg[2]};
                for (int k = 0; k < 3; k++) {
                    o[k] = Rg[k] * rel[0] + Rg[3 + k] * rel[1] + Rg[6 + k] * rel[2];
                    v[k] = Rg[k] * dw[0] + Rg[3 + k] * dw[1] + Rg[6 + k] * dw[2];
                }
                double t0, t1;
                if (!ray_interval(m, g, o, v, &t0, &t1)) continue;
                if (t0 >= znear && t0 < best) best = t0;   /* dc has z = -1: t is the distance along the optical axis */
            }
            if (best < zfar) hits++;
            out[(size_t)ri * W + j] = (float)best;
        }
    return hits;
}

int orc_render_depth(const orc_data* d, int cam, int H, int W, float* out) { return orc_render_depth_rows(d, cam, H, W, 0, 1, H, out); }

/* outward unit normal (geom frame) of convex geom g where the ray o + t v enters it at parameter t0 */
static void entry_normal(const orc_model* m, int g, const double* o, const double* v, double t0, double* n) {
    const double* sz = m->geom_size + 3 * g;
    double p[3] = {o[0] + t0 * v[0], o[1] + t0 * v[1], o[2] + t0 * v[2]};
    n[0] = n[1] = n[2] = 0;
    switch (m->geom_type[g]) {
    case ORC_SPHERE: {
        double r = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
        for (int k = 0; k < 3; k++) n[k] = p[k] / r;
        break;
    }
    case ORC_BOX: { /* the face whose slab the entry point lies on: largest |p_k| / size_k */
        int a = 0;
        double best = -1;
        for (int k = 0; k < 3; k++) { double q = fabs(p[k]) / sz[k]; if (q > best) { best = q; a = k; } }
        n[a] = p[a] > 0 ? 1 : -1;
        break;
    }
    case ORC_CYLINDER: { /* cap when the entry point is nearer (relatively) to a cap than to the side */
        double rr = sqrt(p[0] * p[0] + p[1] * p[1]) / sz[0], hh = fabs(p[2]) / sz[1];
        if (hh > rr) n[2] = p[2] > 0 ? 1 : -1;
        else { double r = sqrt(p[0] * p[0] + p[1] * p[1]); n[0] = p[0] / r; n[1] = p[1] / r; }
        break;
    }
    case ORC_MESH: { /* the entry face: largest crossing among the faces the ray approaches from outside */
        const double* P = m->hull_plane + 4 * m->geom_hplane[2 * g];
        int np = m->geom_hplane[2 * g + 1];
        double lo = -1e30;
        for (int k = 0; k < np; k++) {
            const double* q = P + 4 * k;
            double nv = q[0] * v[0] + q[1] * v[1] + q[2] * v[2], no = q[3] - (q[0] * o[0] + q[1] * o[1] + q[2] * o[2]);
            if (nv < 0 && no / nv > lo) { lo = no / nv; n[0] = q[0]; n[1] = q[1]; n[2] = q[2]; }
        }
        break;
    }
    default: break;
    }
}

/* Colour image u8[H][W][3] of camera `cam` (env.py:180-188 pixels, :195-200 render; MuJoCo OpenGL [EXT]): the nearest
 * proxy surface of orc_render_depth, flat material colour (geom_rgba), Lambert terms of the scene's headlight
 * (scene.xml:9: ambient 0.3, diffuse 0.6, at the camera, along each pixel's ray) and of its directional light
 * (scene.xml:48, default diffuse 0.7), sum clamped to 1; rays that hit nothing show the skybox gradient (scene.xml:34)
 * by the ray's elevation.  No textures, shadows, specular terms, haze or transparency: PARITY UNPINNED against the
 * reference's pixels.  depth (optional) receives the same values as orc_render_depth. */
int orc_render_rgb(const orc_data* d, int cam, int H, int W, unsigned char* out, float* depth) {
    const orc_model* m = d->m;
    if (cam < 0 || cam >= m->ncam || !m->geom_rgba || !m->render_light) return -1;
    int b = m->cam_body[cam];
    double Rl[9], Rc[9], pc[3];
    quat2mat(m->cam_quat + 4 * cam, Rl);
    const double *Rb = d->xmat + 9 * b, *pb = d->xpos + 3 * b, *cp = m->cam_pos + 3 * cam;
    for (int i = 0; i < 3; i++) {
        pc[i] = pb[i] + Rb[3 * i] * cp[0] + Rb[3 * i + 1] * cp[1] + Rb[3 * i + 2] * cp[2];
        for (int j = 0; j < 3; j++) Rc[3 * i + j] = Rb[3 * i] * Rl[j] + Rb[3 * i + 1] * Rl[3 + j] + Rb[3 * i + 2] * Rl[6 + j];
    }
    const double znear = m->cam_clip[0], zfar = m->cam_clip[1];
    const double scale = 2.0 * tan(0.5 * m->cam_fovy[cam] * 3.14159265358979323846 / 180.0) / H;
    const double *L = m->render_light, amb = L[0], hd = L[1], ld = L[2];
    double ldir[3] = {L[4], L[5], L[6]}, ln = sqrt(ldir[0] * ldir[0] + ldir[1] * ldir[1] + ldir[2] * ldir[2]);
    for (int k = 0; k < 3; k++) ldir[k] /= ln;
    int hits = 0;
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            double dc[3] = {(j + 0.5 - 0.5 * W) * scale, -(i + 0.5 - 0.5 * H) * scale, -1.0}, dw[3];
            for (int k = 0; k < 3; k++) dw[k] = Rc[3 * k] * dc[0] + Rc[3 * k + 1] * dc[1] + Rc[3 * k + 2] * dc[2];
            double best = zfar, nw[3] = {0, 0, 0};
            int bg = -1;
            for (int g = 0; g < m->ngeom; g++) {
                if (!m->geom_visible[g]) continue;
                const double *Rg = d->geom_xmat + 9 * g, *pg = d->geom_xpos + 3 * g;
                double o[3], v[3], rel[3] = {pc[0] - pg[0], pc[1] - pg[1], pc[2] - pg[2]};
                for (int k = 0; k < 3; k++) {
                    o[k] = Rg[k] * rel[0] + Rg[3 + k] * rel[1] + Rg[6 + k] * rel[2];
                    v[k] = Rg[k] * dw[0] + Rg[3 + k] * dw[1] + Rg[6 + k] * dw[2];
                }
                double t0, t1;
                if (!ray_interval(m, g, o, v, &t0, &t1)) continue;
                if (t0 >= znear && t0 < best) {
                    double n[3];
                    best = t0; bg = g;
                    entry_normal(m, g, o, v, t0, n);
                    for (int k = 0; k < 3; k++) nw[k] = Rg[3 * k] * n[0] + Rg[3 * k + 1] * n[1] + Rg[3 * k + 2] * n[2];
                }
            }
            double dn = sqrt(dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2]), rgb[3];
            if (bg >= 0) {
                double ch = -(nw[0] * dw[0] + nw[1] * dw[1] + nw[2] * dw[2]) / dn, cl = -(nw[0] * ldir[0] + nw[1] * ldir[1] + nw[2] * ldir[2]);
                double lum = amb + hd * (ch > 0 ? ch : 0) + ld * (cl > 0 ? cl : 0);
                if (lum > 1) lum = 1;
                for (int k = 0; k < 3; k++) rgb[k] = m->geom_rgba[4 * bg + k] * lum;
                hits++;
            } else {
                double w = 0.5 + 0.5 * dw[2] / dn;
                for (int k = 0; k < 3; k++) rgb[k] = L[12 + k] + (L[8 + k] - L[12 + k]) * w;
            }
            for (int k = 0; k < 3; k++) {
                double c = rgb[k] < 0 ? 0 : (rgb[k] > 1 ? 1 : rgb[k]);
                out[((size_t)i * W + j) * 3 + k] = (unsigned char)(c * 255.0 + 0.5);
            }
            if (depth) depth[(size_t)i * W + j] = (float)best;
        }
    return hits;
}
