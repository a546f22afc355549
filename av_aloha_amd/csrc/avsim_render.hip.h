// avsim_render.hip.h -- batched depth images of the model's cameras (BASELINE config 5).
//
// Stands where the reference renders its cameras through MuJoCo's OpenGL pipeline
// (gym_guided_vision/gym_guided_vision/env.py:180-188 get_obs "pixels", :195-200 render): one float32 depth image
// (metres along the optical axis) per env and camera.  Conventions [EXT MuJoCo camera model]: the camera looks along its
// -z, +y is up, fovy is vertical, clip planes znear*extent / zfar*extent (scene.xml:6,13), back faces culled.  The drawn
// surfaces are the collision proxies (boxes, spheres, cylinders, decimated convex hulls in half-space form), exactly what
// oracle/orc_render.c draws.
//
// Two kernels.  k_render_geoms: one wavefront per (env, camera) moves every visible geom into the camera frame, projects its
// vertices to a screen-space octagon, drops the geoms outside the view, writes every polyhedron's faces in camera-ray form, their screen
// boxes and its silhouette edges, ranks the survivors front to back and leaves one 64-byte LIST HEADER per survivor in that order
// (octagon, nearest depth, packed counts, scratch offsets, and the set of image bins that none of its silhouette edges excludes).
// k_render_depth: a block of four wavefronts per bin of 10 x 2 tiles of 32 x 16 pixels: (A) the camera's headers, one per lane, coalesced,
// octagon + bin-mask test, ordered compaction into LDS; (B) the listed polyhedra's faces -- in INVERSE-depth form: 1 / t of a pixel ray
// is affine in the pixel --, face boxes and silhouette edges into an LDS arena, two candidates per wave in flight; (C) the waves take the
// bin's tiles from an LDS counter and cast each against the list, front to back, reading LDS only: per record one face / one edge per
// lane for the tile tests, then two faces per round (v_readlane, v_pk_fma_f32 per pixel pair, v_min3 per pixel), one rcp per pixel at
// the end, one 16-byte non-temporal store per lane and row (8 lanes = one 128-byte line of the image).  HBM-write bound by
// construction: 4 B per pixel out; measured: profiles/r06_experiments.txt.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "avsim_model.h"

namespace avs {


constexpr int REC_W = 32;   // floats per geom record: o_l[3], A[9] (camera dir -> geom frame), c[3], r, size[3], type, plane adr, plane count,
                            // geom id, nearest depth, screen box xl xr yb yt (units of tan), and the extents of x + y and x - y over
                            // the projected vertices [28..31]: with the box an octagon, tight for the long thin frame bars that cross the
                            // image at an angle
#ifndef AVSIM_TILE_R
#define AVSIM_TILE_R 2
#endif
constexpr int TILE_R = AVSIM_TILE_R, NPX = 4 * TILE_R;      // rows of 4 pixels per lane: 2 rows = 32 x 16 tiles (the per-polyhedron tile tests are paid once per 512 pixels)
constexpr int TILE_W = 32, TILE_H = 8 * TILE_R;
#ifndef AVSIM_BIN_TX
#define AVSIM_BIN_TX 10
#define AVSIM_BIN_TY 2
#endif
constexpr int BIN_TX = AVSIM_BIN_TX, BIN_TY = AVSIM_BIN_TY;           // a block's bin: at most 10 x 2 tiles of 32 x 16 pixels; the width is chosen per launch so that the bin columns are equally wide (bin_width: 20 tiles of a 640-pixel row = 2 x 10; the fixed 8 left a third column of 4 with half its block idle).  Measured per 1024 envs x 4 cameras x 480 x 640 (round 3): 10 x 2 4.72 ms, 20 x 2 4.73, 20 x 1 4.86, 8 x 2 5.02, 7 x 2 5.12, 4 x 3 5.22, 5 x 2 and 10 x 1 5.47, 10 x 3 5.59, 8 x 4 6.76; 32 x 32 tiles (TILE_R 4) in 8 x 1 5.19
// bin width in tiles for an image `tiles_x` tiles wide: the smallest number of columns of at most BIN_TX tiles, equally wide
inline int bin_width(int tiles_x) { const int cols = (tiles_x + BIN_TX - 1) / BIN_TX; return (tiles_x + cols - 1) / cols; }
#ifdef AVSIM_RENDER_STATS
__device__ unsigned long long g_rstat[8];   // debug build: tiles, bin-list entries seen, box hits, records cast, entry faces, veto faces, primitives, -
#define RSTAT(i, n) do { if (lane == 0) atomicAdd(&g_rstat[i], (unsigned long long)(n)); } while (0)
#else
#define RSTAT(i, n) ((void)0)
#endif

// Polyhedra of the rasteriser: every visible mesh hull and box as vertices, face planes, the vertices of each face and the edges with
// their two faces (made on the host from the hull's vertices and planes, RenderHost::build).  rg[g][8] = vertex adr / count, plane
// adr / count, edge adr / count of geom g's polyhedron (shared by the geoms of one mesh), and the geom's own offsets into a camera's
// face (tplanes / fbox) and silhouette-edge scratch.
constexpr int RG_W = 8, RVERT_MAX = 32, RFACE_MAX = 64;
struct RenderModel {
    int ngeom, ncam, nbody, nplane, nedge;   // nplane / nedge: faces / edges summed over the visible polyhedra (one camera's scratch)
    const int *rg, *r_fvadr, *r_fvnum, *r_fvidx, *r_edge;
    const unsigned* r_fvmask;      // the vertices of a face as a bit set (polyhedra of at most RVERT_MAX = 32 vertices): one word per face
    const float *r_vert, *r_plane;
    const int *geom_type, *geom_body, *geom_visible, *cam_body;
    const float *geom_pos, *geom_mat, *geom_size, *geom_bcen, *geom_rbound, *cam_pos, *cam_mat, *cam_fovy;   // cam_fovy: tan(fovy / 2) per camera
    const float *geom_rgba, *light;   // colour render: material colours; [ambient, headlight, light, -, light dir (world) 4, sky zenith rgb 4, sky nadir rgb 4]
    float znear, zfar;
};

__device__ inline void mul33(const float* A, const float* B, float* C) {   // C = A B
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// grid (ncam_sel, N), block 64
__global__ void __launch_bounds__(64) k_render_geoms(RenderModel m, const float* __restrict__ xpose, const int* __restrict__ cam_ids, int ncam_sel,
                                                     int H, int W, float* __restrict__ recs, int* __restrict__ counts, int* __restrict__ order, float* __restrict__ tplanes,
                                                     float* __restrict__ camaux, float* __restrict__ fbox, float* __restrict__ sedge) {
    __shared__ float keys[128];
    __shared__ float vcx[RVERT_MAX][64], vcy[RVERT_MAX][64], vcz[RVERT_MAX][64];      // the lane's polyhedron in the camera frame (vertex major: no bank conflicts)
    const int lane = threadIdx.x, cs = blockIdx.x, env = blockIdx.y, cam = cam_ids[cs];
    const float* xb = xpose + (size_t)env * m.nbody * 12;
    // camera pose in the world
    float Rc[9], pc[3];
    {
        const int b = m.cam_body[cam];
        const float *pb = xb + 12 * b, *Rb = pb + 3, *cp = m.cam_pos + 3 * cam;
        mul33(Rb, m.cam_mat + 9 * cam, Rc);
#pragma unroll
        for (int i = 0; i < 3; i++) pc[i] = pb[i] + Rb[3 * i] * cp[0] + Rb[3 * i + 1] * cp[1] + Rb[3 * i + 2] * cp[2];
    }
    if (lane < 8) {   // colour render: the light's direction and the world's up axis, both in the camera frame
        const float* L = m.light + 4;
        const float il = 1.0f / sqrtf(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);
        const int j = lane & 3;
        float v = 0;
        if (j < 3) v = lane < 4 ? (Rc[j] * L[0] + Rc[3 + j] * L[1] + Rc[6 + j] * L[2]) * il : Rc[6 + j];
        camaux[((size_t)env * ncam_sel + cs) * 8 + lane] = v;
    }
    const float scale = 2.0f * m.cam_fovy[cam] / (float)H;       // cam_fovy holds tan(fovy / 2)
    const float tx = 0.5f * W * scale, ty = 0.5f * H * scale, sx = sqrtf(1 + tx * tx), sy = sqrtf(1 + ty * ty);
    float* out = recs + ((size_t)env * ncam_sel + cs) * m.ngeom * REC_W;
    int base = 0;
    for (int g0 = 0; g0 < m.ngeom; g0 += 64) {
        const int g = g0 + lane;
        bool keep = false;
        float rec[REC_W];
        if (g < m.ngeom && m.geom_visible[g]) {
            const int b = m.geom_body[g];
            const float *pb = xb + 12 * b, *Rb = pb + 3, *gp = m.geom_pos + 3 * g, *bc = m.geom_bcen + 3 * g;
            float Rg[9], pg[3], rel[3];
            mul33(Rb, m.geom_mat + 9 * g, Rg);
#pragma unroll
            for (int i = 0; i < 3; i++) { pg[i] = pb[i] + Rb[3 * i] * gp[0] + Rb[3 * i + 1] * gp[1] + Rb[3 * i + 2] * gp[2]; rel[i] = pc[i] - pg[i]; }
            // camera origin in the geom frame, and A = Rg^T Rc (camera-frame direction -> geom frame)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                rec[k] = Rg[k] * rel[0] + Rg[3 + k] * rel[1] + Rg[6 + k] * rel[2];
#pragma unroll
                for (int j = 0; j < 3; j++) rec[3 + 3 * k + j] = Rg[k] * Rc[j] + Rg[3 + k] * Rc[3 + j] + Rg[6 + k] * Rc[6 + j];
            }
            // bounding sphere centre in the camera frame
            float cw[3], c[3];
#pragma unroll
            for (int i = 0; i < 3; i++) cw[i] = pg[i] + Rg[3 * i] * bc[0] + Rg[3 * i + 1] * bc[1] + Rg[3 * i + 2] * bc[2] - pc[i];
#pragma unroll
            for (int j = 0; j < 3; j++) c[j] = Rc[j] * cw[0] + Rc[3 + j] * cw[1] + Rc[6 + j] * cw[2];
            const float r = m.geom_rbound[g] * 1.0001f + 1e-6f;
            rec[12] = c[0]; rec[13] = c[1]; rec[14] = c[2]; rec[15] = r;
            rec[16] = m.geom_size[3 * g]; rec[17] = m.geom_size[3 * g + 1]; rec[18] = m.geom_size[3 * g + 2];
            rec[19] = __int_as_float(m.geom_type[g]);
            rec[20] = 0; rec[21] = 0;
            rec[22] = __int_as_float(g); rec[23] = 0;
            // screen box of the geom's vertices (polyhedron vertices, bounding box of spheres / cylinders)
            float bx0 = 1e30f, bx1 = -1e30f, by0 = 1e30f, by1 = -1e30f, zmin = 1e30f, bu0 = 1e30f, bu1 = -1e30f, bv0 = 1e30f, bv1 = -1e30f;
            bool crossing = false;
            const int type = m.geom_type[g];
            const bool poly = type == 7 || type == 6;
            const int* G = m.rg + RG_W * g;
            const int nvert = poly ? G[1] : 8;
            const float* hv = m.r_vert + 3 * (size_t)(poly ? G[0] : 0);
            const float ex = rec[16], ey = rec[16], ez = type == 5 ? rec[17] : rec[16];
            float c3x = 0, c3y = 0, c3z = 0;
            unsigned behind = 0;                      // vertices nearer than half the near plane distance (or behind the camera)
            const bool big = poly && (nvert > RVERT_MAX || G[3] > RFACE_MAX);      // (no such polyhedron in the models: general path)
            float hn0 = poly && nvert > 0 ? hv[0] : 0.f, hn1 = poly && nvert > 0 ? hv[1] : 0.f, hn2 = poly && nvert > 0 ? hv[2] : 0.f;      // next vertex, fetched one ahead
            for (int k = 0; k < nvert; k++) {
                float pl[3];
                if (poly) {
                    pl[0] = hn0; pl[1] = hn1; pl[2] = hn2;
                    if (k + 1 < nvert) { hn0 = hv[3 * k + 3]; hn1 = hv[3 * k + 4]; hn2 = hv[3 * k + 5]; }
                }
                else { pl[0] = (k & 1) ? ex : -ex; pl[1] = (k & 2) ? ey : -ey; pl[2] = (k & 4) ? ez : -ez; }
                const float d0 = pl[0] - rec[0], d1 = pl[1] - rec[1], d2 = pl[2] - rec[2];
                // p_c = A^T (p_l - o_l)
                const float xc = rec[3] * d0 + rec[6] * d1 + rec[9] * d2, yc = rec[4] * d0 + rec[7] * d1 + rec[10] * d2, zc = rec[5] * d0 + rec[8] * d1 + rec[11] * d2;
                const float depth = -zc;
                if (poly && k < RVERT_MAX) { vcx[k][lane] = xc; vcy[k][lane] = yc; vcz[k][lane] = zc; c3x += xc; c3y += yc; c3z += zc; }
                if (depth < 0.5f * m.znear) { crossing = true; behind |= 1u << (k & 31); continue; }
                const float iz = 1.0f / depth;
                const float sxp = xc * iz, syp = yc * iz;
                bx0 = fminf(bx0, sxp); bx1 = fmaxf(bx1, sxp); by0 = fminf(by0, syp); by1 = fmaxf(by1, syp);
                bu0 = fminf(bu0, sxp + syp); bu1 = fmaxf(bu1, sxp + syp); bv0 = fminf(bv0, sxp - syp); bv1 = fmaxf(bv1, sxp - syp);
                zmin = fminf(zmin, depth);
            }
            if (poly) {
                // (a polyhedron outside the image gets no face data: nothing will look at it)
                const float pd0 = 1e-4f;
                const bool seen = (c[2] - r < -m.znear) && (crossing || (bx0 - pd0 <= tx && bx1 + pd0 >= -tx && by0 - pd0 <= ty && by1 + pd0 >= -ty));
                if (seen) {
                // The polyhedron for the rasteriser: every face in camera-ray form (for the ray (x, y, -1) t: n.v = a x + b y - c, crossing
                // at t = no / n.v, no = d - n.o_l), the screen box of each face seen from outside (the whole image for a face with a
                // vertex behind the near plane), and the silhouette: an edge between a face seen from outside and one seen from
                // inside.  The plane through the eye and such an edge touches the polyhedron, which lies on one side of it: with
                // N = P0 x P1 (camera frame) turned towards the vertices' centroid, a pixel ray (x, y, -1) sees the polyhedron iff
                // N . (x, y, -1) >= 0 for every silhouette edge -- a line in the image, and no vertex needs to be projected, so
                // polyhedra that reach behind the camera (the table, the frame, the camera's own arm) are outlined like the others.
                const int np = G[3], poff = G[6], eoff = G[7];
                const size_t cbase = (size_t)env * ncam_sel + cs;
                float4* tp = reinterpret_cast<float4*>(tplanes) + cbase * m.nplane + poff;
                float4* fb = reinterpret_cast<float4*>(fbox) + cbase * m.nplane + poff;
                float4* se = reinterpret_cast<float4*>(sedge) + cbase * m.nedge + eoff;
                unsigned long long front = 0;
                const float fpad = 1e-4f;
                // (plane and vertex set of the NEXT face are fetched while this one is worked on: the loop is one face per round trip
                // otherwise, the lanes of a wave being at different geoms)
                const float4* PL = reinterpret_cast<const float4*>(m.r_plane) + G[2];
                const unsigned* FM = m.r_fvmask + G[2];
                float4 n_next = np > 0 ? PL[0] : make_float4(0.f, 0.f, 0.f, 0.f);
                unsigned fm_next = np > 0 ? FM[0] : 0u;
                for (int p = 0; p < np; p++) {
                    const float4 n4 = n_next;
                    unsigned fm = fm_next;
                    if (p + 1 < np) { n_next = PL[p + 1]; fm_next = FM[p + 1]; }
                    const float n[4] = {n4.x, n4.y, n4.z, n4.w};
                    float4 q;
                    q.x = n[0] * rec[3] + n[1] * rec[6] + n[2] * rec[9];
                    q.y = n[0] * rec[4] + n[1] * rec[7] + n[2] * rec[10];
                    q.z = n[0] * rec[5] + n[1] * rec[8] + n[2] * rec[11];
                    q.w = n[3] - (n[0] * rec[0] + n[1] * rec[1] + n[2] * rec[2]);
                    tp[p] = q;
                    float4 b = make_float4(1e30f, -1e30f, 1e30f, -1e30f);
                    if (q.w < 0) {
                        if (p < 64) front |= 1ull << p;
                        if (!big) {
                            // (the face's vertices from one word, next to the plane: no chain of dependent loads per face)
                            const bool clipped = (behind & fm) != 0;
                            while (fm) {
                                const int v = __builtin_ctz(fm);
                                fm &= fm - 1;
                                const float iz = -1.0f / vcz[v][lane];
                                const float x = vcx[v][lane] * iz, y = vcy[v][lane] * iz;
                                b.x = fminf(b.x, x); b.y = fmaxf(b.y, x); b.z = fminf(b.z, y); b.w = fmaxf(b.w, y);
                            }
                            b.x -= fpad; b.y += fpad; b.z -= fpad; b.w += fpad;
                            if (clipped) b = make_float4(-1e30f, 1e30f, -1e30f, 1e30f);
                        }
                    }
                    fb[p] = b;
                }
                int nsil = 0;
                if (!big) {
                    const int ne = G[5];
                    const int4* ED = reinterpret_cast<const int4*>(m.r_edge) + G[4];
                    int4 e_next = ne > 0 ? ED[0] : make_int4(0, 0, 0, 0);
                    for (int e = 0; e < ne; e++) {
                        const int4 e4 = e_next;
                        if (e + 1 < ne) e_next = ED[e + 1];
                        const int E[4] = {e4.x, e4.y, e4.z, e4.w};
                        if ((((front >> E[2]) ^ (front >> E[3])) & 1ull) == 0) continue;
                        const float x0 = vcx[E[0]][lane], y0 = vcy[E[0]][lane], z0 = vcz[E[0]][lane], x1 = vcx[E[1]][lane], y1 = vcy[E[1]][lane], z1 = vcz[E[1]][lane];
                        float A = y0 * z1 - z0 * y1, B = z0 * x1 - x0 * z1, Cz = x0 * y1 - y0 * x1;
                        const float il = rsqrtf(fmaxf(A * A + B * B + Cz * Cz, 1e-30f)) * ((A * c3x + B * c3y + Cz * c3z < 0) ? -1.0f : 1.0f);
                        se[nsil++] = make_float4(A * il, B * il, -Cz * il, 0.0f);          // N . (x, y, -1) = A x + B y - Cz
                    }
                }
                rec[16] = __int_as_float(eoff); rec[17] = __int_as_float(nsil); rec[18] = __int_as_float(big ? 1 : 0);
                rec[20] = __int_as_float(poff); rec[21] = __int_as_float(np);
                }
                rec[19] = __int_as_float(7);                                   // boxes are drawn as polyhedra too
            }
            if (type != 7 && type != 6) {   // spheres / cylinders: the corners of their bounding box are not on the surface; box only
                bu0 = bv0 = -1e30f; bu1 = bv1 = 1e30f;
            }
            if (crossing) { bx0 = by0 = bu0 = bv0 = -1e30f; bx1 = by1 = bu1 = bv1 = 1e30f; zmin = 0; }
            const float pad = 1e-4f;
            rec[24] = bx0 - pad; rec[25] = bx1 + pad; rec[26] = by0 - pad; rec[27] = by1 + pad; rec[23] = zmin * 0.9999f - 1e-5f;
            rec[28] = bu0 - 2 * pad; rec[29] = bu1 + 2 * pad; rec[30] = bv0 - 2 * pad; rec[31] = bv1 + 2 * pad;
            keep = (c[2] - r < -m.znear) && rec[24] <= tx && rec[25] >= -tx && rec[26] <= ty && rec[27] >= -ty;
        }
        const unsigned long long bal = __ballot(keep);
        if (keep) {
            const int k = base + __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
            for (int q = 0; q < REC_W; q++) out[k * REC_W + q] = rec[q];
            keys[k] = rec[23];
        }
        base += __popcll(bal);
    }
    if (lane == 0) counts[(size_t)env * ncam_sel + cs] = base;
    __syncthreads();
    // front-to-back order by rank sort on the nearest depth (ties by index)
    int* ord = order + ((size_t)env * ncam_sel + cs) * m.ngeom;
    for (int i = lane; i < base; i += 64) {
        const float ki = keys[i];
        int rank = 0;
        for (int j = 0; j < base; j++) { const float kj = keys[j]; rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
        ord[rank] = i;
    }
}

// grid (views of the pass), block 256.  The list headers k_render_depth reads: one 64-byte header per kept record in FRONT-TO-BACK order -- the screen
// octagon, the nearest depth, a packed word (type | general path << 4 | faces << 6 | silhouette edges << 13; k_render_depth adds the staging bits),
// the record index, the face / edge offsets into the camera's scratch -- and the record's BIN MASK: the set of k_render_depth's bins (bin_tx x
// BIN_TY tiles, numbered row by row; 128 bits, all ones for an image with more bins) that none of its silhouette edges excludes.  The octagon
// alone leaves 17.7 candidates per bin (the long frame bars cross the image at an angle), the mask 4.3.  A wave per record, lane = bin: the
// record's edges are loaded one per lane and handed round by v_readlane; only records whose octagon's box meets more than three bins are tested.
// (Inside k_render_geoms -- one wave per view, LDS-bound at five waves per CU -- the same pass cost 0.8 ms per 16384 views; here it hides behind
// the other waves: profiles/r06_experiments.txt 1.)
__global__ void __launch_bounds__(256) k_render_heads(const float* __restrict__ recs, const int* __restrict__ counts, const int* __restrict__ order,
                                                      const float4* __restrict__ sedges, int nedge, const float* __restrict__ cam_fovy, const int* __restrict__ cam_ids,
                                                      int ncam_sel, int ngeom, int H, int W, float4* __restrict__ heads, const int bin_tx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t cb = blockIdx.x;
    const int cs = (int)(cb % ncam_sel);
    const float scale = 2.0f * cam_fovy[cam_ids[cs]] / (float)H;
    const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
    const int nbx = (tiles_x + bin_tx - 1) / bin_tx, nby = (tiles_y + BIN_TY - 1) / BIN_TY, nb = nbx * nby;
#if defined(AVSIM_RDBG) && AVSIM_RDBG == 4
    const bool binmask = false;      // (experiment: no bin masks)
#else
    const bool binmask = nb <= 128;
#endif
    float cxl[2], cxr[2], cyt[2], cyb[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int b = lane + 64 * h, bx = b % nbx, by = b / nbx;
        const int px0 = bx * bin_tx * TILE_W, px1 = px0 + bin_tx * TILE_W < W ? px0 + bin_tx * TILE_W : W;
        const int py0 = by * BIN_TY * TILE_H, py1 = py0 + BIN_TY * TILE_H < H ? py0 + BIN_TY * TILE_H : H;
        cxl[h] = (px0 - 0.5f * W) * scale; cxr[h] = (px1 - 0.5f * W) * scale; cyt[h] = -(py0 - 0.5f * H) * scale; cyb[h] = -(py1 - 0.5f * H) * scale;
    }
    const int cnt = counts[cb];
    const float* R = recs + cb * ngeom * REC_W;
    const float bw = bin_tx * TILE_W * scale, bh = BIN_TY * TILE_H * scale, x0 = -0.5f * W * scale, y0 = 0.5f * H * scale;
    for (int r = wave; r < cnt; r += 4) {
        const int i = __builtin_amdgcn_readfirstlane(order[cb * ngeom + r]);
        const float* rec = R + (size_t)i * REC_W;                 // wave-uniform: scalar loads
        const int ty_ = __float_as_int(rec[19]);
        const int gen = ty_ == 7 && __float_as_int(rec[18]) != 0 ? 1 : 0;
        const int np_ = ty_ == 7 ? __float_as_int(rec[21]) : 0, ns_ = ty_ == 7 && !gen ? __float_as_int(rec[17]) : 0;
        unsigned long long m0 = ~0ull, m1 = ~0ull;
        if (binmask && ty_ == 7 && !gen && ns_ > 0) {
            const int c0 = max(0, (int)floorf((rec[24] - x0) / bw)), c1 = min(nbx - 1, (int)floorf((rec[25] - x0) / bw));
            const int r0 = max(0, (int)floorf((y0 - rec[27]) / bh)), r1 = min(nby - 1, (int)floorf((y0 - rec[26]) / bh));
            if ((c1 - c0 + 1) * (r1 - r0 + 1) > 3) {
                const int ns = ns_ > 64 ? 64 : ns_;          // (a polyhedron of <= 32 vertices has <= 32 silhouette edges; more are left to the tiles)
                const float4 eg = (sedges + cb * nedge + __float_as_int(rec[16]))[lane < ns ? lane : 0];
                bool out0 = false, out1 = false;
                for (int e = 0; e < ns; e++) {
                    const float ea = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eg.x), e));
                    const float eb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eg.y), e));
                    const float ec = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eg.z), e));
                    // the bin's four corner rays all outside this edge (its largest corner value is negative): the bin cannot see the polyhedron
                    out0 = out0 || (fmaxf(ea * cxl[0], ea * cxr[0]) + fmaxf(eb * cyt[0], eb * cyb[0]) + ec < 0);
                    if (nb > 64) out1 = out1 || (fmaxf(ea * cxl[1], ea * cxr[1]) + fmaxf(eb * cyt[1], eb * cyb[1]) + ec < 0);
                }
                m0 = __ballot(!out0);
                if (nb > 64) m1 = __ballot(!out1);
            }
        }
        if (lane == 0) {
            float4* hd = heads + (cb * ngeom + r) * 4;
            hd[0] = make_float4(rec[24], rec[25], rec[26], rec[27]);
            hd[1] = make_float4(rec[28], rec[29], rec[30], rec[31]);
            hd[2] = make_float4(rec[23], __int_as_float((ty_ & 15) | (gen << 4) | ((np_ > 127 ? 127 : np_) << 6) | ((ns_ > 127 ? 127 : ns_) << 13)), __int_as_float(i),
                                __int_as_float(ty_ == 7 ? (__float_as_int(rec[20]) | (__float_as_int(rec[16]) << 16)) : 0));
            hd[3] = make_float4(__int_as_float((int)(unsigned)m0), __int_as_float((int)(unsigned)(m0 >> 32)), __int_as_float((int)(unsigned)m1), __int_as_float((int)(unsigned)(m1 >> 32)));
        }
    }
}

// max over the 64 lanes, broadcast: DPP butterflies inside each row of 16, then four v_readlane
// max / min as ONE instruction: fmaxf / fminf on a loop-carried value come with a canonicalising v_max x, x, x in front (IEEE quieting
// of signalling NaNs, which the compiler cannot rule out for a phi); none of the depths and edge values here is a NaN that matters
__device__ __forceinline__ float vmax1(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin1(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
typedef float v2f __attribute__((ext_vector_type(2)));      // two pixels per v_pk_fma_f32

__device__ inline float wave_max(float x) {
#define AVS_DPP_MAX(ctrl) x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true)))
    AVS_DPP_MAX(0xB1);    // quad_perm [1,0,3,2]
    AVS_DPP_MAX(0x4E);    // quad_perm [2,3,0,1]
    AVS_DPP_MAX(0x141);   // row_half_mirror
    AVS_DPP_MAX(0x140);   // row_mirror
#undef AVS_DPP_MAX
    const int xi = __builtin_bit_cast(int, x);
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)), b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)), d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// parametric interval of the ray t * v from origin o (geom frame) inside the convex geom; v differs per pixel
__device__ inline bool ray_prim(int type, const float* sz, const float* o, const float* v, float* t0) {
    float lo = -1e30f, hi = 1e30f;
    if (type == 2) {   // sphere
        const float a = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], b = o[0] * v[0] + o[1] * v[1] + o[2] * v[2];
        const float c = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] - sz[0] * sz[0], disc = b * b - a * c;
        if (disc < 0) return false;
        const float s = sqrtf(disc), ia = 1.0f / a;
        lo = (-b - s) * ia; hi = (-b + s) * ia;
    } else if (type == 6) {   // box
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (v[k] == 0) { if (fabsf(o[k]) > sz[k]) return false; continue; }
            const float iv = 1.0f / v[k];
            float ta = (-sz[k] - o[k]) * iv, tb = (sz[k] - o[k]) * iv;
            if (ta > tb) { const float t = ta; ta = tb; tb = t; }
            lo = fmaxf(lo, ta); hi = fminf(hi, tb);
        }
    } else {   // cylinder: axis z, radius sz[0], half height sz[1]
        const float a = v[0] * v[0] + v[1] * v[1], b = o[0] * v[0] + o[1] * v[1], c = o[0] * o[0] + o[1] * o[1] - sz[0] * sz[0];
        if (a > 0) {
            const float disc = b * b - a * c;
            if (disc < 0) return false;
            const float s = sqrtf(disc), ia = 1.0f / a;
            lo = (-b - s) * ia; hi = (-b + s) * ia;
        } else if (c > 0) return false;
        if (v[2] == 0) { if (fabsf(o[2]) > sz[1]) return false; }
        else {
            const float iv = 1.0f / v[2];
            float ta = (-sz[1] - o[2]) * iv, tb = (sz[1] - o[2]) * iv;
            if (ta > tb) { const float t = ta; ta = tb; tb = t; }
            lo = fmaxf(lo, ta); hi = fminf(hi, tb);
        }
    }
    if (lo > hi) return false;
    *t0 = lo;
    return true;
}

// One 32 x 8 pixel tile by one wavefront: lane -> 4 pixels (x0 .. x0+3, y).  `blist` / `cnt` = the records of the tile's bin in
// front-to-back order (LDS, built by k_render_depth).  RGB: the cast also remembers which record (and hull face) each pixel sees,
// and an epilogue shades it (flat material colour, Lambert terms of the headlight along the ray and of the scene's directional
// light, sky gradient where nothing is hit) into u8[H][W][3]
// The bin's list lives in LDS (k_render_depth): per entry the screen octagon (hA: box, hB: x + y / x - y extents), hC = (nearest depth, packed
// word, record index, -), and -- polyhedra -- the faces in camera-ray form, their screen boxes and the silhouette edges in `arena`.  packed:
// bits 0-3 type, 4 general path, 5 staged, 6-12 faces, 13-19 silhouette edges, 20-31 arena offset (float4 slots).
constexpr int BL_MAX = 128;      // list entries per bin (the models have <= 91 collision geoms; RenderHost::launch refuses more)
#ifndef AVSIM_RD_ARENA4
#define AVSIM_RD_ARENA4 1600
#endif
constexpr int ARENA4 = AVSIM_RD_ARENA4;     // float4 slots of a bin's staging arena: 25 KB (with the headers 31.3 KB per block: five blocks = five waves per SIMD at 82 VGPRs)
template <bool RGB>
__device__ __forceinline__ void render_tile(const int lane, const int tx0, const int ty0, const int cs, const int env, const float4* hA, const float4* hB, const float4* hC, const float4* arena, const int cnt,
                                            const float* __restrict__ R, const float4* __restrict__ tplanes, const float4* __restrict__ fboxes, const float4* __restrict__ sedges, int nplane, int nedge, const float scale, int ncam_sel, int H,
                                            int W, float znear, float zfar, float* __restrict__ out, const float* __restrict__ geom_rgba, const float* __restrict__ light,
                                            const float* __restrict__ camaux, unsigned char* __restrict__ out_rgb) {
    // lane -> 4 adjacent pixels in each of TILE_R rows (rows 8 apart, so that a row of the tile is still written by 8 lanes)
    const int px = tx0 + 4 * (lane & 7), py0 = ty0 + (lane >> 3);
    // tile pyramid: x in [xl, xr], y in [yb, yt] at z = -1
    const int x1 = tx0 + TILE_W < W ? tx0 + TILE_W : W, y1 = ty0 + TILE_H < H ? ty0 + TILE_H : H;
    const float xl = (tx0 - 0.5f * W) * scale, xr = (x1 - 0.5f * W) * scale, yt = -(ty0 - 0.5f * H) * scale, yb = -(y1 - 0.5f * H) * scale;
    // The tile works in INVERSE depth s = 1 / t: for a face (a, b, c, no) the crossing of the pixel ray (x, y, -1) is t = no / (a x + b y - c), so
    // s = (a / no) x + (b / no) y - c / no is affine in the pixel -- one fma per pixel and face instead of fma + rcp + mul, the nearest surface
    // is the LARGEST s, the entry face of a polyhedron the one with the SMALLEST s, and one rcp per pixel at the very end gives the depth.
    float dyr[TILE_R], dx[4], best[NPX];      // best: inverse depth of what the pixel sees so far (1 / zfar: nothing; off-image pixels hold a value no surface beats)
    int win[NPX];     // RGB: record index | entry face << 8 of what the pixel sees; pixel q = 4 * row + column
    const float sfar0 = 1.0f / zfar, sznear = 1.0f / znear;
#pragma unroll
    for (int r = 0; r < TILE_R; r++) dyr[r] = -(py0 + 8 * r + 0.5f - 0.5f * H) * scale;
#pragma unroll
    for (int q = 0; q < 4; q++) dx[q] = (px + q + 0.5f - 0.5f * W) * scale;
#pragma unroll
    for (int q = 0; q < NPX; q++) { best[q] = (px + (q & 3) < W && py0 + 8 * (q >> 2) < H) ? sfar0 : 1e30f; win[q] = -1; }
    const v2f dxa = {dx[0], dx[1]}, dxb = {dx[2], dx[3]};
    float far = zfar, sfar = sfar0;    // farthest current depth over the tile's pixels and its inverse (off-image pixels do not count)
    RSTAT(0, 1); RSTAT(1, cnt);
    for (int k0 = 0; k0 < cnt; k0 += 64) {
        bool hit = false;
        float zmine = 0;
        int pmine = 0, kmine = 0;
        if (k0 + lane < cnt) {
            // one entry per lane, from LDS: the octagon test, and what the tile needs to know of the entry should it be hit (a dropped entry
            // carries an empty box)
            const float4 ba = hA[k0 + lane], bo = hB[k0 + lane], hc = hC[k0 + lane];
            // (one compare of the largest violation instead of eight compares and seven mask ANDs: the scalar unit's share of the tile -- mask logic,
            // loop control, readlane targets: 374 SALU next to 636 VALU instructions per tile -- was as long as the vector unit's)
            hit = vmax3(vmax3(ba.x - xr, xl - ba.y, ba.z - yt), vmax3(yb - ba.w, bo.x - (xr + yt), (xl + yb) - bo.y), fmaxf(bo.z - (xr - yb), (xl - yt) - bo.w)) <= 0.0f;
            zmine = hc.x; pmine = __float_as_int(hc.y); kmine = __float_as_int(hc.z);
        }
        unsigned long long mask = __ballot(hit);
        RSTAT(2, __popcll(mask));
        while (mask) {
            const int pos = __builtin_ctzll(mask);
            mask &= mask - 1;
            // front to back: nothing behind this geom's nearest vertex can win once every pixel of the tile is nearer
            if (__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zmine), pos)) >= far) { mask = 0; k0 = cnt; break; }
            const int pk = __builtin_amdgcn_readlane(pmine, pos), k = __builtin_amdgcn_readlane(kmine, pos);
            const float* rec = R + (size_t)k * REC_W;    // wave-uniform: scalar loads (primitives and unstaged polyhedra only)
            const int type = pk & 15;
            if (type == 7) {
                // A convex polyhedron (mesh hull or box), rasterised from what k_render_geoms projected once per camera:
                //  * every face seen from outside has a screen box; only those whose box meets the tile can cover one of its pixels
                //    (the faces seen from outside tile the silhouette without overlap), and the depth of a pixel is the LARGEST
                //    crossing t = no / (a x + b y - c) over them -- faces that do not cover the pixel cross earlier, so extra
                //    candidates are harmless and no per-face inside test is needed;
                //  * whether a pixel sees the polyhedron at all is decided by the silhouette edges, lines in the image, positive
                //    inside: an edge with all four tile corners outside discards the polyhedron for the tile, one with all four
                //    inside needs no per-pixel test, the (few) others are evaluated per pixel.
                // One face and one edge per lane for the tile tests; the kept ones are fetched from those lanes' registers.
                const size_t cb = (size_t)env * ncam_sel + cs;
                const bool staged = (pk >> 5) & 1;
                const int np = staged ? (pk >> 6) & 127 : __float_as_int(rec[21]), aoff = (pk >> 20) & 4095;
                const float4* P = tplanes + cb * nplane + (staged ? 0 : __float_as_int(rec[20]));      // (read when the entry is not staged)
                float lo[NPX];    // fast path: the polyhedron's inverse depth at the pixel (smallest over its faces seen from outside); general path: entry depth
                float em[NPX];    // the smallest silhouette-edge value of the pixel so far (negative: outside the polyhedron); a float,
                                  // not a flag: an array of bools is packed into bytes by the compiler and unpacked again in every round
                int face[NPX];
                const bool genp = (pk >> 4) & 1;
#pragma unroll
                for (int q = 0; q < NPX; q++) { lo[q] = genp ? -1e30f : 1e30f; em[q] = 1e30f; face[q] = 0; }
                if (!genp) {
                    const int nsil = staged ? (pk >> 13) & 127 : __float_as_int(rec[17]);
                    const bool onf = lane < np, one = lane < nsil;
                    float4 fl, bb, eg;
                    if (staged) {          // one LDS round trip: the bin staged this polyhedron's faces (inverse-depth form), face boxes and silhouette edges
                        fl = arena[aoff + (onf ? lane : 0)]; bb = arena[aoff + np + (onf ? lane : 0)]; eg = arena[aoff + 2 * np + (one ? lane : 0)];
                    } else {
                        const float4* FB = fboxes + cb * nplane + __float_as_int(rec[20]);
                        const float4* SE = sedges + cb * nedge + __float_as_int(rec[16]);
                        fl = P[onf ? lane : 0]; bb = FB[onf ? lane : 0]; eg = SE[one ? lane : 0];
                        const float iw = __builtin_amdgcn_rcpf(fl.w);
                        fl = make_float4(fl.x * iw, fl.y * iw, -fl.z * iw, fl.w);
                    }
                    const bool keepF = onf && fmaxf(fmaxf(bb.x - xr, xl - bb.y), fmaxf(bb.z - yt, yb - bb.w)) <= 0.0f;
                    // the edge function over the tile's four corners: its largest and smallest value (affine: max / min of the x part + of the y part)
                    const float exa = eg.x * xl, exb = eg.x * xr, eya = eg.y * yb + eg.z, eyb = eg.y * yt + eg.z;
                    const float emax = fmaxf(exa, exb) + fmaxf(eya, eyb), emin = fminf(exa, exb) + fminf(eya, eyb);
                    const bool allout = emax < 0, allin = emin >= 0;
                    if (__any(one && allout)) continue;
                    unsigned long long mF = __ballot(keepF), mS = __ballot(one && !allin);
                    if (!mF) continue;
                    {   // nothing of this polyhedron is nearer in this tile than what the tile already shows: a pixel's inverse depth is that of its
                        // covering face, affine in the pixel, so over the tile it is not above the largest corner value of the candidate faces
                        // (a face some corner ray does not approach -- s <= 0 there -- gives no bound)
                        const float sxa = fl.x * xl, sxb = fl.x * xr, sya = fl.y * yb + fl.z, syb = fl.y * yt + fl.z;
                        const float smax = fmaxf(sxa, sxb) + fmaxf(sya, syb), smin = fminf(sxa, sxb) + fminf(sya, syb);
                        if (!__any(keepF && (!(smin > 0) || smax > sfar))) continue;
                    }
                    RSTAT(3, 1); RSTAT(4, __popcll(mF)); RSTAT(5, __popcll(mS));
                    // two faces per round: six readlanes, a packed fma per pixel pair and face, one v_min3 per pixel
                    while (mF) {
                        const int p0 = __builtin_ctzll(mF);
                        mF &= mF - 1;
                        int p1 = p0;
                        if (!RGB && mF) { p1 = __builtin_ctzll(mF); mF &= mF - 1; }
                        const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.x), p0));
                        const float b0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.y), p0));
                        const float c0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.z), p0));
                        const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.x), p1));
                        const float b1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.y), p1));
                        const float c1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.z), p1));
                        const v2f a0v = {a0, a0}, a1v = {a1, a1};
#pragma unroll
                        for (int r = 0; r < TILE_R; r++) {
                            const float n0 = b0 * dyr[r] + c0, n1 = b1 * dyr[r] + c1;
                            const v2f n0v = {n0, n0}, n1v = {n1, n1};
                            const v2f s0a = a0v * dxa + n0v, s0b = a0v * dxb + n0v, s1a = a1v * dxa + n1v, s1b = a1v * dxb + n1v;
                            // (a pixel inside the silhouette approaches every face seen from outside: s > 0 there; elsewhere the value is unused)
                            if (RGB) {
                                if (s0a.x < lo[4 * r]) face[4 * r] = p0;
                                if (s0a.y < lo[4 * r + 1]) face[4 * r + 1] = p0;
                                if (s0b.x < lo[4 * r + 2]) face[4 * r + 2] = p0;
                                if (s0b.y < lo[4 * r + 3]) face[4 * r + 3] = p0;
                            }
                            lo[4 * r] = vmin3(lo[4 * r], s0a.x, s1a.x); lo[4 * r + 1] = vmin3(lo[4 * r + 1], s0a.y, s1a.y);
                            lo[4 * r + 2] = vmin3(lo[4 * r + 2], s0b.x, s1b.x); lo[4 * r + 3] = vmin3(lo[4 * r + 3], s0b.y, s1b.y);
                        }
                    }
                    while (mS) {
                        const int e0 = __builtin_ctzll(mS);
                        mS &= mS - 1;
                        int e1 = e0;
                        if (mS) { e1 = __builtin_ctzll(mS); mS &= mS - 1; }
                        const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eg.x), e0));
                        const float b0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eg.y), e0));
                        const float c0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eg.z), e0));
                        const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eg.x), e1));
                        const float b1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eg.y), e1));
                        const float c1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, eg.z), e1));
                        const v2f a0v = {a0, a0}, a1v = {a1, a1};
#pragma unroll
                        for (int r = 0; r < TILE_R; r++) {
                            const float n0 = b0 * dyr[r] + c0, n1 = b1 * dyr[r] + c1;
                            const v2f n0v = {n0, n0}, n1v = {n1, n1};
                            const v2f s0a = a0v * dxa + n0v, s0b = a0v * dxb + n0v, s1a = a1v * dxa + n1v, s1b = a1v * dxb + n1v;
                            em[4 * r] = vmin3(em[4 * r], s0a.x, s1a.x); em[4 * r + 1] = vmin3(em[4 * r + 1], s0a.y, s1a.y);
                            em[4 * r + 2] = vmin3(em[4 * r + 2], s0b.x, s1b.x); em[4 * r + 3] = vmin3(em[4 * r + 3], s0b.y, s1b.y);
                        }
                    }
                } else {
                    // general path (a vertex behind the near plane -- the links around the camera itself --, or more faces / vertices
                    // than the per-lane tables hold): the faces in camera-ray form, one per lane, evaluated on the four corner rays of
                    // the tile.  A crossing t = no / (a x + b y - c) is a ratio of affine functions, so over the tile it takes its
                    // extremes at the corners.  A face seen from outside that no corner ray approaches separates the tile from the
                    // polyhedron; the entry of a ray is the LARGEST crossing over the faces seen from outside, so one whose largest
                    // crossing over the tile is below L = the largest of those faces' smallest crossings is never the entry face; the
                    // other faces bound the exit, and only those that can be the exit face of some ray of the tile are kept.
                unsigned long long mF = 0, mB = 0;     // np <= 64 (hulls are decimated to <= 32 vertices); larger hulls keep every face
                float L = -1e30f, U = 1e30f;
                bool sep = false;
                float4 fl = make_float4(0.f, 0.f, 0.f, 0.f);      // this lane's face: the casting loops fetch the kept faces from here (v_readlane)
                if (np <= 64) {
                    const bool on = lane < np;
                    const float4 f = staged ? arena[aoff + (on ? lane : 0)] : P[on ? lane : 0];
                    fl = f;
                    const float n00 = f.x * xl + f.y * yb - f.z, n10 = f.x * xr + f.y * yb - f.z, n01 = f.x * xl + f.y * yt - f.z, n11 = f.x * xr + f.y * yt - f.z;
                    const bool front = on && f.w < 0;
                    const bool allneg = n00 < 0 && n10 < 0 && n01 < 0 && n11 < 0, allpos = n00 >= 0 && n10 >= 0 && n01 >= 0 && n11 >= 0;
                    sep = front && allpos;
                    const float t00 = f.w * __builtin_amdgcn_rcpf(n00), t10 = f.w * __builtin_amdgcn_rcpf(n10), t01 = f.w * __builtin_amdgcn_rcpf(n01), t11 = f.w * __builtin_amdgcn_rcpf(n11);
                    float tmn = -1e30f, tmx = 1e30f;
                    if (front && allneg) { tmn = fminf(fminf(t00, t10), fminf(t01, t11)); tmx = fmaxf(fmaxf(t00, t10), fmaxf(t01, t11)); }
                    L = wave_max(front ? tmn : -1e30f);
                    U = wave_max(front ? tmx : -1e30f);
                    const bool keepF = front && tmx >= L - (1e-5f * fabsf(L) + 1e-6f);
                    // the other faces (camera inside their half space) bound the EXIT of a ray, the smallest crossing over the faces it
                    // leaves through; an entry is valid iff it is not beyond the exit.  Mirror image of the entry faces: a face whose
                    // smallest crossing over the tile is above X = the smallest of the faces' largest crossings is never the exit face
                    // (nor is one that no corner ray leaves through); and none matters when even X is beyond the farthest entry U
                    float bmin = 1e30f, bmax = 1e30f;
                    const bool back = on && !front;
                    const bool allpos_s = n00 > 0 && n10 > 0 && n01 > 0 && n11 > 0, allneg_b = n00 <= 0 && n10 <= 0 && n01 <= 0 && n11 <= 0;
                    if (back && allpos_s) { bmin = fminf(fminf(t00, t10), fminf(t01, t11)); bmax = fmaxf(fmaxf(t00, t10), fmaxf(t01, t11)); }
                    else if (back && !allneg_b) bmin = -1e30f;     // sign change inside the tile: keep, no bound from it
                    const float X = -wave_max(back ? -bmax : -1e30f);
                    const bool keepB = back && !allneg_b && bmin <= X + (1e-5f * fabsf(X) + 1e-6f) && bmin <= U + (1e-5f * fabsf(U) + 1e-6f);
                    mF = __ballot(keepF);
                    mB = __ballot(keepB);
                } else {
                    for (int p = lane; p < np; p += 64) {
                        const float4 f = P[p];
                        sep = sep || (f.w < 0 && f.x * xl + f.y * yt - f.z >= 0 && f.x * xr + f.y * yt - f.z >= 0 && f.x * xl + f.y * yb - f.z >= 0 &&
                                      f.x * xr + f.y * yb - f.z >= 0);
                    }
                }
                if (__any(sep)) continue;
                if (np <= 64 && L >= far) continue;
                RSTAT(6, 1);        // nothing of this hull in the tile is nearer than what the tile already shows
                // Pass 1, candidate entry faces: the entry is the largest crossing, a ray that does not approach such a face misses.
                // Pass 2, vetoing faces: the entry point must lie behind them (lo * n.v <= no; no division).
                if (np <= 64) {
                    while (mF) {
                        const int p = __builtin_ctzll(mF);
                        mF &= mF - 1;
                        float4 f;
                        f.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.x), p));
                        f.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.y), p));
                        f.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.z), p));
                        f.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.w), p));
                        float nb[TILE_R];
#pragma unroll
                        for (int r = 0; r < TILE_R; r++) nb[r] = f.y * dyr[r] - f.z;
#pragma unroll
                        for (int q = 0; q < NPX; q++) {
                            const float nv = nb[q >> 2] + f.x * dx[q & 3];
                            const float t = f.w * __builtin_amdgcn_rcpf(nv);
                            if (!(nv < 0)) em[q] = -1.0f;
                            if (RGB) { if (t > lo[q]) face[q] = p; }
                            lo[q] = vmax1(lo[q], t);
                        }
                    }
                } else {
                    for (int p = 0; p < np; p++) {
                        const float4 f = P[p];
                        if (!(f.w < 0)) continue;
                        float nb[TILE_R];
#pragma unroll
                        for (int r = 0; r < TILE_R; r++) nb[r] = f.y * dyr[r] - f.z;
#pragma unroll
                        for (int q = 0; q < NPX; q++) {
                            const float nv = nb[q >> 2] + f.x * dx[q & 3];
                            const float t = f.w * __builtin_amdgcn_rcpf(nv);
                            if (!(nv < 0)) em[q] = -1.0f;
                            if (RGB) { if (t > lo[q]) face[q] = p; }
                            lo[q] = vmax1(lo[q], t);
                        }
                    }
                }
                // no ray of the tile can still improve on what it already sees: skip the validity pass
                {
                    bool need = false;
#pragma unroll
                    for (int q = 0; q < NPX; q++) need = need || (em[q] >= 0 && lo[q] >= znear && lo[q] * best[q] < 1.0f);      // (best holds inverse depths)
                    if (!__any(need)) continue;
                }
                if (np <= 64) {
                    while (mB) {
                        const int p = __builtin_ctzll(mB);
                        mB &= mB - 1;
                        float4 f;
                        f.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.x), p));
                        f.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.y), p));
                        f.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.z), p));
                        f.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fl.w), p));
                        float nb[TILE_R];
#pragma unroll
                        for (int r = 0; r < TILE_R; r++) nb[r] = f.y * dyr[r] - f.z;
#pragma unroll
                        for (int q = 0; q < NPX; q++) if (!(lo[q] * (nb[q >> 2] + f.x * dx[q & 3]) <= f.w)) em[q] = -1.0f;
                    }
                } else {
                    for (int p = 0; p < np; p++) {
                        const float4 f = P[p];
                        if (f.w < 0) continue;
                        float nb[TILE_R];
#pragma unroll
                        for (int r = 0; r < TILE_R; r++) nb[r] = f.y * dyr[r] - f.z;
#pragma unroll
                        for (int q = 0; q < NPX; q++) if (!(lo[q] * (nb[q >> 2] + f.x * dx[q & 3]) <= f.w)) em[q] = -1.0f;
                    }
                }
#pragma unroll
                for (int q = 0; q < NPX; q++) lo[q] = lo[q] > 0 ? __builtin_amdgcn_rcpf(lo[q]) : 1e30f;      // entry depth -> inverse depth
                }
                // the pixel sees the polyhedron (inside every silhouette edge), beyond the near plane, nearer than what it has
                float bm0 = best[0];
#pragma unroll
                for (int q = 1; q < NPX; q++) bm0 = fminf(bm0, best[q]);
#pragma unroll
                for (int q = 0; q < NPX; q++) {
                    float c = em[q] >= 0 ? lo[q] : 0.0f;      // (two selects and a max per pixel: no mask logic on the scalar unit)
                    c = lo[q] <= sznear ? c : 0.0f;
                    if (RGB) { if (c > best[q]) win[q] = k | (face[q] << 8); }
                    best[q] = vmax1(best[q], c);
                }
                {
                    float bm = best[0];
#pragma unroll
                    for (int q = 1; q < NPX; q++) bm = fminf(bm, best[q]);
                    if (__any(bm != bm0)) {      // some pixel of the tile came nearer: the tile's farthest depth may have
                        sfar = -wave_max(-bm);
                        far = __builtin_amdgcn_rcpf(sfar);
                    }
                }
            } else {
                // direction in the geom frame: A (dx, dy, -1)
                RSTAT(6, 1);
                const float o[3] = {rec[0], rec[1], rec[2]};
                float va[3];
#pragma unroll
                for (int i = 0; i < 3; i++) va[i] = rec[3 + 3 * i];
                const float sz[3] = {rec[16], rec[17], rec[18]};
#pragma unroll
                for (int q = 0; q < NPX; q++) {
                    const float dyq = dyr[q >> 2], dxq = dx[q & 3];
                    const float v[3] = {rec[4] * dyq - rec[5] + va[0] * dxq, rec[7] * dyq - rec[8] + va[1] * dxq, rec[10] * dyq - rec[11] + va[2] * dxq};
                    float t0;
                    if (ray_prim(type, sz, o, v, &t0) && t0 >= znear && t0 * best[q] < 1.0f) { best[q] = 1.0f / t0; if (RGB) win[q] = k; }
                }
                {
                    float bm = best[0];
#pragma unroll
                    for (int q = 1; q < NPX; q++) bm = fminf(bm, best[q]);
                    sfar = -wave_max(-bm);
                    far = __builtin_amdgcn_rcpf(sfar);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NPX; q++) best[q] = best[q] == sfar0 ? zfar : __builtin_amdgcn_rcpf(best[q]);      // (the far plane itself where nothing was hit)
    if (!RGB) {
#pragma unroll
        for (int r = 0; r < TILE_R; r++) {
            const int py = py0 + 8 * r;
            if (py < H) {
                float* dst = out + (((size_t)env * ncam_sel + cs) * H + py) * W + px;
                if (px + 3 < W && (W & 3) == 0) {
#ifndef AVSIM_RD_NO_NT
                    // streamed out (non-temporal; round 6: 11.21 -> 11.04 ms): the image is written once and not read by this kernel; keeping its 20 GB out of the L2 leaves the headers and faces there
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    const v4f v = {best[4 * r], best[4 * r + 1], best[4 * r + 2], best[4 * r + 3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(dst));
#else
                    *reinterpret_cast<float4*>(dst) = make_float4(best[4 * r], best[4 * r + 1], best[4 * r + 2], best[4 * r + 3]);
#endif
                }
                else {
#pragma unroll
                    for (int q = 0; q < 4; q++) if (px + q < W) dst[q] = best[4 * r + q];
                }
            }
        }
        return;
    }
    const float* aux = camaux + ((size_t)env * ncam_sel + cs) * 8;
    const float amb = light[0], hd = light[1], ld = light[2];
#pragma unroll
    for (int r = 0; r < TILE_R; r++) {
        const int py = py0 + 8 * r;
        const float dy_ = dyr[r];
        if (py >= H) continue;
    unsigned char col[12];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float idn = rsqrtf(dx[q] * dx[q] + dy_ * dy_ + 1.0f);
        float rgb[3];
        if (win[4 * r + q] >= 0) {
            const int k = win[4 * r + q] & 255;
            const float* rec = R + (size_t)k * REC_W;
            const int type = __float_as_int(rec[19]);
            float n[3];    // outward normal in the camera frame
            if (type == 7) {
                const float4 f = tplanes[((size_t)env * ncam_sel + cs) * nplane + __float_as_int(rec[20]) + (win[4 * r + q] >> 8)];
                n[0] = f.x; n[1] = f.y; n[2] = f.z;
            } else {
                float p[3], ng[3] = {0, 0, 0};
                const float t = best[4 * r + q];
#pragma unroll
                for (int i = 0; i < 3; i++) p[i] = rec[i] + t * (rec[3 + 3 * i] * dx[q] + rec[3 + 3 * i + 1] * dy_ - rec[3 + 3 * i + 2]);
                if (type == 2) {
                    const float ir = rsqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
                    ng[0] = p[0] * ir; ng[1] = p[1] * ir; ng[2] = p[2] * ir;
                } else if (type == 6) {   // the face whose slab the entry point lies on
                    const float q0 = fabsf(p[0]) / rec[16], q1 = fabsf(p[1]) / rec[17], q2 = fabsf(p[2]) / rec[18];
                    const int a = (q1 > q0) ? ((q2 > q1) ? 2 : 1) : ((q2 > q0) ? 2 : 0);
                    ng[a] = p[a] > 0 ? 1.0f : -1.0f;
                } else {                  // cylinder: cap or side
                    const float rr = sqrtf(p[0] * p[0] + p[1] * p[1]);
                    if (fabsf(p[2]) / rec[17] > rr / rec[16]) ng[2] = p[2] > 0 ? 1.0f : -1.0f;
                    else { ng[0] = p[0] / rr; ng[1] = p[1] / rr; }
                }
#pragma unroll
                for (int j = 0; j < 3; j++) n[j] = rec[3 + j] * ng[0] + rec[6 + j] * ng[1] + rec[9 + j] * ng[2];   // A^T
            }
            const float ch = -(n[0] * dx[q] + n[1] * dy_ - n[2]) * idn, cl = -(n[0] * aux[0] + n[1] * aux[1] + n[2] * aux[2]);
            const float lum = fminf(1.0f, amb + hd * fmaxf(ch, 0.0f) + ld * fmaxf(cl, 0.0f));
            const float* c = geom_rgba + 4 * __float_as_int(rec[22]);
            rgb[0] = c[0] * lum; rgb[1] = c[1] * lum; rgb[2] = c[2] * lum;
        } else {
            const float w = 0.5f + 0.5f * (aux[4] * dx[q] + aux[5] * dy_ - aux[6]) * idn;
#pragma unroll
            for (int j = 0; j < 3; j++) rgb[j] = light[12 + j] + (light[8 + j] - light[12 + j]) * w;
        }
#pragma unroll
        for (int j = 0; j < 3; j++) col[3 * q + j] = (unsigned char)(fminf(fmaxf(rgb[j], 0.0f), 1.0f) * 255.0f + 0.5f);
    }
    unsigned char* dst = out_rgb + ((((size_t)env * ncam_sel + cs) * H + py) * W + px) * 3;
    if (px + 3 < W && (W & 3) == 0) {
        unsigned int w3[3];
#pragma unroll
        for (int j = 0; j < 3; j++) w3[j] = col[4 * j] | (col[4 * j + 1] << 8) | (col[4 * j + 2] << 16) | ((unsigned)col[4 * j + 3] << 24);
        unsigned int* d32 = reinterpret_cast<unsigned int*>(dst);
        d32[0] = w3[0]; d32[1] = w3[1]; d32[2] = w3[2];
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (px + q < W) { dst[3 * q] = col[3 * q]; dst[3 * q + 1] = col[3 * q + 1]; dst[3 * q + 2] = col[3 * q + 2]; }
    }
    }
}

// grid (bins_x * bins_y, ncam_sel, N), block 256 = 4 wavefronts.  A block owns a bin of bin_tx x BIN_TY tiles (bin_width: equally wide columns of at most BIN_TX tiles).
// Round 6: the bin's list and everything the tiles read of its polyhedra live in LDS.  Until round 5 a tile read, per list entry, the record's
// octagon from global memory, and per entry it then cast the record through scalar loads and -- dependent on those -- one face, one face box and one
// silhouette edge per lane: five to seven dependent L2 round trips per tile, with the VALU busy 40 % of the time at four waves per SIMD.
//   A  wave 0: one record per lane in front-to-back order, kept if its screen octagon (box + the extents of x + y and x - y) meets the bin's
//      rectangle; ordered ballot compaction -> cand[].
//   B  all four waves, a candidate each: the record's header (octagon, nearest depth, a packed word) into hA / hB / hC; a polyhedron's faces in
//      camera-ray form, the faces' screen boxes and its silhouette edges (one per lane: at most 64 faces, 32 vertices) into a slice of `arena`
//      (an LDS atomic hands out the slices; a polyhedron that does not fit stays in global memory and the tiles read it from there as before),
//      and the bin-level rejection on the values just loaded: a silhouette edge with the bin's four corner rays outside (general path: a face seen
//      from outside with the four corner rays on its outer side) drops the entry -- its header gets an empty box.
//   C  the waves take the bin's tiles from an LDS counter (a tile under the arm costs several times an empty one) and cast each against the list.
#ifndef AVSIM_RD_WAVES
#define AVSIM_RD_WAVES 1
#endif
template <bool RGB>
__global__ void __launch_bounds__(256, RGB ? 1 : AVSIM_RD_WAVES) k_render_depth(const float* __restrict__ recs, const int* __restrict__ counts, const float4* __restrict__ heads,
                                                      const float4* __restrict__ tplanes, const float4* __restrict__ fboxes, const float4* __restrict__ sedges, int nplane, int nedge,
                                                      const float* __restrict__ cam_fovy, const int* __restrict__ cam_ids, int ncam_sel, int ngeom, int H,
                                                      int W, float znear, float zfar, float* __restrict__ out, const float* __restrict__ geom_rgba, const float* __restrict__ light,
                                                      const float* __restrict__ camaux, unsigned char* __restrict__ out_rgb, const int bin_tx) {
    __shared__ float4 hA[BL_MAX], hB[BL_MAX], hC[BL_MAX];
    __shared__ float4 arena[ARENA4];
    __shared__ int ncand, arena_top, next_tile;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H, bins_x = (tiles_x + bin_tx - 1) / bin_tx, bins_y = (tiles_y + BIN_TY - 1) / BIN_TY;
    // One block per bin.  (Persistent blocks that walk the bins with the grid's stride were tried in round 6 and are SLOWER, 15.1 against 11.1 ms per
    // 4096 envs x 4 cameras: on gfx9 stores and loads share vmcnt, so the next bin's first dependent load waits for the previous bin's tile stores
    // to drain -- with a block per bin that drain overlaps with the other blocks of the CU.)
    const int gbin = blockIdx.x;
    const int bin = gbin % (bins_x * bins_y), view = gbin / (bins_x * bins_y), cs = view % ncam_sel, env = view / ncam_sel;
    const int btx = (bin % bins_x) * bin_tx, bty = (bin / bins_x) * BIN_TY;
    const float scale = 2.0f * cam_fovy[cam_ids[cs]] / (float)H;     // cam_fovy holds tan(fovy / 2)
    const float* R = recs + ((size_t)env * ncam_sel + cs) * ngeom * REC_W;
    const size_t cb = (size_t)env * ncam_sel + cs;
    if (wave == 0) {
        // A: the camera's list headers, front to back, one per lane (coalesced: 64 B each); kept when the octagon meets the bin's rectangle and
        // the record's bin mask (k_render_geoms: no silhouette edge has the bin's four corner rays outside) has this bin's bit
        const int bpx0 = btx * TILE_W, bpy0 = bty * TILE_H;
        const int bpx1 = bpx0 + bin_tx * TILE_W < W ? bpx0 + bin_tx * TILE_W : W, bpy1 = bpy0 + BIN_TY * TILE_H < H ? bpy0 + BIN_TY * TILE_H : H;
        const float xl = (bpx0 - 0.5f * W) * scale, xr = (bpx1 - 0.5f * W) * scale, yt = -(bpy0 - 0.5f * H) * scale, yb = -(bpy1 - 0.5f * H) * scale;
#if defined(AVSIM_RDBG) && AVSIM_RDBG == 1
        const int cnt = 0;          // (experiment: no list at all -- the far plane is stored)
#else
        const int cnt = counts[cb];
#endif
        const float4* HD = heads + cb * ngeom * 4;
        const int mbit = bins_x * bins_y <= 128 ? bin : -1;
        int n = 0;
        for (int k0 = 0; k0 < cnt; k0 += 64) {
            bool keep = false;
            float4 ba, bo, hc;
            if (k0 + lane < cnt) {
                const float4* hd = HD + (size_t)(k0 + lane) * 4;
                ba = hd[0]; bo = hd[1]; hc = hd[2];
                const float4 mk = hd[3];
                const int w = mbit < 0 ? -1 : __float_as_int(mbit < 32 ? mk.x : (mbit < 64 ? mk.y : (mbit < 96 ? mk.z : mk.w)));
                keep = ba.x <= xr && ba.y >= xl && ba.z <= yt && ba.w >= yb && bo.x <= xr + yt && bo.y >= xl + yb && bo.z <= xr - yb && bo.w >= xl - yt && ((w >> (mbit & 31)) & 1);
            }
            const unsigned long long bal = __ballot(keep);
            if (keep) { const int at = n + __popcll(bal & ((1ull << lane) - 1ull)); if (at < BL_MAX) { hA[at] = ba; hB[at] = bo; hC[at] = hc; } }
            n += __popcll(bal);
        }
        if (lane == 0) { ncand = n < BL_MAX ? n : BL_MAX; arena_top = 0; next_tile = 0; }
    }
    __syncthreads();
#if defined(AVSIM_RDBG) && AVSIM_RDBG == 2
    const int cnt = 0;              // (experiment: list made, nothing staged or cast)
#else
    const int cnt = ncand;
#endif
    // B: the listed polyhedra into LDS, a candidate per wave and two at a time, so that the loads of both are in flight together: the faces in
    // inverse-depth form (1 / t of a pixel ray is AFFINE in the pixel: s = (a x + b y - c) / no), their screen boxes, the silhouette edges
    for (int i0 = wave; i0 < cnt; i0 += 8) {
        // (two candidates spelled out: arrays indexed by the unrolled loop variable went to scratch)
#define AVS_STAGE_LOAD(I, FL, BB, EG, ST, OFF, NP, NS)                                                                                          \
        float4 FL = make_float4(0.f, 0.f, 0.f, 1.f), BB = FL, EG = FL;                                                                          \
        bool ST = false;                                                                                                                       \
        int OFF = 0, NP = 0, NS = 0;                                                                                                           \
        if ((I) < cnt) {                                                                                                                       \
            const float4 hc = hC[I];                                                                                                           \
            const int pk = __builtin_amdgcn_readfirstlane(__float_as_int(hc.y)), pe = __builtin_amdgcn_readfirstlane(__float_as_int(hc.w));    \
            NP = (pk >> 6) & 127; NS = (pk >> 13) & 127;                                                                                       \
            if ((pk & 15) == 7 && ((pk >> 4) & 1) == 0 && NP <= 64 && NS <= 64) {                                                              \
                int o = 0;                                                                                                                     \
                if (lane == 0) o = atomicAdd(&arena_top, 2 * NP + NS);                                                                         \
                OFF = __builtin_amdgcn_readfirstlane(o);                                                                                       \
                ST = OFF + 2 * NP + NS <= ARENA4;                                                                                              \
                if (ST) {                                                                                                                      \
                    const int poff = pe & 0xffff, eoff = (pe >> 16) & 0xffff;                                                                  \
                    FL = (tplanes + cb * nplane + poff)[lane < NP ? lane : 0];                                                                 \
                    BB = (fboxes + cb * nplane + poff)[lane < NP ? lane : 0];                                                                  \
                    EG = (sedges + cb * nedge + eoff)[lane < NS ? lane : 0];                                                                   \
                }                                                                                                                              \
            }                                                                                                                                  \
        }
#define AVS_STAGE_STORE(I, FL, BB, EG, ST, OFF, NP, NS)                                                                                         \
        if (ST) {                                                                                                                              \
            if (lane < NP) {                                                                                                                   \
                const float iw = __builtin_amdgcn_rcpf(FL.w);                                                                                  \
                arena[OFF + lane] = make_float4(FL.x * iw, FL.y * iw, -FL.z * iw, FL.w);                                                       \
                arena[OFF + NP + lane] = BB;                                                                                                   \
            }                                                                                                                                  \
            if (lane < NS) arena[OFF + 2 * NP + lane] = EG;                                                                                    \
            if (lane == 0) {                                                                                                                   \
                float4 hc = hC[I];                                                                                                             \
                hc.y = __int_as_float(__float_as_int(hc.y) | (1 << 5) | (OFF << 20));                                                          \
                hC[I] = hc;                                                                                                                    \
            }                                                                                                                                  \
        }
        AVS_STAGE_LOAD(i0, flA, bbA, egA, stA, offA, npA, nsA)
        AVS_STAGE_LOAD(i0 + 4, flB, bbB, egB, stB, offB, npB, nsB)
        AVS_STAGE_STORE(i0, flA, bbA, egA, stA, offA, npA, nsA)
        AVS_STAGE_STORE(i0 + 4, flB, bbB, egB, stB, offB, npB, nsB)
#undef AVS_STAGE_LOAD
#undef AVS_STAGE_STORE
    }
    __syncthreads();
    const int ntile = bin_tx * BIN_TY;
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(&next_tile, 1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= ntile) break;
        const int tix = btx + (t % bin_tx), tiy = bty + (t / bin_tx);
        if (tix >= tiles_x || tiy >= tiles_y) continue;
#if defined(AVSIM_RDBG) && AVSIM_RDBG == 3
        const int cast_cnt = 0;            // (experiment: list and staging made, nothing cast)
#else
        const int cast_cnt = cnt;
#endif
        render_tile<RGB>(lane, tix * TILE_W, tiy * TILE_H, cs, env, hA, hB, hC, arena, cast_cnt, R, tplanes, fboxes, sedges, nplane, nedge, scale, ncam_sel, H, W, znear, zfar, out, geom_rgba, light, camaux, out_rgb);
    }
}

// host side: float image of the geoms / cameras, scratch buffers, launches
struct RenderHost {
    RenderModel m{};
    std::vector<void*> allocs;
    float* d_xpose = nullptr;       // [N][nbody][12] body poses written by the physics kernel's forward pass
    float* d_recs = nullptr;
    int* d_counts = nullptr;
    int* d_order = nullptr;
    float* d_tplanes = nullptr;     // [N][ncam][nplane][4] faces of the polyhedra in camera-ray form
    float* d_fbox = nullptr;        // [N][ncam][nplane][4] screen boxes of the faces seen from outside
    float* d_sedge = nullptr;       // [N][ncam][nedge][4] silhouette edges (lines in the image, positive inside), compacted per polyhedron
    float* d_heads = nullptr;       // [N][ncam][ngeom][16] the kept records' list headers in front-to-back order (octagon, nearest depth, packed word, record, offsets, bin mask)
    float* d_camaux = nullptr;      // [N][16][8] light direction and world up axis in the camera frame
    int* d_cam_ids = nullptr;
    size_t recs_cap = 0, counts_cap = 0;
    int N = 0;
    // option "kernel_timing": HIP events around every k_render_depth launch on the launch stream (avsim_render_kernel_time)
    bool timing = false;
    int env_chunk = 4096;           // option "render_chunk": envs per pass of the two kernels (launch()): bounds the scratch of a bigger batch
    std::vector<hipEvent_t> tev;
    size_t tev_used = 0;
    double tev_ms = 0;             // launches folded out of the event list (at most 1024 pairs are kept: a long run that renders every step with
    long long tev_n = 0;           // "kernel_timing" on does not grow it)

    template <typename T>
    T* up(const std::vector<T>& v) {
        void* p = nullptr;
        if (hipMalloc(&p, (v.size() ? v.size() : 1) * sizeof(T)) != hipSuccess) throw std::runtime_error("hipMalloc failed while uploading the render model");
        if (v.size() && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("hipMemcpy failed while uploading the render model");
        allocs.push_back(p);
        return (T*)p;
    }
    static std::vector<float> tofloat(const std::vector<double>& v) { return std::vector<float>(v.begin(), v.end()); }
    static void quat2mat(const double* q, double* R) {
        double w = q[0], x = q[1], y = q[2], z = q[3];
        R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
        R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
        R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
    }
    // The convex polyhedra the rasteriser draws: vertices, face planes, the vertices of every face and the edges with their two
    // faces.  Mesh hulls bring theirs in the blob (compiler/hull.py hull_topology: qhull's own triangulation, shared by the geoms of
    // one mesh); boxes are made here from their half extents.
    void build_polyhedra(const Blob& b) {
        auto gt = b.i("geom_type"), gv = b.i("geom_visible"), gh = b.i("geom_hull"), hp = b.i("geom_hplane"), he = b.i("geom_hedge");
        auto hfa = b.i("hull_face_vadr"), hfn = b.i("hull_face_vnum"), hfi = b.i("hull_face_vidx"), hed = b.i("hull_edge");
        auto gs = b.f("geom_size"), hv = b.f("hull_vert"), hpl = b.f("hull_plane");
        std::vector<float> rvert, rplane;
        std::vector<int> fvadr, fvnum, fvidx, redge, rg((size_t)m.ngeom * RG_W, 0);
        std::vector<unsigned> fvmask;
        std::map<std::pair<int, int>, std::array<int, 6>> cache;
        m.nplane = 0; m.nedge = 0;
        for (int g = 0; g < m.ngeom; g++) {
            if (!gv[g] || (gt[g] != 7 && gt[g] != 6)) continue;
            const std::pair<int, int> key = gt[g] == 7 ? std::make_pair(gh[2 * g], hp[2 * g]) : std::make_pair(-1, g);
            auto it = cache.find(key);
            if (it == cache.end()) {
                std::array<int, 6> a{(int)rvert.size() / 3, 0, (int)rplane.size() / 4, 0, (int)redge.size() / 4, 0};
                if (gt[g] == 7) {
                    a[1] = gh[2 * g + 1]; a[3] = hp[2 * g + 1]; a[5] = he[2 * g + 1];
                    for (int k = 0; k < 3 * a[1]; k++) rvert.push_back((float)hv[3 * (size_t)gh[2 * g] + k]);
                    for (int k = 0; k < 4 * a[3]; k++) rplane.push_back((float)hpl[4 * (size_t)hp[2 * g] + k]);
                    for (int f = 0; f < a[3]; f++) {
                        fvadr.push_back((int)fvidx.size());
                        fvnum.push_back(hfn[hp[2 * g] + f]);
                        unsigned msk = 0;
                        for (int j = 0; j < hfn[hp[2 * g] + f]; j++) { const int v = hfi[hfa[hp[2 * g] + f] + j]; fvidx.push_back(v); if (v < 32) msk |= 1u << v; }
                        fvmask.push_back(msk);
                    }
                    for (int k = 0; k < 4 * a[5]; k++) redge.push_back(hed[4 * (size_t)he[2 * g] + k]);
                } else {
                    // box: vertex k = corner with bit c of k choosing the sign on axis c; face 2 c + s = axis c, side s
                    a[1] = 8; a[3] = 6; a[5] = 12;
                    const double e[3] = {gs[3 * g], gs[3 * g + 1], gs[3 * g + 2]};
                    for (int k = 0; k < 8; k++) for (int c = 0; c < 3; c++) rvert.push_back((float)(((k >> c) & 1) ? e[c] : -e[c]));
                    for (int c = 0; c < 3; c++)
                        for (int sd = 0; sd < 2; sd++) {
                            float n[4] = {0, 0, 0, (float)e[c]};
                            n[c] = sd ? 1.0f : -1.0f;
                            for (float x : n) rplane.push_back(x);
                            fvadr.push_back((int)fvidx.size());
                            fvnum.push_back(4);
                            unsigned msk = 0;
                            for (int k = 0; k < 8; k++) if (((k >> c) & 1) == sd) { fvidx.push_back(k); msk |= 1u << k; }
                            fvmask.push_back(msk);
                        }
                    for (int u = 0; u < 8; u++)
                        for (int c = 0; c < 3; c++) {
                            const int v = u | (1 << c);
                            if (v == u) continue;
                            // the edge along axis c: its two faces are the sides of the other two axes that u and v share
                            int f[2], nf = 0;
                            for (int o = 0; o < 3; o++) if (o != c) f[nf++] = 2 * o + ((u >> o) & 1);
                            redge.push_back(u); redge.push_back(v); redge.push_back(f[0]); redge.push_back(f[1]);
                        }
                }
                it = cache.emplace(key, a).first;
            }
            const auto& a = it->second;
            int* G = &rg[(size_t)g * RG_W];
            for (int k = 0; k < 6; k++) G[k] = a[k];
            G[6] = m.nplane; G[7] = m.nedge;
            m.nplane += a[3]; m.nedge += a[5];
        }
        m.rg = up(rg); m.r_vert = up(rvert); m.r_plane = up(rplane); m.r_fvadr = up(fvadr); m.r_fvnum = up(fvnum); m.r_fvidx = up(fvidx); m.r_edge = up(redge); m.r_fvmask = up(fvmask);
    }
    void build(const Blob& b, int N_) {
        N = N_;
        m.ngeom = b.scalar("ngeom"); m.nbody = b.scalar("nbody");
        auto cam_body = b.i("cam_body");
        m.ncam = (int)cam_body.size();
        m.geom_type = up(b.i("geom_type")); m.geom_body = up(b.i("geom_body"));
        m.geom_visible = up(b.i("geom_visible")); m.cam_body = up(cam_body);
        auto gq = b.f("geom_quat"), cq = b.f("cam_quat");
        std::vector<double> gm(9 * m.ngeom), cm(9 * m.ncam);
        for (int g = 0; g < m.ngeom; g++) quat2mat(&gq[4 * g], &gm[9 * g]);
        for (int c = 0; c < m.ncam; c++) quat2mat(&cq[4 * c], &cm[9 * c]);
        m.geom_pos = up(tofloat(b.f("geom_pos"))); m.geom_mat = up(tofloat(gm)); m.geom_size = up(tofloat(b.f("geom_size")));
        m.geom_bcen = up(tofloat(b.f("geom_bcenter"))); m.geom_rbound = up(tofloat(b.f("geom_rbound")));
        build_polyhedra(b);
        m.cam_pos = up(tofloat(b.f("cam_pos"))); m.cam_mat = up(tofloat(cm));
        {   // the kernels only need tan(fovy / 2)
            auto fv = b.f("cam_fovy");
            std::vector<float> th(fv.size());
            for (size_t c = 0; c < fv.size(); c++) th[c] = (float)std::tan(0.5 * fv[c] * 3.14159265358979323846 / 180.0);
            m.cam_fovy = up(th);
        }
        m.geom_rgba = up(tofloat(b.f("geom_rgba"))); m.light = up(tofloat(b.f("render_light")));
        d_camaux = up(std::vector<float>((size_t)N * 16 * 8, 0.0f));
        auto clip = b.f("cam_clip");
        m.znear = (float)clip[0]; m.zfar = (float)clip[1];
        d_xpose = up(std::vector<float>((size_t)N * m.nbody * 12, 0.0f));
    }
    void destroy() {
        for (void* p : allocs) (void)hipFree(p);
        allocs.clear();
        for (auto& ev : tev) (void)hipEventDestroy(ev);
        tev.clear(); tev_used = 0;
        if (d_recs) (void)hipFree(d_recs);
        if (d_counts) (void)hipFree(d_counts);
        if (d_order) (void)hipFree(d_order);
        if (d_tplanes) (void)hipFree(d_tplanes);
        if (d_fbox) (void)hipFree(d_fbox);
        if (d_sedge) (void)hipFree(d_sedge);
        if (d_heads) (void)hipFree(d_heads);
        if (d_cam_ids) (void)hipFree(d_cam_ids);
        d_heads = nullptr; d_recs = nullptr; d_counts = nullptr; d_order = nullptr; d_tplanes = nullptr; d_fbox = nullptr; d_sedge = nullptr; d_cam_ids = nullptr; recs_cap = 0; counts_cap = 0;
    }
    // d_out: device float[N][ncam_sel][H][W], or (rgb) u8[N][ncam_sel][H][W][3]; body poses must already be in d_xpose (same stream)
    int launch(hipStream_t st, const int* cam_ids_host, int ncam_sel, int H, int W, void* d_out, bool rgb, std::string& err) {
        if (ncam_sel < 1 || ncam_sel > 16 || H < 1 || W < 1 || H > 4096 || W > 4096) { err = "avsim_render_depth: bad camera count or image size"; return -1; }
        for (int c = 0; c < ncam_sel; c++)
            if (cam_ids_host[c] < 0 || cam_ids_host[c] >= m.ncam) { err = "avsim_render_depth: camera index out of range"; return -1; }
        if (m.ngeom > BL_MAX || m.nplane >= 65536 || m.nedge >= 65536) { err = "avsim_render_depth: the model has more geoms than a bin's list holds (BL_MAX), or more faces / edges than the list header's 16-bit offsets"; return -1; }
        // The envs go through the two kernels in chunks of at most env_chunk (4096): a view's records, face planes, face boxes and silhouette
        // edges are ~100 KB -- 1.6 GB of scratch per 4096 envs x 4 cameras -- and the allocation is the chunk's, not the batch's.  (Smaller chunks,
        // whose scratch would stay in the Infinity Cache between the two kernels, are SLOWER: 15.5 ms per 4096 envs in one pass, 15.9 in four,
        // 17.1 in sixteen -- k_render_depth is not waiting for those reads, and every pass has a tail.)
        const int chunk = N < env_chunk ? N : env_chunk;
        const size_t need = (size_t)chunk * ncam_sel * m.ngeom * REC_W;
        // d_counts is sized by the chunk alone (16 camera slots per env), the other buffers by chunk x cameras: a larger chunk with fewer cameras can
        // leave `need` where it was, so the chunk size that was allocated for is tracked on its own
        if (need > recs_cap || (size_t)chunk > counts_cap) {
            if (d_recs) (void)hipFree(d_recs);
            if (d_counts) (void)hipFree(d_counts);
            if (d_order) (void)hipFree(d_order);
            if (d_tplanes) (void)hipFree(d_tplanes);
            if (d_fbox) (void)hipFree(d_fbox);
            if (d_sedge) (void)hipFree(d_sedge);
            if (d_heads) (void)hipFree(d_heads);
            d_heads = nullptr; d_recs = nullptr; d_counts = nullptr; d_order = nullptr; d_tplanes = nullptr; d_fbox = nullptr; d_sedge = nullptr; recs_cap = 0; counts_cap = 0;
            if (hipMalloc((void**)&d_recs, need * sizeof(float)) != hipSuccess || hipMalloc((void**)&d_counts, (size_t)chunk * 16 * sizeof(int)) != hipSuccess ||
                hipMalloc((void**)&d_order, (size_t)chunk * ncam_sel * m.ngeom * sizeof(int)) != hipSuccess ||
                hipMalloc((void**)&d_tplanes, (size_t)chunk * ncam_sel * (m.nplane + 1) * 4 * sizeof(float)) != hipSuccess ||
                hipMalloc((void**)&d_fbox, (size_t)chunk * ncam_sel * (m.nplane + 1) * 4 * sizeof(float)) != hipSuccess ||
                hipMalloc((void**)&d_sedge, (size_t)chunk * ncam_sel * (m.nedge + 1) * 4 * sizeof(float)) != hipSuccess ||
                hipMalloc((void**)&d_heads, (size_t)chunk * ncam_sel * m.ngeom * 16 * sizeof(float)) != hipSuccess) {
                err = "hipMalloc(render records) failed";
                return -3;
            }
            recs_cap = need;
            counts_cap = (size_t)chunk;
        }
        if (!d_cam_ids && hipMalloc((void**)&d_cam_ids, 16 * sizeof(int)) != hipSuccess) { err = "hipMalloc(camera ids) failed"; return -3; }
        if (hipMemcpyAsync(d_cam_ids, cam_ids_host, ncam_sel * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) { err = "camera id copy failed"; return -3; }
        const int bin_tx = bin_width((W + TILE_W - 1) / TILE_W);
        const int tiles = (((W + TILE_W - 1) / TILE_W + bin_tx - 1) / bin_tx) * (((H + TILE_H - 1) / TILE_H + BIN_TY - 1) / BIN_TY);     // bins
        if (timing) {
            if (tev_used >= 2 * 1024) {
                for (size_t i = 0; i + 1 < tev_used; i += 2) {
                    float ms = 0;
                    (void)hipEventSynchronize(tev[i + 1]);
                    if (hipEventElapsedTime(&ms, tev[i], tev[i + 1]) == hipSuccess) tev_ms += ms;
                }
                tev_n += (long long)(tev_used / 2);
                tev_used = 0;
            }
            while (tev.size() < tev_used + 2) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) { err = "hipEventCreate failed"; return -3; } tev.push_back(ev); }
        }
        for (int e0 = 0; e0 < N; e0 += chunk) {
            const int n = N - e0 < chunk ? N - e0 : chunk;
            const long long total_ll = (long long)tiles * ncam_sel * n;
            if (total_ll > 0x7fffffffLL) { err = "avsim_render_depth: too many bins in one pass (lower render_chunk)"; return -1; }
            const int nblk = (int)total_ll;
            const float* xp = (const float*)d_xpose + (size_t)e0 * m.nbody * 12;
            float* aux = d_camaux + (size_t)e0 * 16 * 8;
            hipLaunchKernelGGL(k_render_geoms, dim3(ncam_sel, n), dim3(64), 0, st, m, xp, (const int*)d_cam_ids, ncam_sel, H, W, d_recs, d_counts, d_order, d_tplanes, aux, d_fbox, d_sedge);
            hipLaunchKernelGGL(k_render_heads, dim3(ncam_sel * n), dim3(256), 0, st, (const float*)d_recs, (const int*)d_counts, (const int*)d_order, (const float4*)d_sedge, m.nedge, m.cam_fovy,
                               (const int*)d_cam_ids, ncam_sel, m.ngeom, H, W, (float4*)d_heads, bin_tx);
            if (timing && e0 == 0) (void)hipEventRecord(tev[tev_used], st);       // (the image kernel's time; with more than one chunk the later chunks' set-up kernels are inside)
            const size_t px = (size_t)e0 * ncam_sel * H * W;
            if (rgb)
                hipLaunchKernelGGL(k_render_depth<true>, dim3(nblk), dim3(256), 0, st, (const float*)d_recs, (const int*)d_counts, (const float4*)d_heads, (const float4*)d_tplanes, (const float4*)d_fbox, (const float4*)d_sedge, m.nplane, m.nedge,
                                   m.cam_fovy, (const int*)d_cam_ids, ncam_sel, m.ngeom, H, W, m.znear, m.zfar, (float*)nullptr, m.geom_rgba, m.light, (const float*)aux, (unsigned char*)d_out + px * 3, bin_tx);
            else
                hipLaunchKernelGGL(k_render_depth<false>, dim3(nblk), dim3(256), 0, st, (const float*)d_recs, (const int*)d_counts, (const float4*)d_heads, (const float4*)d_tplanes, (const float4*)d_fbox, (const float4*)d_sedge, m.nplane, m.nedge,
                                   m.cam_fovy, (const int*)d_cam_ids, ncam_sel, m.ngeom, H, W, m.znear, m.zfar, (float*)d_out + px, m.geom_rgba, m.light, (const float*)aux, (unsigned char*)nullptr, bin_tx);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { err = std::string("render kernel launch: ") + hipGetErrorString(e); return -3; }
        if (timing) { (void)hipEventRecord(tev[tev_used + 1], st); tev_used += 2; }
#ifdef AVSIM_RENDER_STATS
        {
            unsigned long long hst[8];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpyFromSymbol(hst, HIP_SYMBOL(g_rstat), sizeof hst);
            fprintf(stderr, "render stats per tile: list %.1f, box hits %.2f, hulls cast %.2f (entry faces %.2f, veto faces %.2f), primitives %.2f; tiles %llu\n",
                    (double)hst[1] / hst[0], (double)hst[2] / hst[0], (double)hst[3] / hst[0], (double)hst[4] / hst[0], (double)hst[5] / hst[0], (double)hst[6] / hst[0], hst[0]);
            memset(hst, 0, sizeof hst);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rstat), hst, sizeof hst);
        }
#endif
        return 0;
    }
};

}  // namespace avs
