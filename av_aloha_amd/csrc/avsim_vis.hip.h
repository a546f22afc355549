// avsim_vis.hip.h -- colour images of the VISUAL meshes: what the reference's cameras show through MuJoCo's
// OpenGL pipeline (gym_guided_vision/gym_guided_vision/env.py:180-188 get_obs "pixels", :195-200 render) -- the robot's visual
// meshes (assets/aloha_sim.xml class "visual"), the frame and the table with its texture (assets/scene.xml), the task objects --
// as a software triangle rasteriser, one image per (env, camera).
//
// Scene = the model's instances of the decimated mesh library (compiler/vismesh.py: <= 20 k triangles per scene), expanded once on
// the host into body-frame vertices and triangles.  One workgroup (16 wavefronts, one per CU) per view, persistent over the views:
//   1. camera-from-body transforms into LDS, every vertex into the camera frame (global scratch of the workgroup's slot);
//   2. triangle set-up, one triangle per thread: clipped against the near plane (a camera sits inside its own mount), projected,
//      its three edge functions and the plane of 1 / depth normalised into a 16-float record, flat Lambert shade (headlight +
//      the scene's directional light, the terms of avsim_render.hip.h's proxy image) folded into an rgb8 colour;
//   3. binning into 8 x 8 pixel tiles: count (LDS counters, tile-vs-edge test) -> prefix sum -> fill; big triangles by tile rows (vis_bin);
//   4. one wavefront per tile, lane = pixel: the tile's records are wave-uniform SCALAR loads, the depth test is on 1 / depth with the
//      record index as tie-break (the image does not depend on the order of the lists); the winner's colour -- for the textured
//      table through the triangle's texture-coordinate planes (set-up) -- or the sky gradient.
// Back faces are culled.  Round 5, both optional (avsim_set_option "render_shadows", "render_samples"): SHADOWS of the scene's directional light
// (scene.xml:48) from a depth map rendered from the light, one per env (k_vis_shadow: heights above the plane normal to the light over the
// light's shadow box, 512 x 512 texels, atomicMax of ordered keys), looked up per sample; and 2 x 2 SUPERSAMPLING (MuJoCo's offscreen buffer
// is multisampled, <quality offsamples> default 4 [EXT]): every lane tests the four samples at +-1/4 pixel of its pixel against the tile's
// records and averages their colours.  The light's specular term (Blinn, viewer at infinity) is part of the flat shade.  No per-vertex lighting or transparency: parity with the reference's OpenGL pixels is unpinned.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "avsim_model.h"

namespace avs {

#ifndef VIS_THREADS_N
#define VIS_THREADS_N 1024
#endif
// One workgroup of VIS_THREADS per view, 16 wavefronts per CU in all.  Sixteen wavefronts on ONE view (1024 threads, one workgroup per CU) rather
// than four views of four: a view in flight holds ~0.5 MB of records, lists and camera-frame vertices next to its 0.9 MB image, and 4 x 256
// of them overflow the 256 MB Infinity Cache -- the tile stage's dependent loads then wait for HBM (6.7 ms per 1024 views against 3.6 ms per
// 512: profiles/r05_experiments.txt section 8).
constexpr int VIS_THREADS = VIS_THREADS_N, VIS_WG_PER_CU = 1024 / VIS_THREADS_N, VIS_SHADOW_THREADS = 1024;
constexpr int VIS_TILE = 8, VIS_MAXBODY = 64, VIS_MAXTILES = 16384, VIS_REC = 16, VIS_TEXCAP = 512;

struct VisScene {
    int nvert, ntri, nbody, ncam;
    const float* vert;      // [nvert][3] body frame
    const int* vbody;       // [nvert]
    const int* tri;         // [ntri][3]
    const float* rgb;       // [ntri][3]
    const float* uv;        // [ntri][6]
    const int* tex;         // [ntri]
    const float* tnorm;     // [ntri][9] the corners' lighting normals, body frame (compiler/vismesh.py corner_normals); null: a library without them
    int smooth;             // option "render_smooth" (default 1 where the library has the normals): light the corners and interpolate (Gouraud, as MuJoCo's fixed-function GL [EXT]); 0: one shade per triangle
    const unsigned* texel;  // [VIS_TEX][VIS_TEX] r | g << 8 | b << 16, row 0 = top
    int texn;
    const int* cam_body;
    const float *cam_pos, *cam_mat, *cam_fovy;   // cam_fovy: tan(fovy / 2)
    const float* light;     // as RenderModel::light; [3] half extent of the light's shadow box (0: none), [7] [11] [15] its centre (world)
    float znear;
    int shn;                // texels per side of a shadow map (option "render_shadow_size": 512 default, 1024, 2048)
    const unsigned* shmap;  // shadow maps [N][shn][shn]: ordered keys of the largest height towards the light (null: no shadows)
    float le1[3], le2[3], lw[3];   // light frame: e1, e2 span the plane normal to the unit light direction lw (oracle/orc_vis.c light_frame)
    float sh_s0, sh_t0, sh_itex;   // light-space corner of the shadow box, texels per metre
    float spec_k, spec_n;          // the directional light's specular term: light specular x material specular, exponent (render_light[16], [17]; 0: none)
};
constexpr int VIS_SM = 512;       // default side of the shadow map
__device__ __host__ inline unsigned vis_hkey(float f) { unsigned b; memcpy(&b, &f, 4); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }      // order-preserving
__device__ inline float vis_hval(unsigned k) { const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; return __uint_as_float(b); }

// per-slot scratch (a workgroup's view in flight)
struct VisScratch {
    float4* vcam;     // [slots][nvert]
    float4* rec;      // [slots][reccap][4]
    int* bbox;        // [slots][reccap][4] tile ranges
    int* list;        // [slots][listcap]
    int* bigq;        // [slots][reccap]: the records whose boxes cover more than 16 tiles (vis_bin)
    float4* grec;     // [slots][reccap][3]: smooth shading: per record the planes of shade x w, shade-without-the-light x w, specular x w over the image (w = 1 / depth): value at a sample = plane / w
    float4* trec;     // [slots][VIS_TEXCAP][3]: per textured triangle the planes U, V, D over the image: texture coordinate = U / D, V / D at a sample
    int* flags;       // [nviews][8]: overflow bits (0: records, 1: lists), shader-clock cycles / 1024 of the five stages, records, list entries
    int reccap, listcap;
};

__device__ inline unsigned vis_pack(float r, float g, float b) {
    const unsigned R = (unsigned)(fminf(fmaxf(r, 0.0f), 1.0f) * 255.0f + 0.5f), G = (unsigned)(fminf(fmaxf(g, 0.0f), 1.0f) * 255.0f + 0.5f),
                   B = (unsigned)(fminf(fmaxf(b, 0.0f), 1.0f) * 255.0f + 0.5f);
    return R | (G << 8) | (B << 16);
}

// is the tile [x0, x0 + 8) x [y0, y0 + 8) (pixel centres) entirely outside one of the triangle's edges?
// (sample positions: the pixel centres, and with supersampling 1/4 pixel either side of them: the tests take the wider range)
__device__ inline bool vis_tile_outside(const float4 r0, const float4 r1, const float4 r2, float x0, float y0) {
    const float xa = x0 + 0.25f, xb = x0 + VIS_TILE - 0.25f, ya = y0 + 0.25f, yb = y0 + VIS_TILE - 0.25f;
    const float e0 = r0.x * (r0.x > 0 ? xb : xa) + r0.y * (r0.y > 0 ? yb : ya) + r0.z;
    const float e1 = r0.w * (r0.w > 0 ? xb : xa) + r1.x * (r1.x > 0 ? yb : ya) + r1.y;
    const float e2 = r1.z * (r1.z > 0 ? xb : xa) + r1.w * (r1.w > 0 ? yb : ya) + r2.x;
    return e0 < 0 || e1 < 0 || e2 < 0;
}

// is the tile entirely INSIDE the triangle (all three edge functions >= 0 at every pixel centre)?  Such an entry needs no edge tests.
__device__ inline bool vis_tile_inside(const float4 r0, const float4 r1, const float4 r2, float x0, float y0) {
    const float xa = x0 + 0.25f, xb = x0 + VIS_TILE - 0.25f, ya = y0 + 0.25f, yb = y0 + VIS_TILE - 0.25f;
    const float e0 = r0.x * (r0.x > 0 ? xa : xb) + r0.y * (r0.y > 0 ? ya : yb) + r0.z;
    const float e1 = r0.w * (r0.w > 0 ? xa : xb) + r1.x * (r1.x > 0 ? ya : yb) + r1.y;
    const float e2 = r1.z * (r1.z > 0 ? xa : xb) + r1.w * (r1.w > 0 ? ya : yb) + r2.x;
    return e0 >= 0 && e1 >= 0 && e2 >= 0;
}

// Scalar loads for the tile stage.  A tile's records are the same for all 64 lanes (pixels) of its wave: fetched by the SCALAR unit they
// land in SGPRs, where the per-pixel arithmetic reads them as operands -- no vector load + v_readlane per word (15 of the ~35 VALU
// instructions per tile-list entry until round 4).  The compiler does not emit s_load for memory the kernel itself has written, and it
// assumes an inline asm's outputs are ready when the asm ends: so every asm block below issues its loads AND waits for them
// (s_waitcnt lgkmcnt(0)), four records at a time so that the round trip is paid once per four entries; the scalar cache is invalidated
// once per view (s_dcache_inv), after the barrier that follows the vector stores of the records and lists.
typedef int vis_v4i __attribute__((ext_vector_type(4)));
typedef int vis_v16i __attribute__((ext_vector_type(16)));
__device__ inline const void* vis_uni(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    return (const void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v));
}
__device__ inline vis_v4i vis_sload4(const void* p) {
    vis_v4i r;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(vis_uni(p)) : "memory");
    return r;
}
__device__ inline void vis_sload16x4(const void* p0, const void* p1, const void* p2, const void* p3, vis_v16i& r0, vis_v16i& r1, vis_v16i& r2, vis_v16i& r3) {
    asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %5, 0x0\n\ts_load_dwordx16 %2, %6, 0x0\n\ts_load_dwordx16 %3, %7, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3) : "s"(vis_uni(p0)), "s"(vis_uni(p1)), "s"(vis_uni(p2)), "s"(vis_uni(p3)) : "memory");
}

// the twelve words of four records that the tile loop reads (three edge functions and the plane of 1 / depth; words 12-15 -- colours, bias --
// are fetched for the winner only): 48 SGPRs instead of 64 per batch, which the kernel's other wave-uniform values no longer spill around
typedef int vis_v8i __attribute__((ext_vector_type(8)));
typedef float vis_f2 __attribute__((ext_vector_type(2)));
struct VisRec12 { vis_v8i a; vis_v4i b; };
__device__ inline void vis_sload12x4(const void* p0, const void* p1, const void* p2, const void* p3, VisRec12& r0, VisRec12& r1, VisRec12& r2, VisRec12& r3) {
    asm volatile("s_load_dwordx8 %0, %8, 0x0\n\ts_load_dwordx4 %1, %8, 0x20\n\ts_load_dwordx8 %2, %9, 0x0\n\ts_load_dwordx4 %3, %9, 0x20\n\t"
                 "s_load_dwordx8 %4, %10, 0x0\n\ts_load_dwordx4 %5, %10, 0x20\n\ts_load_dwordx8 %6, %11, 0x0\n\ts_load_dwordx4 %7, %11, 0x20\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(r0.a), "=&s"(r0.b), "=&s"(r1.a), "=&s"(r1.b), "=&s"(r2.a), "=&s"(r2.b), "=&s"(r3.a), "=&s"(r3.b)
                 : "s"(vis_uni(p0)), "s"(vis_uni(p1)), "s"(vis_uni(p2)), "s"(vis_uni(p3)) : "memory");
}

// Tile lists.  Records are taken 64 at a time by a wavefront, one per lane.  A triangle whose box covers a few tiles is walked by its
// own lane.  A big one (the table top covers every tile of the overhead view, a frame bar crosses the image) goes to a queue that the
// workgroup then works through: the entries dealt round-robin over the wavefronts (the big triangles come in runs of one mesh: the table's
// 92), a queued triangle's tile ROWS over the lanes, the row's span of tiles from the three edge functions -- a thin bar across the image has a
// box of thousands of tiles and touches a hundred.  The queue is built by the counting pass and reused by the fill pass.
// FILL = false counts (cnt[tile]++), FILL = true appends the record index at cur[tile]++.
template <bool FILL>
__device__ inline void vis_bin(const float4* __restrict__ rec, const int* __restrict__ bbox, int nrec, int* cnt, int* __restrict__ list, int listcap, int tw, int lane, int wave,
                               int& flag, int* __restrict__ bigq, int* bign) {
    auto visit = [&](const float4 q0, const float4 q1, const float4 q2, int tx, int ty, int idx) {
        if (vis_tile_outside(q0, q1, q2, (float)(tx * VIS_TILE), (float)(ty * VIS_TILE))) return;
        const int p = atomicAdd(&cnt[ty * tw + tx], 1);
        // (bit 31 of a list entry: the tile lies entirely inside the triangle, the tile stage skips the edge functions)
        if (FILL) { if (p < listcap) list[p] = idx | (vis_tile_inside(q0, q1, q2, (float)(tx * VIS_TILE), (float)(ty * VIS_TILE)) ? (int)0x80000000 : 0); else flag |= 2; }
    };
    for (int i0 = wave * 64; i0 < nrec; i0 += VIS_THREADS) {
        const int i = i0 + lane;
        if (i >= nrec) continue;
        const int4 bb = ((const int4*)bbox)[i];
        const int nt = (bb.y - bb.x + 1) * (bb.w - bb.z + 1);
        if (nt > 16) {
            if (!FILL) bigq[atomicAdd(bign, 1)] = i;        // (at most one entry per record: the queue has the records' capacity)
            continue;
        }
        const float4 r0 = rec[4 * i], r1 = rec[4 * i + 1], r2 = rec[4 * i + 2];
        for (int ty = bb.z; ty <= bb.w; ty++)
            for (int tx = bb.x; tx <= bb.y; tx++) visit(r0, r1, r2, tx, ty, i);
    }
    if (!FILL) { __threadfence_block(); __syncthreads(); }      // the queue is complete (the fill pass comes after barriers of its own)
    const int nbig = *bign;
    constexpr int NW = VIS_THREADS / 64;
    // The queue: 64 entries per wavefront at a time, one per lane (entry, record and box arrive with two round trips for the 64, not per entry),
    // then one after the other, lane = tile row of the box: the row's span of tiles follows from the three edge functions (a thin bar across
    // the image has a box of thousands of tiles and touches a hundred: walking the box was most of both passes)
    // (the entries are dealt round-robin over the wavefronts: a mesh's big triangles sit next to each other in the queue)
    for (int qb = 0; qb * NW + wave < nbig; qb += 64) {
        const int q = (qb + lane) * NW + wave, mi = q < nbig ? bigq[q] : 0;
        float4 m0 = make_float4(0, 0, 0, 0), m1 = m0, m2 = m0;
        int4 mb = make_int4(0, -1, 0, -1);
        if (q < nbig) { m0 = rec[4 * mi]; m1 = rec[4 * mi + 1]; m2 = rec[4 * mi + 2]; mb = ((const int4*)bbox)[mi]; }
        const int ne = min(64, (nbig - wave - qb * NW + NW - 1) / NW);
        for (int src = 0; src < ne; src++) {
            float4 r0, r1, r2;
            r0.x = __shfl(m0.x, src, 64); r0.y = __shfl(m0.y, src, 64); r0.z = __shfl(m0.z, src, 64); r0.w = __shfl(m0.w, src, 64);
            r1.x = __shfl(m1.x, src, 64); r1.y = __shfl(m1.y, src, 64); r1.z = __shfl(m1.z, src, 64); r1.w = __shfl(m1.w, src, 64);
            r2.x = __shfl(m2.x, src, 64); r2.y = 0; r2.z = 0; r2.w = 0;
            const int4 bb = make_int4(__shfl(mb.x, src, 64), __shfl(mb.y, src, 64), __shfl(mb.z, src, 64), __shfl(mb.w, src, 64));
            const int i = __shfl(mi, src, 64);
            const float ea[3] = {r0.x, r0.w, r1.z}, eb[3] = {r0.y, r1.x, r1.w}, ec[3] = {r0.z, r1.y, r2.x};
            for (int row = bb.z + lane; row <= bb.w; row += 64) {
                const float ya = (float)(row * VIS_TILE) + 0.25f, yb = (float)(row * VIS_TILE + VIS_TILE) - 0.25f;
                float lo = (float)(bb.x * VIS_TILE), hi = (float)(bb.y * VIS_TILE + VIS_TILE);
                bool none = false;
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    const float v = eb[e] * (eb[e] > 0 ? yb : ya) + ec[e];        // the edge function's largest value over the row, less its x term
                    const float x = -v * __builtin_amdgcn_rcpf(ea[e]);
                    if (ea[e] > 0) lo = fmaxf(lo, x);
                    else if (ea[e] < 0) hi = fminf(hi, x);
                    else none = none || v < 0;
                }
                if (none || !(lo <= hi + 0.5f)) continue;
                // tiles whose samples [8 tx + 1/4, 8 tx + 7 3/4] reach into [lo, hi] (widened by half a pixel for the rounding of the reciprocal;
                // a tile more on either side does no harm: visit() tests exactly)
                const int tx0 = max(bb.x, (int)floorf((lo - 0.5f - (VIS_TILE - 0.25f)) * (1.0f / VIS_TILE))), tx1 = min(bb.y, (int)floorf((hi + 0.5f - 0.25f) * (1.0f / VIS_TILE)));
                for (int tx = tx0; tx <= tx1; tx++) visit(r0, r1, r2, tx, row, i);
            }
        }
    }
}

// Shadow map of one env per workgroup: every triangle of the scene, in the light's frame (s, t across, h towards the light), rasterised over
// the texel centres it covers with atomicMax of its height's ordered key.  Small triangles by their own thread; a triangle whose box holds
// more than 64 texels goes to an LDS queue that the workgroup empties whenever it could overflow -- one queued triangle per wavefront, its
// box's texels over the lanes.  (Round 5's first version took queue overflow to the triangle's own thread: the slot-insertion scene has
// 2 746 such triangles, 2.0 of its 2.35 M texel tests, for 1 024 slots, and single lanes walked boxes of 42 496 texels: 27.9 ms per 1 024
// envs where this takes ~2.)
constexpr int VIS_SHQ = 2048;          // (> VIS_SHADOW_THREADS: a pass of the workgroup adds at most one entry per thread)
__global__ void __launch_bounds__(VIS_SHADOW_THREADS) k_vis_shadow(VisScene S, const float* __restrict__ xpose, unsigned* __restrict__ shmap, int N) {
    __shared__ float Q[VIS_SHQ * 9];
    __shared__ int nq;
    const int env = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (env >= N) return;
    const int SM = S.shn;
    unsigned* map = shmap + (size_t)env * SM * SM;
    // (a launch of few envs splits every map into bands of rows, one workgroup each: blockIdx.y; every band walks all triangles)
    const int rows = SM / (int)gridDim.y, yb0 = (int)blockIdx.y * rows, yb1 = yb0 + rows - 1;
    for (int i = tid; i < rows * SM / 4; i += VIS_SHADOW_THREADS) ((uint4*)(map + (size_t)yb0 * SM))[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) nq = 0;
    __threadfence_block();
    __syncthreads();
    const float* xb = xpose + (size_t)env * S.nbody * 12;
    auto raster = [&](const float* P, int first, int stride) {      // P: three (x, y, h) in texel units / metres
        const float x0 = P[0], y0 = P[1], x1 = P[3], y1 = P[4], x2 = P[6], y2 = P[7];
        const float area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
        if (!(fabsf(area) > 1e-12f)) return;
        const float ia = 1.0f / area;
        const int ix0 = max(0, (int)ceilf(fminf(x0, fminf(x1, x2)) - 0.5f)), ix1 = min(SM - 1, (int)floorf(fmaxf(x0, fmaxf(x1, x2)) - 0.5f));
        const int iy0 = max(yb0, (int)ceilf(fminf(y0, fminf(y1, y2)) - 0.5f)), iy1 = min(yb1, (int)floorf(fmaxf(y0, fmaxf(y1, y2)) - 0.5f));
        if (ix0 > ix1 || iy0 > iy1) return;
        const int bw = ix1 - ix0 + 1, n = bw * (iy1 - iy0 + 1);
        // barycentrics and the height as functions affine in the texel centre
        const float a0 = (y1 - y2) * ia, b0 = (x2 - x1) * ia, c0 = (x1 * y2 - x2 * y1) * ia;
        const float a1 = (y2 - y0) * ia, b1 = (x0 - x2) * ia, c1 = (x2 * y0 - x0 * y2) * ia;
        const float h0 = P[2], h1 = P[5], h2 = P[8];
        const float ibw = 1.0f / (float)bw;
        for (int k = first; k < n; k += stride) {
            int ry = (int)(((float)k + 0.5f) * ibw), rx = k - ry * bw;            // k / bw, k % bw through the reciprocal, put right where it rounded across a row
            if (rx < 0) { ry--; rx += bw; } else if (rx >= bw) { ry++; rx -= bw; }
            const float fx = (float)(ix0 + rx) + 0.5f, fy = (float)(iy0 + ry) + 0.5f;
            const float l0 = a0 * fx + (b0 * fy + c0), l1 = a1 * fx + (b1 * fy + c1), l2 = 1.0f - l0 - l1;
            if (fminf(fminf(l0, l1), l2) >= 0.0f) atomicMax(&map[(iy0 + ry) * SM + ix0 + rx], vis_hkey(l0 * h0 + l1 * h1 + l2 * h2));
        }
    };
    for (int t0 = 0; t0 < S.ntri; t0 += VIS_SHADOW_THREADS) {
        const int t = t0 + tid;
        if (t < S.ntri) {
            float P[9];
            for (int c = 0; c < 3; c++) {
                const int v = S.tri[3 * t + c];
                const float *pb = xb + 12 * S.vbody[v], *Rb = pb + 3;
                const float x = S.vert[3 * v], y = S.vert[3 * v + 1], z = S.vert[3 * v + 2];
                const float w[3] = {Rb[0] * x + Rb[1] * y + Rb[2] * z + pb[0], Rb[3] * x + Rb[4] * y + Rb[5] * z + pb[1], Rb[6] * x + Rb[7] * y + Rb[8] * z + pb[2]};
                P[3 * c] = (w[0] * S.le1[0] + w[1] * S.le1[1] + w[2] * S.le1[2] - S.sh_s0) * S.sh_itex;
                P[3 * c + 1] = (w[0] * S.le2[0] + w[1] * S.le2[1] + w[2] * S.le2[2] - S.sh_t0) * S.sh_itex;
                P[3 * c + 2] = -(w[0] * S.lw[0] + w[1] * S.lw[1] + w[2] * S.lw[2]);
            }
            const float ylo = fmaxf(fminf(P[1], fminf(P[4], P[7])), (float)yb0), yhi = fminf(fmaxf(P[1], fmaxf(P[4], P[7])), (float)(yb1 + 1));      // (the box inside this band)
            const float bx = fmaxf(P[0], fmaxf(P[3], P[6])) - fminf(P[0], fminf(P[3], P[6])), by = yhi - ylo;
            if (by >= 0.0f) {              // (else: the triangle lies outside this band)
                if ((bx + 1.0f) * (by + 1.0f) > 64.0f) {
                    const int slot = atomicAdd(&nq, 1);          // (< VIS_SHQ: the queue is emptied while a pass of the workgroup still fits)
                    for (int c = 0; c < 9; c++) Q[9 * slot + c] = P[c];
                } else raster(P, 0, 1);
            }
        }
        __syncthreads();
        const int n = nq;
        if (n > VIS_SHQ - VIS_SHADOW_THREADS || t0 + VIS_SHADOW_THREADS >= S.ntri) {      // (the same for every thread)
            for (int q = wave; q < n; q += VIS_SHADOW_THREADS / 64) raster(Q + 9 * q, lane, 64);
            __syncthreads();
            if (tid == 0) nq = 0;
            __syncthreads();
        }
    }
}

#ifdef VIS_NO_OCC4
#define VIS_OCC_ATTR
#else
#define VIS_OCC_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))      // 128 VGPRs: sixteen wavefronts per CU
#endif
template <int SS, bool SH, bool SM>       // SS x SS samples per pixel; SH: shadows of the directional light (S.shmap); SM: smooth shading (a template flag: as a run-time one it
                                         // cost the flat path 16 % -- registers of both paths live in the tile loop at the 128-VGPR cap)
__global__ void __launch_bounds__(VIS_THREADS) VIS_OCC_ATTR k_vis_render(VisScene S, VisScratch X, const float* __restrict__ xpose, const int* __restrict__ cam_ids, int ncam_sel,
                                                            int N, int H, int W, unsigned char* __restrict__ out, int cam_major) {
    __shared__ float Rcb[VIS_MAXBODY * 12];
    __shared__ float cam[36];                // [31..33] Blinn's half vector of the directional light (viewer at infinity), camera frame; Rc (9), pc (3), light dir in the camera frame (3), world up in the camera frame (3), scale; [19..30] the light's frame
                                             // and shadow box (VisScene le1, le2, lw, sh_s0, sh_t0, sh_itex: read from here, not from SGPRs, by the few lanes that need them)
    extern __shared__ int vis_dyn[];         // toff[ntile + 1]: tile -> first list entry (after the scan), counters before; tcur[ntile]
    __shared__ int wsum[2 * (VIS_THREADS / 64)];
    __shared__ int texcnt;                   // textured triangles of this view so far (slots of trec)
    __shared__ int nbig;                     // records in the queue of big triangles (vis_bin)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tw = (W + VIS_TILE - 1) / VIS_TILE, th = (H + VIS_TILE - 1) / VIS_TILE, ntile = tw * th;
    const int nviews = N * ncam_sel, slot = blockIdx.x;
    int* const toff = vis_dyn;
    int* const tcur = vis_dyn + ntile + 1;
    float4* vcam = X.vcam + (size_t)slot * S.nvert;
    float4* rec = X.rec + (size_t)slot * X.reccap * 4;
    int* bbox = X.bbox + (size_t)slot * X.reccap * 4;
    int* list = X.list + (size_t)slot * X.listcap;
    float4* trec = X.trec + (size_t)slot * VIS_TEXCAP * 3;
    float4* grec = X.grec + (size_t)slot * X.reccap * 3;
    constexpr bool smooth = SM;
    int* bigq = X.bigq + (size_t)slot * X.reccap;
    for (int view = blockIdx.x; view < nviews; view += gridDim.x) {
        const int env = view / ncam_sel, cs = view - env * ncam_sel, cid = cam_ids[cs];
        const float* xb = xpose + (size_t)env * S.nbody * 12;
        __syncthreads();     // the previous view's tiles are done with the shared tables
        const long long tc0 = __builtin_readcyclecounter();
        if (tid == 0) {
            const int b = S.cam_body[cid];
            const float *pb = xb + 12 * b, *Rb = pb + 3, *cp = S.cam_pos + 3 * cid, *cm = S.cam_mat + 9 * cid;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) cam[3 * i + j] = Rb[3 * i] * cm[j] + Rb[3 * i + 1] * cm[3 + j] + Rb[3 * i + 2] * cm[6 + j];
            for (int i = 0; i < 3; i++) cam[9 + i] = pb[i] + Rb[3 * i] * cp[0] + Rb[3 * i + 1] * cp[1] + Rb[3 * i + 2] * cp[2];
            const float* L = S.light + 4;
            const float il = 1.0f / sqrtf(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);
            for (int j = 0; j < 3; j++) {
                cam[12 + j] = (cam[j] * L[0] + cam[3 + j] * L[1] + cam[6 + j] * L[2]) * il;      // Rc^T l
                cam[15 + j] = cam[6 + j];                                                          // Rc^T e_z
            }
            cam[18] = 2.0f * S.cam_fovy[cid] / (float)H;
            {   // half vector between the direction TO the light and the viewer at infinity on the optical axis (fixed-function GL [EXT]); camera frame
                const float hx = -cam[12], hy = -cam[13], hz = -cam[14] + 1.0f, hn = hx * hx + hy * hy + hz * hz;
                const float ih = hn > 1e-24f ? rsqrtf(hn) : 0.0f;
                cam[31] = hx * ih; cam[32] = hy * ih; cam[33] = hz * ih;
            }
            texcnt = 0;
            nbig = 0;
            for (int k = 0; k < 3; k++) { cam[19 + k] = S.le1[k]; cam[22 + k] = S.le2[k]; cam[25 + k] = S.lw[k]; }
            cam[28] = S.sh_s0; cam[29] = S.sh_t0; cam[30] = S.sh_itex;
        }
        for (int t = tid; t <= ntile; t += VIS_THREADS) toff[t] = 0;
        __syncthreads();
        // 1. camera-from-body transforms, vertices into the camera frame
        for (int b = tid; b < S.nbody; b += VIS_THREADS) {
            const float *pb = xb + 12 * b, *Rb = pb + 3;
            float* o = Rcb + 12 * b;
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) o[3 * i + j] = cam[i] * Rb[j] + cam[3 + i] * Rb[3 + j] + cam[6 + i] * Rb[6 + j];      // Rc^T Rb
                o[9 + i] = cam[i] * (pb[0] - cam[9]) + cam[3 + i] * (pb[1] - cam[10]) + cam[6 + i] * (pb[2] - cam[11]);
            }
        }
        __syncthreads();
        for (int v = tid; v < S.nvert; v += VIS_THREADS) {
            const float* o = Rcb + 12 * S.vbody[v];
            const float x = S.vert[3 * v], y = S.vert[3 * v + 1], z = S.vert[3 * v + 2];
            vcam[v] = make_float4(o[0] * x + o[1] * y + o[2] * z + o[9], o[3] * x + o[4] * y + o[5] * z + o[10], o[6] * x + o[7] * y + o[8] * z + o[11], 0.0f);
        }
        __threadfence_block();
        __syncthreads();
        const long long tc1 = __builtin_readcyclecounter();
        // 2. triangle set-up
        const float scale = cam[18], iscale = 1.0f / scale, znear = S.znear;
        const float amb = S.light[0], hd = S.light[1], ld = S.light[2];
        int flag = 0, nrec_run = 0;
        for (int t0 = 0; t0 < S.ntri; t0 += VIS_THREADS) {
            const int t = t0 + tid;
            int nout = 0;
            float px[4], py[4], pw[4];
            float att[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};      // smooth shading: shade, shade without the light's term, specular at the (clipped) corners
            unsigned colour = 0, colour2 = 0;
            float sbias = -1.0f;
            int tag = t;               // word 13 of the record: the triangle, or for a textured one its slot in trec
            if (t < S.ntri) {
                const float4 a = vcam[S.tri[3 * t]], b = vcam[S.tri[3 * t + 1]], c = vcam[S.tri[3 * t + 2]];
                const float da = -a.z, db = -b.z, dc = -c.z;
                if (!(da < znear && db < znear && dc < znear)) {
                    // clip the triangle against depth = znear (Sutherland-Hodgman on one plane: 3 or 4 corners)
                    const float P[3][3] = {{a.x, a.y, a.z}, {b.x, b.y, b.z}, {c.x, c.y, c.z}};
                    const float D[3] = {da, db, dc};
                    float Q[4][3];
                    int nq = 0;
                    float A3[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
                    if (smooth) {
                        // the three corners lit with their own normals (body frame -> camera frame through the body's rotation), as the flat shade below
                        // lights the centroid: headlight term along the ray to the corner, the light's diffuse and specular terms
                        const float* o = Rcb + 12 * S.vbody[S.tri[3 * t]];
                        const float* tn = S.tnorm + 9 * (size_t)t;
#pragma unroll
                        for (int i = 0; i < 3; i++) {
                            const float nx = o[0] * tn[3 * i] + o[1] * tn[3 * i + 1] + o[2] * tn[3 * i + 2], ny = o[3] * tn[3 * i] + o[4] * tn[3 * i + 1] + o[5] * tn[3 * i + 2],
                                        nz = o[6] * tn[3 * i] + o[7] * tn[3 * i + 1] + o[8] * tn[3 * i + 2];
                            const float ig = rsqrtf(fmaxf(P[i][0] * P[i][0] + P[i][1] * P[i][1] + P[i][2] * P[i][2], 1e-30f));
                            const float chv = fmaxf(-(nx * P[i][0] + ny * P[i][1] + nz * P[i][2]) * ig, 0.0f), clv = -(nx * cam[12] + ny * cam[13] + nz * cam[14]);
                            A3[i][0] = fminf(1.0f, amb + hd * chv + ld * fmaxf(clv, 0.0f));
                            A3[i][1] = fminf(1.0f, amb + hd * chv);
                            float sp = 0.0f;
                            if (clv > 0.0f && S.spec_k > 0.0f) {
                                const float nh = nx * cam[31] + ny * cam[32] + nz * cam[33];
                                if (nh > 0.0f) sp = S.spec_k * exp2f(S.spec_n * log2f(nh));
                            }
                            A3[i][2] = sp;
                        }
                    }
                    for (int i = 0; i < 3; i++) {
                        const int j = i == 2 ? 0 : i + 1;
                        const bool in_i = D[i] >= znear, in_j = D[j] >= znear;
                        if (in_i) { Q[nq][0] = P[i][0]; Q[nq][1] = P[i][1]; Q[nq][2] = P[i][2]; for (int l = 0; l < 3; l++) att[nq][l] = A3[i][l]; nq++; }
                        if (in_i != in_j) {
                            const float s = (znear - D[i]) / (D[j] - D[i]);
                            Q[nq][0] = P[i][0] + s * (P[j][0] - P[i][0]); Q[nq][1] = P[i][1] + s * (P[j][1] - P[i][1]); Q[nq][2] = -znear;
                            for (int l = 0; l < 3; l++) att[nq][l] = A3[i][l] + s * (A3[j][l] - A3[i][l]);
                            nq++;
                        }
                    }
                    for (int i = 0; i < nq; i++) {
                        const float w = -1.0f / Q[i][2];
                        px[i] = Q[i][0] * w * iscale + 0.5f * W;
                        py[i] = -Q[i][1] * w * iscale + 0.5f * H;
                        pw[i] = w;
                    }
                    nout = nq >= 3 ? nq - 2 : 0;
                    // flat shade from the (unclipped) triangle's normal in the camera frame, turned towards the camera
                    float n[3] = {(b.y - a.y) * (c.z - a.z) - (b.z - a.z) * (c.y - a.y), (b.z - a.z) * (c.x - a.x) - (b.x - a.x) * (c.z - a.z),
                                  (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x)};
                    const float g[3] = {(a.x + b.x + c.x) * (1.0f / 3), (a.y + b.y + c.y) * (1.0f / 3), (a.z + b.z + c.z) * (1.0f / 3)};
                    const float nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2], gg = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
                    if (!(nn > 0) || !(gg > 0)) nout = 0;
                    else {
                        const float in = rsqrtf(nn), ig = rsqrtf(gg);
                        // back faces are not drawn (MuJoCo's renderer culls them too [EXT]): the meshes are closed surfaces wound
                        // outwards, a face turned away from the camera is hidden by a front face -- half the records and list entries
                        const float ch = -(n[0] * g[0] + n[1] * g[1] + n[2] * g[2]) * in * ig;
                        if (!(ch > 0)) nout = 0;
                        const float cl = -(n[0] * cam[12] + n[1] * cam[13] + n[2] * cam[14]) * in;
                        const float lum = fminf(1.0f, amb + hd * ch + ld * fmaxf(cl, 0.0f)), lum2 = fminf(1.0f, amb + hd * ch);
                        // the light's specular term: light specular x material specular (S.spec_k) x (n . h)^exponent, white; it goes with the
                        // light's diffuse term in shadow
                        float spec = 0.0f;
                        if (cl > 0.0f && S.spec_k > 0.0f) {
                            const float nh = (n[0] * cam[31] + n[1] * cam[32] + n[2] * cam[33]) * in;
                            if (nh > 0.0f) spec = S.spec_k * exp2f(S.spec_n * log2f(nh));
                        }
                        if (S.tex[t]) {      // textured: the record carries the shade, the colour comes from the texture at the sample
                            colour = 0x80000000u | ((unsigned)(fminf(spec, 1.0f) * 32767.0f + 0.5f) << 16) | (unsigned)(lum * 65535.0f + 0.5f);      // bit 31, specular 15 bits, shade 16 bits
                            colour2 = 0x80000000u | (unsigned)(lum2 * 65535.0f + 0.5f);
                            if (nout > 0) {
                                // Texture coordinates over the image without going back to the vertices per pixel: a point s d of the triangle's plane
                                // (d = the sample's ray) has barycentrics s M^-1 d, M = [a b c]; the rows of M^-1 are b x c, c x a, a x b over det(M),
                                // so (u, v) = (sum_i u_i r_i . d, sum_i v_i r_i . d) / (sum_i r_i . d) -- three functions affine in the pixel, and the
                                // determinant cancels.  One slot of three float4 per textured triangle (the table: ~200 of them).
                                const int ts = atomicAdd(&texcnt, 1);
                                if (ts < VIS_TEXCAP) {
                                    tag = ts;
                                    const float r0[3] = {b.y * c.z - b.z * c.y, b.z * c.x - b.x * c.z, b.x * c.y - b.y * c.x};
                                    const float r1[3] = {c.y * a.z - c.z * a.y, c.z * a.x - c.x * a.z, c.x * a.y - c.y * a.x};
                                    const float r2[3] = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
                                    const float* uv = S.uv + 6 * t;
#pragma unroll
                                    for (int f = 0; f < 3; f++) {
                                        const float k0 = f == 0 ? uv[0] : f == 1 ? uv[1] : 1.0f, k1 = f == 0 ? uv[2] : f == 1 ? uv[3] : 1.0f, k2 = f == 0 ? uv[4] : f == 1 ? uv[5] : 1.0f;
                                        const float vx = k0 * r0[0] + k1 * r1[0] + k2 * r2[0], vy = k0 * r0[1] + k1 * r1[1] + k2 * r2[1], vz = k0 * r0[2] + k1 * r1[2] + k2 * r2[2];
                                        trec[3 * ts + f] = make_float4(vx * scale, -vy * scale, -vx * scale * 0.5f * W + vy * scale * 0.5f * H - vz, 0.0f);
                                    }
                                } else { tag = VIS_TEXCAP - 1; flag |= 1; }
                            }
                        }
                        else if (smooth) colour = vis_pack(S.rgb[3 * t], S.rgb[3 * t + 1], S.rgb[3 * t + 2]);      // (the material's colour: the shade comes from the record's planes at the sample)
                        else { colour = vis_pack(S.rgb[3 * t] * lum + spec, S.rgb[3 * t + 1] * lum + spec, S.rgb[3 * t + 2] * lum + spec); colour2 = vis_pack(S.rgb[3 * t] * lum2, S.rgb[3 * t + 1] * lum2, S.rgb[3 * t + 2] * lum2); }
                        // in the light's shadow the light's term goes (colour2); the depth map's texels are S.sh_itex^-1 wide, so a lit surface
                        // tilted by theta against the light lies up to texel x tan(theta) below its own texel's height: slope-scaled bias.
                        // A surface that faces away from the light has no such term to lose: bias < 0 = no look-up
                        if (smooth) colour2 = (A3[0][2] > 0.0f || A3[1][2] > 0.0f || A3[2][2] > 0.0f) ? 1u : 0u;      // (smooth mode: word 14 says whether the record has a specular plane worth fetching)
                        sbias = cl > 0.0f ? 1e-3f + 1.5f * sqrtf(fmaxf(0.0f, 1.0f - cl * cl)) / fmaxf(cl, 0.05f) / S.sh_itex : -1.0f;
                    }
                }
            }
            // the (at most two) records of this thread's triangle
            float4 R[2][4], G[2][3];
            int4 B[2];
            bool keep[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                keep[k] = k < nout;
                B[k] = make_int4(0, -1, 0, -1);
                if (keep[k]) {
                    const int i0 = 0, i1 = k + 1, i2 = k + 2;
                    const float x0 = px[i0], y0 = py[i0], x1 = px[i1], y1 = py[i1], x2 = px[i2], y2 = py[i2];
                    const float area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
                    const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2)), ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
                    // pixel centres covered by the bounding box
                    // pixels with a sample inside the bounding box (pixel centres; with supersampling the samples sit 1/4 pixel off them)
                    constexpr float so = SS > 1 ? 0.25f : 0.0f;
                    const int ix0 = max(0, (int)ceilf(fmaxf(xmin, -1e6f) - 0.5f - so)), ix1 = min(W - 1, (int)floorf(fminf(xmax, 1e6f) - 0.5f + so));
                    const int iy0 = max(0, (int)ceilf(fmaxf(ymin, -1e6f) - 0.5f - so)), iy1 = min(H - 1, (int)floorf(fminf(ymax, 1e6f) - 0.5f + so));
                    if (!(fabsf(area) > 1e-12f) || ix0 > ix1 || iy0 > iy1 || !(xmax - xmin < 1e7f) || !(ymax - ymin < 1e7f)) keep[k] = false;
                    else {
                        const float ia = 1.0f / area;
                        // lambda_0 = edge (1 -> 2), lambda_1 = edge (2 -> 0), lambda_2 = edge (0 -> 1), each / area: >= 0 inside either winding
                        const float a0 = (y1 - y2) * ia, b0 = (x2 - x1) * ia, c0 = (x1 * y2 - x2 * y1) * ia;
                        const float a1 = (y2 - y0) * ia, b1 = (x0 - x2) * ia, c1 = (x2 * y0 - x0 * y2) * ia;
                        const float a2 = (y0 - y1) * ia, b2 = (x1 - x0) * ia, c2 = (x0 * y1 - x1 * y0) * ia;
                        const float w0 = pw[i0], w1 = pw[i1], w2 = pw[i2];
                        R[k][0] = make_float4(a0, b0, c0, a1);
                        R[k][1] = make_float4(b1, c1, a2, b2);
                        R[k][2] = make_float4(c2, a0 * w0 + a1 * w1 + a2 * w2, b0 * w0 + b1 * w1 + b2 * w2, c0 * w0 + c1 * w1 + c2 * w2);
                        R[k][3] = make_float4(__uint_as_float(colour), __int_as_float(tag), __uint_as_float(colour2), sbias);
                        // smooth shading: a corner attribute q interpolates perspective-correctly as sum(lambda_i q_i w_i) / sum(lambda_i w_i): the numerator's plane
#pragma unroll
                        for (int l = 0; l < 3; l++) {
                            const float q0 = att[i0][l] * w0, q1 = att[i1][l] * w1, q2 = att[i2][l] * w2;
                            G[k][l] = make_float4(a0 * q0 + a1 * q1 + a2 * q2, b0 * q0 + b1 * q1 + b2 * q2, c0 * q0 + c1 * q1 + c2 * q2, 0.0f);
                        }
                        B[k] = make_int4(ix0 / VIS_TILE, ix1 / VIS_TILE, iy0 / VIS_TILE, iy1 / VIS_TILE);
                    }
                }
            }
            // ordered compaction of the block's records (the record index follows the triangle index): one barrier per pass, the
            // per-wave totals double-buffered, the running total kept by every thread
            const unsigned long long m0 = __ballot(keep[0]), m1 = __ballot(keep[1]), lower = (1ull << lane) - 1ull;
            int* ws = wsum + (VIS_THREADS / 64) * ((t0 / VIS_THREADS) & 1);
            if (lane == 0) ws[wave] = __popcll(m0) + __popcll(m1);
            __syncthreads();
            int idx = nrec_run;
            for (int w2_ = 0; w2_ < VIS_THREADS / 64; w2_++) { if (w2_ < wave) idx += ws[w2_]; nrec_run += ws[w2_]; }
            idx += __popcll(m0 & lower) + __popcll(m1 & lower);
#pragma unroll
            for (int k = 0; k < 2; k++)
                if (keep[k]) {
                    if (idx < X.reccap) {
                        rec[4 * idx] = R[k][0]; rec[4 * idx + 1] = R[k][1]; rec[4 * idx + 2] = R[k][2]; rec[4 * idx + 3] = R[k][3];
                        if (smooth) { grec[3 * idx] = G[k][0]; grec[3 * idx + 1] = G[k][1]; grec[3 * idx + 2] = G[k][2]; }
                        ((int4*)bbox)[idx] = B[k];
                    } else flag |= 1;
                    idx++;
                }
        }
        const int nrec = nrec_run < X.reccap ? nrec_run : X.reccap;
        __threadfence_block();
        __syncthreads();
        const long long tc2 = __builtin_readcyclecounter();
        vis_bin<false>(rec, bbox, nrec, toff, list, X.listcap, tw, lane, wave, flag, bigq, &nbig);
        __syncthreads();
        // 3. exclusive scan of the tile counts (one wave), then the fill pass
        if (wave == 0) {
            int carry = 0;
            for (int t0 = 0; t0 < ntile; t0 += 64) {
                const int t = t0 + lane;
                const int c = t < ntile ? toff[t] : 0;
                int s = c;
                for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(s, o, 64); if (lane >= o) s += y; }
                if (t < ntile) { toff[t] = carry + s - c; tcur[t] = carry + s - c; }
                carry += __shfl(s, 63, 64);
            }
            if (lane == 0) toff[ntile] = carry;
        }
        __syncthreads();
        const long long tc3 = __builtin_readcyclecounter();
        vis_bin<true>(rec, bbox, nrec, tcur, list, X.listcap, tw, lane, wave, flag, bigq, &nbig);
        // (the stores of the records and lists are acknowledged by L2, which is where the scalar cache reads them from: a workgroup-scope
        // release waits for exactly that.  An agent-scope fence here wrote the XCD's whole L2 back -- other views' images included -- once
        // per view: 270 of the fill stage's 465 k cycles)
        __threadfence_block();
        __syncthreads();
        const long long tc4 = __builtin_readcyclecounter();
        // 4. tiles: one wavefront each, lane = pixel
        unsigned char* img = out + (size_t)(cam_major ? cs * N + env : view) * H * W * 3;      // [N][cam] or (option render_cam_major) [cam][N]
        const unsigned* shenv = SH ? S.shmap + (size_t)env * S.shn * S.shn : nullptr;
        // (records and lists were written with vector stores: the barrier above made them visible in L2, this drops what the scalar cache
        // still holds of the slot's previous view)
        __builtin_amdgcn_s_dcache_inv();
        int tx = wave % tw, ty = wave / tw;          // this wave's tile, stepped along with the tile index (no division per tile)
        const float lxf = (float)(lane & 7) + 0.5f, lyf = (float)(lane >> 3) + 0.5f;
        for (int tile = wave; tile < ntile; tile += VIS_THREADS / 64) {
            const int ix = tx * VIS_TILE + (lane & 7), iy = ty * VIS_TILE + (lane >> 3);
            const float fx = (float)(tx * VIS_TILE) + lxf, fy = (float)(ty * VIS_TILE) + lyf;
            const vis_f2 xs2 = {fx - 0.25f, fx + 0.25f};          // (supersampling: the two sample columns, the two sample rows)
            const float ysm = fy - 0.25f, ysp = fy + 0.25f;
            tx += VIS_THREADS / 64;
            while (tx >= tw) { tx -= tw; ty++; }
            const int e0 = __builtin_amdgcn_readfirstlane(toff[tile]), e1 = __builtin_amdgcn_readfirstlane(min(toff[tile + 1], X.listcap));
            constexpr int NS = SS * SS;               // samples per pixel: the centre, or the four points 1/4 pixel off it
            // per sample: the nearest record so far -- 1 / depth and the record's index, which also breaks ties (the records are in triangle
            // order, the image does not depend on the order of the lists); colours, shadow bias and texture come from the winner's record
            // after the loop (three selects per entry and sample less than carrying them along)
            float bw[NS];
            int bi[NS];
#pragma unroll
            for (int q = 0; q < NS; q++) { bw[q] = 0.0f; bi[q] = 0x7fffffff; }
            for (int eb = e0; eb < e1; eb += 4) {
                // four list entries and their records through the scalar unit; entries past the tile's end repeat the batch's first: a record
                // tested twice changes nothing
                const int n = e1 - eb;
                const vis_v4i id = vis_sload4(list + eb);
                const int i0 = id[0], i1 = n > 1 ? id[1] : id[0], i2 = n > 2 ? id[2] : id[0], i3 = n > 3 ? id[3] : id[0];
                VisRec12 R0, R1, R2, R3;
                vis_sload12x4(rec + 4 * (size_t)(i0 & 0x7fffffff), rec + 4 * (size_t)(i1 & 0x7fffffff), rec + 4 * (size_t)(i2 & 0x7fffffff), rec + 4 * (size_t)(i3 & 0x7fffffff), R0, R1, R2, R3);
                auto test = [&](const VisRec12& R, int idw) {
                    const int i = idw & 0x7fffffff;
                    const bool edges = idw >= 0;          // (wave-uniform: the tile is not entirely inside this triangle)
                    const float wc = __int_as_float(R.b[1]) * fx + (__int_as_float(R.b[2]) * fy + __int_as_float(R.b[3]));
                    float l0c = 0, l1c = 0, l2c = 0;
                    if (edges) {
                        l0c = __int_as_float(R.a[0]) * fx + (__int_as_float(R.a[1]) * fy + __int_as_float(R.a[2]));
                        l1c = __int_as_float(R.a[3]) * fx + (__int_as_float(R.a[4]) * fy + __int_as_float(R.a[5]));
                        l2c = __int_as_float(R.a[6]) * fx + (__int_as_float(R.a[7]) * fy + __int_as_float(R.b[0]));
                    }
                    // (w > 0 needs no test of its own: bw starts at 0.  The conditions are combined with & and |, not && and ||: lane masks in
                    // SGPRs, no exec-mask round trip per entry)
                    if (NS == 1) {
                        const bool in = fminf(fminf(l0c, l1c), l2c) >= 0.0f;
                        const bool better = in & ((wc > bw[0]) | ((wc == bw[0]) & (i < bi[0])));
                        bw[0] = better ? wc : bw[0]; bi[0] = better ? i : bi[0];
                    } else {
                        // the four samples sit at (-+1/4, -+1/4) off the centre: a plane a x + b y + c is taken at the two sample columns at once
                        // (v_pk_fma_f32: the x pair of the tile against the plane's value at x = 0 on the upper and on the lower sample row) --
                        // two scalar and two packed fma per plane for the four samples
                        auto p4 = [&](float pa, float pb, float pc, float* o) {
                            const float tm = pb * ysm + pc, tp = pb * ysp + pc;
                            const vis_f2 va = {pa, pa};
                            const vis_f2 lo = __builtin_elementwise_fma(va, xs2, (vis_f2){tm, tm}), hi = __builtin_elementwise_fma(va, xs2, (vis_f2){tp, tp});
                            o[0] = lo.x; o[1] = lo.y; o[2] = hi.x; o[3] = hi.y;
                        };
                        float w[4];
                        p4(__int_as_float(R.b[1]), __int_as_float(R.b[2]), __int_as_float(R.b[3]), w);
                        float m[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (edges) {
                            float e0[4], e1[4], e2[4];
                            p4(__int_as_float(R.a[0]), __int_as_float(R.a[1]), __int_as_float(R.a[2]), e0);
                            p4(__int_as_float(R.a[3]), __int_as_float(R.a[4]), __int_as_float(R.a[5]), e1);
                            p4(__int_as_float(R.a[6]), __int_as_float(R.a[7]), __int_as_float(R.b[0]), e2);
#pragma unroll
                            for (int q = 0; q < 4; q++) m[q] = fminf(fminf(e0[q], e1[q]), e2[q]);
                        }
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const bool better = (m[q] >= 0.0f) & ((w[q] > bw[q]) | ((w[q] == bw[q]) & (i < bi[q])));
                            bw[q] = better ? w[q] : bw[q]; bi[q] = better ? i : bi[q];
                        }
                    }
                };
                test(R0, i0); test(R1, i1); test(R2, i2); test(R3, i3);
            }
            if (ix < W && iy < H) {
                // the colour of sample q: its record's, in the light's shadow without the light's term, from the texture, or the sky's
                auto shade = [&](int q) -> unsigned {
                    const float ox = NS == 1 ? 0.0f : ((q & 1) ? 0.25f : -0.25f), oy = NS == 1 ? 0.0f : ((q & 2) ? 0.25f : -0.25f);
                    const float sx = fx + ox, sy = fy + oy;
                    const float dx = (sx - 0.5f * W) * scale, dy = -(sy - 0.5f * H) * scale;
                    unsigned col;
                    if (bi[q] != 0x7fffffff) {
                        const float4 r3 = rec[4u * (unsigned)bi[q] + 3u];      // colour, triangle or texture slot, colour without the light's term, shadow bias
                        col = __float_as_uint(r3.x);
                        bool shadowed = false;
                        if (SH && r3.w >= 0.0f) {
                            // the sample's surface point in the world: depth 1 / w along the optical axis; its place and height in the light's frame
                            const float depth = __builtin_amdgcn_rcpf(bw[q]), px = dx * depth, py = dy * depth, pz = -depth;
                            const float wx = cam[9] + cam[0] * px + cam[1] * py + cam[2] * pz, wy = cam[10] + cam[3] * px + cam[4] * py + cam[5] * pz, wz = cam[11] + cam[6] * px + cam[7] * py + cam[8] * pz;
                            const float su = (wx * cam[19] + wy * cam[20] + wz * cam[21] - cam[28]) * cam[30], sv = (wx * cam[22] + wy * cam[23] + wz * cam[24] - cam[29]) * cam[30];
                            const float hh = -(wx * cam[25] + wy * cam[26] + wz * cam[27]);
                            if (su >= 0.0f && sv >= 0.0f && su < (float)S.shn && sv < (float)S.shn) {
                                const unsigned key = shenv[(unsigned)((int)sv * S.shn + (int)su)];
                                if (key != 0u && vis_hval(key) > hh + r3.w) { if (!smooth) col = __float_as_uint(r3.z); shadowed = true; }
                            }
                        }
                        float glum = 0.0f, gspec = 0.0f;
                        if (smooth) {      // the shade at THIS sample from the record's planes (lit corners, interpolated): plane / w
                            const float4* gp = grec + 3u * (unsigned)bi[q];
                            const float4 g0 = gp[shadowed ? 1 : 0];
                            const float dep = __builtin_amdgcn_rcpf(bw[q]);
                            glum = fminf(fmaxf((g0.x * sx + (g0.y * sy + g0.z)) * dep, 0.0f), 1.0f);
                            if (!shadowed && __float_as_uint(r3.z) != 0u) { const float4 g2 = gp[2]; gspec = fmaxf((g2.x * sx + (g2.y * sy + g2.z)) * dep, 0.0f); }
                            if (!(col & 0x80000000u)) col = vis_pack((float)(col & 255u) * (glum / 255.0f) + gspec, (float)((col >> 8) & 255u) * (glum / 255.0f) + gspec, (float)((col >> 16) & 255u) * (glum / 255.0f) + gspec);
                        }
                        if (col & 0x80000000u) {
                            // textured: (u, v) = (U, V) / D with the triangle's three planes over the image (set-up), then the texel
                            const float lum = smooth ? glum : (float)(col & 0xffffu) * (1.0f / 65535.0f), spc = smooth ? gspec : (float)((col >> 16) & 0x7fffu) * (1.0f / 32767.0f);
                            const float4* tp = trec + 3 * __float_as_int(r3.y);
                            const float4 pu = tp[0], pv = tp[1], pd = tp[2];
                            const float den = pd.x * sx + (pd.y * sy + pd.z);
                            const float iden = fabsf(den) > 1e-30f ? __builtin_amdgcn_rcpf(den) : 0.0f;
                            const float tu = (pu.x * sx + (pu.y * sy + pu.z)) * iden, tv = (pv.x * sx + (pv.y * sy + pv.z)) * iden;
                            const float fu = tu - floorf(tu), fv = tv - floorf(tv);
                            const int txi = min(S.texn - 1, (int)(fu * S.texn)), tyi = min(S.texn - 1, (int)((1.0f - fv) * S.texn));
                            const unsigned tx_ = S.texel[tyi * S.texn + txi];
                            col = vis_pack((float)(tx_ & 255u) * (lum / 255.0f) + spc, (float)((tx_ >> 8) & 255u) * (lum / 255.0f) + spc, (float)((tx_ >> 16) & 255u) * (lum / 255.0f) + spc);
                        }
                    } else {
                        const float idn = rsqrtf(dx * dx + dy * dy + 1.0f);
                        const float w = 0.5f + 0.5f * (cam[15] * dx + cam[16] * dy - cam[17]) * idn;
                        col = vis_pack(S.light[12] + (S.light[8] - S.light[12]) * w, S.light[13] + (S.light[9] - S.light[13]) * w, S.light[14] + (S.light[10] - S.light[14]) * w);
                    }
                    return col;
                };
                unsigned col;
                if (NS == 1) col = shade(0);
                else {
                    // A fragment -- the samples of a pixel that one triangle wins, or that see the sky -- is shaded ONCE, at its first sample,
                    // as a multisampled GL buffer does [EXT]; the pixel is the mean of its four samples' colours.  Pixels inside one triangle
                    // (most of them) cost one shading, not four; a wave whose pixels all do skips the other three altogether.
                    unsigned c[NS];
                    c[0] = shade(0);
#pragma unroll
                    for (int q = 1; q < NS; q++) {
                        bool need = true;
                        unsigned cq = 0;
#pragma unroll
                        for (int p2 = q - 1; p2 >= 0; p2--)
                            if (bi[q] == bi[p2]) { cq = c[p2]; need = false; }
                        if (need) cq = shade(q);
                        c[q] = cq;
                    }
                    float accr = 0.0f, accg = 0.0f, accb = 0.0f;
#pragma unroll
                    for (int q = 0; q < NS; q++) { accr += (float)(c[q] & 255u); accg += (float)((c[q] >> 8) & 255u); accb += (float)((c[q] >> 16) & 255u); }
                    col = (unsigned)(accr * (1.0f / NS) + 0.5f) | ((unsigned)(accg * (1.0f / NS) + 0.5f) << 8) | ((unsigned)(accb * (1.0f / NS) + 0.5f) << 16);
                }
                unsigned char* d = img + (unsigned)((iy * W + ix) * 3);
                d[0] = (unsigned char)(col & 255u); d[1] = (unsigned char)((col >> 8) & 255u); d[2] = (unsigned char)((col >> 16) & 255u);
            }
        }
        if (flag) atomicOr(&X.flags[8 * view], flag);
        if (tid == 0) {
            const long long tc5 = __builtin_readcyclecounter();
            int* f = X.flags + 8 * view;
            f[1] = (int)((tc1 - tc0) >> 10); f[2] = (int)((tc2 - tc1) >> 10); f[3] = (int)((tc3 - tc2) >> 10); f[4] = (int)((tc4 - tc3) >> 10); f[5] = (int)((tc5 - tc4) >> 10);
            f[6] = nrec; f[7] = toff[ntile];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct VisHost {
    // the model's instances (kept from the model blob), the scene once a library is loaded
    std::vector<int> inst_mesh, inst_body, inst_tex;
    std::vector<double> inst_pos, inst_mat, inst_scale, inst_rgba;
    bool have_inst = false, loaded = false;
    VisScene S{};
    VisScratch X{};
    std::vector<void*> allocs;
    int slots = 0, max_slots = 0, nviews_cap = 0, last_nviews = 0;     // scratch slots allocated / the most a launch uses (VIS_WG_PER_CU x CUs); flag rows allocated / written by the last launch
    bool attr_done = false;
    int* d_cam_ids = nullptr;
    int samples = 1;                 // option "render_samples": 1, or 4 = 2 x 2 supersampling
    bool cam_major = false;          // option "render_cam_major": the images as [cam][N][H][W][3] (every camera's batch contiguous) instead of [N][cam][H][W][3]
    int shadow_size = VIS_SM;        // option "render_shadow_size": 512 | 1024 | 2048 texels per side (MuJoCo's own map is 8192 wide, scene.xml:12; 2.3 mm texels at 512)
    bool smooth_opt = false;         // option "render_smooth": corners lit with their own normals and interpolated (needs a library with lib_tnorm); on in the gym / Cartesian facades, off in a bare handle (like shadows and multisampling: the raw kernel timings stay comparable)
    bool shadows = false;            // option "render_shadows": the directional light casts shadows (depth map from the light, one per env)
    unsigned* d_shmap = nullptr;
    int shmap_envs = 0;
    unsigned long long shmap_ver = 0;   // the state version (avsim_api.hip) the shadow maps were rendered for
    double light_host[20] = {0};     // the model's render_light (lights, sky, shadow box, specular term)

    template <typename T>
    T* up(const std::vector<T>& v) {
        void* p = nullptr;
        if (hipMalloc(&p, (v.size() ? v.size() : 1) * sizeof(T)) != hipSuccess) throw std::runtime_error("hipMalloc failed while uploading the visual scene");
        if (v.size() && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("hipMemcpy failed while uploading the visual scene");
        allocs.push_back(p);
        return (T*)p;
    }
    void keep_instances(const Blob& b) {
        try {
            inst_mesh = b.i("vis_inst_mesh"); inst_body = b.i("vis_inst_body"); inst_tex = b.i("vis_inst_tex");
            inst_pos = b.f("vis_inst_pos"); inst_mat = b.f("vis_inst_mat"); inst_scale = b.f("vis_inst_scale"); inst_rgba = b.f("vis_inst_rgba");
            { auto L = b.f("render_light"); for (size_t k = 0; k < 20 && k < L.size(); k++) light_host[k] = L[k]; }
            have_inst = true;
        } catch (const std::exception&) { have_inst = false; }      // a model compiled without the visual scene: the proxy image only
    }
    // library blob (models/visual_meshes.avv) x instances -> the scene's triangles in body frames, uploaded
    void load(const Blob& lib, int nbody, const float* d_cam_pos, const float* d_cam_mat, const float* d_cam_fovy, const int* d_cam_body, const float* d_light, float znear) {
        if (!have_inst) throw std::runtime_error("the model blob carries no visual instances (vis_inst_*): recompile it with av_aloha_amd.compiler.compile");
        if (loaded) return;
        auto vadr = lib.i("lib_vadr"), vnum = lib.i("lib_vnum"), tadr = lib.i("lib_tadr"), tnum = lib.i("lib_tnum"), ltri = lib.i("lib_tri"), ltex = lib.i("lib_tex");
        auto lvert = lib.f("lib_vert"), luv = lib.f("lib_uv");
        std::vector<double> ltn;
        try { ltn = lib.f("lib_tnorm"); } catch (const std::exception&) { ltn.clear(); }      // (a library from before round 6: flat shading only)
        std::vector<float> vert, rgb, uv, tnorm;
        std::vector<int> vbody, tri, tex;
        const int ninst = (int)inst_mesh.size();
        for (int k = 0; k < ninst; k++) {
            const int mid = inst_mesh[k];
            if (mid < 0 || mid >= (int)vadr.size()) throw std::runtime_error("visual instance refers to a mesh the library lacks");
            const int v0 = (int)vbody.size();
            const double *sc = &inst_scale[3 * k], *R = &inst_mat[9 * k], *p = &inst_pos[3 * k];
            for (int v = 0; v < vnum[mid]; v++) {
                const double* q = &lvert[3 * (size_t)(vadr[mid] + v)];
                const double s[3] = {q[0] * sc[0], q[1] * sc[1], q[2] * sc[2]};
                for (int i = 0; i < 3; i++) vert.push_back((float)(R[3 * i] * s[0] + R[3 * i + 1] * s[1] + R[3 * i + 2] * s[2] + p[i]));
                vbody.push_back(inst_body[k]);
            }
            const bool flip = sc[0] * sc[1] * sc[2] < 0;
            for (int t = 0; t < tnum[mid]; t++) {
                const int* f = &ltri[3 * (size_t)(tadr[mid] + t)];
                const double* u = &luv[6 * (size_t)(tadr[mid] + t)];
                const int o[3] = {0, flip ? 2 : 1, flip ? 1 : 2};
                for (int c = 0; c < 3; c++) { tri.push_back(v0 + f[o[c]]); uv.push_back((float)u[2 * o[c]]); uv.push_back((float)u[2 * o[c] + 1]); }
                for (int c = 0; c < 3; c++) rgb.push_back((float)inst_rgba[4 * k + c]);
                if (!ltn.empty())
                    for (int c = 0; c < 3; c++) {      // normals go with the inverse transpose: n / scale, then the instance's rotation
                        const double* n = &ltn[9 * (size_t)(tadr[mid] + t) + 3 * o[c]];
                        const double m[3] = {n[0] / sc[0], n[1] / sc[1], n[2] / sc[2]};
                        double r[3];
                        for (int i = 0; i < 3; i++) r[i] = R[3 * i] * m[0] + R[3 * i + 1] * m[1] + R[3 * i + 2] * m[2];
                        const double l = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
                        for (int i = 0; i < 3; i++) tnorm.push_back((float)(l > 0 ? r[i] / l : (i == 2 ? 1.0 : 0.0)));
                    }
                tex.push_back(inst_tex[k]);
            }
        }
        if (nbody > VIS_MAXBODY) throw std::runtime_error("visual renderer: more than 64 bodies");
        S.nvert = (int)vbody.size(); S.ntri = (int)tex.size(); S.nbody = nbody;
        S.vert = up(vert); S.vbody = up(vbody); S.tri = up(tri); S.rgb = up(rgb); S.uv = up(uv); S.tex = up(tex);
        S.tnorm = tnorm.empty() ? nullptr : up(tnorm);
        S.smooth = smooth_opt ? 1 : 0;
        std::vector<unsigned> texel(ltex.begin(), ltex.end());
        S.texel = up(texel);
        S.texn = (int)std::lround(std::sqrt((double)texel.size()));
        S.cam_body = d_cam_body; S.cam_pos = d_cam_pos; S.cam_mat = d_cam_mat; S.cam_fovy = d_cam_fovy; S.light = d_light; S.znear = znear;
        {   // the light's frame and its shadow box (oracle/orc_vis.c light_frame; render_light[3] half extent, [7] [11] [15] centre)
            const double* L = light_host;
            double lw[3] = {L[4], L[5], L[6]}, ln = std::sqrt(lw[0] * lw[0] + lw[1] * lw[1] + lw[2] * lw[2]);
            if (!(ln > 0)) { lw[0] = 0; lw[1] = 0; lw[2] = -1; ln = 1; }
            for (int k = 0; k < 3; k++) lw[k] /= ln;
            const double ax[3] = {std::fabs(lw[0]) < 0.9 ? 1.0 : 0.0, std::fabs(lw[0]) < 0.9 ? 0.0 : 1.0, 0.0};
            double d = ax[0] * lw[0] + ax[1] * lw[1] + ax[2] * lw[2], e1[3], e2[3];
            for (int k = 0; k < 3; k++) e1[k] = ax[k] - d * lw[k];
            d = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
            for (int k = 0; k < 3; k++) e1[k] /= d;
            e2[0] = lw[1] * e1[2] - lw[2] * e1[1]; e2[1] = lw[2] * e1[0] - lw[0] * e1[2]; e2[2] = lw[0] * e1[1] - lw[1] * e1[0];
            for (int k = 0; k < 3; k++) { S.le1[k] = (float)e1[k]; S.le2[k] = (float)e2[k]; S.lw[k] = (float)lw[k]; }
            const double half = L[3] > 0 ? L[3] : 1.0, c[3] = {L[7], L[11], L[15]};
            S.sh_s0 = (float)(c[0] * e1[0] + c[1] * e1[1] + c[2] * e1[2] - half);
            S.sh_t0 = (float)(c[0] * e2[0] + c[1] * e2[1] + c[2] * e2[2] - half);
            S.shn = shadow_size;
            S.sh_itex = (float)(shadow_size / (2.0 * half));
            S.shmap = nullptr;
            S.spec_k = (float)L[16]; S.spec_n = (float)L[17];
        }
        loaded = true;
    }
    void destroy() {
        for (void* p : allocs) (void)hipFree(p);
        allocs.clear();
        for (void* p : {(void*)X.vcam, (void*)X.rec, (void*)X.bbox, (void*)X.list, (void*)X.trec, (void*)X.grec, (void*)X.bigq, (void*)X.flags, (void*)d_cam_ids, (void*)d_shmap}) if (p) (void)hipFree(p);
        X = VisScratch{}; d_cam_ids = nullptr; d_shmap = nullptr; shmap_envs = 0; shmap_ver = 0; loaded = false; slots = 0; max_slots = 0; nviews_cap = 0; last_nviews = 0;
    }
    // overflow flags of the last launch, OR over the views (bit 0: triangle records, bit 1: tile lists); synchronises the stream
    int launch(hipStream_t st, int N, const float* d_xpose, const int* cam_ids_host, int ncam_sel, int ncam_model, int H, int W, void* d_out, std::string& err, unsigned long long state_ver = 0) {
        if (!loaded) { err = "avsim_render_rgb: no visual scene loaded (avsim_load_visual)"; return -1; }
        if (ncam_sel < 1 || ncam_sel > 16 || H < 1 || W < 1) { err = "avsim_render_rgb: bad camera count or image size"; return -1; }
        const int ntile = ((W + VIS_TILE - 1) / VIS_TILE) * ((H + VIS_TILE - 1) / VIS_TILE);
        if (ntile > VIS_MAXTILES) { err = "avsim_render_rgb: the visual-mesh image is limited to 16384 tiles of 8 x 8 pixels (1024 x 1024)"; return -1; }
        for (int c = 0; c < ncam_sel; c++)
            if (cam_ids_host[c] < 0 || cam_ids_host[c] >= ncam_model) { err = "avsim_render_rgb: camera index out of range"; return -1; }
        const int nviews = N * ncam_sel;
        if (!max_slots) {
            int dev = 0, cus = 256;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
            max_slots = VIS_WG_PER_CU * cus;
            X.reccap = S.ntri + 2048;                  // near-plane clipping can split a triangle in two
            X.listcap = 8 * S.ntri + 4 * VIS_MAXTILES;
            if (hipMalloc((void**)&d_cam_ids, 16 * sizeof(int)) != hipSuccess) { err = "hipMalloc(visual render camera ids) failed"; max_slots = 0; return -3; }
        }
        // per-slot scratch (camera-frame vertices, triangle records, their boxes, tile lists: ~2.8 MB a slot for the 20 k-triangle
        // scenes) for as many workgroups as this call can use -- a single env with one camera holds one slot, not one per CU
        // (2.9 GB per handle: a vector env of single-env handles ran out of HBM); regrown when a later call has more views
        const int want = nviews < max_slots ? nviews : max_slots;
        if (want > slots) {
            if (hipStreamSynchronize(st) != hipSuccess) { err = "visual render: stream synchronisation failed"; return -3; }
            for (void* p : {(void*)X.vcam, (void*)X.rec, (void*)X.bbox, (void*)X.list, (void*)X.trec, (void*)X.grec, (void*)X.bigq}) if (p) (void)hipFree(p);
            X.vcam = nullptr; X.rec = nullptr; X.bbox = nullptr; X.list = nullptr; X.trec = nullptr; X.grec = nullptr; X.bigq = nullptr; slots = 0;
            if (hipMalloc((void**)&X.vcam, (size_t)want * S.nvert * sizeof(float4)) != hipSuccess || hipMalloc((void**)&X.rec, (size_t)want * X.reccap * 4 * sizeof(float4)) != hipSuccess ||
                hipMalloc((void**)&X.bbox, (size_t)want * X.reccap * 4 * sizeof(int)) != hipSuccess || hipMalloc((void**)&X.list, ((size_t)want * X.listcap + 16) * sizeof(int)) != hipSuccess ||      // (+ 16: the tile stage reads its list four entries at a time)
                hipMalloc((void**)&X.trec, (size_t)want * VIS_TEXCAP * 3 * sizeof(float4)) != hipSuccess || hipMalloc((void**)&X.grec, (size_t)want * X.reccap * 3 * sizeof(float4)) != hipSuccess || hipMalloc((void**)&X.bigq, (size_t)want * X.reccap * sizeof(int)) != hipSuccess) {
                for (void* p : {(void*)X.vcam, (void*)X.rec, (void*)X.bbox, (void*)X.list, (void*)X.trec, (void*)X.grec, (void*)X.bigq}) if (p) (void)hipFree(p);
                X.vcam = nullptr; X.rec = nullptr; X.bbox = nullptr; X.list = nullptr; X.trec = nullptr; X.grec = nullptr; X.bigq = nullptr;
                err = "hipMalloc(visual render scratch) failed"; return -3;
            }
            slots = want;
        }
        if (nviews > nviews_cap) {
            if (X.flags) (void)hipFree(X.flags);
            X.flags = nullptr;
            if (hipMalloc((void**)&X.flags, (size_t)nviews * 8 * sizeof(int)) != hipSuccess) { err = "hipMalloc(visual render flags) failed"; nviews_cap = 0; return -3; }
            nviews_cap = nviews;
        }
        last_nviews = nviews;
        if (hipMemsetAsync(X.flags, 0, (size_t)nviews * 8 * sizeof(int), st) != hipSuccess || hipMemcpyAsync(d_cam_ids, cam_ids_host, ncam_sel * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) { err = "visual render set-up copy failed"; return -3; }
        const int grid = nviews < slots ? nviews : slots;
        const size_t shmem = (size_t)(2 * ntile + 1) * sizeof(int);
        if (!attr_done) {
            bool ok = true;
#define VIS_ATTR(SS_, SH_, SM_) ok = ok && hipFuncSetAttribute((const void*)k_vis_render<SS_, SH_, SM_>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) == hipSuccess
            VIS_ATTR(1, false, false); VIS_ATTR(2, false, false); VIS_ATTR(1, true, false); VIS_ATTR(2, true, false);
            VIS_ATTR(1, false, true); VIS_ATTR(2, false, true); VIS_ATTR(1, true, true); VIS_ATTR(2, true, true);
#undef VIS_ATTR
            if (!ok) { err = "hipFuncSetAttribute(visual render) failed"; return -3; }
            attr_done = true;
        }
        S.shmap = nullptr;
        if (shadows && light_host[3] > 0) {
            // one depth map from the light per env (shared by the env's cameras): shadow_size^2 keys, 1 MB an env at 512
            if (S.shn != shadow_size) {        // (option changed since the scene was loaded: the texel scale with it)
                S.sh_itex *= (float)shadow_size / (float)S.shn;
                S.shn = shadow_size;
                shmap_envs = 0;                // reallocate
            }
            if (N > shmap_envs) {
                if (hipStreamSynchronize(st) != hipSuccess) { err = "visual render: stream synchronisation failed"; return -3; }
                if (d_shmap) (void)hipFree(d_shmap);
                d_shmap = nullptr; shmap_envs = 0;
                if (hipMalloc((void**)&d_shmap, (size_t)N * shadow_size * shadow_size * sizeof(unsigned)) != hipSuccess) { err = "hipMalloc(shadow maps) failed"; return -3; }
                shmap_envs = N;
                shmap_ver = 0;
            }
            S.shmap = d_shmap;
            if (state_ver == 0 || shmap_ver != state_ver) {      // (the maps of this state may be there already: an earlier call for other cameras)
                // few envs: the maps in bands of rows (a workgroup each), so that one env's map is not one workgroup's work (16 bands for one env)
                int bands = 1;
                while (bands < 16 && N * bands < 256) bands *= 2;
                hipLaunchKernelGGL(k_vis_shadow, dim3(N, bands), dim3(VIS_SHADOW_THREADS), 0, st, S, d_xpose, d_shmap, N);
                shmap_ver = state_ver;
            }
        }
        const bool sh = S.shmap != nullptr;
        const bool sm = S.smooth != 0 && S.tnorm != nullptr;
#define VIS_LAUNCH(SS_, SH_) do { if (sm) hipLaunchKernelGGL((k_vis_render<SS_, SH_, true>), dim3(grid), dim3(VIS_THREADS), shmem, st, S, X, d_xpose, (const int*)d_cam_ids, ncam_sel, N, H, W, (unsigned char*)d_out, cam_major ? 1 : 0); \
                                  else hipLaunchKernelGGL((k_vis_render<SS_, SH_, false>), dim3(grid), dim3(VIS_THREADS), shmem, st, S, X, d_xpose, (const int*)d_cam_ids, ncam_sel, N, H, W, (unsigned char*)d_out, cam_major ? 1 : 0); } while (0)
        if (samples > 1) { if (sh) VIS_LAUNCH(2, true); else VIS_LAUNCH(2, false); }
        else { if (sh) VIS_LAUNCH(1, true); else VIS_LAUNCH(1, false); }
#undef VIS_LAUNCH
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { err = std::string("visual render kernel launch: ") + hipGetErrorString(e); return -3; }
        return 0;
    }
};

}  // namespace avs
