// The device's serial narrow-phase routines (av_aloha_amd/csrc/avsim_collide.hip.h: box_box, narrow = sphere / MPR dispatch)
// compiled for the host with the oracle's floating-point rules (-ffp-contract=off), behind a C entry point with the signature of
// orc_narrow (oracle/orc.h).  Test infrastructure: tests/test_host_collide.py compares the two on random poses.
#include <vector>

#include "../../av_aloha_amd/csrc/avsim_collide.hip.h"

using namespace avs;

template <typename T>
static void fill(Shape<T>& s, int type, const double* size, const double* pos, const double* mat, const T* tab, int hbase, int R, int hovf) {
    s.type = type;
    for (int k = 0; k < 3; k++) { s.size[k] = (T)size[k]; s.pos[k] = (T)pos[k]; s.center[k] = (T)pos[k]; s.lc[k] = 0; s.lh[k] = (T)1e9; }
    for (int k = 0; k < 9; k++) s.mat[k] = (T)mat[k];
    s.hull = tab;
    s.hbase = hbase;
    s.hR = R;
    s.hovf = hovf;
    s.nh = 0;
}

// tab / ntab: the model's support tables as the device holds them (tests/test_host_collide.py expands the blob's chull_cells / chull_cand as
// PhysHost::build does: ntab words); hbase1 / hbase2: the two hulls' first cell records, R1 / R2 their cube-map resolutions, hovf: the entry
// index at which the overflow part starts
template <typename T>
static int run(int t1, const double* size1, const double* pos1, const double* mat1, int hbase1, int R1, const double* c1,
               int t2, const double* size2, const double* pos2, const double* mat2, int hbase2, int R2, const double* c2,
               const double* tab, int ntab, int hovf, double rb1, double rb2, double* dist, double* pos, double* normal) {
    static thread_local std::vector<T> c4;
    c4.assign(tab, tab + (size_t)ntab);
    Shape<T> a, b;
    fill(a, t1, size1, pos1, mat1, c4.data(), hbase1, R1, hovf);
    fill(b, t2, size2, pos2, mat2, c4.data(), hbase2, R2, hovf);
    if (c1) for (int k = 0; k < 3; k++) a.center[k] = (T)c1[k];
    if (c2) for (int k = 0; k < 3; k++) b.center[k] = (T)c2[k];
    T scr[SLOT_W], work[56], ovf[BOX_OVF_W];
    int n;
    const bool isbox = t1 == G_BOX && t2 == G_BOX;
    if (isbox) n = box_box(a, b, scr, work, ovf);
    else {
        n = narrow(a, b, scr);
        // multiccd for the convex (MPR) pairs, spheres excluded -- the kernel's rule (Env::collide_i)
        if (n == 1 && t1 != G_SPHERE && t2 != G_SPHERE)
            n = multiccd_serial(a, b, MultiCcd<T>::reltol * (T)(rb1 < rb2 ? rb1 : rb2), scr, scr + SLOT_P, scr + SLOT_N);
    }
    for (int k = 0; k < n; k++) {
        const bool o = isbox && k >= BOX_SLOTC;         // points 4..7 of a box-box manifold: the overflow record
        dist[k] = o ? ovf[4 * (k - BOX_SLOTC)] : scr[k];
        for (int c = 0; c < 3; c++) pos[3 * k + c] = o ? ovf[4 * (k - BOX_SLOTC) + 1 + c] : scr[SLOT_P + 3 * k + c];
    }
    for (int c = 0; c < 3; c++) normal[c] = n ? scr[SLOT_N + c] : 0;
    return n;
}

extern "C" int dev_narrow_f64(int t1, const double* size1, const double* pos1, const double* mat1, int hbase1, int R1, const double* c1,
                              int t2, const double* size2, const double* pos2, const double* mat2, int hbase2, int R2, const double* c2,
                              const double* tab, int ntab, int hovf, double rb1, double rb2, double* dist, double* pos, double* normal) {
    return run<double>(t1, size1, pos1, mat1, hbase1, R1, c1, t2, size2, pos2, mat2, hbase2, R2, c2, tab, ntab, hovf, rb1, rb2, dist, pos, normal);
}
extern "C" int dev_narrow_f32(int t1, const double* size1, const double* pos1, const double* mat1, int hbase1, int R1, const double* c1,
                              int t2, const double* size2, const double* pos2, const double* mat2, int hbase2, int R2, const double* c2,
                              const double* tab, int ntab, int hovf, double rb1, double rb2, double* dist, double* pos, double* normal) {
    return run<float>(t1, size1, pos1, mat1, hbase1, R1, c1, t2, size2, pos2, mat2, hbase2, R2, c2, tab, ntab, hovf, rb1, rb2, dist, pos, normal);
}
