// avsim_api.hip -- the C-ABI of libavsim.so (include/avsim.h) and the kernel launches behind it.
// gfx950 only; there is no CPU path in this library.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/avsim.h"
#include "avsim_ik.hip.h"
#include "avsim_model.h"
#include "avsim_phys.hip.h"
#include "avsim_render.hip.h"
#include "avsim_vis.hip.h"

using namespace avs;

static thread_local std::string g_create_error;

#define HIPCHK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (h)->set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return AVSIM_EHIP;                                                                       \
        }                                                                                            \
    } while (0)

// Every entry point runs on the handle's device and puts the caller's current device back on return (a torch process keeps
// its own current device; one process per GPU is the intended deployment, several handles per process still work).
struct DevGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;      // a failed switch is an error of the entry point (AVS_ON_DEVICE), never a silent run on the caller's device
    explicit DevGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DevGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define AVS_ON_DEVICE(h)                                                                                          \
    DevGuard dev_guard_((h)->device);                                                                       \
    if (dev_guard_.err != hipSuccess) {                                                                     \
        (h)->set_error("hipSetDevice(%d) failed: %s", (h)->device, hipGetErrorString(dev_guard_.err));      \
        return AVSIM_EHIP;                                                                                  \
    }

constexpr size_t EV_RING = 1024;      // HIP event pairs kept per timing list before they are folded into a running sum
struct avsim {
    int device = 0;
    uint32_t flags = 0;
    int N = 0;
    bool io_device = false, f64 = false;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[16] = {};
    std::string err;
    // model
    int nq = 0, nv = 0, nu = 0, nj = 0, nobj = 0, task_id = 0, num_arms = 3, max_reward = 0;
    IkParams ik;
    std::vector<int> obj_qadr, obs_qadr;
    std::vector<double> qpos_home, ctrl_home;
    PhysHost phys;  // device model image + launch configuration (avsim_phys.hip.h)
    RenderHost render;   // depth renderer (avsim_render.hip.h)
    VisHost vis;         // colour images of the visual meshes (avsim_vis.hip.h), once avsim_load_visual has run
    // the state's version: bumped by everything that writes qpos (reset, the steps, set_state); the image calls skip their pose pass and the shadow
    // map when they already hold this version's (a facade that fetches its cameras one call at a time repeats neither)
    unsigned long long state_ver = 1, xpose_ver = 0;
    bool render_proxies = false;   // option "render_proxies": avsim_render_rgb draws the collision proxies even with a visual scene loaded
    // device state (real = float, or double with AVSIM_F64_PHYSICS)
    void *d_qpos = nullptr, *d_qvel = nullptr, *d_ctrl = nullptr, *d_warm = nullptr;
    int* d_latch = nullptr;
    // device scratch for host-pointer I/O
    void* d_io[8] = {};
    size_t d_io_sz[8] = {};
    // kernel timing
    // kernel timing: pairs of HIP events recorded around every physics launch on the launch stream, read
    // back (and only then synchronised) by avsim_kernel_time
    bool ktiming = false;
    std::vector<hipEvent_t> kev;
    size_t kev_used = 0;
    double kev_ms = 0;            // launches folded out of the event list (it holds at most EV_RING pairs: a long run with kernel_timing on
    int64_t kev_n = 0;            // and nobody asking does not grow it)

    void set_error(const char* fmt, ...) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
    }
    size_t rsz() const { return f64 ? 8 : 4; }
    int io_buf(int slot, size_t bytes, void** out) {
        if (d_io_sz[slot] < bytes) {
            if (d_io[slot]) (void)hipFree(d_io[slot]);
            d_io[slot] = nullptr;
            d_io_sz[slot] = 0;
            hipError_t e = hipMalloc(&d_io[slot], bytes);
            if (e != hipSuccess) {
                set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
                return AVSIM_EHIP;
            }
            d_io_sz[slot] = bytes;
        }
        *out = d_io[slot];
        return 0;
    }
    // stage an input: returns device pointer holding `bytes` of `p` (host or device according to io mode)
    int in(int slot, const void* p, size_t bytes, const void** out) {
        if (io_device) { *out = p; return 0; }
        void* d;
        int rc = io_buf(slot, bytes, &d);
        if (rc) return rc;
        hipError_t e = hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) { set_error("H2D copy failed: %s", hipGetErrorString(e)); return AVSIM_EHIP; }
        *out = d;
        return 0;
    }
    int out_begin(int slot, void* p, size_t bytes, void** dev) {
        if (io_device) { *dev = p; return 0; }
        return io_buf(slot, bytes, dev);
    }
    // Large results to pageable host memory (the images of avsim_render_*: 0.9 MB per 480 x 640 colour view).  hipMemcpy into pageable memory
    // runs at ~5 GB/s here (measured: 354 MB of pixels in 66 of the 73 ms of a 64-env gym step).  Above 16 MB the copy is pipelined instead: the
    // device buffer goes in 32 MB chunks by DMA into two pinned staging buffers while the previous chunk is copied on into the caller's memory by
    // four threads.  Synchronous, like the plain path once finish() has run.
    static constexpr size_t PIN_CHUNK = 32u << 20;
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    int out_end_pipelined(const char* src, char* dst, size_t bytes) {
        for (int k = 0; k < 2; k++) {
            if (!pin[k] && hipHostMalloc(&pin[k], PIN_CHUNK, hipHostMallocDefault) != hipSuccess) { pin[k] = nullptr; return 1; }      // (1: fall back to the plain copy)
            if (!pin_ev[k] && hipEventCreateWithFlags(&pin_ev[k], hipEventDisableTiming) != hipSuccess) { pin_ev[k] = nullptr; return 1; }
        }
        const size_t nchunk = (bytes + PIN_CHUNK - 1) / PIN_CHUNK;
        auto issue = [&](size_t c) -> hipError_t {
            const size_t off = c * PIN_CHUNK, n = bytes - off < PIN_CHUNK ? bytes - off : PIN_CHUNK;
            hipError_t e = hipMemcpyAsync(pin[c & 1], src + off, n, hipMemcpyDeviceToHost, stream);
            return e != hipSuccess ? e : hipEventRecord(pin_ev[c & 1], stream);
        };
        hipError_t e = issue(0);
        for (size_t c = 0; c < nchunk && e == hipSuccess; c++) {
            if ((e = hipEventSynchronize(pin_ev[c & 1])) != hipSuccess) break;
            if (c + 1 < nchunk && (e = issue(c + 1)) != hipSuccess) break;          // (the other buffer: its previous contents were copied out in the last round)
            const size_t off = c * PIN_CHUNK, n = bytes - off < PIN_CHUNK ? bytes - off : PIN_CHUNK;
            const char* from = (const char*)pin[c & 1];
            constexpr int NT = 4;
            const size_t part = ((n + NT - 1) / NT + 4095) & ~(size_t)4095;
            std::thread th[NT - 1];
            int started = 0;
            for (int t = 1; t < NT; t++) {
                const size_t a = (size_t)t * part;
                if (a >= n) break;
                const size_t m = n - a < part ? n - a : part;
                th[started++] = std::thread([=] { std::memcpy(dst + off + a, from + a, m); });
            }
            std::memcpy(dst + off, from, n < part ? n : part);
            for (int t = 0; t < started; t++) th[t].join();
        }
        if (e != hipSuccess) { set_error("D2H copy failed: %s", hipGetErrorString(e)); return AVSIM_EHIP; }
        return 0;
    }
    int out_end(int slot, void* p, size_t bytes) {
        if (io_device || !p) return 0;
        if (bytes >= ((size_t)16 << 20)) {
            const int rc = out_end_pipelined((const char*)d_io[slot], (char*)p, bytes);
            if (rc != 1) return rc;
        }
        hipError_t e = hipMemcpyAsync(p, d_io[slot], bytes, hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) { set_error("D2H copy failed: %s", hipGetErrorString(e)); return AVSIM_EHIP; }
        return 0;
    }
    int finish() {  // host-pointer mode is synchronous
        if (io_device) return 0;
        hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { set_error("stream sync failed: %s", hipGetErrorString(e)); return AVSIM_EHIP; }
        return 0;
    }
};

// ------------------------------------------------------------------------------------------------
// IK kernels: one problem per lane; blockIdx.y selects the arm so a wave never mixes 6- and 7-DoF code
// ------------------------------------------------------------------------------------------------
template <int NJ>
__global__ void __launch_bounds__(64) k_fk_jac(IkParams P, int arm, int n, const double* __restrict__ q, double* __restrict__ Tout,
                                               double* __restrict__ Jout) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double th[NJ];
#pragma unroll
    for (int k = 0; k < NJ; k++) th[k] = q[(size_t)i * NJ + k];
    if (Tout) {
        double R[9], p[3];
        fk<double, NJ>(P.arm[arm], th, R, p);
        double* T = Tout + (size_t)i * 16;
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int c = 0; c < 3; c++) T[4 * r + c] = R[3 * r + c];
            T[4 * r + 3] = p[r];
        }
        T[12] = T[13] = T[14] = 0;
        T[15] = 1;
    }
    if (Jout) {
        double J[6][NJ];
        jac<double, NJ>(P.arm[arm], th, J);
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int k = 0; k < NJ; k++) Jout[((size_t)i * 6 + r) * NJ + k] = J[r][k];
    }
}

template <int NJ>
__global__ void __launch_bounds__(64) k_ik(IkParams P, int arm, int controller, int iters, int n, const double* __restrict__ q,
                                           const double* __restrict__ pos, const double* __restrict__ quat,
                                           double* __restrict__ qout) {
    // DiffIK: one problem per lane.  GradIK: one problem per 16-lane row (gradik spreads its cost evaluations over the row)
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = controller == 1 ? gt >> 4 : gt;
    const bool live = i < n;
    const int ii = live ? i : n - 1;          // idle rows of the last block shadow the last problem (wave-wide shuffles inside gradik)
    double th[NJ], out[NJ], tp[3], qx[4], Rt[9];
#pragma unroll
    for (int k = 0; k < NJ; k++) th[k] = q[(size_t)ii * NJ + k];
#pragma unroll
    for (int k = 0; k < 3; k++) tp[k] = pos[(size_t)ii * 3 + k];
    qx[0] = quat[(size_t)ii * 4 + 1]; qx[1] = quat[(size_t)ii * 4 + 2]; qx[2] = quat[(size_t)ii * 4 + 3];
    qx[3] = quat[(size_t)ii * 4 + 0];  // wxyz_to_xyzw (diff_ik.py:59)
    quat2mat_xyzw(qx, Rt);
    if (controller == 0) {
        if (!live) return;
        diffik<double, NJ>(P, arm, th, tp, Rt, iters, out);
    } else {
        if constexpr (NJ == 6) gradik<double>(P, arm, th, tp, Rt, iters, out);
        if (!live || (threadIdx.x & 15) != 0) return;
    }
#pragma unroll
    for (int k = 0; k < NJ; k++) qout[(size_t)i * NJ + k] = out[k];
}

// sim_env.py:277-301: Cartesian action -> ctrl, IK seeded with the MEASURED qpos
// The GradIK arms of the reference mode in a kernel of their own, one env per 16-lane row, gradik inlined: as an out-of-line function
// next to the DiffIK code it spilled 3.3 KB per lane at 512 registers.  (Capped at 256 registers for two waves per SIMD it is no
// faster: 50 iterations x 3 rounds of f64 forward kinematics are VALU-bound, not latency-bound.)
template <typename real>
__global__ void __launch_bounds__(64) k_gradik_ctrl(IkParams P, int N, int nq, int nu, const double* __restrict__ act,
                                                                                          const real* __restrict__ qpos, real* __restrict__ ctrl) {
    const int arm = blockIdx.y;
    const int i0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool live = i0 < N;
    const int i = live ? i0 : N - 1;
    const double* a = act + (size_t)i * 23 + (arm == 0 ? 0 : 8);
    double tp[3] = {a[0], a[1], a[2]}, qx[4] = {a[4], a[5], a[6], a[3]}, Rt[9];
    quat2mat_xyzw(qx, Rt);
    const real* qp = qpos + (size_t)i * nq;
    real* c = ctrl + (size_t)i * nu + (arm == 0 ? 0 : 7);
    const IkArm& A = P.arm[arm];
    double th[6], out[6];
#pragma unroll
    for (int k = 0; k < 6; k++) th[k] = (double)qp[A.qadr[k]];
    gradik<double>(P, arm, th, tp, Rt, P.grad_iters, out);
    if (!live || (threadIdx.x & 15) != 0) return;
#pragma unroll
    for (int k = 0; k < 6; k++) c[k] = (real)out[k];
    const double trig = a[7];  // sim_env.py:300-301: unnorm(1 - trigger)
    c[6] = (real)((1.0 - trig) * (P.grip_hi - P.grip_lo) + P.grip_lo);
}

template <typename real>
__global__ void __launch_bounds__(64) k_cart_ctrl(IkParams P, int mode, int arm0, int N, int nq, int nu, const double* __restrict__ act,
                                                  const real* __restrict__ qpos, real* __restrict__ ctrl) {
    // blockIdx.y + arm0 = arm.  One env per lane, except the GradIK arms of the reference mode: one env per 16-lane row
    const int arm = blockIdx.y + arm0;
    (void)mode;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double* a = act + (size_t)i * 23 + (arm == 0 ? 0 : (arm == 1 ? 8 : 16));
    double tp[3] = {a[0], a[1], a[2]}, qx[4] = {a[4], a[5], a[6], a[3]}, Rt[9];
    quat2mat_xyzw(qx, Rt);
    const real* qp = qpos + (size_t)i * nq;
    real* c = ctrl + (size_t)i * nu + (arm == 0 ? 0 : (arm == 1 ? 7 : 14));
    const IkArm& A = P.arm[arm];
    if (arm == 2) {
        double th[7], out[7];
#pragma unroll
        for (int k = 0; k < 7; k++) th[k] = (double)qp[A.qadr[k]];
        diffik<double, 7>(P, 2, th, tp, Rt, P.diff_iters, out);
#pragma unroll
        for (int k = 0; k < 7; k++) c[k] = (real)out[k];
    } else {
        double th[6], out[6];
#pragma unroll
        for (int k = 0; k < 6; k++) th[k] = (double)qp[A.qadr[k]];
        diffik<double, 6>(P, arm, th, tp, Rt, P.diff_iters, out);       // (the GradIK arms of the reference mode: k_gradik_ctrl)
#pragma unroll
        for (int k = 0; k < 6; k++) c[k] = (real)out[k];
        double trig = a[7];  // sim_env.py:300-301: unnorm(1 - trigger)
        c[6] = (real)((1.0 - trig) * (P.grip_hi - P.grip_lo) + P.grip_lo);
    }
}

// ------------------------------------------------------------------------------------------------
// small state kernels
// ------------------------------------------------------------------------------------------------
template <typename real>
__global__ void k_reset(int N, int nq, int nv, int nu, int nobj, const unsigned char* __restrict__ mask,
                        const double* __restrict__ obj, const double* __restrict__ qhome, const double* __restrict__ chome,
                        const int* __restrict__ objadr, real* qpos, real* qvel, real* ctrl, real* warm, int* latch, double* obj_keep) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (mask && !mask[i]) return;
    for (int k = 0; k < nq; k++) qpos[(size_t)i * nq + k] = (real)qhome[k];
    for (int o = 0; o < nobj; o++)
        for (int k = 0; k < 7; k++) {
            const double v = obj[((size_t)i * nobj + o) * 7 + k];
            qpos[(size_t)i * nq + objadr[o] + k] = (real)v;
            obj_keep[((size_t)i * nobj + o) * 7 + k] = v;      // where a diverged env of this episode is put back (check_divergence)
        }
    for (int k = 0; k < nv; k++) { qvel[(size_t)i * nv + k] = 0; warm[(size_t)i * nv + k] = 0; }
    for (int k = 0; k < nu; k++) ctrl[(size_t)i * nu + k] = (real)chome[k];
    latch[i] = 0;
}

template <typename A, typename B>
__global__ void k_convert(size_t n, const A* __restrict__ a, B* __restrict__ b) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = (B)a[i];
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* avsim_last_error(const avsim_t* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static void fill_ik(const Blob& b, IkParams& P) {
    std::memset(&P, 0, sizeof P);
    auto n = b.i("ik_n");
    auto w0 = b.f("ik_w0"), p0 = b.f("ik_p0"), s0 = b.f("ik_site0"), rg = b.f("ik_range");
    auto qa = b.i("ik_qadr");
    auto home = b.f("qpos_home");
    for (int a = 0; a < 3; a++) {
        IkArm& A = P.arm[a];
        A.n = n[a];
        for (int i = 0; i < 7; i++) {
            const double* w = &w0[(a * 7 + i) * 3];
            const double* p = &p0[(a * 7 + i) * 3];
            for (int k = 0; k < 3; k++) A.w[i][k] = w[k];
            // kinematics.py:12  v0 = -cross(w0, p0)
            A.v[i][0] = -(w[1] * p[2] - w[2] * p[1]);
            A.v[i][1] = -(w[2] * p[0] - w[0] * p[2]);
            A.v[i][2] = -(w[0] * p[1] - w[1] * p[0]);
            A.lo[i] = rg[(a * 7 + i) * 2];
            A.hi[i] = rg[(a * 7 + i) * 2 + 1];
            A.qadr[i] = qa[a * 7 + i];
        }
        for (int k = 0; k < 12; k++) A.site0[k] = s0[a * 16 + k];
    }
    // DiffIK parameters: sim_env.py:125-138 (middle arm); manipulators reuse the gains with q0 = home pose
    P.k_pos = 0.9; P.k_ori = 0.9; P.damping = 1.0e-4; P.max_angvel = 3.14; P.dt = 0.04; P.diff_iters = 10;
    const double kn[7] = {10.0, 10.0, 10.0, 10.0, 5.0, 5.0, 5.0};
    for (int a = 0; a < 3; a++)
        for (int i = 0; i < P.arm[a].n; i++) {
            P.k_null[a][i] = kn[i];
            P.q0[a][i] = home[P.arm[a].qadr[i]];
        }
    // GradIK parameters: sim_env.py:89-122
    P.g_step = 1e-4; P.g_min_delta = 1e-12; P.grad_iters = 50; P.g_pw = 500.0; P.g_rw = 100.0;
    P.g_pthr = 1e-3; P.g_rthr = 1e-3; P.g_maxp = 0.1; P.g_maxr = 0.3; P.g_joint_p = 0.9;
    const double jc[6] = {10.0, 10.0, 1.0, 50.0, 1.0, 1.0};
    for (int i = 0; i < 6; i++) { P.g_jcw[i] = jc[i]; P.g_jdw[i] = 50.0; }
    auto gr = b.f("grip_range");
    P.grip_lo = gr[0];
    P.grip_hi = gr[1];
}

int avsim_create(const void* blob, size_t nbytes, int num_envs, int device, uint32_t flags, avsim_t** out) {
    if (!out) return AVSIM_EINVAL;
    *out = nullptr;
    if (!blob || num_envs <= 0) { g_create_error = "avsim_create: bad arguments"; return AVSIM_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_create_error = "avsim_create: no usable HIP device (this library has no CPU path)";
        return AVSIM_ENODEV;
    }
    std::unique_ptr<avsim> h(new avsim);
    h->device = device;
    h->flags = flags;
    h->N = num_envs;
    h->io_device = flags & AVSIM_IO_DEVICE;
    h->f64 = flags & AVSIM_F64_PHYSICS;
    DevGuard dev_guard_(device);
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return AVSIM_EHIP; }
    try {
        Blob b(blob, nbytes);
        h->nq = b.scalar("nq"); h->nv = b.scalar("nv"); h->nu = b.scalar("nu");
        h->task_id = b.scalar("task_id"); h->num_arms = b.scalar("num_arms");
        h->nj = h->num_arms == 3 ? 21 : 14;
        static const int mx[5] = {4, 4, 5, 3, 4};  // env.py:423, 509, 598, 699, 788
        h->max_reward = mx[h->task_id];
        h->obj_qadr = b.i("objects_qposadr");
        h->nobj = (int)h->obj_qadr.size();
        h->obs_qadr = b.i("obs_qposadr");
        h->qpos_home = b.f("qpos_home");
        h->ctrl_home = b.f("ctrl_home");
        fill_ik(b, h->ik);
        std::string perr;
        if (!h->phys.init(b, num_envs, h->f64, perr)) { g_create_error = perr; return AVSIM_EMODEL; }
        h->render.build(b, num_envs);
        h->vis.keep_instances(b);
    } catch (const std::exception& ex) {
        g_create_error = std::string("avsim_create: ") + ex.what();
        return AVSIM_EMODEL;
    }
    e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); return AVSIM_EHIP; }
    h->own_stream = true;
    for (auto& ev : h->ev) (void)hipEventCreate(&ev);
    size_t N = num_envs, r = h->rsz();
    if (hipMalloc(&h->d_qpos, N * h->nq * r) != hipSuccess || hipMalloc(&h->d_qvel, N * h->nv * r) != hipSuccess ||
        hipMalloc(&h->d_ctrl, N * h->nu * r) != hipSuccess || hipMalloc(&h->d_warm, N * h->nv * r) != hipSuccess ||
        hipMalloc((void**)&h->d_latch, N * sizeof(int)) != hipSuccess) {
        g_create_error = "avsim_create: hipMalloc of the state arrays failed";
        avsim_destroy(h.release());
        return AVSIM_EHIP;
    }
    *out = h.release();
    // bring every env to the home pose with objects at their model default pose
    std::vector<double> obj((size_t)num_envs * (*out)->nobj * 7);
    for (int i = 0; i < num_envs; i++)
        for (int o = 0; o < (*out)->nobj; o++)
            for (int k = 0; k < 7; k++) obj[((size_t)i * (*out)->nobj + o) * 7 + k] = (*out)->qpos_home[(*out)->obj_qadr[o] + k];
    bool save = (*out)->io_device;
    (*out)->io_device = false;
    int rc = avsim_reset(*out, nullptr, obj.data());
    (*out)->io_device = save;
    if (rc) {
        g_create_error = (*out)->err;
        avsim_destroy(*out);
        *out = nullptr;
        return rc;
    }
    return AVSIM_OK;
}

void avsim_destroy(avsim_t* h) {
    if (!h) return;
    DevGuard dev_guard_(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->phys.destroy();
    h->render.destroy();
    h->vis.destroy();
    for (void* p : {h->d_qpos, h->d_qvel, h->d_ctrl, h->d_warm, (void*)h->d_latch})
        if (p) (void)hipFree(p);
    for (void* p : h->d_io)
        if (p) (void)hipFree(p);
    for (int k = 0; k < 2; k++) {
        if (h->pin[k]) (void)hipHostFree(h->pin[k]);
        if (h->pin_ev[k]) (void)hipEventDestroy(h->pin_ev[k]);
    }
    for (auto& ev : h->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : h->kev) (void)hipEventDestroy(ev);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int avsim_dims(const avsim_t* h, int32_t d[AVSIM_NDIMS]) {
    if (!h || !d) return AVSIM_EINVAL;
    d[0] = h->nq; d[1] = h->nv; d[2] = h->nu; d[3] = h->nj; d[4] = h->nobj; d[5] = h->max_reward; d[6] = h->N;
    d[7] = h->task_id; d[8] = h->phys.maxcon; d[9] = h->phys.maxefc;
    d[10] = (int32_t)h->phys.lds_bytes();
    d[11] = (int32_t)(160 * 1024 / (h->phys.lds_bytes() ? h->phys.lds_bytes() : 1));
    return AVSIM_OK;
}

int avsim_set_option(avsim_t* h, const char* name, double value) {
    if (!h || !name) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);               // "maxefc" / "maxcon" / "profile_phases" allocate on the handle's device
    if (!std::strcmp(name, "kernel_timing")) { h->ktiming = value != 0; h->render.timing = value != 0; return AVSIM_OK; }
    if (!std::strcmp(name, "render_proxies")) { h->render_proxies = value != 0; return AVSIM_OK; }
    if (!std::strcmp(name, "render_samples")) { if (value != 1 && value != 4) { h->set_error("render_samples is 1 or 4"); return AVSIM_EINVAL; } h->vis.samples = (int)value; return AVSIM_OK; }
    if (!std::strcmp(name, "render_shadows")) { h->vis.shadows = value != 0; return AVSIM_OK; }
    if (!std::strcmp(name, "render_smooth")) { h->vis.smooth_opt = value != 0; h->vis.S.smooth = value != 0 ? 1 : 0; return AVSIM_OK; }
    if (!std::strcmp(name, "render_shadow_size")) { if (value != 512 && value != 1024 && value != 2048) { h->set_error("render_shadow_size is 512, 1024 or 2048"); return AVSIM_EINVAL; } h->vis.shadow_size = (int)value; return AVSIM_OK; }
    if (!std::strcmp(name, "render_cam_major")) { h->vis.cam_major = value != 0; return AVSIM_OK; }
    if (!std::strcmp(name, "render_chunk")) { if (value < 1) { h->set_error("render_chunk is a number of envs >= 1"); return AVSIM_EINVAL; } h->render.env_chunk = (int)value; return AVSIM_OK; }
    if (!std::strcmp(name, "diffik_iters")) { h->ik.diff_iters = (int)value; return AVSIM_OK; }
    if (!std::strcmp(name, "gradik_iters")) { h->ik.grad_iters = (int)value; return AVSIM_OK; }
    try {
        if (h->phys.set_option(name, value)) {
            if (!std::strcmp(name, "num_joints")) h->nj = (int)value;
            return AVSIM_OK;
        }
    } catch (const std::exception& ex) {
        h->set_error("avsim_set_option(%s): %s", name, ex.what());
        return AVSIM_EHIP;
    }
    h->set_error("avsim_set_option: unknown option '%s'", name);
    return AVSIM_EINVAL;
}

int avsim_sync(avsim_t* h) {
    if (!h) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return AVSIM_OK;
}

int avsim_set_stream(avsim_t* h, void* s) {
    if (!h) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)s;
    h->own_stream = false;
    return AVSIM_OK;
}

int avsim_event_record(avsim_t* h, int slot) {
    if (!h || slot < 0 || slot >= 16) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    HIPCHK(h, hipEventRecord(h->ev[slot], h->stream));
    return AVSIM_OK;
}

int avsim_event_elapsed_ms(avsim_t* h, int a, int b, float* ms) {
    if (!h || !ms || a < 0 || a >= 16 || b < 0 || b >= 16) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    HIPCHK(h, hipEventSynchronize(h->ev[b]));
    HIPCHK(h, hipEventElapsedTime(ms, h->ev[a], h->ev[b]));
    return AVSIM_OK;
}

int avsim_kernel_time(avsim_t* h, int reset, double* total_ms, int64_t* launches) {
    if (!h) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    double tot = 0;
    for (size_t i = 0; i + 1 < h->kev_used; i += 2) {
        float ms = 0;
        HIPCHK(h, hipEventSynchronize(h->kev[i + 1]));
        HIPCHK(h, hipEventElapsedTime(&ms, h->kev[i], h->kev[i + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot + h->kev_ms;
    if (launches) *launches = (int64_t)(h->kev_used / 2) + h->kev_n;
    if (reset) { h->kev_used = 0; h->kev_ms = 0; h->kev_n = 0; }
    return AVSIM_OK;
}

int avsim_render_kernel_time(avsim_t* h, int reset, double* total_ms, int64_t* launches) {
    if (!h) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    double tot = 0;
    for (size_t i = 0; i + 1 < h->render.tev_used; i += 2) {
        float ms = 0;
        HIPCHK(h, hipEventSynchronize(h->render.tev[i + 1]));
        HIPCHK(h, hipEventElapsedTime(&ms, h->render.tev[i], h->render.tev[i + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot + h->render.tev_ms;
    if (launches) *launches = (int64_t)(h->render.tev_used / 2) + h->render.tev_n;
    if (reset) { h->render.tev_used = 0; h->render.tev_ms = 0; h->render.tev_n = 0; }
    return AVSIM_OK;
}

int avsim_observe(avsim_t* h, double* agent_pos, int32_t* reward, uint8_t* success) {
    if (!h) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    int rc;
    void *dap = nullptr, *drw = nullptr, *dsu = nullptr;
    size_t N = h->N;
    if (agent_pos && (rc = h->out_begin(4, agent_pos, sizeof(double) * N * h->nj, &dap))) return rc;
    if (reward && (rc = h->out_begin(5, reward, sizeof(int32_t) * N, &drw))) return rc;
    if (success && (rc = h->out_begin(6, success, N, &dsu))) return rc;
    h->phys.force_reward = 1;
    rc = h->phys.launch(h->stream, h->N, 0, nullptr, h->nj, h->d_qpos, h->d_qvel, h->d_ctrl, h->d_warm, h->d_latch, (double*)dap, (int32_t*)drw,
                        (uint8_t*)dsu, h->err);
    h->phys.force_reward = 0;
    if (rc) return rc;
    if ((rc = h->out_end(4, agent_pos, sizeof(double) * N * h->nj))) return rc;
    if ((rc = h->out_end(5, reward, sizeof(int32_t) * N))) return rc;
    if ((rc = h->out_end(6, success, N))) return rc;
    return h->finish();
}

int avsim_fk_jac(avsim_t* h, int arm, int n, const double* q, double* T, double* J) {
    if (!h || arm < 0 || arm > 2 || n <= 0 || !q) { if (h) h->set_error("avsim_fk_jac: bad arguments"); return AVSIM_EINVAL; }
    AVS_ON_DEVICE(h);
    int nj = h->ik.arm[arm].n, rc;
    const void* dq;
    void *dT = nullptr, *dJ = nullptr;
    if ((rc = h->in(0, q, sizeof(double) * n * nj, &dq))) return rc;
    if (T && (rc = h->out_begin(1, T, sizeof(double) * n * 16, &dT))) return rc;
    if (J && (rc = h->out_begin(2, J, sizeof(double) * n * 6 * nj, &dJ))) return rc;
    dim3 grid((n + 63) / 64);
    if (nj == 6) hipLaunchKernelGGL(k_fk_jac<6>, grid, dim3(64), 0, h->stream, h->ik, arm, n, (const double*)dq, (double*)dT, (double*)dJ);
    else hipLaunchKernelGGL(k_fk_jac<7>, grid, dim3(64), 0, h->stream, h->ik, arm, n, (const double*)dq, (double*)dT, (double*)dJ);
    HIPCHK(h, hipGetLastError());
    if ((rc = h->out_end(1, T, sizeof(double) * n * 16))) return rc;
    if ((rc = h->out_end(2, J, sizeof(double) * n * 6 * nj))) return rc;
    return h->finish();
}

int avsim_ik(avsim_t* h, int arm, int controller, int max_iters, int n, const double* q, const double* pos,
             const double* quat, double* qout) {
    if (!h || arm < 0 || arm > 2 || n <= 0 || !q || !pos || !quat || !qout || controller < 0 || controller > 1) {
        if (h) h->set_error("avsim_ik: bad arguments");
        return AVSIM_EINVAL;
    }
    if (controller == 1 && arm == 2) { h->set_error("avsim_ik: GradIK is defined for the 6-DoF manipulators only"); return AVSIM_EINVAL; }
    AVS_ON_DEVICE(h);
    int nj = h->ik.arm[arm].n, rc;
    int iters = max_iters > 0 ? max_iters : (controller == 0 ? h->ik.diff_iters : h->ik.grad_iters);
    const void *dq, *dp, *dqt;
    void* dout;
    if ((rc = h->in(0, q, sizeof(double) * n * nj, &dq))) return rc;
    if ((rc = h->in(1, pos, sizeof(double) * n * 3, &dp))) return rc;
    if ((rc = h->in(2, quat, sizeof(double) * n * 4, &dqt))) return rc;
    if ((rc = h->out_begin(3, qout, sizeof(double) * n * nj, &dout))) return rc;
    dim3 grid(controller == 1 ? (n + 3) / 4 : (n + 63) / 64);
    if (nj == 6)
        hipLaunchKernelGGL(k_ik<6>, grid, dim3(64), 0, h->stream, h->ik, arm, controller, iters, n, (const double*)dq, (const double*)dp,
                           (const double*)dqt, (double*)dout);
    else
        hipLaunchKernelGGL(k_ik<7>, grid, dim3(64), 0, h->stream, h->ik, arm, controller, iters, n, (const double*)dq, (const double*)dp,
                           (const double*)dqt, (double*)dout);
    HIPCHK(h, hipGetLastError());
    if ((rc = h->out_end(3, qout, sizeof(double) * n * nj))) return rc;
    return h->finish();
}

}  // extern "C"

template <typename real>
static int reset_impl(avsim_t* h, const uint8_t* mask, const double* obj) {
    int rc;
    const void *dm = nullptr, *dobj;
    if (mask && (rc = h->in(0, mask, h->N, &dm))) return rc;
    if ((rc = h->in(1, obj, sizeof(double) * h->N * h->nobj * 7, &dobj))) return rc;
    hipLaunchKernelGGL(k_reset<real>, dim3((h->N + 63) / 64), dim3(64), 0, h->stream, h->N, h->nq, h->nv, h->nu, h->nobj,
                       (const unsigned char*)dm, (const double*)dobj, h->phys.d_qpos_home, h->phys.d_ctrl_home, h->phys.d_obj_qadr,
                       (real*)h->d_qpos, (real*)h->d_qvel, (real*)h->d_ctrl, (real*)h->d_warm, h->d_latch, h->phys.d_obj_reset);
    HIPCHK(h, hipGetLastError());
    // mj_forward (env.py:244, 538): refresh kinematics + contacts of the new state, no time stepping
    if ((rc = h->phys.launch(h->stream, h->N, 0, nullptr, h->nj, h->d_qpos, h->d_qvel, h->d_ctrl, h->d_warm, h->d_latch, nullptr, nullptr,
                             nullptr, h->err)))
        return rc;
    return h->finish();
}

extern "C" {

int avsim_reset(avsim_t* h, const uint8_t* mask, const double* obj_qpos) {
    if (!h || !obj_qpos) { if (h) h->set_error("avsim_reset: obj_qpos is required"); return AVSIM_EINVAL; }
    AVS_ON_DEVICE(h);
    h->state_ver++;
    return h->f64 ? reset_impl<double>(h, mask, obj_qpos) : reset_impl<float>(h, mask, obj_qpos);
}

static int step_common(avsim_t* h, const float* d_action, int nsub, double* agent_pos, int32_t* reward, uint8_t* success) {
    int rc;
    h->state_ver++;
    void *dap = nullptr, *drw = nullptr, *dsu = nullptr;
    size_t N = h->N;
    if (agent_pos && (rc = h->out_begin(4, agent_pos, sizeof(double) * N * h->nj, &dap))) return rc;
    if (reward && (rc = h->out_begin(5, reward, sizeof(int32_t) * N, &drw))) return rc;
    if (success && (rc = h->out_begin(6, success, N, &dsu))) return rc;
    if (h->ktiming) {
        if (h->kev_used >= 2 * EV_RING) {      // fold the recorded pairs into the running sum (waits for the newest of them: once per EV_RING launches)
            for (size_t i = 0; i + 1 < h->kev_used; i += 2) {
                float ms = 0;
                HIPCHK(h, hipEventSynchronize(h->kev[i + 1]));
                HIPCHK(h, hipEventElapsedTime(&ms, h->kev[i], h->kev[i + 1]));
                h->kev_ms += ms;
            }
            h->kev_n += (int64_t)(h->kev_used / 2);
            h->kev_used = 0;
        }
        while (h->kev.size() < h->kev_used + 2) { hipEvent_t e; HIPCHK(h, hipEventCreate(&e)); h->kev.push_back(e); }
        HIPCHK(h, hipEventRecord(h->kev[h->kev_used], h->stream));
    }
    if ((rc = h->phys.launch(h->stream, h->N, nsub, d_action, h->nj, h->d_qpos, h->d_qvel, h->d_ctrl, h->d_warm, h->d_latch, (double*)dap,
                             (int32_t*)drw, (uint8_t*)dsu, h->err)))
        return rc;
    if (h->ktiming) {
        HIPCHK(h, hipEventRecord(h->kev[h->kev_used + 1], h->stream));
        h->kev_used += 2;
    }
    if ((rc = h->out_end(4, agent_pos, sizeof(double) * N * h->nj))) return rc;
    if ((rc = h->out_end(5, reward, sizeof(int32_t) * N))) return rc;
    if ((rc = h->out_end(6, success, N))) return rc;
    return h->finish();
}

int avsim_step(avsim_t* h, const float* action, int nsub, double* agent_pos, int32_t* reward, uint8_t* success) {
    if (!h || !action || nsub < 0) { if (h) h->set_error("avsim_step: bad arguments"); return AVSIM_EINVAL; }
    AVS_ON_DEVICE(h);
    const void* da;
    int rc;
    if ((rc = h->in(0, action, sizeof(float) * h->N * h->nj, &da))) return rc;
    return step_common(h, (const float*)da, nsub, agent_pos, reward, success);
}

int avsim_step_ctrl(avsim_t* h, int nsub, double* agent_pos, int32_t* reward, uint8_t* success) {
    if (!h || nsub < 0) { if (h) h->set_error("avsim_step_ctrl: bad arguments"); return AVSIM_EINVAL; }
    AVS_ON_DEVICE(h);
    return step_common(h, nullptr, nsub, agent_pos, reward, success);
}

int avsim_step_cartesian(avsim_t* h, const double* action23, int ik_mode, int nsub, double* agent_pos, int32_t* reward,
                         uint8_t* success) {
    if (!h || !action23 || nsub < 0 || (ik_mode != AVSIM_IK_REFERENCE && ik_mode != AVSIM_IK_DLS)) {
        if (h) h->set_error("avsim_step_cartesian: bad arguments");
        return AVSIM_EINVAL;
    }
    if (h->num_arms != 3) { h->set_error("avsim_step_cartesian: the 23-D Cartesian action drives three arms (sim_env.py:277-282)"); return AVSIM_EINVAL; }
    AVS_ON_DEVICE(h);
    const void* da;
    int rc;
    if ((rc = h->in(0, action23, sizeof(double) * h->N * 23, &da))) return rc;
    auto launch = [&](dim3 grid, int arm0) {
        if (h->f64)
            hipLaunchKernelGGL(k_cart_ctrl<double>, grid, dim3(64), 0, h->stream, h->ik, ik_mode, arm0, h->N, h->nq, h->nu, (const double*)da,
                               (const double*)h->d_qpos, (double*)h->d_ctrl);
        else
            hipLaunchKernelGGL(k_cart_ctrl<float>, grid, dim3(64), 0, h->stream, h->ik, ik_mode, arm0, h->N, h->nq, h->nu, (const double*)da,
                               (const float*)h->d_qpos, (float*)h->d_ctrl);
    };
    if (ik_mode == AVSIM_IK_DLS) launch(dim3((h->N + 63) / 64, 3), 0);
    else {
        // GradIK on the two manipulators: 16 lanes per env
        if (h->f64) hipLaunchKernelGGL(k_gradik_ctrl<double>, dim3((h->N + 3) / 4, 2), dim3(64), 0, h->stream, h->ik, h->N, h->nq, h->nu, (const double*)da, (const double*)h->d_qpos, (double*)h->d_ctrl);
        else hipLaunchKernelGGL(k_gradik_ctrl<float>, dim3((h->N + 3) / 4, 2), dim3(64), 0, h->stream, h->ik, h->N, h->nq, h->nu, (const double*)da, (const float*)h->d_qpos, (float*)h->d_ctrl);
        launch(dim3((h->N + 63) / 64, 1), 2);     // DiffIK on the camera arm: one env per lane
    }
    HIPCHK(h, hipGetLastError());
    return step_common(h, nullptr, nsub, agent_pos, reward, success);
}

// ---- state access --------------------------------------------------------------------------------
static int put(avsim_t* h, int slot, const double* src, void* dst, size_t n) {
    if (!src) return 0;
    const void* d;
    int rc;
    if ((rc = h->in(slot, src, sizeof(double) * n, &d))) return rc;
    unsigned g = (unsigned)((n + 255) / 256);
    if (h->f64) hipLaunchKernelGGL((k_convert<double, double>), dim3(g), dim3(256), 0, h->stream, n, (const double*)d, (double*)dst);
    else hipLaunchKernelGGL((k_convert<double, float>), dim3(g), dim3(256), 0, h->stream, n, (const double*)d, (float*)dst);
    return 0;
}
static int get(avsim_t* h, int slot, double* dst, const void* src, size_t n) {
    if (!dst) return 0;
    void* d;
    int rc;
    if ((rc = h->out_begin(slot, dst, sizeof(double) * n, &d))) return rc;
    unsigned g = (unsigned)((n + 255) / 256);
    if (h->f64) hipLaunchKernelGGL((k_convert<double, double>), dim3(g), dim3(256), 0, h->stream, n, (const double*)src, (double*)d);
    else hipLaunchKernelGGL((k_convert<float, double>), dim3(g), dim3(256), 0, h->stream, n, (const float*)src, (double*)d);
    return h->out_end(slot, dst, sizeof(double) * n);
}

int avsim_get_state(avsim_t* h, double* qpos, double* qvel, double* ctrl, double* warm) {
    if (!h) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    size_t N = h->N;
    int rc;
    if ((rc = get(h, 0, qpos, h->d_qpos, N * h->nq))) return rc;
    if ((rc = get(h, 1, qvel, h->d_qvel, N * h->nv))) return rc;
    if ((rc = get(h, 2, ctrl, h->d_ctrl, N * h->nu))) return rc;
    if ((rc = get(h, 3, warm, h->d_warm, N * h->nv))) return rc;
    HIPCHK(h, hipGetLastError());
    return h->finish();
}

int avsim_set_state(avsim_t* h, const double* qpos, const double* qvel, const double* ctrl, const double* warm) {
    if (!h) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    h->state_ver++;
    size_t N = h->N;
    int rc;
    if ((rc = put(h, 0, qpos, h->d_qpos, N * h->nq))) return rc;
    if ((rc = put(h, 1, qvel, h->d_qvel, N * h->nv))) return rc;
    if ((rc = put(h, 2, ctrl, h->d_ctrl, N * h->nu))) return rc;
    if ((rc = put(h, 3, warm, h->d_warm, N * h->nv))) return rc;
    HIPCHK(h, hipGetLastError());
    if ((rc = h->phys.launch(h->stream, h->N, 0, nullptr, h->nj, h->d_qpos, h->d_qvel, h->d_ctrl, h->d_warm, h->d_latch, nullptr, nullptr,
                             nullptr, h->err)))
        return rc;
    return h->finish();
}

int avsim_get_latch(avsim_t* h, int32_t* latch) {
    if (!h || !latch) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    HIPCHK(h, hipMemcpyAsync(latch, h->d_latch, sizeof(int32_t) * (size_t)h->N, h->io_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    return h->finish();
}

int avsim_set_latch(avsim_t* h, const int32_t* latch) {
    if (!h || !latch) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    HIPCHK(h, hipMemcpyAsync(h->d_latch, latch, sizeof(int32_t) * (size_t)h->N, h->io_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    return h->finish();
}

// The object poses a diverged env falls back to (mj_checkPos-style reset): written by avsim_reset; state next to the
// latch for a handle that continues another one's episode (hide / show_middle_arm, checkpoints): double[N][nobj][7]
int avsim_get_reset_poses(avsim_t* h, double* obj_qpos) {
    if (!h || !obj_qpos) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    HIPCHK(h, hipMemcpyAsync(obj_qpos, h->phys.d_obj_reset, sizeof(double) * (size_t)h->N * h->nobj * 7, h->io_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    return h->finish();
}

int avsim_set_reset_poses(avsim_t* h, const double* obj_qpos) {
    if (!h || !obj_qpos) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    HIPCHK(h, hipMemcpyAsync(h->phys.d_obj_reset, obj_qpos, sizeof(double) * (size_t)h->N * h->nobj * 7, h->io_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    return h->finish();
}

int avsim_set_qpos(avsim_t* h, const double* qpos) {
    if (!h || !qpos) return AVSIM_EINVAL;
    return avsim_set_state(h, qpos, nullptr, nullptr, nullptr);
}

int avsim_get_contacts(avsim_t* h, int32_t* ncon, int32_t* pairs, double* dist) {
    if (!h) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    size_t N = h->N, cap = h->phys.maxcon;
    if (h->io_device) {
        if (ncon) HIPCHK(h, hipMemcpyAsync(ncon, h->phys.d_ncon, N * 4, hipMemcpyDeviceToDevice, h->stream));
        if (pairs) HIPCHK(h, hipMemcpyAsync(pairs, h->phys.d_cpairs, N * cap * 8, hipMemcpyDeviceToDevice, h->stream));
        if (dist) HIPCHK(h, hipMemcpyAsync(dist, h->phys.d_cdist, N * cap * 8, hipMemcpyDeviceToDevice, h->stream));
        return AVSIM_OK;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (ncon) HIPCHK(h, hipMemcpy(ncon, h->phys.d_ncon, N * 4, hipMemcpyDeviceToHost));
    if (pairs) HIPCHK(h, hipMemcpy(pairs, h->phys.d_cpairs, N * cap * 8, hipMemcpyDeviceToHost));
    if (dist) HIPCHK(h, hipMemcpy(dist, h->phys.d_cdist, N * cap * 8, hipMemcpyDeviceToHost));
    return AVSIM_OK;
}

/* debug: per-env cycle counters of the 8 physics phases of the last launch (option "profile_phases"); int64[N][10] */
int avsim_get_phase_cycles(avsim_t* h, int64_t* out) {
    if (!h || !out || !h->phys.d_prof) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->phys.d_prof, (size_t)h->N * avs::PROF_W * 8, h->io_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return AVSIM_OK;
}

int avsim_get_diag(avsim_t* h, int32_t* diag) {
    if (!h || !diag) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    size_t N = h->N;
    if (h->io_device) {
        HIPCHK(h, hipMemcpyAsync(diag, h->phys.d_diag, N * 16, hipMemcpyDeviceToDevice, h->stream));
        return AVSIM_OK;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(diag, h->phys.d_diag, N * 16, hipMemcpyDeviceToHost));
    return AVSIM_OK;
}

}  // extern "C"

extern "C" {

// E6 (env.py:180-188 get_obs pixels / :195-200 render) as depth images: forward pass of the physics kernel (nsub = 0) exports
// the body poses, then the two render kernels run on the same stream.
static int render_images(avsim_t* h, const int32_t* cam_ids, int ncam, int height, int width, void* out, bool rgb) {
    const char* who = rgb ? "avsim_render_rgb" : "avsim_render_depth";
    if (!h || !cam_ids || !out) { if (h) h->set_error("%s: bad arguments", who); return AVSIM_EINVAL; }
    AVS_ON_DEVICE(h);
    int rc;
    void* dout = nullptr;
    const size_t bytes = (rgb ? 3 : sizeof(float)) * (size_t)h->N * ncam * height * width;
    if ((rc = h->out_begin(7, out, bytes, &dout))) return rc;
    if (h->xpose_ver != h->state_ver) {          // body poses of the current state (a forward pass of the physics kernel, no substep)
        h->phys.d_xpose = h->render.d_xpose;
        rc = h->phys.launch(h->stream, h->N, 0, nullptr, h->nj, h->d_qpos, h->d_qvel, h->d_ctrl, h->d_warm, h->d_latch, nullptr, nullptr, nullptr, h->err);
        h->phys.d_xpose = nullptr;
        if (rc) return rc;
        h->xpose_ver = h->state_ver;
    }
    if (rgb && h->vis.cam_major && !(h->vis.loaded && !h->render_proxies)) { h->set_error("option render_cam_major applies to the visual scene's images only (avsim_load_visual, render_proxies 0)"); return AVSIM_EINVAL; }
    if (rgb && h->vis.loaded && !h->render_proxies)
        rc = h->vis.launch(h->stream, h->N, h->render.d_xpose, (const int*)cam_ids, ncam, h->render.m.ncam, height, width, dout, h->err, h->state_ver);
    else
        rc = h->render.launch(h->stream, (const int*)cam_ids, ncam, height, width, dout, rgb, h->err);
    if (rc) return rc < -1 ? AVSIM_EHIP : AVSIM_EINVAL;
    if ((rc = h->out_end(7, out, bytes))) return rc;
    return h->finish();
}
int avsim_render_depth(avsim_t* h, const int32_t* cam_ids, int ncam, int height, int width, float* out) {
    return render_images(h, cam_ids, ncam, height, width, out, false);
}
// E6 as colour images of the same proxies (env.py:180-188 "pixels" u8[H][W][3], :195-200 render)
int avsim_render_rgb(avsim_t* h, const int32_t* cam_ids, int ncam, int height, int width, uint8_t* out) {
    return render_images(h, cam_ids, ncam, height, width, out, true);
}

// The visual scene of avsim_render_rgb: the mesh library (models/visual_meshes.avv, compiler/vismesh.py) against the instances the
// model blob carries.  From then on avsim_render_rgb draws the visual meshes (option "render_proxies" 1: the collision proxies again).
int avsim_load_visual(avsim_t* h, const void* library_blob, size_t nbytes) {
    if (!h || !library_blob) { if (h) h->set_error("avsim_load_visual: bad arguments"); return AVSIM_EINVAL; }
    AVS_ON_DEVICE(h);
    try {
        Blob lib(library_blob, nbytes);
        h->vis.load(lib, h->render.m.nbody, h->render.m.cam_pos, h->render.m.cam_mat, h->render.m.cam_fovy, h->render.m.cam_body, h->render.m.light, h->render.m.znear);
    } catch (const std::exception& e) {
        h->set_error("avsim_load_visual: %s", e.what());
        return AVSIM_EMODEL;
    }
    return AVSIM_OK;
}
// triangles / vertices of the loaded visual scene (0 when none is loaded); overflow flags of the last visual render: bit 0 a view ran
// out of triangle records, bit 1 out of tile-list entries (the image then lacks triangles); synchronises the stream
int avsim_visual_info(avsim_t* h, int32_t info[4]) {
    if (!h || !info) return AVSIM_EINVAL;
    AVS_ON_DEVICE(h);
    info[0] = h->vis.loaded ? h->vis.S.ntri : 0; info[1] = h->vis.loaded ? h->vis.S.nvert : 0; info[2] = 0; info[3] = h->vis.have_inst ? (int)h->vis.inst_mesh.size() : 0;
    if (h->vis.loaded && h->vis.X.flags && h->vis.last_nviews > 0) {      // the views of the LAST launch only: rows beyond them hold an earlier, larger call's flags
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::vector<int> f((size_t)h->vis.last_nviews * 8);
        HIPCHK(h, hipMemcpy(f.data(), h->vis.X.flags, f.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t v = 0; v < f.size(); v += 8) info[2] |= f[v];
    }
    return AVSIM_OK;
}

// debug: per-view records of the last visual render, int32[nviews][8] = {overflow bits, cycles / 1024 of the stages transform, set-up,
// count, fill, tiles, triangle records, tile-list entries}; nviews = num_envs x cameras of that call
int avsim_visual_profile(avsim_t* h, int32_t* out, int nviews) {
    if (!h || !out || nviews < 0 || nviews > h->vis.nviews_cap || !h->vis.X.flags) { if (h) h->set_error("avsim_visual_profile: no visual render of that size yet"); return AVSIM_EINVAL; }
    AVS_ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->vis.X.flags, (size_t)nviews * 8 * sizeof(int), hipMemcpyDeviceToHost));
    return AVSIM_OK;
}

int avsim_camera_count(const avsim_t* h) { return h ? h->render.m.ncam : 0; }

// R1 (env.py:425-863, get_reward x5) on caller-supplied contact lists: the class-bit predicate the step kernel applies
// to its own contacts (reward_pair_flags / reward_from_flags), one thread per list; host pointers
int avsim_reward_from_pairs(avsim_t* h, const int32_t* geom_pairs, int nsets, int cap, int32_t* latch, int32_t* reward) {
    if (!h || (!geom_pairs && cap > 0) || !reward || nsets < 0 || cap < 0) { if (h) h->set_error("avsim_reward_from_pairs: bad arguments"); return AVSIM_EINVAL; }
    if (nsets == 0) return AVSIM_OK;
    AVS_ON_DEVICE(h);
    int *dp = nullptr, *dl = nullptr, *dr = nullptr;
    const size_t pb = sizeof(int) * (size_t)nsets * (cap ? cap : 1) * 2, nb = sizeof(int) * (size_t)nsets;
    HIPCHK(h, hipMalloc(&dp, pb + 2 * nb));
    dl = dp + (pb / sizeof(int)); dr = dl + nsets;
    hipError_t e = hipSuccess;
    if (cap) e = hipMemcpyAsync(dp, geom_pairs, sizeof(int) * (size_t)nsets * cap * 2, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = latch ? hipMemcpyAsync(dl, latch, nb, hipMemcpyHostToDevice, h->stream) : hipMemsetAsync(dl, 0, nb, h->stream);
    if (e == hipSuccess) {
        GLB_PTR(const int) gc = h->phys.f64 ? h->phys.md.geom_class : h->phys.mf.geom_class;
        const int ng = h->phys.f64 ? h->phys.md.ngeom : h->phys.mf.ngeom, task = h->phys.f64 ? h->phys.md.task_id : h->phys.mf.task_id;
        hipLaunchKernelGGL(k_reward_pairs, dim3((nsets + 255) / 256), dim3(256), 0, h->stream, gc, ng, task, dp, nsets, cap, dl, dr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(reward, dr, nb, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && latch) e = hipMemcpyAsync(latch, dl, nb, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(dp);
    if (e != hipSuccess) { h->set_error("avsim_reward_from_pairs: %s", hipGetErrorString(e)); return AVSIM_EHIP; }
    return AVSIM_OK;
}

}  // extern "C"
