// avsim_phys.hip.h -- the batched physics kernel: nsub substeps of forward dynamics + contact solve +
// integration for one env per group of G lanes, all working state in LDS.
//
// Replaces `self._physics.step(nstep=20)` (gym_guided_vision/gym_guided_vision/env.py:218, MuJoCo
// mj_step [EXT]) plus the obs/reward tail of env.py:220-224.  Stages follow SURVEY.md 8(a) P1..P9:
//   P1 kinematics  P2 CRB mass matrix + per-tree Cholesky  P5 RNE bias  P6 position servos
//   P7 smooth acceleration  P3 collision (bounding-sphere broad phase over the compiled pair list,
//   narrow phase one pair per lane)  P4 soft-constraint rows (equality, dry friction, limits,
//   elliptic contacts)  P8 projected Gauss-Seidel in acceleration space + noslip sweeps
//   P9 semi-implicit Euler with implicit joint damping.
// Layout: a block is ONE wavefront (64 lanes) holding 64/G envs; lanes of a group cooperate through
// LDS and wave-level fences only (no block barrier is needed inside a single wave).
// Constraint rows are stored sparse by kinematic tree: every row touches at most two trees of <= 8 dofs,
// so J and B = J M^-1 are 16 words each and the PGS row update is a 16-lane (one DPP row) dot product.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "avsim_collide.hip.h"
#include "avsim_math.hip.h"
#include "avsim_model.h"

namespace avs {

enum { J_FREE = 0, J_BALL = 1, J_SLIDE = 2, J_HINGE = 3 };
enum { R_EQ = 0, R_FLOSS = 1, R_LIMIT = 2, R_CONTACT = 3 };
constexpr int TREE_W = 8;     // max dofs of one kinematic tree (8, 8, 7, 6, 6 here)
constexpr int ROW_W = 2 * TREE_W;
// LDS stride of the per-row solver record: odd, so that the row-per-lane loops (lane i reads word k of row i) spread over
// all 32 banks instead of hammering 4 of them (stride 8)
constexpr int ROW_S = ROW_W;       // Jacobian rows live in global memory (L2-resident scratch): 64-byte rows, no banks to dodge
constexpr int RS_S = 9;            // solver record: 8 words used
constexpr int CAND_MAX = 256;  // exact broad-phase survivors per substep (narrow-phase work list, global scratch)
constexpr int PROF_W = 26;   // words per env of the phase profile: 8 phases, broad / narrow, 16 solver / probe slots
constexpr int ANC_MAX = 64;    // LDS ints of the kinematics' pointer-jumping table (one per body)
constexpr int NEAR_MAX = 512;  // Verlet neighbour list: pairs within reach + skin, rebuilt when a geom moved > skin/2

template <typename real>
struct DevModel {
    int nq, nv, nu, nbody, njnt, ngeom, npair, ntree, neq, nfloss, nlimited, nment, task_id, nj, msize;
    real timestep, gravity[3], impratio, grip_lo, grip_hi;
    int noslip_iters;
    int noslip_trees;             // 1 (default): noslip passes whose contacts all touch one kinematic tree run per tree, the trees' chains side by side (option "noslip_trees")
    int qcqp_tridiag;             // sliding contacts' multiplier iteration: 0 MuJoCo's Cholesky per iterate (f64 default), 1 the same iterates through the tridiagonal form, 2 tridiagonal form + secular-equation steps (f32 default); option "qcqp_tridiag"
    int newton_early_exit;        // 1 (default): Newton leaves without the confirming gradient evaluation after an exact step inside one quadratic piece (option "newton_early_exit")
    int newton_component;         // 1 (default): in scenes with rows that couple two trees Newton's dense factorisation / substitutions take the coupled component only; 0: all nv columns (option "newton_component"; same bits)
    int noslip_per_tree;          // 1: the dry-friction rows of the noslip pass go per kinematic tree (needs <= 8 trees); option "noslip_per_tree"
    int solver, newton_iters;     // 0 = PGS (dual), 1 = Newton (primal, the reference's default solver)
    real newton_tol, nscale;      // MuJoCo tolerance and 1/(meaninertia*nv) scaling of the termination tests
    real ls_tolerance;            // Newton's line search stops at |phi'(alpha)| < ls_tolerance |phi'(0)| (option "ls_tolerance"; default 1e-10 in f64, 1e-4 in f32 -- MuJoCo's mjOption.ls_tolerance is 0.01 of another scale [EXT])
    int ls_iterations;            // ... after at most this many evaluations beyond phi'(0) (option "ls_iterations"; default 50 = MuJoCo's mjOption.ls_iterations [EXT])
    // bodies
    GLB_PTR(const int) body_parent;
    GLB_PTR(const int) body_jntadr;
    GLB_PTR(const int) body_jntnum;
    GLB_PTR(const int) body_dofadr;
    GLB_PTR(const int) body_dofnum;
    GLB_PTR(const int) body_tree;
    GLB_PTR(const int) body_dofmask;
    GLB_PTR(const int) body_last;
    GLB_PTR(const real) body_pos;
    GLB_PTR(const real) body_quat;
    GLB_PTR(const real) body_mass;
    GLB_PTR(const real) body_ipos;
    GLB_PTR(const real) body_inertia;
    GLB_PTR(const real) body_invweight0;
    GLB_PTR(const real) static_xpos;
    GLB_PTR(const real) static_xmat;
    GLB_PTR(const int) tree_bodyadr;
    GLB_PTR(const int) tree_bodylist;
    GLB_PTR(const int) tree_dofadr;
    GLB_PTR(const int) tree_dofnum;
    GLB_PTR(const int) tree_madr;
    // joints / dofs
    GLB_PTR(const int) jnt_type;
    GLB_PTR(const int) jnt_qposadr;
    GLB_PTR(const int) jnt_dofadr;
    GLB_PTR(const int) jnt_actfrclimited;
    GLB_PTR(const int) limited_jnt;
    GLB_PTR(const real) jnt_pos;
    GLB_PTR(const real) jnt_axis;
    GLB_PTR(const real) jnt_range;
    GLB_PTR(const real) jnt_actfrcrange;
    GLB_PTR(const real) jnt_solref;
    GLB_PTR(const real) jnt_solimp;
    GLB_PTR(const real) jnt_margin;
    GLB_PTR(const int) dof_body;
    GLB_PTR(const int) dof_parent;
    GLB_PTR(const int) dof_tree;
    GLB_PTR(const int) dof_jnt;
    GLB_PTR(const int) floss_dof;
    GLB_PTR(const int) ment_i;
    GLB_PTR(const int) ment_j;
    GLB_PTR(const real) dof_armature;
    GLB_PTR(const real) dof_damping;
    GLB_PTR(const real) dof_frictionloss;
    GLB_PTR(const real) dof_invweight0;
    GLB_PTR(const real) dof_solref;
    GLB_PTR(const real) dof_solimp;
    // actuators, equalities
    GLB_PTR(const int) act_dof;
    GLB_PTR(const int) act_qposadr;
    GLB_PTR(const int) act_ctrllimited;
    GLB_PTR(const real) act_kp;
    GLB_PTR(const real) act_kv;
    GLB_PTR(const real) act_gear;
    GLB_PTR(const real) act_ctrlrange;
    GLB_PTR(const int) eq_dof1;
    GLB_PTR(const int) eq_dof2;
    GLB_PTR(const int) eq_qpos1;
    GLB_PTR(const int) eq_qpos2;
    GLB_PTR(const real) eq_polycoef;
    GLB_PTR(const real) eq_solref;
    GLB_PTR(const real) eq_solimp;
    GLB_PTR(const real) qpos0;
    GLB_PTR(const real) qpos_home;   // state an env falls back to when its simulation diverges
    int nobj;                        // free objects; their qpos addresses and the poses the env's episode started with (avsim_reset)
    GLB_PTR(const int) obj_qadr;
    const double* obj_reset;         // double[N][nobj][7]
    // geoms
    GLB_PTR(const int) geom_type;
    GLB_PTR(const int) geom_body;
    GLB_PTR(const int) geom_hull;
    GLB_PTR(const int) geom_class;
    GLB_PTR(const int) geom_static;
    GLB_PTR(const real) geom_pos;
    GLB_PTR(const real) geom_mat;
    GLB_PTR(const real) geom_size;
    GLB_PTR(const real) geom_cpos;
    GLB_PTR(const real) geom_rbound;
    GLB_PTR(const real) geom_xpos0;
    GLB_PTR(const real) geom_xmat0;
    GLB_PTR(const real) geom_cen0;
    GLB_PTR(const real) geom_aabb0;
    GLB_PTR(const real) geom_lbox;
    GLB_PTR(const real) hull_vert;     // the collision hulls' support tables: a record of eight candidate entries per cube-map cell, then the overflow part (avsim_collide.hip.h Shape)
    int hull_ovf;                      // entry index at which the overflow part starts; geom_hull[g] = (first cell record of the geom's hull, cube-map resolution R)
    // pairs
    GLB_PTR(const int) pair_geom;
    GLB_PTR(const int) pair_condim;
    GLB_PTR(const real) pair_friction;
    GLB_PTR(const real) pair_solref;
    GLB_PTR(const real) pair_solimp;
    GLB_PTR(const real) pair_margin;
    GLB_PTR(const real) pair_gap;
    // constraint Jacobian rows of every env: real[N][maxefc][16], rewritten each substep (kept out of LDS so that more envs
    // fit on a CU; the working set of the resident envs stays in L2)
    real* rJ_glob;
    real* rB_glob;      // rows of J M^-1, same layout (the Gauss-Seidel sweeps apply force changes through them)
    real* gA_glob;      // intra-group couplings real[N][maxgrp][16] (write once per substep, read by the Gauss-Seidel sweeps)
    int* near_glob;     // Verlet neighbour lists int[N][NEAR_MAX] (rebuilt when a geom moved more than skin / 2)
    int* cand_glob;     // exact broad-phase survivors int[N][CAND_MAX]
    real* gref_glob;    // geom centres at the last rebuild, real[N][ngeom][3]
    real* bxo_glob;     // box-box overflow records real[N][64][BOX_OVF_W]: points 4..7 of a pair's manifold (avsim_collide.hip.h)
    // observation
    GLB_PTR(const int) obs_qposadr;
    GLB_PTR(const real) obs_offset;
    GLB_PTR(const real) obs_scale;
};

struct MOff {
    int nreal, nint;
    int body_parent;
    int body_jntadr;
    int body_jntnum;
    int body_dofadr;
    int body_dofnum;
    int body_tree;
    int body_dofmask;
    int body_last;
    int tree_bodyadr;
    int tree_bodylist;
    int tree_dofadr;
    int tree_dofnum;
    int tree_madr;
    int jnt_type;
    int jnt_qposadr;
    int jnt_dofadr;
    int jnt_actfrclimited;
    int limited_jnt;
    int dof_body;
    int dof_parent;
    int dof_tree;
    int dof_jnt;
    int floss_dof;
    int ment_i;
    int ment_j;
    int act_dof;
    int act_qposadr;
    int act_ctrllimited;
    int geom_type;
    int geom_body;
    int geom_static;
    int body_pos;
    int body_quat;
    int body_mass;
    int body_ipos;
    int body_inertia;
    int body_invweight0;
    int jnt_pos;
    int jnt_axis;
    int jnt_range;
    int jnt_actfrcrange;
    int jnt_margin;
    int dof_armature;
    int dof_damping;
    int dof_frictionloss;
    int dof_invweight0;
    int act_kp;
    int act_kv;
    int act_gear;
    int act_ctrlrange;
    int geom_cpos;
    int geom_rbound;
};

// per-env LDS layout (offsets in reals / ints)
struct Layout {
    int qpos, qvel, ctrl, warm, xpos, xmat, xipos, cdof, gcen, M, L, Minv, bias, fsm, asm_, qacc, fcon, nH, ng, ndl, njv, U, nreal;
    // scratch union U, phase A
    int cinert, cvel, cacc, cfrc, binert;   // binert: the bodies' own spatial inertias (cinert becomes the composites)
    // phase B
    int cdist, cpos, cnrm, rowS, scr;   // scr: narrow-phase scratch (overlays rowS: 64 result slots of 20 words + 9 box work areas)
    // ints
    int cand, cpair, cefc, rmeta, rowI, gI, misc, nprof, nint;
    int maxgrp;
    int maxcon, maxefc;
    int expcon;                   // stride of the contact export arrays (the full capacity, whatever this layout's own)
    int gefc, ggrp;               // rows / groups per env in the global scratch (the full capacities)
    int bytes_per_env;
};

// everything the kernel needs to know about the model: read through a constant-address-space pointer, so that the ~90 table
// pointers and ~100 offsets are scalar-loaded where they are used instead of being held (and spilled) for the whole kernel
template <typename real>
struct KArgs {
    DevModel<real> m;
    Layout lay;
    MOff mo;
};
template <typename real>
using KPtr = const KArgs<real> __attribute__((address_space(4)))*;
// start of a phase: forget what was loaded through ka so far (keeps the live ranges of model scalars inside one phase)
#define PHASE_LAUNDER()                                                                                \
    do {                                                                                               \
        const unsigned long long p_ = (unsigned long long)ka;                                          \
        unsigned lo_ = __builtin_amdgcn_readfirstlane((unsigned)p_), hi_ = __builtin_amdgcn_readfirstlane((unsigned)(p_ >> 32)); \
        asm volatile("" : "+s"(lo_), "+s"(hi_));                                                       \
        ka = (decltype(ka))(((unsigned long long)hi_ << 32) | lo_);                                    \
    } while (0)
// the env's LDS regions through pointers whose provenance (LDS) is visible inside a non-inlined member function: without it
// every access through the members `r` / `ii` is a flat_load / flat_store
#if defined(__HIP_DEVICE_COMPILE__)
#define AVS_ASSUME_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void*)(p)))
#else
#define AVS_ASSUME_LDS(p) ((void)(p))
#endif
#define LDS_BASES()                                                                       \
    real* const r = this->r; AVS_ASSUME_LDS(r);     \
    int* const ii = this->ii; AVS_ASSUME_LDS(ii);   \
    (void)r; (void)ii
#define PHASE_BEGIN() PHASE_LAUNDER(); LDS_BASES()



// Synchronisation between the steps of a phase.  One env is one wavefront, so no barrier between waves is needed; the fence is
// workgroup scope: it drains the wave's outstanding stores (vmcnt / lgkmcnt) before the next step's loads, and the steps hand data
// from lane to lane through the global row scratch as well as through LDS.  A wavefront-scope fence (compiler ordering only; build
// with -DAVS_SYNC_SCOPE='"wavefront"') passes every parity and determinism test and is 1 % faster; the conservative scope is kept.
#ifndef AVS_SYNC_SCOPE
#define AVS_SYNC_SCOPE "workgroup"
#endif
#define GSYNC()                                              \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, AVS_SYNC_SCOPE); \
        __builtin_amdgcn_wave_barrier();                     \
    } while (0)

// all-reduce over the 16 lanes of a DPP row
AVS_DEV float row16_sum(float x) {
    asm volatile("" : "+v"(x));      // (as in oct_sum: every lane of the row gets the same sum to the last bit)
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x128, 0xf, 0xf, true));  // row_ror:8
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x124, 0xf, 0xf, true));  // row_ror:4
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x122, 0xf, 0xf, true));  // row_ror:2
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x121, 0xf, 0xf, true));  // row_ror:1
    return x;
}
AVS_DEV double row16_sum(double x) {
    x += __shfl_xor(x, 8, 16);
    x += __shfl_xor(x, 4, 16);
    x += __shfl_xor(x, 2, 16);
    x += __shfl_xor(x, 1, 16);
    return x;
}

// sum over the 64 lanes of the wave, result broadcast (through an SGPR) to every lane
AVS_DEV float wave_sum(float x) {
    x = row16_sum(x);
    int xi = __builtin_bit_cast(int, x);
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1,3
    xi = __builtin_bit_cast(int, x);
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2,3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
AVS_DEV double wave_sum(double x) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

#define LDS_PTR(T) __attribute__((address_space(3))) T*

// ------------------------------------------------------------------------------------------------
// P8 inner loop, one env per wave (G == 64): projected Gauss-Seidel with the acceleration vector held in
// registers, one dof per lane.  Per row: every lane fetches its J/B entry from the row's two tree windows
// (chain-independent LDS reads), the row residual is a wave-wide DPP sum, the update is a register FMA.
// Separate noinline function so that the sweep loop gets its own register allocation (no spills).
// ------------------------------------------------------------------------------------------------
// Per-row solver record in LDS (8 reals): reference acceleration, regulariser, 1/(diag+R), 1/diag for the noslip
// sweeps (0 = the row does not take part in them), clamp bounds, current force, 1/mu (friction rows of contacts).
template <typename real>
struct RowS {
    real aref, R, inv, invn, lo, hi, f, muinv;
};

// (The argument goes through an empty asm: were it visibly a product a * b, the f32 build -- FMA contraction on -- would turn the
// first step into fma(a, b, neighbour's ROUNDED product), a different number in the two lanes of a pair, and the eight lanes of an octet
// would no longer hold the same sum to the last bit.  noslip_trees takes decisions on such sums in every lane of the octet: with the
// contraction the lanes of an octet disagreed about a sliding contact that sits on its cone, |w|^2 - r^2 ~ 0.)
AVS_DEV float oct_sum(float x) {   // sum over each aligned group of 8 lanes
    asm volatile("" : "+v"(x));
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));  // row_half_mirror
    return x;
}
AVS_DEV double oct_sum(double x) {
    x += __shfl_xor(x, 1, 8);
    x += __shfl_xor(x, 2, 8);
    x += __shfl_xor(x, 4, 8);
    return x;
}
// value of lane J (0..7) of every aligned group of 8 lanes, to all 8 lanes of the group: a quad broadcast, then the other quad's
// copy through the half-row mirror -- two DPP moves and a select instead of a ds_bpermute round trip through the LDS crossbar
template <int J>
AVS_DEV int oct_bcast_i(int x) {
    constexpr int q = J & 3, ctrl = q | (q << 2) | (q << 4) | (q << 6);
    const int a = __builtin_amdgcn_mov_dpp(x, ctrl, 0xf, 0xf, true);      // quad_perm:[q,q,q,q] (every lane has a source: the destination needs no initial value)
    const int b = __builtin_amdgcn_mov_dpp(a, 0x141, 0xf, 0xf, true);     // row_half_mirror: lane i <- lane 7 - i of its 8
    const bool hi = (threadIdx.x & 4) != 0;
    return hi == (J >= 4) ? a : b;
}
template <int J> AVS_DEV float oct_bcast(float x) { return __builtin_bit_cast(float, oct_bcast_i<J>(__builtin_bit_cast(int, x))); }
template <int J> AVS_DEV double oct_bcast(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const unsigned lo = (unsigned)oct_bcast_i<J>((int)(unsigned)b), hi = (unsigned)oct_bcast_i<J>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
// the same with the source lane in a loop counter of an unrolled loop
template <typename T>
AVS_DEV T oct_bcast_n(T x, int j) {
    switch (j & 7) {
        case 0: return oct_bcast<0>(x); case 1: return oct_bcast<1>(x); case 2: return oct_bcast<2>(x); case 3: return oct_bcast<3>(x);
        case 4: return oct_bcast<4>(x); case 5: return oct_bcast<5>(x); case 6: return oct_bcast<6>(x); default: return oct_bcast<7>(x);
    }
}
AVS_DEV float lane_get(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }
AVS_DEV double lane_get(double x, int l) {   // l is wave-uniform at every call site
    const long long b = __builtin_bit_cast(long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}

constexpr int GRP_MAX = 6;   // rows per Gauss-Seidel group (a condim-6 contact is one group)
// per-group record in global scratch: [0, 15) couplings (packed lower triangle), then for a contact with >= 3 friction rows the
// inverse of its friction block for the noslip pass, transposed (entry k of row r at word 16 + 8 k + r, so that row r's lane fetches
// six separate words), and the rows' singular flags at words 56 ..
constexpr int GA_W = 64, GA_Q = 16, GA_QW = 6;   // group record: 15 couplings, then the noslip block transposed: word 8 k + r = entry k of row r
// convergence thresholds of the multiplier iteration in the noslip QCQP: MuJoCo's absolute 1e-10 in double; in float a relative
// part on top, since v.v - r^2 and the multiplier carry 1e-7 relative rounding
// t / d with 1 / d at hand.  The product kernel (float) multiplies: a division there is ten instructions (range scaling around
// v_rcp_f32), and the multiplier iteration of the noslip QCQP is thirty divisions per step.  The parity kernel (double) divides, as
// the oracle does.
template <typename T> AVS_DEV T qdiv(T t, T d, T dinv) { return sizeof(T) == 4 ? t * dinv : t / d; }
AVS_DEV float qrsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
AVS_DEV double qrsqrt(double x) { return 1.0 / sqrt(x); }
template <typename T> struct QTol;
template <> struct QTol<double> { static constexpr double abs = 1e-10, rel = 0.0; };
template <> struct QTol<float> { static constexpr float abs = 1e-10f, rel = 2e-6f; };

// ------------------------------------------------------------------------------------------------
// P8 inner loop, one env per wavefront: projected Gauss-Seidel over GROUPS of up to 6 consecutive rows.
// A group's rows are relaxed in the reference order, but their mutual coupling A_rs = J_r M^-1 J_s^T is
// precomputed (gA, lower triangle), so the sweep is mathematically identical to row-by-row GS while
//   - the <=6 row residuals J_r.qacc are formed together: each 16-lane DPP row handles one matrix row
//     (its two 8-dof tree windows), 4 rows per pass;
//   - the sequential part runs on lanes 0..5 (lane r owns row r: force, bounds, couplings) with one
//     v_readlane + one FMA per step;
//   - the acceleration update is a returnless LDS atomic add per (row, dof), again 4 rows per pass.
// A contact is exactly one group, so the elliptic-cone projection of its friction block is local to the group.
// Separate noinline function: the sweep loop gets its own register allocation.
// rowI[i] = two 13-bit dof windows of row i: first dof (6) | count (4) | kinematic tree (3).
// B = J M^-1 is read from the global row scratch next to J (written by the row's lane in make_constraints).
// ------------------------------------------------------------------------------------------------
template <bool B> struct BoolTag { static constexpr bool value = B; };

// The dry-friction rows of a noslip sweep, all kinematic trees at once (see pgs_groups)
template <typename real>
struct NoslipLead {
    LDS_PTR(const real) Minv;        // 8 x 8 inverse inertia block per tree
    LDS_PTR(const int) tadr;         // tree -> first dof, dof count
    LDS_PTR(const int) tnum;
    LDS_PTR(const int) floss_dof;    // dry-friction row -> dof (increasing)
    LDS_PTR(int) dmap;               // nv ints of dead LDS: dof -> row
    LDS_PTR(int) prof;               // probe slots (8 ints) or null
    int tridiag;                     // the multiplier iteration of a sliding contact's QCQP on the tridiagonal form (option "qcqp_tridiag")
    int ntree, nv, neq, nfloss, nlg; // nlg = groups of leading (non-contact) rows; -1: no per-tree pass (more than 8 trees)
};

// A struct argument of an out-of-line function is passed in memory (12 dwords per lane through the wave's private segment, per call): the
// noslip functions take it as three 4-vectors, which travel in registers.
typedef int nl_v4 __attribute__((ext_vector_type(4)));
#define NL_PARAMS nl_v4 nl_a_, nl_v4 nl_b_, nl_v4 nl_c_
#define NL_ARGS(nl) nl_pack_a(nl), nl_pack_b(nl), nl_pack_c(nl)
#define NL_UNPACK(real) NoslipLead<real> nl = nl_unpack<real>(nl_a_, nl_b_, nl_c_)
template <typename real> AVS_DEV nl_v4 nl_pack_a(const NoslipLead<real>& n) {
    return nl_v4{(int)(__SIZE_TYPE__)n.Minv, (int)(__SIZE_TYPE__)n.tadr, (int)(__SIZE_TYPE__)n.tnum, (int)(__SIZE_TYPE__)n.floss_dof};
}
template <typename real> AVS_DEV nl_v4 nl_pack_b(const NoslipLead<real>& n) { return nl_v4{(int)(__SIZE_TYPE__)n.dmap, (int)(__SIZE_TYPE__)n.prof, n.tridiag, n.ntree}; }
template <typename real> AVS_DEV nl_v4 nl_pack_c(const NoslipLead<real>& n) { return nl_v4{n.nv, n.neq, n.nfloss, n.nlg}; }
template <typename real> AVS_DEV NoslipLead<real> nl_unpack(nl_v4 a, nl_v4 b, nl_v4 c) {
    NoslipLead<real> n;
    n.Minv = (LDS_PTR(const real))(__SIZE_TYPE__)(unsigned)a.x; n.tadr = (LDS_PTR(const int))(__SIZE_TYPE__)(unsigned)a.y; n.tnum = (LDS_PTR(const int))(__SIZE_TYPE__)(unsigned)a.z;
    n.floss_dof = (LDS_PTR(const int))(__SIZE_TYPE__)(unsigned)a.w; n.dmap = (LDS_PTR(int))(__SIZE_TYPE__)(unsigned)b.x; n.prof = (LDS_PTR(int))(__SIZE_TYPE__)(unsigned)b.y;
    n.tridiag = b.z; n.ntree = b.w; n.nv = c.x; n.neq = c.y; n.nfloss = c.z; n.nlg = c.w;
    return n;
}

// arguments of a non-kernel function arrive in VGPRs: the wave-uniform ones go back to SGPRs (scalar branches, scalar addressing)
template <typename T> AVS_DEV LDS_PTR(T) uni_lds(LDS_PTR(T) p) {
    return (LDS_PTR(T))(unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)p);
}
template <typename T> AVS_DEV GLB_PTR(T) uni_glb(GLB_PTR(T) p) {
    const unsigned long long v = (unsigned long long)p;
    return (GLB_PTR(T))(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v));
}

// Step of the multiplier iteration of a sliding contact's QCQP, |y(la)| = r with y = -(A + la I)^-1 b.  MuJoCo's mju_QCQP takes
// Newton steps on |y|^2 - r^2 from la = 0 (val / (2 y.w), w = (A + la I)^-1 y); that function is strongly convex in la and the
// iteration needs about nine steps on a dragging arm.  1 / |y(la)| is nearly linear in la (the trust-region secular equation, More
// and Sorensen 1983), so Newton on 1 / r - 1 / |y| reaches the SAME root in two to four steps, also monotonically from the left:
// delta = (|y| - r) / r * |y|^2 / (y.w), with |y| - r taken as val / (|y| + r).  Product mode (f32, option qcqp_tridiag = 2); the
// f64 parity kernel keeps MuJoCo's steps.
template <typename T> AVS_DEV T secular_step(T val, T r2, T r, T yw) {
    const T yy = val + r2, ny = sqrt(yy);
    return val * yy / ((ny + r) * r * yw);
}

template <typename real>
__device__ AVS_OUTLINE_2 void pgs_groups(LDS_PTR(real) rowS, LDS_PTR(const int) rowI, GLB_PTR(const real) rJ, GLB_PTR(const real) rB,
                                                     LDS_PTR(real) q, LDS_PTR(const int) gI, GLB_PTR(const real) gA, int ngrp, int iters,
                                                     int noslip_iters, real noslip_tol_scaled, NL_PARAMS) {
    NL_UNPACK(real);
    const long long tent = __builtin_readcyclecounter();
    rowS = uni_lds(rowS); rowI = uni_lds(rowI); q = uni_lds(q); gI = uni_lds(gI);
    rJ = uni_glb(rJ); rB = uni_glb(rB); gA = uni_glb(gA);
    ngrp = __builtin_amdgcn_readfirstlane(ngrp); iters = __builtin_amdgcn_readfirstlane(iters); noslip_iters = __builtin_amdgcn_readfirstlane(noslip_iters);
    noslip_tol_scaled = lane_get(noslip_tol_scaled, 0);
    nl.Minv = uni_lds(nl.Minv); nl.tadr = uni_lds(nl.tadr); nl.tnum = uni_lds(nl.tnum); nl.floss_dof = uni_lds(nl.floss_dof); nl.dmap = uni_lds(nl.dmap); nl.prof = uni_lds(nl.prof);
    nl.ntree = __builtin_amdgcn_readfirstlane(nl.ntree); nl.nv = __builtin_amdgcn_readfirstlane(nl.nv); nl.neq = __builtin_amdgcn_readfirstlane(nl.neq);
    nl.nfloss = __builtin_amdgcn_readfirstlane(nl.nfloss); nl.nlg = __builtin_amdgcn_readfirstlane(nl.nlg); nl.tridiag = __builtin_amdgcn_readfirstlane(nl.tridiag);
    // lane 8 d + k serves row d of the group (d < 6) and dof slot k of BOTH of the row's tree windows
    const int lane = threadIdx.x & 63, d = lane >> 3, k8 = lane & 7, dr = d < GRP_MAX ? d : 0;
    const int lr = lane < GRP_MAX ? lane : 0;          // row slot owned in the sequential phase
    const int tri = lr * (lr - 1) / 2;                 // offset of that row in the packed lower triangle
    if (ngrp <= 0) return;
    // ---- dry-friction rows in the noslip sweeps ----
    // A dry-friction row is a unit row (J = e_dof), so its residual is qacc[dof] - aref, its J M^-1 is a row of the tree's inverse
    // inertia block, and rows of different trees do not interact: instead of walking them six at a time through the group
    // machinery, lane 8 t + i takes dof i of tree t and every tree relaxes its rows in row order at the same time (eight
    // steps for all of them; x follows qacc[dof] through the updates of its tree, which is all a later row's residual sees).
    // Contact blocks couple trees and come after all dry-friction rows in mj_solNoSlip's sweep [EXT], so the order of the
    // updates that can see each other is the reference's.
    const bool fl = noslip_iters > 0 && nl.nlg >= 0;
    const int ft = lane >> 3, fi = lane & 7, fgb = lane & ~7;
    bool fm = false;
    int fdof = 0, frow = -1;
    real mrow[TREE_W], faref = 0, finv = 0, fdiag = 0, flo = 0, fhi = 0;
#pragma unroll
    for (int s = 0; s < TREE_W; s++) mrow[s] = 0;
    if (fl) {
        for (int k = lane; k < nl.nv; k += 64) nl.dmap[k] = -1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int r = lane; r < nl.nfloss; r += 64) nl.dmap[nl.floss_dof[r]] = nl.neq + r;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        fm = ft < nl.ntree && fi < nl.tnum[ft < nl.ntree ? ft : 0];
        fdof = fm ? nl.tadr[ft] + fi : 0;
        frow = fm ? nl.dmap[fdof] : -1;
        if (fm) {
#pragma unroll
            for (int s = 0; s < TREE_W; s++) mrow[s] = nl.Minv[64 * ft + 8 * fi + s];
        }
        LDS_PTR(const real) FS = rowS + RS_S * (frow >= 0 ? frow : 0);
        faref = FS[0]; finv = frow >= 0 ? FS[3] : real(0); flo = FS[4]; fhi = FS[5];
        fdiag = finv != 0 ? real(1) / finv : real(0);
    }
    real imp = 0, imp_c = 0;      // improvement of the dual cost over the current noslip sweep (per-lane parts, uniform part)
#ifdef AVSIM_PROBE_ROWS
    LDS_PTR(int) prof = (LDS_PTR(int))nullptr;
#else
    LDS_PTR(int) prof = nl.prof;
#endif
    const long long tq0 = prof ? __builtin_readcyclecounter() : 0;
    int nstep = 0, nsweep = 0;
    auto floss_sweep = [&]() {
        const long long tf0 = prof ? __builtin_readcyclecounter() : 0;
        real x = fm ? q[fdof] : real(0);
        LDS_PTR(real) FS = rowS + RS_S * (frow >= 0 ? frow : 0);
        real f = frow >= 0 ? FS[6] : real(0);
#pragma unroll
        for (int s = 0; s < TREE_W; s++) {
            // dry-friction rows of mj_solNoSlip [EXT]: clamped scalar update, undone when it would raise the cost (costChange)
            const real res = x - faref;
            const real fs = tmin(tmax(f - res * finv, flo), fhi);
            const real dl_ = fs - f, ch = dl_ * (real(0.5) * dl_ * fdiag + res);      // (lanes without a row never take the update)
            const bool take = frow >= 0 && fi == s && !(ch > real(1e-10));
            if (take) { imp -= ch; f = fs; }
            const real ds = oct_bcast_n(take ? dl_ : real(0), s);
            x += mrow[s] * ds;
        }
        if (fm) q[fdof] = x;
        if (frow >= 0) FS[6] = f;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        nsweep++;
    };
    const int first_ns = fl ? nl.nlg : 0;               // first group of a noslip sweep
    const int per_ns = ngrp - first_ns;                 // (contact groups only when the dry-friction rows go per tree)
    const int noslip_only = (fl && per_ns <= 0) ? noslip_iters : 0;
    if (noslip_only) noslip_iters = 0;   // no contact: the noslip sweeps are the dry-friction rows alone, after the PGS sweeps (if any)
    // software pipeline: group headers three groups ahead, row windows two ahead, the J / J M^-1 entries (global memory, L2
    // latency) one group ahead, so that a step only waits for LDS
    auto jb_load = [&](int hdr, int win, real& ja, real& jb, real& ba, real& bb) {
        const int st = hdr & 0xffff, cn = (hdr >> 16) & 15;
        const bool in = d < cn;
        const int rw = st + (in ? d : 0);
        const bool a = in && k8 < ((win >> 6) & 15), b = in && k8 < ((win >> 19) & 15);
        GLB_PTR(const real) Jr = rJ + ROW_S * rw + k8;
        GLB_PTR(const real) Br = rB + ROW_S * rw + k8;
        // unconditional loads (every row record is 16 words, zero padded; a lane outside the group reads the group's first row): no
        // exec-mask detours, and nothing touches the values before the step that uses them, which masks them (inA / inB)
        (void)a; (void)b;
        // a contact between one kinematic tree and the world keeps J M^-1 in the idle second window of its J record (make_constraints):
        // one 64-byte line per row instead of two, and the J M^-1 buffer is not touched at all by such contacts (group-uniform branch)
        const bool packed = ((hdr >> 24) & 1) && ((__builtin_amdgcn_readfirstlane(win) >> 19) & 15) == 0;
        ja = Jr[0];
        const real x = Jr[TREE_W];
        if (packed) { jb = 0; ba = x; bb = 0; }
        else { jb = x; ba = Br[0]; bb = Br[TREE_W]; }
    };
    // wrap-around of the group index at the end of a sweep: a noslip sweep starts at first_ns
    auto nextg = [&](int gq, int itq) { return gq + 1 < ngrp ? gq + 1 : (itq + 1 >= iters ? first_ns : 0); };
    int g = iters > 0 ? 0 : first_ns, it = 0;
    if (g >= ngrp) g = 0;      // (no step will run)
    int gi = __builtin_amdgcn_readfirstlane(gI[g]);
    int gin = __builtin_amdgcn_readfirstlane(gI[nextg(g, it)]);
    int ginn = __builtin_amdgcn_readfirstlane(gI[nextg(nextg(g, it), it)]);
    int ra = rowI[(gi & 0xffff) + dr], ran = rowI[(gin & 0xffff) + dr];
    real JA, JB, BA, BB;
    jb_load(gi, ra, JA, JB, BA, BB);
    real ac[GRP_MAX - 1];
#pragma unroll
    for (int s = 0; s < GRP_MAX - 1; s++) ac[s] = gA[GA_W * g + tri + s];
    // noslip QCQP rows of the group (lanes 1..5 = friction rows), prefetched with the couplings
    const int qr = (lane >= 1 && lane <= 5) ? lane - 1 : 0;
    real qc[GA_QW], qn[GA_QW];
#pragma unroll
    for (int s = 0; s < GA_QW; s++) qc[s] = gA[GA_W * g + GA_Q + 8 * s + qr];
    const int total = iters * ngrp + noslip_iters * per_ns;
    // the first group's look-ahead loads land here: with one of them still pending at the loop entry, the compiler's wait for it
    // inside the loop body (in-order counter, sized for the first pass) would also drain the loads every later step has just issued
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
    if (fl && iters == 0 && noslip_iters > 0) floss_sweep();
    for (int step = 0; step < total; step++) {
        nstep++;
        const bool noslip = it >= iters;
        const int start = gi & 0xffff, cnt = (gi >> 16) & 15;
        const bool contact = (gi >> 24) & 1;
        const int g1 = nextg(g, it), g2 = nextg(g1, it), g3 = nextg(g2, it);
        // ---- issue the look-ahead reads and this group's LDS reads in one batch ----
        const int ginnn_v = gI[g3];
        const int rann = rowI[(ginn & 0xffff) + dr];
        real JAn, JBn, BAn, BBn;
        jb_load(gin, ran, JAn, JBn, BAn, BBn);
        const bool inr = d < cnt;
        const int adrA = (ra & 63) + k8, adrB = ((ra >> 13) & 63) + k8;
        const bool inA = inr && k8 < ((ra >> 6) & 15), inB = inr && k8 < ((ra >> 19) & 15);
        const real qA = q[inA ? adrA : 0], qB = q[inB ? adrB : 0];
        const bool mine = lane < cnt;
        LDS_PTR(real) S = rowS + RS_S * (start + (mine ? lane : 0));
        const real aref = S[0], R = S[1], inv2 = S[2], inv3 = S[3], lo = S[4], hi = S[5], f0 = S[6], muinv = S[7];
        real a[GRP_MAX - 1], an[GRP_MAX - 1];
#pragma unroll
        for (int s = 0; s < GRP_MAX - 1; s++) an[s] = 0;
        // next group's couplings (global memory): the PGS sweeps and the leading rows' groups use them in every step, a noslip step on
        // a contact only when the contact slides (the multiplier iteration below then fetches them itself)
        const bool next_noslip = (g + 1 >= ngrp ? it + 1 : it) >= iters;
        if (!(next_noslip && ((gin >> 24) & 1))) {
#pragma unroll
            for (int s = 0; s < GRP_MAX - 1; s++) an[s] = gA[GA_W * g1 + tri + s];
        }
        if (noslip_iters > 0) {
#pragma unroll
            for (int s = 0; s < GA_QW; s++) qn[s] = gA[GA_W * g1 + GA_Q + 8 * s + qr];
        }
#pragma unroll
        for (int s = 0; s < GRP_MAX - 1; s++) a[s] = (mine && s < lane) ? ac[s] : real(0);
        // ---- row residuals J_r . qacc: every row is summed over its 8 lanes, lane r then fetches row r's sum ----
        const real x = oct_sum((inA ? JA * qA : real(0)) + (inB ? JB * qB : real(0)));
        const real dot = __shfl(x, 8 * lr, 64);
        real f = f0;
        if (noslip && contact) {
            // ---- mj_solNoSlip [EXT], friction block of an elliptic contact: the exact minimiser of 1/2 x'Ax + x'b over the cone
            // section sum (x_j / mu_j)^2 <= fn^2 (mju_QCQP2 / mju_QCQP), fn = the normal force held fixed.  Lane r owns row r
            // (row 0 = normal); the block (A lower triangle from the couplings, diagonal 1 / invn), b and mu are gathered from
            // the owner lanes and the small Newton iteration on the multiplier runs redundantly on every lane ----
            const int n = cnt - 1;
            bool done = false;
            if (n >= 3) {
                // multiplier 0 first: the unconstrained minimiser f - A^-1 res, through the inverse of the friction block made with the
                // rows (make_constraints: lane r = 1..n holds row r - 1 of A^-1).  Inside the cone section this is mju_QCQP's answer,
                // and its cost change is dl . (A dl / 2 + res) = dl . res / 2 because A dl = -res.
                const real fn = lane_get(f0, 0), r2 = fn * fn;
                const bool row = lane >= 1 && lane <= n;
                const real res_r = row ? dot - aref : real(0);
                if (!(fn < real(1e-15)) && lane_get(qc[5], 1) == real(0)) {
                    real t = 0;
#pragma unroll
                    for (int k = 0; k < 5; k++) t += qc[k] * lane_get(res_r, k + 1);
                    const real dl_ = row ? -t : real(0), vr = f0 + dl_;
                    const real w = row ? vr * muinv : real(0);
                    const real val = lane_get(oct_sum(w * w), 0) - r2;
                    if (val < QTol<real>::abs + QTol<real>::rel * r2) {
                        const real change = real(0.5) * lane_get(oct_sum(dl_ * res_r), 0);
                        if (!(change > real(1e-10))) {
                            imp_c -= change;
                            if (row) f = vr;
                        }
                        done = true;
                    }
                }
            }
            if (n >= 1 && !done) {
                const long long tsl0 = prof ? __builtin_readcyclecounter() : 0; int nit_q = 0; long long tsl1 = 0;
                const real fn = lane_get(f0, 0);
                real acs[GRP_MAX - 1];      // this group's couplings, fetched here: the sliding case pays the memory round trip, not every step
#pragma unroll
                for (int s = 0; s < GRP_MAX - 1; s++) acs[s] = gA[GA_W * g + tri + s];
                real Aq[5][5], bq[5], dq[5], oldf[5], resq[5], v[5];
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const bool in = j < n;
                    resq[j] = in ? lane_get(dot - aref, j + 1) : real(0);
                    oldf[j] = in ? lane_get(f0, j + 1) : real(0);
                    const real mi = lane_get(muinv, j + 1);
                    dq[j] = in ? real(1) / mi : real(1);
                    const real di = lane_get(inv3, j + 1);
                    Aq[j][j] = in ? real(1) / di : real(1);
#pragma unroll
                    for (int k = 0; k < j; k++) { const real c = in ? lane_get(acs[k + 1], j + 1) : real(0); Aq[j][k] = c; Aq[k][j] = c; }
                }
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    real t = resq[j];
#pragma unroll
                    for (int k = 0; k < 5; k++) t -= (j < n && k < n) ? Aq[j][k] * oldf[k] : real(0);
                    bq[j] = t;
                    v[j] = 0;
                }
                if (prof) { asm volatile("" :: "v"(bq[0]), "v"(bq[1]), "v"(bq[2]), "v"(bq[3]), "v"(bq[4])); tsl1 = __builtin_readcyclecounter(); }
                if (!(fn < real(1e-15))) {
                    const real r2 = fn * fn, vtol = QTol<real>::abs + QTol<real>::rel * r2;
                    real la = 0;
                    bool active;
                    if (n == 2) {
                        const real b1 = bq[0] * dq[0], b2 = bq[1] * dq[1];
                        const real A11 = Aq[0][0] * dq[0] * dq[0], A22 = Aq[1][1] * dq[1] * dq[1], A12 = Aq[1][0] * dq[0] * dq[1];
                        real v1 = 0, v2 = 0;
                        bool singular = false;
                        for (int iter = 0; iter < 20; iter++) {
                            const real det = (A11 + la) * (A22 + la) - A12 * A12;
                            if (det < real(1e-10)) { singular = true; break; }
                            const real detinv = real(1) / det, P11 = (A22 + la) * detinv, P22 = (A11 + la) * detinv, P12 = -A12 * detinv;
                            v1 = -P11 * b1 - P12 * b2;
                            v2 = -P12 * b1 - P22 * b2;
                            const real val = v1 * v1 + v2 * v2 - r2;
                            if (val < vtol) break;
                            const real yw = P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2;
                            const real delta = nl.tridiag == 2 ? secular_step(val, r2, fn, yw) : val / (2 * yw);
                            if (delta < QTol<real>::abs + QTol<real>::rel * la) break;
                            la += delta;
                        }
                        v[0] = singular ? real(0) : v1 * dq[0];
                        v[1] = singular ? real(0) : v2 * dq[1];
                        active = !singular && la != 0;
                    } else {
                        real y[5];
                        bool singular = false;
                        if (nl.tridiag) {
                            // mju_QCQP's Newton iteration on the multiplier needs y = -(As + la I)^-1 bs and (As + la I)^-1 y for a dozen
                            // values of la.  MuJoCo factors As + la I afresh every time (a 5 x 5 Cholesky: a chain of ~150 dependent
                            // instructions, 1.2 k cycles, redundantly on every lane).  Here As is reduced once to tridiagonal form by
                            // three Householder reflections, As = H T H^T; (As + la I)^-1 = H (T + la I)^-1 H^T, |y| and y . w are the
                            // same in the rotated basis, and T + la I is factored by the four-step LDL^T recurrence: the same iterates
                            // up to rounding, a quarter of the instructions per iterate.  Singularity is MuJoCo's pivot rule at la = 0
                            // (the flag make_constraints left with the block's inverse; the pivots only grow with la).
                            real As[5][5], cs[5], w[5], hv[3][5], hb[3];
#pragma unroll
                            for (int j = 0; j < 5; j++) {
                                cs[j] = j < n ? bq[j] * dq[j] : real(0);
                                y[j] = 0;
#pragma unroll
                                for (int k = 0; k < 5; k++) As[j][k] = (j < n && k < n) ? Aq[j][k] * dq[j] * dq[k] : (j == k ? real(1) : real(0));
                            }
                            singular = lane_get(qc[5], 1) != real(0);
                            if (!singular) {
#pragma unroll
                                for (int k = 0; k < 3; k++) {
                                    real sigma = 0;
#pragma unroll
                                    for (int i = k + 2; i < 5; i++) sigma += As[i][k] * As[i][k];
                                    const real x0 = As[k + 1][k];
#pragma unroll
                                    for (int i = 0; i < 5; i++) hv[k][i] = 0;
                                    hb[k] = 0;
                                    if (sigma != real(0)) {       // (wave-uniform: the column is tridiagonal already otherwise)
                                        const real nrm = sqrt(x0 * x0 + sigma), alpha = x0 > 0 ? -nrm : nrm;
                                        hv[k][k + 1] = x0 - alpha;
#pragma unroll
                                        for (int i = k + 2; i < 5; i++) hv[k][i] = As[i][k];
                                        const real beta = real(2) / (hv[k][k + 1] * hv[k][k + 1] + sigma);
                                        hb[k] = beta;
                                        real pv[5], vp = 0;
#pragma unroll
                                        for (int i = k + 1; i < 5; i++) {
                                            real t = 0;
#pragma unroll
                                            for (int j = k + 1; j < 5; j++) t += As[i][j] * hv[k][j];
                                            pv[i] = beta * t;
                                            vp += hv[k][i] * pv[i];
                                        }
                                        const real K = real(0.5) * beta * vp;
#pragma unroll
                                        for (int i = k + 1; i < 5; i++) pv[i] -= K * hv[k][i];
#pragma unroll
                                        for (int i = k + 1; i < 5; i++)
#pragma unroll
                                            for (int j = k + 1; j <= i; j++) { As[i][j] -= hv[k][i] * pv[j] + pv[i] * hv[k][j]; As[j][i] = As[i][j]; }
                                        As[k + 1][k] = alpha; As[k][k + 1] = alpha;
#pragma unroll
                                        for (int i = k + 2; i < 5; i++) { As[i][k] = 0; As[k][i] = 0; }
                                        // right-hand side into the rotated basis
                                        real t = 0;
#pragma unroll
                                        for (int i = k + 1; i < 5; i++) t += hv[k][i] * cs[i];
                                        t *= beta;
#pragma unroll
                                        for (int i = k + 1; i < 5; i++) cs[i] -= t * hv[k][i];
                                    }
                                }
                                const real b0 = As[1][0], b1 = As[2][1], b2 = As[3][2], b3 = As[4][3];
                                for (int iter = 0; iter < 20; iter++) {
                                    nit_q++;
                                    // LDL^T of T + la I (la on the contact's own dimensions only, as in mju_QCQP's padded loops)
                                    const real d0 = As[0][0] + la, r0 = real(1) / d0, l0 = b0 * r0;
                                    const real d1 = As[1][1] + la - l0 * b0, r1 = real(1) / d1, l1 = b1 * r1;
                                    const real d2 = As[2][2] + la - l1 * b1, r2_ = real(1) / d2, l2 = b2 * r2_;
                                    const real d3 = As[3][3] + (3 < n ? la : real(0)) - l2 * b2, r3 = real(1) / d3, l3 = b3 * r3;
                                    const real d4 = As[4][4] + (4 < n ? la : real(0)) - l3 * b3, r4 = real(1) / d4;
                                    real z0 = -cs[0], z1 = -cs[1] - l0 * z0, z2 = -cs[2] - l1 * z1, z3 = -cs[3] - l2 * z2, z4 = -cs[4] - l3 * z3;
                                    y[4] = z4 * r4; y[3] = z3 * r3 - l3 * y[4]; y[2] = z2 * r2_ - l2 * y[3]; y[1] = z1 * r1 - l1 * y[2]; y[0] = z0 * r0 - l0 * y[1];
                                    real val = -r2;
#pragma unroll
                                    for (int i = 0; i < 5; i++) val += y[i] * y[i];
                                    if (val < vtol) break;
                                    z0 = y[0]; z1 = y[1] - l0 * z0; z2 = y[2] - l1 * z1; z3 = y[3] - l2 * z2; z4 = y[4] - l3 * z3;
                                    w[4] = z4 * r4; w[3] = z3 * r3 - l3 * w[4]; w[2] = z2 * r2_ - l2 * w[3]; w[1] = z1 * r1 - l1 * w[2]; w[0] = z0 * r0 - l0 * w[1];
                                    real yw = 0;
#pragma unroll
                                    for (int i = 0; i < 5; i++) yw += y[i] * w[i];
                                    const real delta = nl.tridiag == 2 ? secular_step(val, r2, fn, yw) : val / (2 * yw);
                                    if (delta < QTol<real>::abs + QTol<real>::rel * la) break;
                                    la += delta;
                                }
                                // back to the contact's basis: y <- H y
#pragma unroll
                                for (int k = 2; k >= 0; k--) {
                                    real t = 0;
#pragma unroll
                                    for (int i = k + 1; i < 5; i++) t += hv[k][i] * y[i];
                                    t *= hb[k];
#pragma unroll
                                    for (int i = k + 1; i < 5; i++) y[i] -= t * hv[k][i];
                                }
                            }
                        } else {
                            // MuJoCo's own evaluation (the f64 parity mode's default: the oracle's arithmetic, operation for operation)
                            real As[5][5], bs[5], L[5][5], w[5], rd[5];
#pragma unroll
                            for (int j = 0; j < 5; j++) {
                                bs[j] = j < n ? bq[j] * dq[j] : real(0);
                                y[j] = 0;
#pragma unroll
                                for (int k = 0; k < 5; k++) As[j][k] = (j < n && k < n) ? Aq[j][k] * dq[j] * dq[k] : (j == k ? real(1) : real(0));
                            }
                            for (int iter = 0; iter < 20; iter++) {
                                nit_q++;
#pragma unroll
                                for (int j = 0; j < 5; j++) {
                                    real dd = As[j][j] + (j < n ? la : real(0));
#pragma unroll
                                    for (int k = 0; k < j; k++) dd -= L[j][k] * L[j][k];
                                    if (j < n && dd < real(1e-10)) singular = true;
                                    dd = tmax(dd, real(1e-30));
                                    if (sizeof(real) == 4) { rd[j] = qrsqrt(dd); dd = dd * rd[j]; }
                                    else { dd = sqrt(dd); rd[j] = real(1) / dd; }
                                    L[j][j] = dd;
#pragma unroll
                                    for (int i = j + 1; i < 5; i++) {
                                        real t = As[i][j];
#pragma unroll
                                        for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
                                        L[i][j] = qdiv(t, dd, rd[j]);
                                    }
                                }
                                if (singular) break;
#pragma unroll
                                for (int i = 0; i < 5; i++) { real t = -bs[i]; for (int k = 0; k < i; k++) t -= L[i][k] * y[k]; y[i] = qdiv(t, L[i][i], rd[i]); }
#pragma unroll
                                for (int i = 4; i >= 0; i--) { real t = y[i]; for (int k = i + 1; k < 5; k++) t -= L[k][i] * y[k]; y[i] = qdiv(t, L[i][i], rd[i]); }
                                real val = -r2;
#pragma unroll
                                for (int i = 0; i < 5; i++) val += y[i] * y[i];
                                if (val < vtol) break;
#pragma unroll
                                for (int i = 0; i < 5; i++) { real t = y[i]; for (int k = 0; k < i; k++) t -= L[i][k] * w[k]; w[i] = qdiv(t, L[i][i], rd[i]); }
#pragma unroll
                                for (int i = 4; i >= 0; i--) { real t = w[i]; for (int k = i + 1; k < 5; k++) t -= L[k][i] * w[k]; w[i] = qdiv(t, L[i][i], rd[i]); }
                                real deriv = 0;
#pragma unroll
                                for (int i = 0; i < 5; i++) deriv += y[i] * w[i];
                                deriv *= -2;
                                const real delta = -val / deriv;
                                if (delta < QTol<real>::abs + QTol<real>::rel * la) break;
                                la += delta;
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 5; j++) v[j] = (singular || !(j < n)) ? real(0) : y[j] * dq[j];
                        active = !singular && la != 0;
                    }
                    if (active) {       // exactly onto the ellipsoid
                        real sq = 0;
#pragma unroll
                        for (int j = 0; j < 5; j++) sq += j < n ? v[j] * v[j] / (dq[j] * dq[j]) : real(0);
                        const real sc = sqrt(r2 / tmax(real(1e-15), sq));
#pragma unroll
                        for (int j = 0; j < 5; j++) v[j] *= sc;
                    }
                }
                // change of the cost [EXT: costChange]; an update that would raise it by more than 1e-10 is not made
                real change = 0;
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    real t = 0;
#pragma unroll
                    for (int k = 0; k < 5; k++) t += (j < n && k < n) ? Aq[j][k] * (v[k] - oldf[k]) : real(0);
                    change += j < n ? (v[j] - oldf[j]) * (real(0.5) * t + resq[j]) : real(0);
                }
                if (!(change > real(1e-10))) {
                    imp_c -= change;
#pragma unroll
                    for (int j = 0; j < 5; j++) if (lane == j + 1 && j < n) f = v[j];
                }
                if (prof && lane == 0) { prof[2] += (int)(__builtin_readcyclecounter() - tsl0); prof[3] += 1; prof[4] += nit_q; prof[1] += (int)(tsl1 - tsl0); }
            }
        } else {
        // ---- sequential relaxation on lanes 0..cnt-1 (lane r = row start + r) ----
        const real inv = noslip ? inv3 : inv2;
        real res = dot - aref + (noslip ? real(0) : R * f0);
#pragma unroll
        for (int s = 0; s < GRP_MAX; s++) {
            real fs = tmin(tmax(f0 - res * inv, lo), hi);
            if (noslip) {
                // dry-friction rows of mj_solNoSlip [EXT]: clamped scalar update, undone when it would raise the cost (costChange)
                const real dl_ = fs - f0, ch = inv != 0 ? dl_ * (real(0.5) * dl_ / inv + res) : real(0);
                if (ch > real(1e-10)) fs = f0;
                else if (lane == s && s < cnt) imp -= ch;
            }
            const real ds = s < cnt ? lane_get(fs - f0, s) : real(0);
            if (lane == s && s < cnt) f = fs;
            if (s < GRP_MAX - 1) res += a[s < GRP_MAX - 1 ? s : 0] * ds;
        }
        // ---- elliptic cone: scale the friction block back when sliding (PGS sweeps) ----
        if (contact && cnt > 1) {
            const real fn = lane_get(f, 0);
            const real t = (mine && lane >= 1) ? f * muinv : real(0);
            const real s2 = lane_get(oct_sum(t * t), 0);
            if (s2 > fn * fn) {
                const real sc = fn / sqrt(s2);
                if (mine && lane >= 1) f *= sc;
            }
        }
        }
        if (mine) S[6] = f;
        const real delta = mine ? f - f0 : real(0);
        // ---- qacc += B_r^T delta_r ----
        const real dd = __shfl(delta, d, 64);
        if (inA) __hip_atomic_fetch_add(q + adrA, BA * dd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (inB) __hip_atomic_fetch_add(q + adrB, BB * dd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // only this wave touches the env's LDS record, and a wave's LDS operations execute in issue order: a wavefront-scope fence
        // keeps the compiler from moving the next step's reads above the atomics without draining the look-ahead global loads
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        gi = gin;
        gin = ginn;
        ginn = __builtin_amdgcn_readfirstlane(ginnn_v);
        ra = ran;
        ran = rann;
        JA = JAn; JB = JBn; BA = BAn; BB = BBn;
#pragma unroll
        for (int s = 0; s < GRP_MAX - 1; s++) ac[s] = an[s];
#pragma unroll
        for (int s = 0; s < GA_QW; s++) qc[s] = qn[s];
        const bool sweep_end = g + 1 >= ngrp;
        g = g1;
        if (sweep_end) {
            // end of a sweep; a noslip sweep that improved the cost by less than noslip_tolerance ends the pass [EXT]
            if (noslip) {
                // contact blocks add their (wave-uniform) change on every lane, dry-friction rows on their own lane only: lane 0's
                // share of the former plus the lanes' own parts
                const real uni = lane_get(imp_c, 0);
                if (uni + lane_get(wave_sum(imp), 0) < noslip_tol_scaled) break;
            }
            imp = 0;
            imp_c = 0;
            it++;
            if (fl && it >= iters && step + 1 < total) floss_sweep();
        }
    }
    if (prof && lane == 0) { prof[0] += nstep; prof[5] += (int)(__builtin_readcyclecounter() - tent); }
    if (fl && per_ns <= 0) {
        for (int sw = 0; sw < noslip_only; sw++) {
            imp = 0;
            floss_sweep();
            if (lane_get(wave_sum(imp), 0) < noslip_tol_scaled) break;
        }
    }
}


// The friction block of a SLIDING contact in noslip_trees: mju_QCQP's multiplier iteration on the Householder-tridiagonal form of the
// scaled block, as in pgs_groups (same operations in the same order), the block gathered by octet broadcasts instead of wave-wide
// ones; redundantly on the eight lanes of the octet.  Returns the new force of this lane's row (lanes 1..n), *change = the cost change.
template <typename real>
__device__ AVS_OUTLINE_3 real qcqp_slide_octet(GLB_PTR(const real) gA, int g, int n, int fi, real res_r, real f0, real muinv, real invn, real qc5, real fn,
                                                          int tridiag, real* change_out) {
    struct { int tridiag; } nl{tridiag};
    struct { real f0, muinv, invn; real qc[GA_QW]; int g; } cur;
    cur.f0 = f0; cur.muinv = muinv; cur.invn = invn; cur.qc[5] = qc5; cur.g = g;
    const real r2 = fn * fn;
    const int tri = fi <= 5 ? fi * (fi - 1) / 2 : 0;
    real acs[GRP_MAX - 1];
#pragma unroll
    for (int k = 0; k < GRP_MAX - 1; k++) acs[k] = gA[GA_W * cur.g + tri + k];
    real Aq[5][5], bq[5], dq[5], oldf[5], resq[5], v[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const bool in = j < n;
        resq[j] = in ? oct_bcast_n(res_r, j + 1) : real(0);
        oldf[j] = in ? oct_bcast_n(cur.f0, j + 1) : real(0);
        const real mi = oct_bcast_n(cur.muinv, j + 1);
        dq[j] = in ? real(1) / mi : real(1);
        const real di = oct_bcast_n(cur.invn, j + 1);
        Aq[j][j] = in ? real(1) / di : real(1);
#pragma unroll
        for (int k = 0; k < j; k++) { const real c = in ? oct_bcast_n(acs[k + 1], j + 1) : real(0); Aq[j][k] = c; Aq[k][j] = c; }
    }
#pragma unroll
    for (int j = 0; j < 5; j++) {
        real t = resq[j];
#pragma unroll
        for (int k = 0; k < 5; k++) t -= (j < n && k < n) ? Aq[j][k] * oldf[k] : real(0);
        bq[j] = t;
        v[j] = 0;
    }
    const real vtol = QTol<real>::abs + QTol<real>::rel * r2;
    real la = 0, y[5];
    bool singular = oct_bcast<1>(cur.qc[5]) != real(0);
    real As[5][5], cs[5], w[5], hv[3][5], hb[3];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        cs[j] = j < n ? bq[j] * dq[j] : real(0);
        y[j] = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) As[j][k] = (j < n && k < n) ? Aq[j][k] * dq[j] * dq[k] : (j == k ? real(1) : real(0));
    }
    if (!singular) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            real sigma = 0;
#pragma unroll
            for (int i = k + 2; i < 5; i++) sigma += As[i][k] * As[i][k];
            const real x0 = As[k + 1][k];
#pragma unroll
            for (int i = 0; i < 5; i++) hv[k][i] = 0;
            hb[k] = 0;
            if (sigma != real(0)) {
                const real nrm = sqrt(x0 * x0 + sigma), alpha = x0 > 0 ? -nrm : nrm;
                hv[k][k + 1] = x0 - alpha;
#pragma unroll
                for (int i = k + 2; i < 5; i++) hv[k][i] = As[i][k];
                const real beta = real(2) / (hv[k][k + 1] * hv[k][k + 1] + sigma);
                hb[k] = beta;
                real pv[5], vp = 0;
#pragma unroll
                for (int i = k + 1; i < 5; i++) {
                    real t = 0;
#pragma unroll
                    for (int j = k + 1; j < 5; j++) t += As[i][j] * hv[k][j];
                    pv[i] = beta * t;
                    vp += hv[k][i] * pv[i];
                }
                const real K = real(0.5) * beta * vp;
#pragma unroll
                for (int i = k + 1; i < 5; i++) pv[i] -= K * hv[k][i];
#pragma unroll
                for (int i = k + 1; i < 5; i++)
#pragma unroll
                    for (int j = k + 1; j <= i; j++) { As[i][j] -= hv[k][i] * pv[j] + pv[i] * hv[k][j]; As[j][i] = As[i][j]; }
                As[k + 1][k] = alpha; As[k][k + 1] = alpha;
#pragma unroll
                for (int i = k + 2; i < 5; i++) { As[i][k] = 0; As[k][i] = 0; }
                real t = 0;
#pragma unroll
                for (int i = k + 1; i < 5; i++) t += hv[k][i] * cs[i];
                t *= beta;
#pragma unroll
                for (int i = k + 1; i < 5; i++) cs[i] -= t * hv[k][i];
            }
        }
        const real b0 = As[1][0], b1 = As[2][1], b2 = As[3][2], b3 = As[4][3];
        for (int iter = 0; iter < 20; iter++) {
            const real d0 = As[0][0] + la, r0 = real(1) / d0, l0 = b0 * r0;
            const real d1 = As[1][1] + la - l0 * b0, r1 = real(1) / d1, l1 = b1 * r1;
            const real d2 = As[2][2] + la - l1 * b1, r2_ = real(1) / d2, l2 = b2 * r2_;
            const real d3 = As[3][3] + (3 < n ? la : real(0)) - l2 * b2, r3 = real(1) / d3, l3 = b3 * r3;
            const real d4 = As[4][4] + (4 < n ? la : real(0)) - l3 * b3, r4 = real(1) / d4;
            real z0 = -cs[0], z1 = -cs[1] - l0 * z0, z2 = -cs[2] - l1 * z1, z3 = -cs[3] - l2 * z2, z4 = -cs[4] - l3 * z3;
            y[4] = z4 * r4; y[3] = z3 * r3 - l3 * y[4]; y[2] = z2 * r2_ - l2 * y[3]; y[1] = z1 * r1 - l1 * y[2]; y[0] = z0 * r0 - l0 * y[1];
            real val = -r2;
#pragma unroll
            for (int i = 0; i < 5; i++) val += y[i] * y[i];
            if (val < vtol) break;
            z0 = y[0]; z1 = y[1] - l0 * z0; z2 = y[2] - l1 * z1; z3 = y[3] - l2 * z2; z4 = y[4] - l3 * z3;
            w[4] = z4 * r4; w[3] = z3 * r3 - l3 * w[4]; w[2] = z2 * r2_ - l2 * w[3]; w[1] = z1 * r1 - l1 * w[2]; w[0] = z0 * r0 - l0 * w[1];
            real yw = 0;
#pragma unroll
            for (int i = 0; i < 5; i++) yw += y[i] * w[i];
            const real delta = nl.tridiag == 2 ? secular_step(val, r2, fn, yw) : val / (2 * yw);
            if (delta < QTol<real>::abs + QTol<real>::rel * la) break;
            la += delta;
        }
#pragma unroll
        for (int k = 2; k >= 0; k--) {
            real t = 0;
#pragma unroll
            for (int i = k + 1; i < 5; i++) t += hv[k][i] * y[i];
            t *= hb[k];
#pragma unroll
            for (int i = k + 1; i < 5; i++) y[i] -= t * hv[k][i];
        }
    }
#pragma unroll
    for (int j = 0; j < 5; j++) v[j] = (singular || !(j < n)) ? real(0) : y[j] * dq[j];
    if (!singular && la != 0) {       // exactly onto the ellipsoid
        real sq = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) sq += j < n ? v[j] * v[j] / (dq[j] * dq[j]) : real(0);
        const real sc = sqrt(r2 / tmax(real(1e-15), sq));
#pragma unroll
        for (int j = 0; j < 5; j++) v[j] *= sc;
    }
    real change = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        real t = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) t += (j < n && k < n) ? Aq[j][k] * (v[k] - oldf[k]) : real(0);
        change += j < n ? (v[j] - oldf[j]) * (real(0.5) * t + resq[j]) : real(0);
    }
    *change_out = change;
    real fv = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) if (fi == j + 1 && j < n) fv = v[j];
    return fv;
}

// ------------------------------------------------------------------------------------------------
// The noslip pass when every contact touches ONE kinematic tree (objects resting on the table, an arm on the table; no contact
// between two trees -- the scene of the headline workload until a gripper closes on something).  mj_solNoSlip's sweep relaxes the
// dry-friction rows, then the contacts in order; rows and contacts of different trees share no dof, so the sweep falls apart into
// one independent chain per tree, and pgs_groups' "one contact per step for the whole wave" (280 instructions a step) leaves that
// on the table.  Here octet t of the wave owns tree t for the whole pass: lane 8 t + k keeps qacc[dof k of tree t] in a register
// (as the per-tree dry-friction steps already do) and the octets walk their trees' contacts side by side, one contact per step and
// octet -- the steps of a sweep are the LONGEST chain, not the number of contacts, and a step needs no LDS round trip for the
// acceleration, no atomics and no wave-wide broadcasts:
//   row residuals   lane k holds column k of the contact's J (6 loads, one step ahead): 6 products, 6 octet sums (DPP);
//   friction block  the multiplier-0 candidate f - A^-1 res of mju_QCQP through the block inverse made with the rows (lane r = row r,
//                   the residuals come over by octet broadcasts), its cone test and cost change by two octet sums; condim-3 contacts
//                   run mju_QCQP2's two-by-two multiplier iteration, redundantly on the lanes of the octet;
//   update          x_k += sum_r B[r][k] dl_r with the contact's J M^-1 column in registers.
// Same arithmetic per contact as pgs_groups' noslip steps (same octet sums, same candidate, same tests), the order of the contacts
// of a tree is the reference's, and trees do not interact: the results differ from pgs_groups' by the rounding of the acceleration
// update (a chain of FMAs here, LDS atomic adds of rounded products there) and of the sweep's improvement sum.
// A contact with >= 3 friction rows that SLIDES (candidate outside its cone section) needs the multiplier iteration: the function
// then gives up -- it has written nothing but the rows' spare word -- and the caller runs pgs_groups from the untouched state.
// Returns 1 when the pass is done, 0 for "use pgs_groups" (two-tree contact, more than 64 contacts, sliding contact, odd layout).
// ------------------------------------------------------------------------------------------------
template <typename real>
__device__ AVS_OUTLINE_1 int noslip_trees(LDS_PTR(real) rowS, LDS_PTR(const int) rowI, LDS_PTR(const int) cefc, GLB_PTR(const real) rJ,
                                                      LDS_PTR(real) q, LDS_PTR(const int) gI, GLB_PTR(const real) gA, int ncon, int nefc, int noslip_iters,
                                                      real noslip_tol_scaled, NL_PARAMS) {
    NL_UNPACK(real);
    rowS = uni_lds(rowS); rowI = uni_lds(rowI); cefc = uni_lds(cefc); q = uni_lds(q); gI = uni_lds(gI);
    rJ = uni_glb(rJ); gA = uni_glb(gA);
    ncon = __builtin_amdgcn_readfirstlane(ncon); nefc = __builtin_amdgcn_readfirstlane(nefc); noslip_iters = __builtin_amdgcn_readfirstlane(noslip_iters);
    noslip_tol_scaled = lane_get(noslip_tol_scaled, 0);
    nl.Minv = uni_lds(nl.Minv); nl.tadr = uni_lds(nl.tadr); nl.tnum = uni_lds(nl.tnum); nl.floss_dof = uni_lds(nl.floss_dof); nl.dmap = uni_lds(nl.dmap);
    nl.ntree = __builtin_amdgcn_readfirstlane(nl.ntree); nl.nv = __builtin_amdgcn_readfirstlane(nl.nv); nl.neq = __builtin_amdgcn_readfirstlane(nl.neq);
    nl.nfloss = __builtin_amdgcn_readfirstlane(nl.nfloss); nl.nlg = __builtin_amdgcn_readfirstlane(nl.nlg); nl.tridiag = __builtin_amdgcn_readfirstlane(nl.tridiag);
    const int lane = threadIdx.x & 63, ft = lane >> 3, fi = lane & 7;
    if (noslip_iters <= 0 || nl.nlg < 0 || ncon <= 0 || ncon > 64) return 0;
    // ---- the contacts' trees; every contact must have its rows, one tree, and the group pgs_groups would give it ----
    // (a contact without rows -- the reward-only pins of the needle and of the hook -- is in no chain and has no group: the groups
    // count the contacts WITH rows)
    int ctree = -1;
    bool bad = false, two_ = false;
    const int ce_l = lane < ncon ? cefc[lane] : -1;
    const unsigned long long hasrows = __ballot(ce_l >= 0);
    const int g_lane = nl.nlg + __popcll(hasrows & ((1ull << lane) - 1ull));      // the group of this lane's contact
    if (ce_l >= 0) {
        const int ce = ce_l, g = g_lane;
        const int h = ce & 0xffff, ra = rowI[h];
        ctree = (ra >> 10) & 7;
        two_ = ((ra >> 19) & 15) != 0;
        bad = two_ || ctree >= nl.ntree || (gI[g] & 0xffff) != h || ((gI[g] >> 16) & 15) != (ce >> 16) || (ce >> 16) > GRP_MAX || (ce >> 16) == 2;
    }
    LDS_PTR(int) prof = uni_lds(nl.prof);
    if (__any(bad)) { if (prof && lane == 0) prof[1] += 1; return __any(two_) ? 2 : 0; }      // 2: a contact between two trees (noslip_trees2)
    unsigned long long chain = 0;      // this octet's contacts (bit c = contact c), walked in index order
    int nstep = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const unsigned long long m = __ballot(ctree == t);
        if (ft == t) chain = m;
        const int n = __popcll(m);
        nstep = n > nstep ? n : nstep;
    }
    // ---- dry-friction rows, as in pgs_groups: lane 8 t + i = dof i of tree t ----
    for (int k = lane; k < nl.nv; k += 64) nl.dmap[k] = -1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int r = lane; r < nl.nfloss; r += 64) nl.dmap[nl.floss_dof[r]] = nl.neq + r;
    // the spare word of every row record takes the forces of the pass; the rows' own word is written when the pass has succeeded
    for (int i = lane; i < nefc; i += 64) rowS[RS_S * i + 8] = rowS[RS_S * i + 6];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool fm = ft < nl.ntree && fi < nl.tnum[ft < nl.ntree ? ft : 0];
    const int fdof = fm ? nl.tadr[ft] + fi : 0;
    const int frow = fm ? nl.dmap[fdof] : -1;
    real mrow[TREE_W];
#pragma unroll
    for (int s = 0; s < TREE_W; s++) mrow[s] = fm ? nl.Minv[64 * ft + 8 * fi + s] : real(0);
    LDS_PTR(real) FS = rowS + RS_S * (frow >= 0 ? frow : 0);
    const real faref = FS[0], finv = frow >= 0 ? FS[3] : real(0), flo = FS[4], fhi = FS[5];
    const real fdiag = finv != 0 ? real(1) / finv : real(0);
    real ff = frow >= 0 ? FS[6] : real(0);
    real x = fm ? q[fdof] : real(0);
    // ---- contact data, one step ahead ----
    struct CD { int h, dim, g; real J[GRP_MAX], B[GRP_MAX], qc[GA_QW], ar[GRP_MAX], a1, aref, invn, f0, muinv; };
    auto fetch = [&](unsigned long long& rem, CD& d) {
        const bool on = rem != 0;
        const int c = on ? __builtin_ctzll(rem) : 0;
        rem &= rem - 1;
        const int ce = cefc[c];
        d.h = on ? (ce & 0xffff) : 0; d.dim = on ? (ce >> 16) : 0; d.g = __shfl(g_lane, c, 64);
        // the six records from the contact's first row on, whatever its row count: one address, constant offsets (what lies past
        // the contact -- the next contact's rows, the env's spare capacity, for the launch's last env the J M^-1 half of the
        // buffer -- is masked where it is used)
        GLB_PTR(const real) R = rJ + ROW_S * d.h + fi;
#pragma unroll
        for (int r = 0; r < GRP_MAX; r++) { d.J[r] = R[ROW_S * r]; d.B[r] = R[ROW_S * r + TREE_W]; }
        const int qr = (fi >= 1 && fi <= 5) ? fi - 1 : 0;
        GLB_PTR(const real) Q = gA + GA_W * d.g + GA_Q + qr;
#pragma unroll
        for (int s = 0; s < GA_QW; s++) d.qc[s] = Q[8 * s];
        d.a1 = gA[GA_W * d.g + 2];      // the coupling of the two friction rows of a condim-3 contact (rows 2 and 1)
        LDS_PTR(const real) S0 = rowS + RS_S * d.h;
#pragma unroll
        for (int r = 1; r < GRP_MAX; r++) d.ar[r] = S0[RS_S * r];       // the rows' reference accelerations, for every lane
        LDS_PTR(const real) S = S0 + RS_S * (fi < d.dim ? fi : 0);
        d.aref = S[0]; d.invn = S[3]; d.f0 = S[8]; d.muinv = S[7];
    };
    bool slid = false;
    for (int sweep = 0; sweep < noslip_iters; sweep++) {
        real imp = 0;
        // dry-friction rows of mj_solNoSlip [EXT]: clamped scalar update, undone when it would raise the cost (costChange)
#pragma unroll
        for (int s = 0; s < TREE_W; s++) {
            const real res = x - faref;
            const real fs = tmin(tmax(ff - res * finv, flo), fhi);
            const real dl_ = fs - ff, ch = dl_ * (real(0.5) * dl_ * fdiag + res);
            const bool take = frow >= 0 && fi == s && !(ch > real(1e-10));
            if (take) { imp -= ch; ff = fs; }
            const real ds = oct_bcast_n(take ? dl_ : real(0), s);
            x += mrow[s] * ds;
        }
        // the contacts of this octet's tree, in order
        unsigned long long rem = chain;
        CD ca, cb;      // two sets, used alternately: the next contact's data lands in one while the other is worked on
        auto one_step = [&](const CD& cur) {
            const int dim = cur.dim, n = dim - 1;
            const bool row = fi >= 1 && fi < dim;
            // row residuals J_r . qacc: every lane ends up with all of them
            real res_r = 0, ra[5];
#pragma unroll
            for (int r = 1; r < GRP_MAX; r++) {
                const real sr = oct_sum(r < dim ? cur.J[r] * x : real(0));
                ra[r - 1] = r < dim ? sr - cur.ar[r] : real(0);
                if (fi == r) res_r = ra[r - 1];
            }
            const real fn = oct_bcast<0>(cur.f0), r2 = fn * fn;
            real f = cur.f0;
            if (n >= 3) {
                // multiplier 0 first: the unconstrained minimiser f - A^-1 res (the inverse made with the rows); inside the cone section
                // this is mju_QCQP's answer, and its cost change is dl . res / 2
                bool done = false;
                if (!(fn < real(1e-15)) && oct_bcast<1>(cur.qc[5]) == real(0)) {
                    real t = 0;
#pragma unroll
                    for (int k = 0; k < 5; k++) t += cur.qc[k] * ra[k];
                    const real dl_ = row ? -t : real(0), vr = cur.f0 + dl_;
                    const real w = row ? vr * cur.muinv : real(0);
                    const real val = oct_sum(w * w) - r2;
                    if (val < QTol<real>::abs + QTol<real>::rel * r2) {
                        const real change = real(0.5) * oct_sum(dl_ * res_r);
                        if (!(change > real(1e-10))) {
                            if (fi == 0) imp -= change;
                            if (row) f = vr;
                        }
                        done = true;
                    }
                }
                // No normal force: mj_solNoSlip takes the friction forces to zero (subject to the cost test).  They are zero already when
                // the primal solution had no normal force either -- nothing to do; anything else, and the multiplier iteration of a
                // sliding contact, is pgs_groups' business
                if (!done && fn < real(1e-15) && oct_sum(row ? fabs(cur.f0) : real(0)) == real(0)) done = true;
                if (!done && (nl.tridiag == 0 || fn < real(1e-15))) slid = true;
                else if (__builtin_expect(!done, 0)) {
                    // the contact slides: the multiplier iteration, out of line (rare; its forty-odd live values would otherwise sit in
                    // the registers of every step)
                    real change;
                    const real fv = qcqp_slide_octet<real>(gA, cur.g, n, fi, res_r, cur.f0, cur.muinv, cur.invn, cur.qc[5], fn, nl.tridiag, &change);
                    if (!(change > real(1e-10))) {
                        if (fi == 0) imp -= change;
                        if (row) f = fv;
                    }
                }
            } else if (n == 2) {
                // mju_QCQP2 [EXT] as pgs_groups evaluates it, on the lanes of the octet
                const real resq0 = ra[0], resq1 = ra[1];
                const real of0 = oct_bcast<1>(cur.f0), of1 = oct_bcast<2>(cur.f0);
                const real dq0 = real(1) / oct_bcast<1>(cur.muinv), dq1 = real(1) / oct_bcast<2>(cur.muinv);
                const real A00 = real(1) / oct_bcast<1>(cur.invn), A11 = real(1) / oct_bcast<2>(cur.invn), A10 = oct_bcast<2>(cur.a1);
                real bq0 = resq0, bq1 = resq1;
                bq0 -= A00 * of0; bq0 -= A10 * of1;
                bq1 -= A10 * of0; bq1 -= A11 * of1;
                real v0 = 0, v1 = 0;
                if (!(fn < real(1e-15))) {
                    const real vtol = QTol<real>::abs + QTol<real>::rel * r2;
                    real la = 0;
                    const real b1 = bq0 * dq0, b2 = bq1 * dq1;
                    const real A11s = A00 * dq0 * dq0, A22s = A11 * dq1 * dq1, A12s = A10 * dq0 * dq1;
                    real y1 = 0, y2 = 0;
                    bool singular = false;
                    for (int iter = 0; iter < 20; iter++) {
                        const real det = (A11s + la) * (A22s + la) - A12s * A12s;
                        if (det < real(1e-10)) { singular = true; break; }
                        const real detinv = real(1) / det, P11 = (A22s + la) * detinv, P22 = (A11s + la) * detinv, P12 = -A12s * detinv;
                        y1 = -P11 * b1 - P12 * b2;
                        y2 = -P12 * b1 - P22 * b2;
                        const real val = y1 * y1 + y2 * y2 - r2;
                        if (val < vtol) break;
                        const real yw = P11 * y1 * y1 + 2 * P12 * y1 * y2 + P22 * y2 * y2;
                        const real delta = nl.tridiag == 2 ? secular_step(val, r2, fn, yw) : val / (2 * yw);
                        if (delta < QTol<real>::abs + QTol<real>::rel * la) break;
                        la += delta;
                    }
                    v0 = singular ? real(0) : y1 * dq0;
                    v1 = singular ? real(0) : y2 * dq1;
                    if (!singular && la != 0) {       // exactly onto the ellipsoid
                        const real sq = v0 * v0 / (dq0 * dq0) + v1 * v1 / (dq1 * dq1);
                        const real sc = sqrt(r2 / tmax(real(1e-15), sq));
                        v0 *= sc; v1 *= sc;
                    }
                }
                const real change = (v0 - of0) * (real(0.5) * (A00 * (v0 - of0) + A10 * (v1 - of1)) + resq0) + (v1 - of1) * (real(0.5) * (A10 * (v0 - of0) + A11 * (v1 - of1)) + resq1);
                if (!(change > real(1e-10)) && dim == 3) {
                    if (fi == 0) imp -= change;
                    if (fi == 1) f = v0;
                    if (fi == 2) f = v1;
                }
            }
            // x += B^T (f - f0); the normal row does not move
            const real dl = row ? f - cur.f0 : real(0);
#pragma unroll
            for (int r = 1; r < GRP_MAX; r++) x += (r < dim ? cur.B[r] : real(0)) * oct_bcast_n(dl, r);
            if (row) rowS[RS_S * (cur.h + fi) + 8] = f;
        };
        fetch(rem, ca);
        for (int step = 0; step < nstep; step += 2) {
            fetch(rem, cb);
            one_step(ca);
            if (step + 1 < nstep) {
                fetch(rem, ca);
                one_step(cb);
            }
        }
        if (__any(slid)) { if (prof && lane == 0) prof[2] += 1; return 0; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the next sweep reads the forces this one has stored
        __builtin_amdgcn_wave_barrier();
        if (lane_get(wave_sum(imp), 0) < noslip_tol_scaled) break;
    }
    // ---- the pass has succeeded: accelerations and forces to their places ----
    if (prof && lane == 0) prof[3] += 1;
    if (fm) q[fdof] = x;
    if (frow >= 0) FS[8] = ff;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < nefc; i += 64) rowS[RS_S * i + 6] = rowS[RS_S * i + 8];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return 1;
}

// noslip_trees for passes with a contact BETWEEN TWO kinematic trees (a gripper holding something, an object lying on another): such a
// contact sits in both trees' chains and is relaxed by both octets in the same step -- its level is one more than the larger of its
// two predecessors' levels, a tree has at most one contact per level --, each octet sums its tree's half of the row residuals and gets
// the other half from its partner (ds_bpermute; first tree's part + second tree's part on both sides, so that the two octets go on
// with identical numbers), both run the friction block, each moves its own tree's accelerations, the first tree's octet stores the
// forces; a sliding contact runs the multiplier iteration on both octets (identical inputs, identical results: oct_sum keeps FMA
// contraction out of the sums).  A function of its own, entered when noslip_trees returns 2: with the two-tree bookkeeping in
// noslip_trees the headline workload, which has no such contact, lost 1.6 %; this way 0.3 %.
template <typename real>
__device__ AVS_OUTLINE_1 int noslip_trees2(LDS_PTR(real) rowS, LDS_PTR(const int) rowI, LDS_PTR(const int) cefc, GLB_PTR(const real) rJ, GLB_PTR(const real) rB,
                                                      LDS_PTR(real) q, LDS_PTR(const int) gI, GLB_PTR(const real) gA, int ncon, int nefc, int noslip_iters,
                                                      real noslip_tol_scaled, NL_PARAMS) {
    NL_UNPACK(real);
    rowS = uni_lds(rowS); rowI = uni_lds(rowI); cefc = uni_lds(cefc); q = uni_lds(q); gI = uni_lds(gI);
    rJ = uni_glb(rJ); rB = uni_glb(rB); gA = uni_glb(gA);
    ncon = __builtin_amdgcn_readfirstlane(ncon); nefc = __builtin_amdgcn_readfirstlane(nefc); noslip_iters = __builtin_amdgcn_readfirstlane(noslip_iters);
    noslip_tol_scaled = lane_get(noslip_tol_scaled, 0);
    nl.Minv = uni_lds(nl.Minv); nl.tadr = uni_lds(nl.tadr); nl.tnum = uni_lds(nl.tnum); nl.floss_dof = uni_lds(nl.floss_dof); nl.dmap = uni_lds(nl.dmap);
    nl.ntree = __builtin_amdgcn_readfirstlane(nl.ntree); nl.nv = __builtin_amdgcn_readfirstlane(nl.nv); nl.neq = __builtin_amdgcn_readfirstlane(nl.neq);
    nl.nfloss = __builtin_amdgcn_readfirstlane(nl.nfloss); nl.nlg = __builtin_amdgcn_readfirstlane(nl.nlg); nl.tridiag = __builtin_amdgcn_readfirstlane(nl.tridiag);
    const int lane = threadIdx.x & 63, ft = lane >> 3, fi = lane & 7;
    if (noslip_iters <= 0 || nl.nlg < 0 || ncon <= 0 || ncon > 64) return 0;
    // ---- the contacts' trees (one, or two for a contact between two trees); every contact must have its rows and the group
    // pgs_groups would give it ----
    int tA = -1, tB = -1;
    bool bad = false;
    const int ce_l = lane < ncon ? cefc[lane] : -1;
    const unsigned long long hasrows = __ballot(ce_l >= 0);      // (contacts without rows -- reward-only pins -- are in no chain and have no group)
    const int g_lane = nl.nlg + __popcll(hasrows & ((1ull << lane) - 1ull));
    if (ce_l >= 0) {
        const int ce = ce_l, h = ce & 0xffff, ra = rowI[h];
        tA = (ra >> 10) & 7;
        tB = ((ra >> 19) & 15) != 0 ? (ra >> 23) & 7 : -1;
        bad = tA >= nl.ntree || tB >= nl.ntree || tB == tA || (gI[g_lane] & 0xffff) != h || ((gI[g_lane] >> 16) & 15) != (ce >> 16) || (ce >> 16) > GRP_MAX || (ce >> 16) == 2;
    }
    LDS_PTR(int) prof = uni_lds(nl.prof);
    if (__any(bad)) { if (prof && lane == 0) prof[1] += 1; return 0; }
    // The order: a tree's contacts in index order (its chain); a contact between two trees sits in both chains and is relaxed by both
    // octets in the same step, so it waits for its predecessors in BOTH.  Level of a contact = the step it is relaxed in = 1 + the
    // larger of its two predecessors' levels; a tree has at most one contact per level.
    unsigned long long chain = 0, mA = 0, mB = 0;      // this octet's chain; the chains of this lane's contact's trees
    int nstep = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const unsigned long long m = __ballot(tA == t || tB == t);
        if (ft == t) chain = m;
        if (tA == t) mA = m;
        if (tB == t) mB = m;
        const int n = __popcll(m);
        nstep = n > nstep ? n : nstep;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    int lev = ce_l >= 0 ? __popcll(mA & below) : 1 << 20;       // no contact between two trees: the position in the chain
    if (__any(tB >= 0)) {
        const int pA = (mA & below) ? 63 - __builtin_clzll(mA & below) : -1, pB = (tB >= 0 && (mB & below)) ? 63 - __builtin_clzll(mB & below) : -1;
        lev = ce_l >= 0 ? -1 : 1 << 20;
        nstep = 0;
        for (int L = 0; L < 64; L++) {
            const int la = __shfl(lev, pA >= 0 ? pA : 0, 64), lb = __shfl(lev, pB >= 0 ? pB : 0, 64);
            if (lev < 0 && (pA < 0 || la >= 0) && (pB < 0 || lb >= 0)) lev = L;
            nstep = L + 1;
            if (!__any(lev < 0)) break;
        }
    }
    // ---- dry-friction rows, as in pgs_groups: lane 8 t + i = dof i of tree t ----
    for (int k = lane; k < nl.nv; k += 64) nl.dmap[k] = -1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int r = lane; r < nl.nfloss; r += 64) nl.dmap[nl.floss_dof[r]] = nl.neq + r;
    // the spare word of every row record takes the forces of the pass; the rows' own word is written when the pass has succeeded
    for (int i = lane; i < nefc; i += 64) rowS[RS_S * i + 8] = rowS[RS_S * i + 6];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool fm = ft < nl.ntree && fi < nl.tnum[ft < nl.ntree ? ft : 0];
    const int fdof = fm ? nl.tadr[ft] + fi : 0;
    const int frow = fm ? nl.dmap[fdof] : -1;
    real mrow[TREE_W];
#pragma unroll
    for (int s = 0; s < TREE_W; s++) mrow[s] = fm ? nl.Minv[64 * ft + 8 * fi + s] : real(0);
    LDS_PTR(real) FS = rowS + RS_S * (frow >= 0 ? frow : 0);
    const real faref = FS[0], finv = frow >= 0 ? FS[3] : real(0), flo = FS[4], fhi = FS[5];
    const real fdiag = finv != 0 ? real(1) / finv : real(0);
    real ff = frow >= 0 ? FS[6] : real(0);
    real x = fm ? q[fdof] : real(0);
    // ---- contact data, one step ahead ----
    // the sweeps, compiled twice: for passes whose contacts all touch one tree (nothing of the two-tree bookkeeping in the steps) and for
    // passes with a contact between two trees
    auto sweeps = [&](auto two_tag) -> int {
    constexpr bool TWO = decltype(two_tag)::value;
    struct CD { int h, dim, g, side, partner; bool two; real J[GRP_MAX], B[GRP_MAX], qc[GA_QW], ar[GRP_MAX], a1, aref, invn, f0, muinv; };
    unsigned long long rem = 0;      // (one-tree passes: what is left of this octet's chain in the current sweep)
    auto fetch = [&](int L, CD& d) {
        // this octet's contact of level L, if it has one; without contacts between two trees the levels are the positions in the chain
        const unsigned long long m = TWO ? __ballot(lev == L) & chain : rem;
        const bool on = m != 0;
        const int c = on ? __builtin_ctzll(m) : 0;
        if (!TWO) rem &= rem - 1;
        const int ce = cefc[c];
        d.h = on ? (ce & 0xffff) : 0; d.dim = on ? (ce >> 16) : 0; d.g = __shfl(g_lane, c, 64);
        // a contact between two trees: this octet's tree is its first (side 0: words 0..7 of the row records) or its second (side 1:
        // words 8..15); its J M^-1 rows are in the second buffer (a one-tree contact keeps them in words 8..15 of the J record)
        d.two = false; d.side = 0; d.partner = ft;
        if (TWO) {
            const int ra_ = rowI[d.h];
            d.two = on && ((ra_ >> 19) & 15) != 0;
            d.side = (d.two && ((ra_ >> 10) & 7) != ft) ? 1 : 0;
            d.partner = d.two ? (d.side ? (ra_ >> 10) & 7 : (ra_ >> 23) & 7) : ft;
        }
        // the six records from the contact's first row on, whatever its row count: one address, constant offsets (what lies past
        // the contact -- the next contact's rows, the env's spare capacity, for the launch's last env the J M^-1 half of the
        // buffer -- is masked where it is used)
        GLB_PTR(const real) R = rJ + ROW_S * d.h + TREE_W * d.side + fi;
        GLB_PTR(const real) RB = (TWO && d.two) ? rB + ROW_S * d.h + TREE_W * d.side + fi : R + TREE_W;
#pragma unroll
        for (int r = 0; r < GRP_MAX; r++) { d.J[r] = R[ROW_S * r]; d.B[r] = RB[ROW_S * r]; }
        const int qr = (fi >= 1 && fi <= 5) ? fi - 1 : 0;
        GLB_PTR(const real) Q = gA + GA_W * d.g + GA_Q + qr;
#pragma unroll
        for (int s = 0; s < GA_QW; s++) d.qc[s] = Q[8 * s];
        d.a1 = gA[GA_W * d.g + 2];      // the coupling of the two friction rows of a condim-3 contact (rows 2 and 1)
        LDS_PTR(const real) S0 = rowS + RS_S * d.h;
#pragma unroll
        for (int r = 1; r < GRP_MAX; r++) d.ar[r] = S0[RS_S * r];       // the rows' reference accelerations, for every lane
        LDS_PTR(const real) S = S0 + RS_S * (fi < d.dim ? fi : 0);
        d.aref = S[0]; d.invn = S[3]; d.f0 = S[8]; d.muinv = S[7];
    };
    bool slid = false;
    for (int sweep = 0; sweep < noslip_iters; sweep++) {
        real imp = 0;
        // dry-friction rows of mj_solNoSlip [EXT]: clamped scalar update, undone when it would raise the cost (costChange)
#pragma unroll
        for (int s = 0; s < TREE_W; s++) {
            const real res = x - faref;
            const real fs = tmin(tmax(ff - res * finv, flo), fhi);
            const real dl_ = fs - ff, ch = dl_ * (real(0.5) * dl_ * fdiag + res);
            const bool take = frow >= 0 && fi == s && !(ch > real(1e-10));
            if (take) { imp -= ch; ff = fs; }
            const real ds = oct_bcast_n(take ? dl_ : real(0), s);
            x += mrow[s] * ds;
        }
        // the contacts of this octet's tree, in order
        CD ca, cb;      // two sets, used alternately: the next contact's data lands in one while the other is worked on
        auto one_step = [&](const CD& cur) {
            const int dim = cur.dim, n = dim - 1;
            const bool row = fi >= 1 && fi < dim;
            // row residuals J_r . qacc: every lane ends up with all of them
            real res_r = 0, ra[5], sr[GRP_MAX];
#pragma unroll
            for (int r = 1; r < GRP_MAX; r++) sr[r] = oct_sum(r < dim ? cur.J[r] * x : real(0));
            if (TWO && __any(cur.two)) {
                // a contact between two trees: the other tree's octet has the other half of every sum (first tree's part + second tree's
                // part, in that order on both sides, so that the two octets go on with identical numbers)
#pragma unroll
                for (int r = 1; r < GRP_MAX; r++) {
                    const real other = __shfl(sr[r], 8 * cur.partner + fi, 64);
                    if (cur.two) sr[r] = cur.side == 0 ? sr[r] + other : other + sr[r];
                }
            }
#pragma unroll
            for (int r = 1; r < GRP_MAX; r++) {
                ra[r - 1] = r < dim ? sr[r] - cur.ar[r] : real(0);
                if (fi == r) res_r = ra[r - 1];
            }
            const real fn = oct_bcast<0>(cur.f0), r2 = fn * fn;
            real f = cur.f0;
            if (n >= 3) {
                // multiplier 0 first: the unconstrained minimiser f - A^-1 res (the inverse made with the rows); inside the cone section
                // this is mju_QCQP's answer, and its cost change is dl . res / 2
                bool done = false;
                if (!(fn < real(1e-15)) && oct_bcast<1>(cur.qc[5]) == real(0)) {
                    real t = 0;
#pragma unroll
                    for (int k = 0; k < 5; k++) t += cur.qc[k] * ra[k];
                    const real dl_ = row ? -t : real(0), vr = cur.f0 + dl_;
                    const real w = row ? vr * cur.muinv : real(0);
                    const real val = oct_sum(w * w) - r2;
                    if (val < QTol<real>::abs + QTol<real>::rel * r2) {
                        const real change = real(0.5) * oct_sum(dl_ * res_r);
                        if (!(change > real(1e-10))) {
                            if (fi == 0 && cur.side == 0) imp -= change;
                            if (row) f = vr;
                        }
                        done = true;
                    }
                }
                // No normal force: mj_solNoSlip takes the friction forces to zero (subject to the cost test).  They are zero already when
                // the primal solution had no normal force either -- nothing to do; anything else, and the multiplier iteration of a
                // sliding contact, is pgs_groups' business
                if (!done && fn < real(1e-15) && oct_sum(row ? fabs(cur.f0) : real(0)) == real(0)) done = true;
                if (!done && (nl.tridiag == 0 || fn < real(1e-15))) slid = true;
                else if (__builtin_expect(!done, 0)) {
                    // the contact slides: the multiplier iteration, out of line (rare; its forty-odd live values would otherwise sit in
                    // the registers of every step)
                    real change;
                    const real fv = qcqp_slide_octet<real>(gA, cur.g, n, fi, res_r, cur.f0, cur.muinv, cur.invn, cur.qc[5], fn, nl.tridiag, &change);
                    if (!(change > real(1e-10))) {
                        if (fi == 0 && cur.side == 0) imp -= change;
                        if (row) f = fv;
                    }
                }
            } else if (n == 2) {
                // mju_QCQP2 [EXT] as pgs_groups evaluates it, on the lanes of the octet
                const real resq0 = ra[0], resq1 = ra[1];
                const real of0 = oct_bcast<1>(cur.f0), of1 = oct_bcast<2>(cur.f0);
                const real dq0 = real(1) / oct_bcast<1>(cur.muinv), dq1 = real(1) / oct_bcast<2>(cur.muinv);
                const real A00 = real(1) / oct_bcast<1>(cur.invn), A11 = real(1) / oct_bcast<2>(cur.invn), A10 = oct_bcast<2>(cur.a1);
                real bq0 = resq0, bq1 = resq1;
                bq0 -= A00 * of0; bq0 -= A10 * of1;
                bq1 -= A10 * of0; bq1 -= A11 * of1;
                real v0 = 0, v1 = 0;
                if (!(fn < real(1e-15))) {
                    const real vtol = QTol<real>::abs + QTol<real>::rel * r2;
                    real la = 0;
                    const real b1 = bq0 * dq0, b2 = bq1 * dq1;
                    const real A11s = A00 * dq0 * dq0, A22s = A11 * dq1 * dq1, A12s = A10 * dq0 * dq1;
                    real y1 = 0, y2 = 0;
                    bool singular = false;
                    for (int iter = 0; iter < 20; iter++) {
                        const real det = (A11s + la) * (A22s + la) - A12s * A12s;
                        if (det < real(1e-10)) { singular = true; break; }
                        const real detinv = real(1) / det, P11 = (A22s + la) * detinv, P22 = (A11s + la) * detinv, P12 = -A12s * detinv;
                        y1 = -P11 * b1 - P12 * b2;
                        y2 = -P12 * b1 - P22 * b2;
                        const real val = y1 * y1 + y2 * y2 - r2;
                        if (val < vtol) break;
                        const real yw = P11 * y1 * y1 + 2 * P12 * y1 * y2 + P22 * y2 * y2;
                        const real delta = nl.tridiag == 2 ? secular_step(val, r2, fn, yw) : val / (2 * yw);
                        if (delta < QTol<real>::abs + QTol<real>::rel * la) break;
                        la += delta;
                    }
                    v0 = singular ? real(0) : y1 * dq0;
                    v1 = singular ? real(0) : y2 * dq1;
                    if (!singular && la != 0) {       // exactly onto the ellipsoid
                        const real sq = v0 * v0 / (dq0 * dq0) + v1 * v1 / (dq1 * dq1);
                        const real sc = sqrt(r2 / tmax(real(1e-15), sq));
                        v0 *= sc; v1 *= sc;
                    }
                }
                const real change = (v0 - of0) * (real(0.5) * (A00 * (v0 - of0) + A10 * (v1 - of1)) + resq0) + (v1 - of1) * (real(0.5) * (A10 * (v0 - of0) + A11 * (v1 - of1)) + resq1);
                if (!(change > real(1e-10)) && dim == 3) {
                    if (fi == 0 && cur.side == 0) imp -= change;
                    if (fi == 1) f = v0;
                    if (fi == 2) f = v1;
                }
            }
            // x += B^T (f - f0); the normal row does not move
            const real dl = row ? f - cur.f0 : real(0);
#pragma unroll
            for (int r = 1; r < GRP_MAX; r++) x += (r < dim ? cur.B[r] : real(0)) * oct_bcast_n(dl, r);
            if (row && cur.side == 0) rowS[RS_S * (cur.h + fi) + 8] = f;
        };
        rem = chain;
        fetch(0, ca);
        for (int step = 0; step < nstep; step += 2) {
            fetch(step + 1, cb);
            one_step(ca);
            if (step + 1 < nstep) {
                fetch(step + 2, ca);
                one_step(cb);
            }
        }
        if (__any(slid)) { if (prof && lane == 0) prof[2] += 1; return 0; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the next sweep reads the forces this one has stored
        __builtin_amdgcn_wave_barrier();
        if (lane_get(wave_sum(imp), 0) < noslip_tol_scaled) break;
    }
    return 1;
    };
    if (!(__any(tB >= 0) ? sweeps(BoolTag<true>{}) : sweeps(BoolTag<false>{}))) return 0;
    // ---- the pass has succeeded: accelerations and forces to their places ----
    if (prof && lane == 0) prof[3] += 1;
    if (fm) q[fdof] = x;
    if (frow >= 0) FS[8] = ff;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < nefc; i += 64) rowS[RS_S * i + 6] = rowS[RS_S * i + 8];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return 1;
}


}  // namespace avs
#include "avsim_newton.hip.h"
namespace avs {

// one 16-word row record to global memory in 16-byte pieces (rows are 64- / 128-byte aligned)
typedef float avs_v4f __attribute__((ext_vector_type(4)));
typedef double avs_v2d __attribute__((ext_vector_type(2)));
AVS_DEV void store_row16(GLB_PTR(float) dst, const float* v) {
#pragma unroll
    for (int q = 0; q < 4; q++) { avs_v4f t = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}; ((GLB_PTR(avs_v4f))dst)[q] = t; }
}
AVS_DEV void store_row16(GLB_PTR(double) dst, const double* v) {
#pragma unroll
    for (int q = 0; q < 8; q++) { avs_v2d t = {v[2 * q], v[2 * q + 1]}; ((GLB_PTR(avs_v2d))dst)[q] = t; }
}

// eight consecutive LDS reals whose address is a multiple of 16 bytes, as two (four) vector reads
AVS_DEV void lds_load8(const float* p, float* v) {
    LDS_PTR(const avs_v4f) q = (LDS_PTR(const avs_v4f))p;
    const avs_v4f a = q[0], b = q[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
AVS_DEV void lds_load8(const double* p, double* v) {
    LDS_PTR(const avs_v2d) q = (LDS_PTR(const avs_v2d))p;
#pragma unroll
    for (int k = 0; k < 4; k++) { const avs_v2d a = q[k]; v[2 * k] = a.x; v[2 * k + 1] = a.y; }
}

template <int G>
AVS_DEV unsigned long long group_mask(int grp) {
    if (G == 64) return ~0ull;
    return ((1ull << G) - 1ull) << (grp * G);
}
// number of set flags in lower lanes of the group, and total
template <int G>
AVS_DEV int group_rank(bool flag, int grp, int lane, int* total) {
    unsigned long long b = __ballot(flag) & group_mask<G>(grp);
    unsigned long long lower = (G == 64) ? ((1ull << lane) - 1ull) : (((1ull << lane) - 1ull) << (grp * G));
    *total = __popcll(b);
    return __popcll(b & lower);
}

template <typename real>
struct SInert {
    real m, h[3], I[6];
};

// spatial inertia of body b about the world origin in world axes
template <typename real>
AVS_DEV void body_inertia(KPtr<real> ka, const real* xmat, const real* xipos, int b, SInert<real>& s) {
    const real* R = xmat + 9 * b;
    GLB_PTR(const real) Ib = ka->m.body_inertia + 6 * b;
    real I3[9] = {Ib[0], Ib[3], Ib[4], Ib[3], Ib[1], Ib[5], Ib[4], Ib[5], Ib[2]}, T[9], Ic[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T[3 * i + j] = R[3 * i] * I3[j] + R[3 * i + 1] * I3[3 + j] + R[3 * i + 2] * I3[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Ic[3 * i + j] = T[3 * i] * R[3 * j] + T[3 * i + 1] * R[3 * j + 1] + T[3 * i + 2] * R[3 * j + 2];
    real ms = ka->m.body_mass[b];
    real c[3] = {xipos[3 * b], xipos[3 * b + 1], xipos[3 * b + 2]};
    real cc = dot3(c, c);
    s.m = ms;
    s.h[0] = ms * c[0]; s.h[1] = ms * c[1]; s.h[2] = ms * c[2];
    s.I[0] = Ic[0] + ms * (cc - c[0] * c[0]); s.I[1] = Ic[4] + ms * (cc - c[1] * c[1]); s.I[2] = Ic[8] + ms * (cc - c[2] * c[2]);
    s.I[3] = Ic[1] - ms * c[0] * c[1]; s.I[4] = Ic[2] - ms * c[0] * c[2]; s.I[5] = Ic[5] - ms * c[1] * c[2];
}

template <typename real>
AVS_DEV void inert_mul(const real* s /* m,h3,I6 */, const real* mv, real* f) {
    const real *w = mv, *v = mv + 3, *h = s + 1, *I = s + 4;
    real hv[3], hw[3];
    cross3(h, v, hv);
    cross3(h, w, hw);
    f[0] = I[0] * w[0] + I[3] * w[1] + I[4] * w[2] + hv[0];
    f[1] = I[3] * w[0] + I[1] * w[1] + I[5] * w[2] + hv[1];
    f[2] = I[4] * w[0] + I[5] * w[1] + I[2] * w[2] + hv[2];
    f[3] = s[0] * v[0] - hw[0]; f[4] = s[0] * v[1] - hw[1]; f[5] = s[0] * v[2] - hw[2];
}

template <typename real>
AVS_DEV void cross_motion(const real* v, const real* s, real* o) {
    real a[3], b[3], c[3];
    cross3(v, s, a);
    cross3(v, s + 3, b);
    cross3(v + 3, s, c);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
    o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
template <typename real>
AVS_DEV void cross_force(const real* v, const real* f, real* o) {
    real a[3], b[3], c[3];
    cross3(v, f, a);
    cross3(v + 3, f + 3, b);
    cross3(v, f + 3, c);
    o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2];
    o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}

// in-place dense Cholesky of an n x n block (row-major, n<=8) and solves with it
template <typename real>
AVS_DEV void chol_block(const real* A, real* L, int n) {
    for (int j = 0; j < n; j++) {
        real dd = A[j * n + j];
        for (int k = 0; k < j; k++) dd -= L[j * n + k] * L[j * n + k];
        dd = sqrt(tmax(dd, real(1e-30)));
        L[j * n + j] = dd;
        real inv = real(1) / dd;
        for (int i = j + 1; i < n; i++) {
            real s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = s * inv;
        }
    }
}
template <typename real>
AVS_DEV void chol_solve_block(const real* L, real* x, int n) {
    for (int i = 0; i < n; i++) {
        real s = x[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
        x[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        real s = x[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
}

// x^p for the impedance curve: the models' solimp power is 2 (MuJoCo's default), a product then; pow() only otherwise
template <typename real>
AVS_DEV real imp_pow(real x, real p) { return p == real(2) ? x * x : pow(x, p); }
// solimp impedance [EXT: getimpedance]
template <typename real>
AVS_DEV real impedance(const real* si, real pos, real margin) {
    real dmin = tclamp(si[0], real(0.0001), real(0.9999)), dmax = tclamp(si[1], real(0.0001), real(0.9999));
    real width = tmax(si[2], real(1e-15)), mid = tclamp(si[3], real(0.0001), real(0.9999)), power = tmax(si[4], real(1));
    if (dmin == dmax) return real(0.5) * (dmin + dmax);
    real x = fabs(pos - margin) / width;
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    real y;
    if (power == 1) y = x;
    else if (x <= mid) y = imp_pow(x / mid, power) * mid;
    else y = 1 - imp_pow((1 - x) / (1 - mid), power) * (1 - mid);
    return dmin + y * (dmax - dmin);
}

// reward predicates over geom class bits (env.py get_reward x5; compile.py geom_class)
AVS_DEV int has_pair(int c1, int c2, int a, int b) { return ((c1 & a) && (c2 & b)) || ((c2 & a) && (c1 & b)); }
// contact-pair predicates of the five get_reward functions (env.py:425-863) on the geom class bits of compile.py: one
// contact contributes flag bits, the staged reward follows from the OR over all contacts (later rules overwrite earlier)
AVS_DEV int reward_pair_flags(int c1, int c2, int t) {
    enum { CL = 1, CR = 2, CT = 4, CA = 8, CB = 16, CC = 32, CD = 64 };
    int f = 0;  // bit flags: 0 tl, 1 tr, 2 a_table, 3 b_table, 4 ab, 5 cd, 6 ad, 7 ac
    if (has_pair(c1, c2, CT, CA)) f |= 4;
    if (has_pair(c1, c2, CT, CB)) f |= 8;
    if (has_pair(c1, c2, CA, CB)) f |= 16;
    if (has_pair(c1, c2, CC, CD)) f |= 32;
    if (has_pair(c1, c2, CA, CD)) f |= 64;
    if (has_pair(c1, c2, CA, CC)) f |= 128;
    if (has_pair(c1, c2, CA, CR)) f |= 2;
    if (t == 0 || t == 3) { if (has_pair(c1, c2, CB, CL)) f |= 1; }
    else { if (has_pair(c1, c2, CA, CL)) f |= 1; }
    return f;
}
AVS_DEV int reward_from_flags(int f, int t, int* latch) {
    bool tl = f & 1, tr = f & 2, a_table = f & 4, b_table = f & 8, ab = f & 16, cd = f & 32, ad = f & 64, ac = f & 128;
    int rw = 0;
    switch (t) {
        case 0:
            if (tl && tr) rw = 1;
            if (tl && tr && !a_table && !b_table) rw = 2;
            if (ab && !a_table && !b_table) rw = 3;
            if (ac) rw = 4;
            break;
        case 1:
            if (tl && tr) rw = 1;
            if (tl && tr && !a_table) rw = 2;
            if (ab && !a_table) rw = 3;
            if (cd) rw = 4;
            break;
        case 2:
            if (cd) *latch = 1;
            if (tr) rw = 1;
            if (tr && !a_table) rw = 2;
            if (ab && !a_table) rw = 3;
            if (*latch) rw = 4;
            if (tl && !tr && !a_table && !ad && *latch) rw = 5;
            break;
        case 3:
            if (tl && tr) rw = 1;
            if (tl && tr && !a_table && !b_table) rw = 2;
            if (cd) rw = 3;
            break;
        default:
            if (tl && tr) rw = 1;
            if (tl && tr && !a_table) rw = 2;
            if (ab && !a_table) rw = 3;
            if (cd) rw = 4;
    }
    return rw;
}
// get_reward on explicit contact lists (one thread per list): geom id pairs int[n][cap][2], -1 = unused slot
#ifndef AVSIM_TU_F64
__global__ void k_reward_pairs(GLB_PTR(const int) geom_class, int ngeom, int task_id, const int* __restrict__ pairs, int n, int cap, int* __restrict__ latch, int* __restrict__ reward) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int f = 0;
    for (int c = 0; c < cap; c++) {
        const int g1 = pairs[((size_t)i * cap + c) * 2], g2 = pairs[((size_t)i * cap + c) * 2 + 1];
        if (g1 < 0 || g2 < 0 || g1 >= ngeom || g2 >= ngeom) continue;
        f |= reward_pair_flags(geom_class[g1], geom_class[g2], task_id);
    }
    int l = latch ? latch[i] : 0;
    reward[i] = reward_from_flags(f, task_id, &l);
    if (latch) latch[i] = l;
}
#endif


template <typename real, int G>
struct Env {
    KPtr<real> ka;   // model, LDS layout and table offsets: one struct in constant memory, re-read per phase (PHASE_BEGIN)
    int env = 0;                    // global env index (row buffer addressing)
    int diverged = 0;               // the state left the representable range during this launch and was put back to the home pose
    int nit_sum = 0, nit_max = 0;   // Newton iterations over the launch's substeps (diagnostics)
    bool profiling = false;
    real* r;  // real region of this env
    int* ii;  // int region of this env
    int lane, grp;
    const real* lr;
    const int* li;
    long long t_broad = 0, t_narrow = 0;
    __device__ Env(KPtr<real> ka_, real* r_, int* i_, int lane_, int grp_, const real* lr_, const int* li_)
        : ka(ka_), r(r_), ii(i_), lane(lane_), grp(grp_), lr(lr_), li(li_) {}
    // hot model tables live in LDS (copied once per block); the accessors rebuild the pointer from the kernarg offset
    AVS_DEV const int* LI() const { const int* p = li; AVS_ASSUME_LDS(p); return p; }
    AVS_DEV const real* LR() const { const real* p = lr; AVS_ASSUME_LDS(p); return p; }
    // (the env's share of the global scratch is sized by the FULL capacities in both layouts of a two-tier handle: envs stepped with
    // the small and with the full record run side by side)
    AVS_DEV GLB_PTR(real) rows_() const { return (GLB_PTR(real))ka->m.rJ_glob + (size_t)env * ka->lay.gefc * ROW_S; }
    AVS_DEV GLB_PTR(real) rowsB_() const { return (GLB_PTR(real))ka->m.rB_glob + (size_t)env * ka->lay.gefc * ROW_S; }
    AVS_DEV GLB_PTR(real) coup_() const { return (GLB_PTR(real))ka->m.gA_glob + (size_t)env * ka->lay.ggrp * GA_W; }
    AVS_DEV GLB_PTR(int) near_() const { return (GLB_PTR(int))ka->m.near_glob + (size_t)env * NEAR_MAX; }
    AVS_DEV GLB_PTR(int) cand_() const { return (GLB_PTR(int))ka->m.cand_glob + (size_t)env * CAND_MAX; }
    AVS_DEV GLB_PTR(real) bxo_() const { return (GLB_PTR(real))ka->m.bxo_glob + (size_t)env * 64 * BOX_OVF_W; }
    AVS_DEV GLB_PTR(real) gref_() const { return (GLB_PTR(real))ka->m.gref_glob + (size_t)env * ka->m.ngeom * 3; }
    AVS_DEV const int* body_parent_() const { return LI() + ka->mo.body_parent; }
    AVS_DEV const int* body_jntadr_() const { return LI() + ka->mo.body_jntadr; }
    AVS_DEV const int* body_jntnum_() const { return LI() + ka->mo.body_jntnum; }
    AVS_DEV const int* body_dofadr_() const { return LI() + ka->mo.body_dofadr; }
    AVS_DEV const int* body_dofnum_() const { return LI() + ka->mo.body_dofnum; }
    AVS_DEV const int* body_tree_() const { return LI() + ka->mo.body_tree; }
    AVS_DEV const int* body_dofmask_() const { return LI() + ka->mo.body_dofmask; }
    AVS_DEV const int* body_last_() const { return LI() + ka->mo.body_last; }
    AVS_DEV const int* tree_bodyadr_() const { return LI() + ka->mo.tree_bodyadr; }
    AVS_DEV const int* tree_bodylist_() const { return LI() + ka->mo.tree_bodylist; }
    AVS_DEV const int* tree_dofadr_() const { return LI() + ka->mo.tree_dofadr; }
    AVS_DEV const int* tree_dofnum_() const { return LI() + ka->mo.tree_dofnum; }
    AVS_DEV const int* tree_madr_() const { return LI() + ka->mo.tree_madr; }
    AVS_DEV const int* jnt_type_() const { return LI() + ka->mo.jnt_type; }
    AVS_DEV const int* jnt_qposadr_() const { return LI() + ka->mo.jnt_qposadr; }
    AVS_DEV const int* jnt_dofadr_() const { return LI() + ka->mo.jnt_dofadr; }
    AVS_DEV const int* jnt_actfrclimited_() const { return LI() + ka->mo.jnt_actfrclimited; }
    AVS_DEV const int* limited_jnt_() const { return LI() + ka->mo.limited_jnt; }
    AVS_DEV const int* dof_body_() const { return LI() + ka->mo.dof_body; }
    AVS_DEV const int* dof_parent_() const { return LI() + ka->mo.dof_parent; }
    AVS_DEV const int* dof_tree_() const { return LI() + ka->mo.dof_tree; }
    AVS_DEV const int* dof_jnt_() const { return LI() + ka->mo.dof_jnt; }
    AVS_DEV const int* floss_dof_() const { return LI() + ka->mo.floss_dof; }
    AVS_DEV const int* ment_i_() const { return LI() + ka->mo.ment_i; }
    AVS_DEV const int* ment_j_() const { return LI() + ka->mo.ment_j; }
    AVS_DEV const int* act_dof_() const { return LI() + ka->mo.act_dof; }
    AVS_DEV const int* act_qposadr_() const { return LI() + ka->mo.act_qposadr; }
    AVS_DEV const int* act_ctrllimited_() const { return LI() + ka->mo.act_ctrllimited; }
    AVS_DEV const int* geom_type_() const { return LI() + ka->mo.geom_type; }
    AVS_DEV const int* geom_body_() const { return LI() + ka->mo.geom_body; }
    AVS_DEV const int* geom_static_() const { return LI() + ka->mo.geom_static; }
    AVS_DEV const real* body_pos_() const { return LR() + ka->mo.body_pos; }
    AVS_DEV const real* body_quat_() const { return LR() + ka->mo.body_quat; }
    AVS_DEV const real* body_mass_() const { return LR() + ka->mo.body_mass; }
    AVS_DEV const real* body_ipos_() const { return LR() + ka->mo.body_ipos; }
    AVS_DEV const real* body_inertia_() const { return LR() + ka->mo.body_inertia; }
    AVS_DEV const real* body_invweight0_() const { return LR() + ka->mo.body_invweight0; }
    AVS_DEV const real* jnt_pos_() const { return LR() + ka->mo.jnt_pos; }
    AVS_DEV const real* jnt_axis_() const { return LR() + ka->mo.jnt_axis; }
    AVS_DEV const real* jnt_range_() const { return LR() + ka->mo.jnt_range; }
    AVS_DEV const real* jnt_actfrcrange_() const { return LR() + ka->mo.jnt_actfrcrange; }
    AVS_DEV const real* jnt_margin_() const { return LR() + ka->mo.jnt_margin; }
    AVS_DEV const real* dof_armature_() const { return LR() + ka->mo.dof_armature; }
    AVS_DEV const real* dof_damping_() const { return LR() + ka->mo.dof_damping; }
    AVS_DEV const real* dof_frictionloss_() const { return LR() + ka->mo.dof_frictionloss; }
    AVS_DEV const real* dof_invweight0_() const { return LR() + ka->mo.dof_invweight0; }
    AVS_DEV const real* act_kp_() const { return LR() + ka->mo.act_kp; }
    AVS_DEV const real* act_kv_() const { return LR() + ka->mo.act_kv; }
    AVS_DEV const real* act_gear_() const { return LR() + ka->mo.act_gear; }
    AVS_DEV const real* act_ctrlrange_() const { return LR() + ka->mo.act_ctrlrange; }
    AVS_DEV const real* geom_cpos_() const { return LR() + ka->mo.geom_cpos; }
    AVS_DEV const real* geom_rbound_() const { return LR() + ka->mo.geom_rbound; }


    // ---- P1 ------------------------------------------------------------------------------------
    // ---- per-tree dense linear algebra on the whole wave: lane 8 t + i owns row i of tree t's block (<= 8 trees of <= 8 dofs) ----
    // Cholesky of every tree block at once: L L^T = A (+ add on the diagonal); the multipliers travel by ds_bpermute inside the
    // tree's 8 lanes.  L goes to Ldst in the same n x n layout as A.  Padded lanes carry an identity row.
    __device__ void tree_chol(const real* A, real* Ldst, const real* diag_add, real diag_scale) {
        const int t = lane >> 3, i = lane & 7, gb = lane & ~7;
        const bool act = t < ka->m.ntree;
        const int n = act ? tree_dofnum_()[t] : 0, base = act ? tree_madr_()[t] : 0, a0 = act ? tree_dofadr_()[t] : 0;
        real row[TREE_W];
#pragma unroll
        for (int k = 0; k < TREE_W; k++) row[k] = (i < n && k <= i) ? A[base + i * n + k] : (k == i ? real(1) : real(0));
        if (diag_add && i < n) {
#pragma unroll
            for (int k = 0; k < TREE_W; k++) if (k == i) row[k] += diag_scale * diag_add[a0 + i];
        }
#pragma unroll
        for (int j = 0; j < TREE_W; j++) {
            const real d = sqrt(tmax(oct_bcast_n(row[j], j), real(1e-30)));
            const real lij = i == j ? d : row[j] / d;
            row[j] = lij;
            const real mul = i > j ? lij : real(0);
#pragma unroll
            for (int k = j + 1; k < TREE_W; k++) row[k] -= mul * oct_bcast_n(lij, k);
        }
        if (i < n) {
#pragma unroll
            for (int k = 0; k < TREE_W; k++) if (k <= i) Ldst[base + i * n + k] = row[k];
        }
    }
    // solves L L^T x = b for every tree: lane 8 t + i passes b_i and gets x_i (L read from LDS)
    __device__ real tree_solve(const real* L, real x) {
        const int t = lane >> 3, i = lane & 7, gb = lane & ~7;
        const bool act = t < ka->m.ntree;
        const int n = act ? tree_dofnum_()[t] : 0, base = act ? tree_madr_()[t] : 0;
        real row[TREE_W], col[TREE_W];
#pragma unroll
        for (int k = 0; k < TREE_W; k++) {
            row[k] = (i < n && k < i) ? L[base + i * n + k] : real(0);
            col[k] = (k < n && k > i) ? L[base + k * n + i] : real(0);
        }
        const real dinv = i < n ? real(1) / L[base + i * n + i] : real(1);
        if (!(i < n)) x = 0;
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
        return x;
    }

    // P1.  One body per lane: every lane builds its body's transform relative to the parent from its joint coordinate, then
    // the chains are multiplied out by pointer jumping (3 rounds for the 8-deep arms) instead of one lane walking each chain.
    // cdof (one dof per lane), inertial frames and geom centres follow from the world poses in parallel.
    __device__ void kinematics() {
        PHASE_BEGIN();
        real *xpos = r + ka->lay.xpos, *xmat = r + ka->lay.xmat, *xipos = r + ka->lay.xipos, *cdof = r + ka->lay.cdof, *qpos = r + ka->lay.qpos;
        int* anc = ii + ka->lay.cand;        // ancestor still to be folded in (-1: the pose is final); collide's list is idle here
        const int nb = ka->m.nbody;
        // local transforms; children of world-welded bodies and free bodies are already in world coordinates
        for (int b = lane; b < nb; b += G) {
            real R[9], pos[3];
            int a = -1;
            if (body_tree_()[b] < 0) {
                for (int k = 0; k < 3; k++) pos[k] = ka->m.static_xpos[3 * b + k];
                for (int k = 0; k < 9; k++) R[k] = ka->m.static_xmat[9 * b + k];
            } else {
                const int p = body_parent_()[b], ja = body_jntadr_()[b], jn = body_jntnum_()[b];
                const int jt = jn > 0 ? jnt_type_()[ja] : -1;
                if (jt == J_FREE) {
                    const int qa = jnt_qposadr_()[ja];
                    real quat[4];
                    for (int k = 0; k < 3; k++) pos[k] = qpos[qa + k];
                    for (int k = 0; k < 4; k++) quat[k] = qpos[qa + 3 + k];
                    quatnorm(quat);
                    quat2mat(quat, R);
                } else {
                    real Rl[9], bq[4] = {body_quat_()[4 * b], body_quat_()[4 * b + 1], body_quat_()[4 * b + 2], body_quat_()[4 * b + 3]};
                    quat2mat(bq, Rl);
                    for (int k = 0; k < 3; k++) pos[k] = body_pos_()[3 * b + k];
                    for (int k = 0; k < 9; k++) R[k] = Rl[k];
                    if (jt == J_HINGE) {
                        const real q = qpos[jnt_qposadr_()[ja]];
                        const real ax[3] = {jnt_axis_()[3 * ja], jnt_axis_()[3 * ja + 1], jnt_axis_()[3 * ja + 2]};
                        const real jp[3] = {jnt_pos_()[3 * ja], jnt_pos_()[3 * ja + 1], jnt_pos_()[3 * ja + 2]};
                        const real sn = sin(q), co = cos(q), oc = 1 - co;
                        const real Rq[9] = {co + ax[0] * ax[0] * oc, ax[0] * ax[1] * oc - ax[2] * sn, ax[0] * ax[2] * oc + ax[1] * sn,
                                            ax[1] * ax[0] * oc + ax[2] * sn, co + ax[1] * ax[1] * oc, ax[1] * ax[2] * oc - ax[0] * sn,
                                            ax[2] * ax[0] * oc - ax[1] * sn, ax[2] * ax[1] * oc + ax[0] * sn, co + ax[2] * ax[2] * oc};
                        for (int i = 0; i < 3; i++)
                            for (int j = 0; j < 3; j++) R[3 * i + j] = Rl[3 * i] * Rq[j] + Rl[3 * i + 1] * Rq[3 + j] + Rl[3 * i + 2] * Rq[6 + j];
                        // the joint anchor stays put: pos = body_pos + Rl jp - (Rl Rq) jp
                        real t0[3], t1[3];
                        mulmat(Rl, jp, t0);
                        mulmat(R, jp, t1);
                        for (int k = 0; k < 3; k++) pos[k] += t0[k] - t1[k];
                    } else if (jt == J_SLIDE) {
                        const real q = qpos[jnt_qposadr_()[ja]];
                        const real ax[3] = {jnt_axis_()[3 * ja], jnt_axis_()[3 * ja + 1], jnt_axis_()[3 * ja + 2]};
                        real t0[3];
                        mulmat(Rl, ax, t0);
                        for (int k = 0; k < 3; k++) pos[k] += t0[k] * q;
                    }
                    if (body_tree_()[p] < 0) {      // parent welded to the world: fold its constant pose in right away
                        GLB_PTR(const real) Rp = ka->m.static_xmat + 9 * p;
                        real Rn[9], t0[3];
                        for (int i = 0; i < 3; i++)
                            for (int j = 0; j < 3; j++) Rn[3 * i + j] = Rp[3 * i] * R[j] + Rp[3 * i + 1] * R[3 + j] + Rp[3 * i + 2] * R[6 + j];
                        mulmat(Rp, pos, t0);
                        for (int k = 0; k < 3; k++) pos[k] = ka->m.static_xpos[3 * p + k] + t0[k];
                        for (int k = 0; k < 9; k++) R[k] = Rn[k];
                    } else a = p;
                }
            }
            for (int k = 0; k < 3; k++) xpos[3 * b + k] = pos[k];
            for (int k = 0; k < 9; k++) xmat[9 * b + k] = R[k];
            anc[b] = a;
        }
        GSYNC();
        // pointer jumping: T_b <- T_anc(b) T_b, anc(b) <- anc(anc(b)), until every chain reaches its root
        for (int round = 0; round < 6; round++) {
            bool any = false;
            real Ra[9], pa[3];
            int a = -1, aa = -1;
            const int b = lane;
            if (b < nb) {
                a = anc[b];
                if (a >= 0) {
                    any = true;
                    aa = anc[a];
                    for (int k = 0; k < 9; k++) Ra[k] = xmat[9 * a + k];
                    for (int k = 0; k < 3; k++) pa[k] = xpos[3 * a + k];
                }
            }
            if (!__any(any)) break;
            GSYNC();
            if (a >= 0) {
                real R[9], pos[3], Rn[9], t0[3];
                for (int k = 0; k < 9; k++) R[k] = xmat[9 * b + k];
                for (int k = 0; k < 3; k++) pos[k] = xpos[3 * b + k];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) Rn[3 * i + j] = Ra[3 * i] * R[j] + Ra[3 * i + 1] * R[3 + j] + Ra[3 * i + 2] * R[6 + j];
                mulmat(Ra, pos, t0);
                for (int k = 0; k < 3; k++) xpos[3 * b + k] = pa[k] + t0[k];
                for (int k = 0; k < 9; k++) xmat[9 * b + k] = Rn[k];
                anc[b] = aa;
            }
            GSYNC();
        }
        // inertial frames (one body per lane), cdof (one dof per lane), geom centres (one geom per lane)
        for (int b = lane; b < nb; b += G) {
            real t3[3] = {0, 0, 0};
            if (body_tree_()[b] >= 0) {
                const real ip[3] = {body_ipos_()[3 * b], body_ipos_()[3 * b + 1], body_ipos_()[3 * b + 2]};
                mulmat(xmat + 9 * b, ip, t3);
                for (int k = 0; k < 3; k++) t3[k] += xpos[3 * b + k];
            }
            for (int k = 0; k < 3; k++) xipos[3 * b + k] = t3[k];
        }
        for (int d = lane; d < ka->m.nv; d += G) {
            const int j = dof_jnt_()[d], b = dof_body_()[d], jt = jnt_type_()[j], k = d - jnt_dofadr_()[j];
            const real *R = xmat + 9 * b, *pos = xpos + 3 * b;
            real cd[6] = {0, 0, 0, 0, 0, 0};
            if (jt == J_FREE) {
                if (k < 3) cd[3 + k] = 1;
                else {
                    const real w[3] = {R[k - 3], R[3 + k - 3], R[6 + k - 3]};
                    real c[3];
                    cross3(pos, w, c);
                    for (int q = 0; q < 3; q++) { cd[q] = w[q]; cd[3 + q] = c[q]; }
                }
            } else {
                const real ax[3] = {jnt_axis_()[3 * j], jnt_axis_()[3 * j + 1], jnt_axis_()[3 * j + 2]};
                real axis[3];
                mulmat(R, ax, axis);
                if (jt == J_HINGE) {
                    const real jp[3] = {jnt_pos_()[3 * j], jnt_pos_()[3 * j + 1], jnt_pos_()[3 * j + 2]};
                    real anchor[3], c[3];
                    mulmat(R, jp, anchor);
                    for (int q = 0; q < 3; q++) anchor[q] += pos[q];
                    cross3(anchor, axis, c);
                    for (int q = 0; q < 3; q++) { cd[q] = axis[q]; cd[3 + q] = c[q]; }
                } else {
                    for (int q = 0; q < 3; q++) cd[3 + q] = axis[q];
                }
            }
            for (int q = 0; q < 6; q++) cdof[6 * d + q] = cd[q];
        }
        real* gcen = r + ka->lay.gcen;
        for (int g = lane; g < ka->m.ngeom; g += G) {
            if (geom_static_()[g]) continue;
            int b = geom_body_()[g];
            real c[3] = {geom_cpos_()[3 * g], geom_cpos_()[3 * g + 1], geom_cpos_()[3 * g + 2]}, t3[3];
            mulmat(xmat + 9 * b, c, t3);
            for (int k = 0; k < 3; k++) gcen[3 * g + k] = xpos[3 * b + k] + t3[k];
        }
        GSYNC();
    }

    // ---- P2 ------------------------------------------------------------------------------------
    __device__ void crb() {
        PHASE_BEGIN();
        real *xmat = r + ka->lay.xmat, *xipos = r + ka->lay.xipos, *cdof = r + ka->lay.cdof, *ci = r + ka->lay.cinert, *M = r + ka->lay.M, *L = r + ka->lay.L;
        for (int b = lane; b < ka->m.nbody; b += G) {
            SInert<real> s;
            if (body_tree_()[b] >= 0) body_inertia(ka, xmat, xipos, b, s);
            else { s.m = 0; for (int k = 0; k < 3; k++) s.h[k] = 0; for (int k = 0; k < 6; k++) s.I[k] = 0; }
            real *o = ci + 10 * b, *ob = r + ka->lay.binert + 10 * b;    // the second copy survives the composite sums (rne_bias reads it)
            o[0] = ob[0] = s.m;
            for (int k = 0; k < 3; k++) o[1 + k] = ob[1 + k] = s.h[k];
            for (int k = 0; k < 6; k++) o[4 + k] = ob[4 + k] = s.I[k];
        }
        for (int i = lane; i < ka->m.msize; i += G) M[i] = 0;
        GSYNC();
        // composite inertias: bodies are numbered depth first, so the subtree of body p is the id range p .. body_last[p]
        {
            real acc[10];
            const int p = lane < ka->m.nbody ? lane : 0;
            const bool dyn = lane < ka->m.nbody && body_tree_()[p] >= 0;
            for (int k = 0; k < 10; k++) acc[k] = 0;
            if (dyn)
                for (int d = p; d <= body_last_()[p]; d++)
                    for (int k = 0; k < 10; k++) acc[k] += ci[10 * d + k];
            GSYNC();
            if (dyn)
                for (int k = 0; k < 10; k++) ci[10 * p + k] = acc[k];
        }
        GSYNC();
        for (int e = lane; e < ka->m.nment; e += G) {
            int i = ment_i_()[e], j = ment_j_()[e], t = dof_tree_()[i], n = tree_dofnum_()[t], a = tree_dofadr_()[t];
            real f[6];
            inert_mul(ci + 10 * dof_body_()[i], cdof + 6 * i, f);
            real v = 0;
            for (int k = 0; k < 6; k++) v += cdof[6 * j + k] * f[k];
            if (i == j) v += dof_armature_()[i];
            real* Mb = M + tree_madr_()[t];
            Mb[(i - a) * n + (j - a)] = v;
            Mb[(j - a) * n + (i - a)] = v;
        }
        GSYNC();
        tree_chol(M, L, (const real*)nullptr, real(0));
        GSYNC();
        // dense per-tree inverse (8x8, zero padded): B = J M^-1 is formed on the fly from it, so rows store only J
        real* Minv = r + ka->lay.Minv;
        for (int w = lane; w < ka->m.ntree * TREE_W; w += G) {
            int t = w >> 3, j = w & 7, n = tree_dofnum_()[t];
            real x[TREE_W];
            const real* Lt = L + tree_madr_()[t];
#pragma unroll
            for (int k = 0; k < TREE_W; k++) x[k] = (k == j && j < n) ? real(1) : real(0);
            // triangular solves unrolled to 8 with predicates: x stays in registers
#pragma unroll
            for (int i = 0; i < TREE_W; i++)
                if (i < n) {
                    real sacc = x[i];
#pragma unroll
                    for (int k = 0; k < i; k++) sacc -= Lt[i * n + k] * x[k];
                    x[i] = sacc / Lt[i * n + i];
                }
#pragma unroll
            for (int i = TREE_W - 1; i >= 0; i--)
                if (i < n) {
                    real sacc = x[i];
#pragma unroll
                    for (int k = i + 1; k < TREE_W; k++) if (k < n) sacc -= Lt[k * n + i] * x[k];
                    x[i] = sacc / Lt[i * n + i];
                }
#pragma unroll
            for (int k = 0; k < TREE_W; k++) Minv[64 * t + 8 * k + j] = (k < n && j < n) ? x[k] : real(0);
        }
        GSYNC();
    }

    // ---- P5 bias -------------------------------------------------------------------------------
    // Recursive Newton-Euler without the recursion: in world coordinates about the origin the body velocity is the plain sum of
    // cdof_j qvel_j over the ancestor dofs (body_dofmask), the bias acceleration the sum of cdof_dot_j qvel_j over the same
    // dofs, and the joint torque the projection of the inertial forces summed over the subtree (a contiguous id range).
    __device__ void rne_bias() {
        PHASE_BEGIN();
        real *xmat = r + ka->lay.xmat, *xipos = r + ka->lay.xipos, *cdof = r + ka->lay.cdof, *qvel = r + ka->lay.qvel;
        real *cvel = r + ka->lay.cvel, *cfrc = r + ka->lay.cfrc, *bias = r + ka->lay.bias;
        real* cdd = r + ka->lay.cinert;     // cdof_dot_j qvel_j per dof (the composite inertias are no longer needed)
        const int nb = ka->m.nbody;
        // body velocities
        for (int b = lane; b < nb; b += G) {
            real v[6] = {0, 0, 0, 0, 0, 0};
            const int t = body_tree_()[b];
            if (t >= 0) {
                const int a0 = tree_dofadr_()[t], n = tree_dofnum_()[t], mask = body_dofmask_()[b];
                // all eight slots unconditionally (reads of an absent slot fall back on the tree's first dof, weight 0): the LDS
                // reads batch up instead of one branch + wait per ancestor dof
#pragma unroll
                for (int k = 0; k < TREE_W; k++) {
                    const bool on = k < n && ((mask >> k) & 1);
                    const int d = on ? a0 + k : a0;
                    const real qd = on ? qvel[d] : real(0);
#pragma unroll
                    for (int q = 0; q < 6; q++) v[q] += cdof[6 * d + q] * qd;
                }
            }
            for (int q = 0; q < 6; q++) cvel[6 * b + q] = v[q];
        }
        GSYNC();
        // cdof_dot_j qvel_j: v x cdof_j with v the body's velocity (hinge / slide; the joint's own term drops out of the cross
        // product), or the translational part of a free body's velocity for its three rotational dofs
        for (int d = lane; d < ka->m.nv; d += G) {
            const int j = dof_jnt_()[d], b = dof_body_()[d];
            real o[6] = {0, 0, 0, 0, 0, 0};
            if (jnt_type_()[j] == J_FREE) {
                const int d0 = jnt_dofadr_()[j];
                if (d - d0 >= 3) {
                    real vt[6] = {0, 0, 0, 0, 0, 0};
                    for (int k = 0; k < 3; k++)
                        for (int q = 0; q < 6; q++) vt[q] += cdof[6 * (d0 + k) + q] * qvel[d0 + k];
                    cross_motion(vt, cdof + 6 * d, o);
                }
            } else {
                cross_motion(cvel + 6 * b, cdof + 6 * d, o);
            }
            const real qd = qvel[d];
            for (int q = 0; q < 6; q++) cdd[6 * d + q] = o[q] * qd;
        }
        GSYNC();
        // inertial force of every body: I a + v x* (I v)
        for (int b = lane; b < nb; b += G) {
            const int t = body_tree_()[b];
            real f[6] = {0, 0, 0, 0, 0, 0};
            if (t >= 0) {
                real a[6] = {0, 0, 0, -ka->m.gravity[0], -ka->m.gravity[1], -ka->m.gravity[2]}, v[6];
                const int a0 = tree_dofadr_()[t], n = tree_dofnum_()[t], mask = body_dofmask_()[b];
#pragma unroll
                for (int k = 0; k < TREE_W; k++) {
                    const bool on = k < n && ((mask >> k) & 1);
                    const int d = on ? a0 + k : a0;
#pragma unroll
                    for (int q = 0; q < 6; q++) { const real c = cdd[6 * d + q]; a[q] += on ? c : real(0); }
                }
                for (int q = 0; q < 6; q++) v[q] = cvel[6 * b + q];
                real sv[10];      // the body's spatial inertia about the world origin, left by crb
                for (int q = 0; q < 10; q++) sv[q] = (r + ka->lay.binert)[10 * b + q];
                real Ia[6], Iv[6], vIv[6];
                inert_mul(sv, a, Ia);
                inert_mul(sv, v, Iv);
                cross_force(v, Iv, vIv);
                for (int q = 0; q < 6; q++) f[q] = Ia[q] + vIv[q];
            }
            for (int q = 0; q < 6; q++) cfrc[6 * b + q] = f[q];
        }
        GSYNC();
        // joint torques: cdof_i . (forces of the subtree of dof i's body)
        for (int i = lane; i < ka->m.nv; i += G) {
            const int b = dof_body_()[i];
            real f[6] = {0, 0, 0, 0, 0, 0};
            for (int d = b; d <= body_last_()[b]; d++)
                for (int q = 0; q < 6; q++) f[q] += cfrc[6 * d + q];
            real sacc = 0;
            for (int q = 0; q < 6; q++) sacc += cdof[6 * i + q] * f[q];
            bias[i] = sacc;
        }
        GSYNC();
    }

    // ---- P5 passive + P6 actuation + P7 smooth acceleration ---------------------------------------
    __device__ void smooth() {
        PHASE_BEGIN();
        real *qpos = r + ka->lay.qpos, *qvel = r + ka->lay.qvel, *ctrl = r + ka->lay.ctrl, *bias = r + ka->lay.bias, *fsm = r + ka->lay.fsm, *as = r + ka->lay.asm_;
        real* act = r + ka->lay.fcon;  // reuse as qfrc_actuator until the solve
        for (int i = lane; i < ka->m.nv; i += G) act[i] = 0;
        GSYNC();
        for (int u = lane; u < ka->m.nu; u += G) {
            real c = ctrl[u];
            if (act_ctrllimited_()[u]) c = tclamp(c, act_ctrlrange_()[2 * u], act_ctrlrange_()[2 * u + 1]);
            int dof = act_dof_()[u];
            real f = act_kp_()[u] * c - act_kp_()[u] * qpos[act_qposadr_()[u]] - act_kv_()[u] * qvel[dof];
            act[dof] += act_gear_()[u] * f;   // one actuator per dof in these models
        }
        GSYNC();
        for (int i = lane; i < ka->m.nv; i += G) {
            int j = dof_jnt_()[i];
            real a = act[i];
            if (jnt_actfrclimited_()[j] && jnt_type_()[j] != J_FREE) a = tclamp(a, jnt_actfrcrange_()[2 * j], jnt_actfrcrange_()[2 * j + 1]);
            real f = -dof_damping_()[i] * qvel[i] - bias[i] + a;
            fsm[i] = f;
            as[i] = f;
        }
        GSYNC();
        {
            const int t = lane >> 3, i = lane & 7;
            const bool mine = t < ka->m.ntree && i < tree_dofnum_()[t];
            const int k = mine ? tree_dofadr_()[t] + i : 0;
            const real x = tree_solve(r + ka->lay.L, mine ? as[k] : real(0));
            GSYNC();
            if (mine) as[k] = x;
        }
        GSYNC();
    }

    __device__ void load_shape(int g, Shape<real>& s) {
        LDS_BASES();
        real *xpos = r + ka->lay.xpos, *xmat = r + ka->lay.xmat, *gcen = r + ka->lay.gcen;
        s.type = geom_type_()[g];
        for (int k = 0; k < 3; k++) s.size[k] = ka->m.geom_size[3 * g + k];
        s.hull = ka->m.hull_vert;
        s.hbase = ka->m.geom_hull[2 * g];
        s.hR = ka->m.geom_hull[2 * g + 1];
        s.hovf = ka->m.hull_ovf;
        s.nh = 0;
        for (int k = 0; k < 3; k++) { s.lc[k] = ka->m.geom_lbox[6 * g + k]; s.lh[k] = ka->m.geom_lbox[6 * g + 3 + k]; }
        if (geom_static_()[g]) {
            for (int k = 0; k < 3; k++) { s.pos[k] = ka->m.geom_xpos0[3 * g + k]; s.center[k] = ka->m.geom_cen0[3 * g + k]; }
            for (int k = 0; k < 9; k++) s.mat[k] = ka->m.geom_xmat0[9 * g + k];
        } else {
            int b = geom_body_()[g];
            real gp[3] = {ka->m.geom_pos[3 * g], ka->m.geom_pos[3 * g + 1], ka->m.geom_pos[3 * g + 2]}, t3[3];
            mulmat(xmat + 9 * b, gp, t3);
            for (int k = 0; k < 3; k++) { s.pos[k] = xpos[3 * b + k] + t3[k]; s.center[k] = gcen[3 * g + k]; }
            const real* Rb = xmat + 9 * b;
            GLB_PTR(const real) Rg = ka->m.geom_mat + 9 * g;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) s.mat[3 * i + j] = Rb[3 * i] * Rg[j] + Rb[3 * i + 1] * Rg[3 + j] + Rb[3 * i + 2] * Rg[6 + j];
        }
    }

    // ---- P3 ------------------------------------------------------------------------------------
    // out of line, on a copy of the object: the members then live in registers (a callee reached through `this` reloads them
    // from memory after every store, and the kernel-argument pointer with them: vector loads instead of s_load)
    // One multiccd perturbation of the pair (pga, pgb) whose first contact sits in result slot `src`; leaves distance and position in
    // entry 1 + pert of that slot (1e30: none).  Out of line with register arguments only: the MPR code
    // is 27 KB, and as a call of mpr_perturbed itself the two Shapes went through the wave's private segment -- 79 dwords per lane,
    // 20 KB written and read back per narrow-phase pass, most of the kernel's spill traffic (profiles/r05_experiments.txt section 7).
    __device__ AVS_OUTLINE_0 static void multi_perturb(KPtr<real> ka_, real* r_, int* i_, int lane_, int grp_, const real* lr_, const int* li_, int env_,
                                                                   int pga, int pgb, int src, int pert) {
        Env e(ka_, r_, i_, lane_, grp_, lr_, li_);       // (an Env by value is an aggregate of 22 dwords: passed in memory)
        e.env = env_;
        LDS_PTR(real) os = (LDS_PTR(real))(e.r + e.ka->lay.scr + SLOT_W * src);
        const real p0[3] = {os[SLOT_P], os[SLOT_P + 1], os[SLOT_P + 2]}, n0[3] = {os[SLOT_N], os[SLOT_N + 1], os[SLOT_N + 2]};
        Shape<real> a, b;
        e.load_shape(pga, a);
        e.load_shape(pgb, b);
        real dk = real(1e30), pk[3] = {0, 0, 0};
        if (!mpr_perturbed(a, b, p0, n0, pert, &dk, pk)) dk = real(1e30);
        os[1 + pert] = dk;
        for (int c = 0; c < 3; c++) os[SLOT_P + 3 * (1 + pert) + c] = pk[c];
    }

    __device__ AVS_OUTLINE void collide() {
        Env e(*this);
        e.collide_i();
        diverged = e.diverged; nit_sum = e.nit_sum; nit_max = e.nit_max; t_broad = e.t_broad; t_narrow = e.t_narrow;
    }
    // the call goes through a throw-away copy: the caller's object never has its address taken and stays in registers too
    __device__ __attribute__((always_inline)) void collide_i() {
        PHASE_BEGIN();
        real* gcen = r + ka->lay.gcen;
        int* misc = ii + ka->lay.misc;
        GLB_PTR(int) cand = cand_();
        int ncand = 0;
        long long tb0 = __builtin_readcyclecounter();
        const real skin = real(0.03);
        GLB_PTR(real) gref = gref_();
        GLB_PTR(int) nearl = near_();
        // pair test with extra reach `pad` (0 = exact broad phase)
        auto pair_hit = [&](int p, real pad) -> bool {
            int g1 = ka->m.pair_geom[2 * p], g2 = ka->m.pair_geom[2 * p + 1];
            bool s1 = geom_static_()[g1], s2 = geom_static_()[g2];
            real mg = ka->m.pair_margin[p] + pad;
            if (s1 || s2) {
                // dynamic bounding sphere against the world AABB of the static geom
                int gs = s1 ? g1 : g2, gd = s1 ? g2 : g1;
                real rd = geom_rbound_()[gd] + mg, d2 = 0;
                for (int k = 0; k < 3; k++) {
                    real c = gcen[3 * gd + k], lo = ka->m.geom_aabb0[6 * gs + k], hi = ka->m.geom_aabb0[6 * gs + 3 + k];
                    real e = c < lo ? lo - c : (c > hi ? c - hi : real(0));
                    d2 += e * e;
                }
                return !(d2 > rd * rd);
            }
            real d[3], rr = geom_rbound_()[g1] + geom_rbound_()[g2] + mg;
            sub3(gcen + 3 * g2, gcen + 3 * g1, d);
            return !(dot3(d, d) > rr * rr);
        };
        // Verlet neighbour list: valid while no dynamic geom centre moved more than skin/2 since it was built
        bool moved = false;
        if (misc[7]) {
            for (int g = lane; g < ka->m.ngeom; g += G)
                if (!geom_static_()[g]) {
                    const real d[3] = {gcen[3 * g] - gref[3 * g], gcen[3 * g + 1] - gref[3 * g + 1], gcen[3 * g + 2] - gref[3 * g + 2]};
                    moved |= dot3(d, d) > real(0.25) * skin * skin;
                }
        }
        bool rebuild = !misc[7] || __any(moved);
        int nnear = misc[6];
        if (rebuild) {
            nnear = 0;
            for (int base = 0; base < ka->m.npair; base += G) {
                int p = base + lane;
                bool hit = p < ka->m.npair && pair_hit(p, skin);
                int tot, rk = group_rank<G>(hit, grp, lane, &tot);
                if (hit && nnear + rk < NEAR_MAX) nearl[nnear + rk] = p;
                nnear += tot;
            }
            for (int i = lane; i < 3 * ka->m.ngeom; i += G) gref[i] = gcen[i];
            if (lane == 0) { misc[6] = nnear < NEAR_MAX ? nnear : NEAR_MAX; misc[7] = nnear <= NEAR_MAX; if (nnear > NEAR_MAX) misc[2] |= 8; }
            GSYNC();
        }
        if (nnear <= NEAR_MAX) {
            for (int base = 0; base < nnear; base += G) {
                int i = base + lane, p = i < nnear ? nearl[i] : 0;
                bool hit = i < nnear && pair_hit(p, real(0));
                int tot, rk = group_rank<G>(hit, grp, lane, &tot);
                if (hit && ncand + rk < CAND_MAX) cand[ncand + rk] = p;
                if (ncand + tot > CAND_MAX && lane == 0) misc[2] |= 4;
                ncand = ncand + tot < CAND_MAX ? ncand + tot : CAND_MAX;
            }
        } else {
            // neighbour list overflow: exact test over the whole compiled pair list
            for (int base = 0; base < ka->m.npair; base += G) {
                int p = base + lane;
                bool hit = p < ka->m.npair && pair_hit(p, real(0));
                int tot, rk = group_rank<G>(hit, grp, lane, &tot);
                if (hit && ncand + rk < CAND_MAX) cand[ncand + rk] = p;
                if (ncand + tot > CAND_MAX && lane == 0) misc[2] |= 4;
                ncand = ncand + tot < CAND_MAX ? ncand + tot : CAND_MAX;
            }
        }
        if (lane == 0) misc[3] = ncand;
        GSYNC();
        long long tb1 = __builtin_readcyclecounter();
        t_broad += tb1 - tb0;
        real *cdist = r + ka->lay.cdist, *cpos = r + ka->lay.cpos, *cnrm = r + ka->lay.cnrm;
        int* cpair = ii + ka->lay.cpair;
        int ncon = 0, ovf = 0;
        // 64 candidate pairs per pass, one per lane, each with a result slot in LDS (the solver records, not live during
        // collision).  Box-box pairs go first, four at a time through the polygon work areas behind the slots, then everything
        // else, then the multiccd perturbations of the convex pairs found in contact: the wave executes the clipping code and the
        // MPR code once each instead of both in every pass.
        int* mlist = ii + ka->lay.cand;      // lanes of the pairs that get the multiccd treatment (the kinematics' table is idle here)
        GLB_PTR(real) bxo = bxo_();          // points 4..7 of the box-box manifolds of this pass, one record per lane
        for (int base = 0; base < ncand; base += G) {
            const int ci = base + lane;
            int n = 0, p = 0, nn = 0;
            LDS_PTR(real) scr = (LDS_PTR(real))(r + ka->lay.scr + SLOT_W * lane);
            const bool valid = ci < ncand;
            if (valid) p = cand[ci];
            const int ga = ka->m.pair_geom[2 * p], gb = ka->m.pair_geom[2 * p + 1];
            const int tya = geom_type_()[ga], tyb = geom_type_()[gb];
            const bool isbox = valid && tya == G_BOX && tyb == G_BOX;
            const long long tn0_ = profiling ? __builtin_readcyclecounter() : 0;
            {
                // box-box pairs, four at a time: every 16-lane row of the wave works on one pair (box_box16), the results land in
                // the result slot of the lane that owns the pair
                // the conservative cull first, one pair per lane: only boxes whose local bounding boxes overlap take a 16-lane row of a
                // pass (resting scenes have more box pairs in reach of each other than in contact: one pass of four instead of two)
                bool boxlive = false;
                if (isbox) {
                    Shape<real> a, b;
                    load_shape(ga, a);
                    load_shape(gb, b);
                    boxlive = !boxes_separated(a, b);
                }
                int nbox, rk = group_rank<G>(boxlive, grp, lane, &nbox);
                const int row = lane >> 4;
                for (int b0 = 0; b0 < nbox; b0 += 4) {
                    int src = 0;
                    bool on = false;
#pragma unroll
                    for (int gg = 0; gg < 4; gg++) {
                        const unsigned long long mm = __ballot(boxlive && rk == b0 + gg);
                        if (row == gg && mm != 0) { src = __builtin_ctzll(mm); on = true; }
                    }
                    const int pga = __shfl(ga, src, 64), pgb = __shfl(gb, src, 64);     // an idle row runs on lane 0's pair, unused
                    Shape<real> a, b;
                    load_shape(pga, a);
                    load_shape(pgb, b);
                    const int n16 = box_box16(a, b, (LDS_PTR(real))(r + ka->lay.scr + SLOT_W * src), (LDS_PTR(real))(r + ka->lay.scr + SLOT_W * G + 56 * row), bxo + BOX_OVF_W * src, lane, on);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int got = __shfl(on ? n16 : 0, 16 * ((rk - b0) & 3), 64);
                    if (boxlive && rk >= b0 && rk < b0 + 4) nn = got;
                }
            }
            const long long tn1 = profiling ? __builtin_readcyclecounter() : 0;
            if (valid && !isbox) {
                Shape<real> a, b;
                load_shape(ga, a);
                load_shape(gb, b);
                nn = narrow(a, b, scr);
            }
            if (profiling) {
                const long long tn2 = __builtin_readcyclecounter();
                if (lane == 0) { (ii + ka->lay.nprof)[14] += (int)(tn1 - tn0_); (ii + ka->lay.nprof)[15] += (int)(tn2 - tn1); }
            }
            {
                // multiccd [EXT: mjc_Convex]: the convex (MPR) pairs found in contact, spheres excluded.  Lane 4 q + k runs perturbation
                // k of the q-th such pair (16 pairs per round) and leaves its contact in entry 1 + k of the owner's slot (distance
                // 1e30: none); the owner then keeps, in the order k = 0..3, those that are distinct from the ones kept before.
                const bool multi = valid && !isbox && nn == 1 && tya != G_SPHERE && tyb != G_SPHERE;
                int nm, mrk = group_rank<G>(multi, grp, lane, &nm);
                if (nm > 0) {
                    if (multi) mlist[mrk] = lane;
                    GSYNC();
                    for (int m0 = 0; m0 < nm; m0 += 16) {
                        const int q = m0 + (lane >> 2), pert = lane & 3;
                        const bool on = q < nm;
                        const int src = on ? mlist[q] : 0;
                        const int pga = __shfl(ga, src, 64), pgb = __shfl(gb, src, 64);
                        if (on) multi_perturb(ka, r, ii, lane, grp, lr, li, env, pga, pgb, src, pert);
                    }
                    GSYNC();
                    if (multi) {
                        const real rb1 = geom_rbound_()[ga], rb2 = geom_rbound_()[gb];
                        const real tol = MultiCcd<real>::reltol * (rb1 < rb2 ? rb1 : rb2);
                        int kept = 1;
                        for (int k = 1; k <= 4; k++) {
                            const real dk = scr[k];
                            const real pk[3] = {scr[SLOT_P + 3 * k], scr[SLOT_P + 3 * k + 1], scr[SLOT_P + 3 * k + 2]};
                            if (dk < real(1e29) && multiccd_distinct(pk, scr + SLOT_P, kept, tol)) {
                                scr[kept] = dk;
                                for (int c = 0; c < 3; c++) scr[SLOT_P + 3 * kept + c] = pk[c];
                                kept++;
                            }
                        }
                        nn = kept;
                    }
                }
            }
            int keepmask = 0;
            GLB_PTR(const real) ov = bxo + BOX_OVF_W * lane;
            if (valid) {
                // drop separated points (margin = 0 here) while keeping order; a box pair's points 4..7 sit in its overflow record
                real mg = ka->m.pair_margin[p];
                for (int k = 0; k < nn; k++)
                    if ((isbox && k >= BOX_SLOTC ? ov[4 * (k - BOX_SLOTC)] : scr[k]) < mg) { keepmask |= 1 << k; n++; }
            }
            int off = 0, tot = 0;
            for (int j = 1; j <= BOX_MAXC; j++) {
                int tj, rj = group_rank<G>(n >= j, grp, lane, &tj);
                if (tj == 0) break;                     // (the env's lanes agree: no pair of this pass has j contacts)
                off += rj;
                tot += tj;
            }
            int w = 0;
            for (int k = 0; k < BOX_MAXC; k++)
                if ((keepmask >> k) & 1) {
                    int c = ncon + off + w;
                    w++;
                    if (c < ka->lay.maxcon) {
                        const bool o = isbox && k >= BOX_SLOTC;
                        cdist[c] = o ? ov[4 * (k - BOX_SLOTC)] : scr[k];
                        cpair[c] = p;
                        for (int q = 0; q < 3; q++) { cpos[3 * c + q] = o ? ov[4 * (k - BOX_SLOTC) + 1 + q] : scr[SLOT_P + 3 * k + q]; cnrm[3 * c + q] = scr[SLOT_N + q]; }
                    }
                }
            if (ncon + tot > ka->lay.maxcon) ovf = 1;
            ncon = ncon + tot < ka->lay.maxcon ? ncon + tot : ka->lay.maxcon;
        }
        if (lane == 0) { misc[0] = ncon; misc[2] |= ovf; }
        GSYNC();
        t_narrow += __builtin_readcyclecounter() - tb1;
    }

    // ---- P4 ------------------------------------------------------------------------------------
#ifndef AVS_NO_SPLIT_PRE
    // the four smooth-dynamics phases and the integration step as out-of-line functions of their own (a register allocation each; measured on one
    // box against the same phases inlined into the kernel's substep loop: config 3 367 -> 375 k env-steps/s, configs 2 / 4 and f64 unchanged)
    __device__ AVS_OUTLINE void pre_phases() {
        Env e(*this);
        e.kinematics(); e.crb(); e.rne_bias(); e.smooth();
    }
    __device__ AVS_OUTLINE void post_phases() {
        Env e(*this);
        e.euler(); e.check_divergence();
        diverged = e.diverged;
    }
#endif
    // out of line, on a copy of the object: the members then live in registers (a callee reached through `this` reloads them
    // from memory after every store, and the kernel-argument pointer with them: vector loads instead of s_load)
    __device__ AVS_OUTLINE void make_constraints() {
        Env e(*this);
        e.make_constraints_i();
        diverged = e.diverged; nit_sum = e.nit_sum; nit_max = e.nit_max; t_broad = e.t_broad; t_narrow = e.t_narrow;
    }
    // the call goes through a throw-away copy: the caller's object never has its address taken and stays in registers too
    __device__ __attribute__((always_inline)) void make_constraints_i() {
        PHASE_BEGIN();
#ifdef AVSIM_PROBE_ROWS
        long long tpr_ = __builtin_readcyclecounter();
#define ROWPROBE(k) do { if (profiling) { const long long t_ = __builtin_readcyclecounter(); if (lane == 0) (ii + ka->lay.nprof)[8 + (k)] += (int)(t_ - tpr_); tpr_ = t_; } } while (0)
#else
#define ROWPROBE(k) ((void)0)
#endif
        real *qpos = r + ka->lay.qpos, *qvel = r + ka->lay.qvel;
        int *misc = ii + ka->lay.misc, *rmeta = ii + ka->lay.rmeta, *cpair = ii + ka->lay.cpair, *cefc = ii + ka->lay.cefc;
        real *cdist = r + ka->lay.cdist, *cpos = r + ka->lay.cpos, *cnrm = r + ka->lay.cnrm;
        int ncon = misc[0];
        // --- row table: meta = type | id<<2 | sub<<12 | dim<<20 (tree ids replace dim once the row is filled) ---
        int nefc = ka->m.neq + ka->m.nfloss;
        for (int i = lane; i < ka->m.neq; i += G) rmeta[i] = R_EQ | (i << 2);
        for (int i = lane; i < ka->m.nfloss; i += G) rmeta[ka->m.neq + i] = R_FLOSS | (i << 2);
        for (int base = 0; base < ka->m.nlimited; base += G) {
            int li = base + lane, lo = 0, hi = 0, j = 0;
            if (li < ka->m.nlimited) {
                j = limited_jnt_()[li];
                real q = qpos[jnt_qposadr_()[j]];
                lo = (q - jnt_range_()[2 * j]) < jnt_margin_()[j];
                hi = (jnt_range_()[2 * j + 1] - q) < jnt_margin_()[j];
            }
            int t1, t2, r1 = group_rank<G>(lo, grp, lane, &t1), r2 = group_rank<G>(hi, grp, lane, &t2);
            int pos = nefc + r1 + r2;   // rows of lower lanes come first; a joint's lower side before its upper side
            if (lo && pos < ka->lay.maxefc) rmeta[pos] = R_LIMIT | (j << 2) | (0 << 12);
            if (hi && pos + lo < ka->lay.maxefc) rmeta[pos + lo] = R_LIMIT | (j << 2) | (1 << 12);
            nefc += t1 + t2;
        }
        int ovf = 0;
        if (nefc > ka->lay.maxefc) { nefc = ka->lay.maxefc; ovf = 1; }
        if (lane == 0) misc[4] = nefc;   // rows before the contacts
        const int nlead0 = nefc;
        int cend = nefc;   // end of the last contact block that fits under the row cap (row offsets are monotonic)
        for (int base = 0; base < ncon; base += G) {
            int c = base + lane, dim = 0;
            if (c < ncon) {
                int p = cpair[c];
                if (cdist[c] < ka->m.pair_margin[p] - ka->m.pair_gap[p]) dim = ka->m.pair_condim[p];
            }
            int off = 0, tot = 0;
            for (int j = 1; j <= 6; j++) {
                int tj, rj = group_rank<G>(dim >= j, grp, lane, &tj);
                off += rj;
                tot += tj;
            }
            int myend = 0;
            if (c < ncon) {
                int first = nefc + off;
                if (dim > 0 && first + dim <= ka->lay.maxefc) {
                    cefc[c] = first | (dim << 16);   // first row | rows
                    myend = first + dim;
                    for (int s = 0; s < dim; s++) rmeta[first + s] = R_CONTACT | (c << 2) | (s << 12) | (dim << 20);
                } else {
                    cefc[c] = -1;
                    if (dim > 0) ovf = 1;
                }
            }
            for (int o = 1; o < G; o <<= 1) { int x = __shfl_xor(myend, o, G); myend = x > myend ? x : myend; int y = __shfl_xor(ovf, o, G); ovf |= y; }
            cend = myend > cend ? myend : cend;
            nefc += tot;
        }
        nefc = cend;
        if (lane == 0) { misc[1] = nefc; if (ovf) misc[2] |= 2; }
        GSYNC();
        ROWPROBE(0);
        // --- fill rows (one row per lane) ---
        GLB_PTR(real) rJ = rows_();
        real *rowS = r + ka->lay.rowS, *warm = r + ka->lay.warm, *Minv = r + ka->lay.Minv, *asm_ = r + ka->lay.asm_;
        const bool newton = ka->m.solver == 1;
        // Newton + noslip with the dry-friction rows relaxed per tree (pgs_groups): nothing reads the J M^-1 rows or the
        // couplings of the leading rows
        const bool lead_slim = newton && lead_per_tree();
        int* rowI = ii + ka->lay.rowI;
        real* Lm = r + ka->lay.L;
        const int *bmask = body_dofmask_(), *tadr = tree_dofadr_(), *tnum = tree_dofnum_();
        const real* cdofp = r + ka->lay.cdof;
        const int nvm = ka->m.nv - 1;
        // Two passes of one body: the leading rows (equality, dry friction, limits: at most two entries, everything they need comes
        // from a handful of LDS words) and the contact rows (dense over the two dof windows).  One loop over all rows would run the
        // dense code for every 64 rows, leading or not.
        auto fill = [&](int i, auto lead_tag) {
            constexpr bool LEAD = decltype(lead_tag)::value;
            int meta = rmeta[i], type = meta & 3, id = (meta >> 2) & 1023, sub = (meta >> 12) & 255, dim = meta >> 20;
            real J[ROW_W];   // kept in registers: every index below is a compile-time constant after unrolling
#pragma unroll
            for (int k = 0; k < ROW_W; k++) J[k] = 0;
            int tA = -1, tB = -1;
            int j1 = -1, j2 = -1;      // slots of the (at most two) non-zero entries of non-contact rows
            real v1 = 0, v2 = 0;
            real pos = 0, margin = 0, diag0 = 0, floss = 0;
            real solref[2], solimp[5];
            bool valid = true;
            if (LEAD && type == R_EQ) {
                GLB_PTR(const real) c = ka->m.eq_polycoef + 5 * id;
                real q1 = qpos[ka->m.eq_qpos1[id]] - ka->m.qpos0[ka->m.eq_qpos1[id]], q2 = qpos[ka->m.eq_qpos2[id]] - ka->m.qpos0[ka->m.eq_qpos2[id]];
                real poly = c[0] + q2 * (c[1] + q2 * (c[2] + q2 * (c[3] + q2 * c[4])));
                real dpoly = c[1] + q2 * (2 * c[2] + q2 * (3 * c[3] + q2 * 4 * c[4]));
                int d1 = ka->m.eq_dof1[id], d2 = ka->m.eq_dof2[id];
                tA = dof_tree_()[d1];
                pos = q1 - poly;
                diag0 = dof_invweight0_()[d1] + dof_invweight0_()[d2];
                j1 = d1 - tree_dofadr_()[tA]; v1 = 1;
                j2 = d2 - tree_dofadr_()[tA]; v2 = -dpoly;   // both joints of a gripper live in the same tree
                for (int k = 0; k < 2; k++) solref[k] = ka->m.eq_solref[2 * id + k];
                for (int k = 0; k < 5; k++) solimp[k] = ka->m.eq_solimp[5 * id + k];
            } else if (LEAD && type == R_FLOSS) {
                int d = floss_dof_()[id];
                tA = dof_tree_()[d];
                diag0 = dof_invweight0_()[d];
                floss = dof_frictionloss_()[d];
                j1 = d - tree_dofadr_()[tA]; v1 = 1;
                for (int k = 0; k < 2; k++) solref[k] = ka->m.dof_solref[2 * d + k];
                for (int k = 0; k < 5; k++) solimp[k] = ka->m.dof_solimp[5 * d + k];
            } else if (LEAD && type == R_LIMIT) {
                int j = id, d = jnt_dofadr_()[j];
                real q = qpos[jnt_qposadr_()[j]];
                tA = dof_tree_()[d];
                pos = sub == 0 ? q - jnt_range_()[2 * j] : jnt_range_()[2 * j + 1] - q;
                margin = jnt_margin_()[j];
                diag0 = dof_invweight0_()[d];
                j1 = d - tree_dofadr_()[tA]; v1 = sub == 0 ? real(1) : real(-1);
                for (int k = 0; k < 2; k++) solref[k] = ka->m.jnt_solref[2 * j + k];
                for (int k = 0; k < 5; k++) solimp[k] = ka->m.jnt_solimp[5 * j + k];
            } else if (!LEAD) {
                int c = id, p = cpair[c];
                int g1 = ka->m.pair_geom[2 * p], g2 = ka->m.pair_geom[2 * p + 1], b1 = geom_body_()[g1], b2 = geom_body_()[g2];
                int t1 = body_tree_()[b1], t2 = body_tree_()[b2];
                tA = t1 >= 0 ? t1 : t2;
                tB = (t1 >= 0 && t2 >= 0 && t2 != t1) ? t2 : -1;
                real n[3] = {cnrm[3 * c], cnrm[3 * c + 1], cnrm[3 * c + 2]}, t1v[3], t2v[3], cp[3] = {cpos[3 * c], cpos[3 * c + 1], cpos[3 * c + 2]};
                make_frame(n, t1v, t2v);
                const int sm = sub % 3;
                real ax[3];
#pragma unroll
                for (int q = 0; q < 3; q++) ax[q] = sm == 0 ? n[q] : (sm == 1 ? t1v[q] : t2v[q]);
                const bool rot = sub >= 3;
                // entry of dof slot k: the axis' component of the dof's motion at the contact point, signed by which of the two bodies
                // the dof moves.  ax . (lin + ang x cp) = ax . lin + (cp x ax) . ang, so the row is one 6-vector W = (cp x ax, ax) --
                // (ax, 0) for a rotational row -- against every slot's cdof: six FMAs per slot, no branch, the body masks as factors
                real cx[3], Wa[3], Wl[3];
                cross3(cp, ax, cx);
#pragma unroll
                for (int q = 0; q < 3; q++) { Wa[q] = rot ? ax[q] : cx[q]; Wl[q] = rot ? real(0) : ax[q]; }
                const int mA = (t1 == tA ? bmask[b1] : 0), pA = (t2 == tA ? bmask[b2] : 0);
                const int mB = (tB >= 0 && t1 == tB ? bmask[b1] : 0), pB = (tB >= 0 ? bmask[b2] : 0);
                const int aA = tadr[tA], aB = tadr[tB >= 0 ? tB : tA];
#pragma unroll
                for (int k = 0; k < TREE_W; k++) {
                    const real* ca = cdofp + 6 * (aA + k < nvm ? aA + k : nvm);
                    const real ea = Wa[0] * ca[0] + Wa[1] * ca[1] + Wa[2] * ca[2] + Wl[0] * ca[3] + Wl[1] * ca[4] + Wl[2] * ca[5];
                    J[k] = real(((pA >> k) & 1) - ((mA >> k) & 1)) * ea;
                }
                if (tB >= 0) {       // second window: contacts between two kinematic trees only
#pragma unroll
                    for (int k = 0; k < TREE_W; k++) {
                        const real* cb = cdofp + 6 * (aB + k < nvm ? aB + k : nvm);
                        const real eb = Wa[0] * cb[0] + Wa[1] * cb[1] + Wa[2] * cb[2] + Wl[0] * cb[3] + Wl[1] * cb[4] + Wl[2] * cb[5];
                        J[TREE_W + k] = real(((pB >> k) & 1) - ((mB >> k) & 1)) * eb;
                    }
                }
                pos = sub == 0 ? cdist[c] : real(0);
                margin = ka->m.pair_margin[p] - ka->m.pair_gap[p];
                diag0 = body_invweight0_()[2 * b1] + body_invweight0_()[2 * b2];
                for (int k = 0; k < 2; k++) solref[k] = ka->m.pair_solref[2 * p + k];
                for (int k = 0; k < 5; k++) solimp[k] = ka->m.pair_solimp[5 * p + k];
            }
            (void)valid;
            const int j2c = j2 >= 0 ? j2 : (j1 >= 0 ? j1 : 0), j1c = j1 >= 0 ? j1 : 0;      // leading rows: the two entries (the second
            const real v2c = j2 >= 0 ? v2 : real(0);                                       // one repeated with value 0 when absent)
            if (LEAD) {
#pragma unroll
                for (int k = 0; k < TREE_W; k++) J[k] += (k == j1 ? v1 : real(0)) + (k == j2 ? v2 : real(0));
            }
            // K, B, impedance, R [EXT: mj_makeImpedance]
            real dmax = tclamp(solimp[1], real(0.0001), real(0.9999));
            real tc = tmax(solref[0], 2 * ka->m.timestep), dr = solref[1];
            real K = real(1) / tmax(real(1e-15), dmax * dmax * tc * tc * dr * dr), Bd = real(2) / tmax(real(1e-15), dmax * tc);
            real imp, R;
            // (a contact's rows all take the impedance at the contact distance: pos is that distance for the normal row)
            const real imp_row = impedance(solimp, (!LEAD && sub > 0) ? cdist[id] : pos, margin);
            if (!LEAD && sub > 0) {
                int c = id, p = cpair[c];
                real imp0 = imp_row;
                real R0 = tmax(real(1e-15), (1 - imp0) * diag0 / imp0);
                real R1 = R0 / tmax(real(1e-15), ka->m.impratio);
                real mu0 = ka->m.pair_friction[5 * p], mur = ka->m.pair_friction[5 * p + sub - 1];
                R = sub == 1 ? R1 : R1 * mu0 * mu0 / tmax(real(1e-15), mur * mur);
                imp = imp0;
                K = 0;
                floss = mur;
            } else {
                imp = imp_row;
                R = tmax(real(1e-15), (1 - imp) * diag0 / imp);
            }
            // velocity along the row, reference acceleration
            real vel = 0;
            if (LEAD) {
                const int a0 = tadr[tA];
                vel = v1 * qvel[a0 + j1c] + v2c * qvel[a0 + j2c];
            } else {
                // J is zero beyond a tree's dofs: clamped reads instead of per-slot branches
                const int a0 = tadr[tA], b0 = tadr[tB >= 0 ? tB : tA];
#pragma unroll
                for (int k = 0; k < TREE_W; k++) vel += J[k] * qvel[a0 + k < nvm ? a0 + k : nvm];
                if (tB >= 0) {
#pragma unroll
                    for (int k = 0; k < TREE_W; k++) vel += J[TREE_W + k] * qvel[b0 + k < nvm ? b0 + k : nvm];
                }
            }
            const real aref = -Bd * vel - K * imp * (pos - margin);
            // B = J M^-1 per tree (dense 8x8 inverse), diag = J B^T
            real dg = 0;
            real Bv[ROW_W];
            if (LEAD) {
                // J M^-1 = v1 (row j1 of the tree's inverse) + v2 (row j2): the diagonal needs four of its entries, the whole row is
                // made only where something reads it (PGS sweeps)
                const real* Mi = Minv + 64 * tA;
                const real s1 = Mi[8 * j1c + j1c] * v1 + Mi[8 * j1c + j2c] * v2c, s2 = Mi[8 * j2c + j1c] * v1 + Mi[8 * j2c + j2c] * v2c;
                dg = v1 * s1 + v2c * s2;
                if (!lead_slim) {
#pragma unroll
                    for (int k = 0; k < TREE_W; k++) { Bv[k] = Mi[8 * k + j1c] * v1 + Mi[8 * k + j2c] * v2c; Bv[TREE_W + k] = 0; }
                }
            } else {
#pragma unroll
            for (int k = 0; k < TREE_W; k++) {
                real mr[TREE_W], sa = 0;
                lds_load8(Minv + 64 * tA + 8 * k, mr);          // rows of the 8 x 8 inverse start on 32-byte boundaries
#pragma unroll
                for (int j = 0; j < TREE_W; j++) sa += mr[j] * J[j];
                dg += J[k] * sa;
                Bv[k] = sa;
                Bv[TREE_W + k] = 0;
            }
            if (tB >= 0) {
#pragma unroll
                for (int k = 0; k < TREE_W; k++) {
                    real mr[TREE_W], sb = 0;
                    lds_load8(Minv + 64 * tB + 8 * k, mr);
#pragma unroll
                    for (int j = 0; j < TREE_W; j++) sb += mr[j] * J[TREE_W + j];
                    dg += J[TREE_W + k] * sb;
                    Bv[TREE_W + k] = sb;
                }
            }
            }
            if (!LEAD && tB < 0) {
                // one-tree contact row: [J (8) | J M^-1 (8)] in one record; every reader of J masks the second window by the row's dof
                // counts (nB = 0), the Gauss-Seidel sweeps and the couplings below know where J M^-1 is
                real JB_[ROW_W];
#pragma unroll
                for (int k = 0; k < TREE_W; k++) { JB_[k] = J[k]; JB_[TREE_W + k] = Bv[k]; }
                store_row16(rJ + ROW_S * i, JB_);
            } else if (!LEAD || !lead_slim) {       // (the Newton solver and the per-tree noslip pass read the leading rows' descriptors)
                store_row16(rJ + ROW_S * i, J);
                store_row16(rowsB_() + ROW_S * i, Bv);
            }
            // warm start: force implied by last step's acceleration, f = -D (J qacc_ws - aref), made feasible per row
            const int a0 = tadr[tA], nA = tnum[tA], b0 = tB >= 0 ? tadr[tB] : 0, nB = tB >= 0 ? tnum[tB] : 0;
            real jw = 0, jas = 0;      // J . warm start, J . qacc_smooth (the Newton solver's two start candidates)
            if (LEAD) {
                jw = v1 * warm[a0 + j1c] + v2c * warm[a0 + j2c];
                jas = v1 * asm_[a0 + j1c] + v2c * asm_[a0 + j2c];
            } else {
#pragma unroll
            for (int k = 0; k < TREE_W; k++) {
                const int da = a0 + k < nvm ? a0 + k : nvm;
                jw += J[k] * warm[da];
                jas += J[k] * asm_[da];
            }
            if (tB >= 0) {
#pragma unroll
                for (int k = 0; k < TREE_W; k++) {
                    const int db = b0 + k < nvm ? b0 + k : nvm;
                    jw += J[TREE_W + k] * warm[db];
                    jas += J[TREE_W + k] * asm_[db];
                }
            }
            }
            const real big = real(1e30);
            real lo = -big, hi = big, muinv = 0;
            bool ns = false;   // takes part in the noslip sweeps (dry friction and contact friction rows)
            if (LEAD && type == R_FLOSS) { lo = -floss; hi = floss; ns = true; }
            else if (LEAD && type == R_LIMIT) lo = 0;
            else if (!LEAD) {
                if (sub == 0) lo = 0;
                else { ns = true; muinv = real(1) / tmax(real(1e-15), floss); }
            }
            real f = -(jw - aref) / R;
            f = tmin(tmax(f, lo), hi);
            real* S = rowS + RS_S * i;
            // word 2: PGS 1 / (A_rr + R); Newton: residual of the warm start, with that of qacc_smooth in word 8
            S[0] = aref; S[1] = R; S[2] = newton ? jw - aref : real(1) / (dg + R); S[3] = ns ? real(1) / tmax(dg, real(1e-15)) : real(0);
            S[4] = lo; S[5] = hi; S[6] = f; S[7] = muinv; S[8] = jas - aref;
            // leading rows: slots of the (at most two) entries and the sign of the first in the row word, the second value in word 7
            // (lead_d1 / lead_d2 / lead_v1 of avsim_newton.hip.h)
            const int lead_bits = !LEAD ? 0 : ((v1 < 0 ? 1 : 0) << 13) | (j1c << 26) | (j2c << 29);
            if (LEAD) S[7] = v2c;
            rowI[i] = a0 | (nA << 6) | (tA << 10) | ((b0 | (nB << 6) | ((tB >= 0 ? tB : 0) << 10)) << 13) | lead_bits;
            rmeta[i] = (meta & 0xfffff) | ((tA + 1) << 20) | ((tB + 1) << 24);
        };
        for (int i = lane; i < nlead0; i += G) fill(i, BoolTag<true>{});
        ROWPROBE(1);
        for (int i = nlead0 + lane; i < nefc; i += G) fill(i, BoolTag<false>{});
        GSYNC();
        ROWPROBE(2);
        // --- Gauss-Seidel groups: the leading non-contact rows in packs of GRP_MAX, then one group per contact ---
        int* gI = ii + ka->lay.gI;
        GLB_PTR(real) gA = coup_();
        const int nlead = misc[4];   // number of equality / dry-friction / limit rows (they precede the contacts)
        int ngrp = 0;
        for (int base = 0; base < nefc; base += G) {
            int i = base + lane;
            bool head = false;
            int cnt = 0, isc = 0;
            if (i < nefc) {
                if (i < nlead) { head = (i % GRP_MAX) == 0; cnt = nlead - i < GRP_MAX ? nlead - i : GRP_MAX; }
                else { int meta = rmeta[i]; head = ((meta >> 12) & 255) == 0; cnt = ka->m.pair_condim[cpair[(meta >> 2) & 1023]]; isc = 1; }
            }
            int tot, rk = group_rank<G>(head, grp, lane, &tot);
            if (head && ngrp + rk < ka->lay.maxgrp) gI[ngrp + rk] = i | (cnt << 16) | (isc << 24);
            ngrp += tot;
        }
        if (ngrp > ka->lay.maxgrp) ngrp = ka->lay.maxgrp;
        if (lane == 0) misc[5] = ngrp;
        GSYNC();
        ROWPROBE(3);
        // couplings A_rs = J_r . B_s (r > s) inside each group, one pair per lane
        // Under the Newton solver only the noslip sweeps use them, and those never move a contact's normal row: a contact group
        // then needs the 10 pairs among its friction rows (the 5 pairs with the normal row are stored as zeros)
        const int nlg = (nlead + GRP_MAX - 1) / GRP_MAX;                  // the groups of leading rows come first
        const int per_c = newton ? 10 : 15, nl15 = lead_slim ? 0 : nlg * 15, nitem = ngrp <= nlg ? (lead_slim ? 0 : ngrp * 15) : nl15 + (ngrp - nlg) * per_c;
        for (int w = lane; w < nitem; w += G) {
            int g, e, rr = 1, ss;
            if (w < nl15 || !newton) {
                g = w / 15; e = w - 15 * g;
                while ((rr + 1) * rr / 2 <= e) rr++;          // e = rr*(rr-1)/2 + ss
                ss = e - rr * (rr - 1) / 2;
            } else {
                const int wc = w - nl15, e2 = wc % 10;
                g = nlg + wc / 10;
                while ((rr + 1) * rr / 2 <= e2) rr++;         // pair (rr, ss) among rows 1..5, shifted down by one
                ss = e2 - rr * (rr - 1) / 2 + 1;
                rr += 1;
                e = rr * (rr - 1) / 2 + ss;
                if (e2 < 5) gA[GA_W * g + (e2 + 1) * e2 / 2] = 0;   // (row e2 + 1, normal row)
            }
            int gi = gI[g], start = gi & 0xffff, cnt = (gi >> 16) & 15;
            real v = 0;
            if (rr < cnt) {
                int ir = start + rr, is = start + ss, ra = rowI[ir], rs = rowI[is];
                // J_r . (J_s M^-1) over the tree windows the two rows share: the stored rows of J M^-1 are zero-padded, and a
                // window of row r matches a window of row s iff it is the same kinematic tree
                GLB_PTR(const real) Jr = rJ + ROW_S * ir;
                const bool packed_s = g >= nlg && ((rs >> 19) & 15) == 0;       // (J M^-1 of a one-tree contact row: second window of its J record)
                GLB_PTR(const real) Bs = packed_s ? (GLB_PTR(const real))(rJ + ROW_S * is + TREE_W) : (GLB_PTR(const real))(rowsB_() + ROW_S * is);
#pragma unroll
                for (int wr = 0; wr < 2; wr++) {
                    const bool on_r = ((ra >> (13 * wr + 6)) & 15) != 0;
                    const int tr = (ra >> (13 * wr + 10)) & 7;
#pragma unroll
                    for (int ws = 0; ws < 2; ws++) {
                        const bool match = on_r && ((rs >> (13 * ws + 6)) & 15) != 0 && ((rs >> (13 * ws + 10)) & 7) == tr;
                        real t = 0;
#pragma unroll
                        for (int k = 0; k < TREE_W; k++) t += Jr[TREE_W * wr + k] * Bs[TREE_W * ws + k];
                        v += match ? t : real(0);
                    }
                }
            }
            gA[GA_W * g + e] = v;
        }
        GSYNC();
        ROWPROBE(4);
        // noslip QCQP rows (mj_solNoSlip [EXT], see pgs_groups): one contact per lane builds the friction block A of its group from the
        // couplings just written and the rows' diagonals, scales it by the friction coefficients, inverts it through its Cholesky
        // factor (qc_inverse of oracle/orc_dyn.c, same loops) and leaves, per friction row, A's row, the inverse's row and mu
        if (ka->m.noslip_iters > 0)
        for (int g = nlg + lane; g < ngrp; g += G) {
            const int gi = gI[g], start = gi & 0xffff, n = ((gi >> 16) & 15) - 1;
            if (n < 3) continue;
            real A[5][5], As[5][5], L[5][5], Inv[5][5], mu[5];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const bool in = j < n;
                const real* S = rowS + RS_S * (start + 1 + (in ? j : 0));
                mu[j] = in ? real(1) / S[7] : real(1);
                A[j][j] = in ? real(1) / S[3] : real(1);
#pragma unroll
                for (int k = 0; k < j; k++) { const real c = in ? gA[GA_W * g + (j + 1) * j / 2 + (k + 1)] : real(0); A[j][k] = c; A[k][j] = c; }
            }
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int j = 0; j < 5; j++) As[i][j] = (i < n && j < n) ? A[i][j] * mu[i] * mu[j] : (i == j ? real(1) : real(0));
            bool singular = false;
#pragma unroll
            for (int j = 0; j < 5; j++) {
                real dd = As[j][j];
#pragma unroll
                for (int k = 0; k < j; k++) dd -= L[j][k] * L[j][k];
                if (j < n && dd < real(1e-10)) singular = true;
                dd = sqrt(tmax(dd, real(1e-30)));
                L[j][j] = dd;
#pragma unroll
                for (int i = j + 1; i < 5; i++) {
                    real t = As[i][j];
#pragma unroll
                    for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
                    L[i][j] = t / dd;
                }
            }
#pragma unroll
            for (int c = 0; c < 5; c++) {
                real x[5];
#pragma unroll
                for (int i = 0; i < 5; i++) { real t = (i == c) ? real(1) : real(0); for (int k = 0; k < i; k++) t -= L[i][k] * x[k]; x[i] = t / L[i][i]; }
#pragma unroll
                for (int i = 4; i >= 0; i--) { real t = x[i]; for (int k = i + 1; k < 5; k++) t -= L[k][i] * x[k]; x[i] = t / L[i][i]; }
#pragma unroll
                for (int i = 0; i < 5; i++) Inv[i][c] = x[i];
            }
#pragma unroll
            for (int j = 0; j < 5; j++) {
                // row j of A^-1 = D (D A D)^-1 D and the singular flag, entry k at word 8 k + j: the row's lane fetches six separate
                // words (a 16-byte load here gets a register copy, and with it a wait for the look-ahead load, in every step)
                GLB_PTR(real) o = gA + GA_W * g + GA_Q + j;
#pragma unroll
                for (int k = 0; k < 5; k++) o[8 * k] = mu[j] * Inv[j][k] * mu[k];
                o[40] = singular ? real(1) : real(0);
            }
        }
        GSYNC();
        ROWPROBE(5);
    }

    // ---- P8 ------------------------------------------------------------------------------------
    // out of line, on a copy of the object: the members then live in registers (a callee reached through `this` reloads them
    // from memory after every store, and the kernel-argument pointer with them: vector loads instead of s_load)
    __device__ AVS_OUTLINE void solve(int pgs_iters, int solver, int newton_iters, real newton_tol, real scale) {
        Env e(*this);
        e.solve_i(pgs_iters, solver, newton_iters, newton_tol, scale);
        diverged = e.diverged; nit_sum = e.nit_sum; nit_max = e.nit_max; t_broad = e.t_broad; t_narrow = e.t_narrow;
    }
    // the call goes through a throw-away copy: the caller's object never has its address taken and stays in registers too
    __device__ __attribute__((always_inline)) void solve_i(int pgs_iters, int solver, int newton_iters, real newton_tol, real scale) {
        PHASE_BEGIN();
        int *misc = ii + ka->lay.misc, *rmeta = ii + ka->lay.rmeta, *cefc = ii + ka->lay.cefc, *rowI = ii + ka->lay.rowI;
        real *qacc = r + ka->lay.qacc, *as = r + ka->lay.asm_, *rowS = r + ka->lay.rowS, *fcon = r + ka->lay.fcon;
        GLB_PTR(real) rJ = rows_();
        int nefc = misc[1], ncon = misc[0];
        real* Minv = r + ka->lay.Minv;
        if (solver == 1) {
            // ---- primal Newton (the reference's MuJoCo default), then the noslip sweeps on the dual ----
            real* warm = r + ka->lay.warm;
            for (int k = lane; k < ka->m.nv; k += G) qacc[k] = warm[k];
            GSYNC();
            // (one or two contacts per lane by the env's own contact count, not by the capacity: the sums then run in the same order
            // whatever LDS layout the env is stepped with -- the two capacity tiers of PhysHost::launch_t give identical results)
            // does any row reach into two kinematic trees?  (second dof window non-empty; wave-uniform)  Then the Hessian is not block
            // diagonal: the instance with the dense factorisation of the coupled component
            bool two_ = false;
            for (int i = lane; i < nefc; i += G) two_ = two_ || ((rowI[i] >> 19) & 15) != 0;
            const bool coupled = __any(two_) != 0 || ka->m.ntree > 8;
#define AVS_NEWTON_ARGS ka, (GLB_PTR(const real))rJ, (LDS_PTR(real))r, (LDS_PTR(int))ii, (LDS_PTR(const int))li, nefc, ncon, misc[4], newton_iters, newton_tol, scale, profiling ? 1 : 0
            int used = __builtin_amdgcn_readfirstlane(ncon) <= 64 ? (coupled ? newton_solve_coupled<real, 1>(AVS_NEWTON_ARGS) : newton_solve<real, 1, false>(AVS_NEWTON_ARGS))
                                                                  : (coupled ? newton_solve_coupled<real, 2>(AVS_NEWTON_ARGS) : newton_solve<real, 2, false>(AVS_NEWTON_ARGS));
#undef AVS_NEWTON_ARGS
            nit_sum += used; nit_max = used > nit_max ? used : nit_max;
            GSYNC();
            long long tn0 = profiling ? __builtin_readcyclecounter() : 0;
            // every contact on one kinematic tree: the trees' chains side by side (noslip_trees); else, or when a contact slides, the
            // general pass
            // (an env whose pass has given up on a sliding contact skips the attempt for the rest of its env-step: contacts that slide
            // go on sliding for a while)
            int done_ = 0;
            if (ka->m.noslip_trees && lead_per_tree() && __builtin_amdgcn_readfirstlane(misc[8]) == 0)
                done_ = noslip_trees<real>((LDS_PTR(real))rowS, (LDS_PTR(const int))rowI, (LDS_PTR(const int))cefc, (GLB_PTR(const real))rJ, (LDS_PTR(real))qacc,
                                           (LDS_PTR(const int))(ii + ka->lay.gI), (GLB_PTR(const real))coup_(), ncon, nefc, ka->m.noslip_iters, real(1e-6) / ka->m.nscale, NL_ARGS(noslip_lead()));
            if (__builtin_amdgcn_readfirstlane(done_) == 2)
                done_ = noslip_trees2<real>((LDS_PTR(real))rowS, (LDS_PTR(const int))rowI, (LDS_PTR(const int))cefc, (GLB_PTR(const real))rJ, (GLB_PTR(const real))rowsB_(), (LDS_PTR(real))qacc,
                                            (LDS_PTR(const int))(ii + ka->lay.gI), (GLB_PTR(const real))coup_(), ncon, nefc, ka->m.noslip_iters, real(1e-6) / ka->m.nscale, NL_ARGS(noslip_lead()));
            if (__builtin_amdgcn_readfirstlane(done_) != 1 && lane == 0) misc[8] = 1;
            if (__builtin_amdgcn_readfirstlane(done_) != 1)
            pgs_groups<real>((LDS_PTR(real))rowS, (LDS_PTR(const int))rowI, (GLB_PTR(const real))rJ, (GLB_PTR(const real))rowsB_(), (LDS_PTR(real))qacc,
                             (LDS_PTR(const int))(ii + ka->lay.gI), (GLB_PTR(const real))coup_(), misc[5], 0, ka->m.noslip_iters, real(1e-6) / ka->m.nscale, NL_ARGS(noslip_lead()));
            if (profiling && lane == 0) (ii + ka->lay.nprof)[6] += (int)(__builtin_readcyclecounter() - tn0);
        } else {
        // warm-start forces of friction blocks back onto their cones (one contact per lane)
        for (int c = lane; c < ncon; c += G) {
            int first = cefc[c];
            if (first < 0) continue;
            first &= 0xffff;
            int dim = ka->m.pair_condim[(ii + ka->lay.cpair)[c]];
            real fn = rowS[RS_S * first + 6], s2 = 0;
            for (int s = 1; s < dim; s++) { real t = rowS[RS_S * (first + s) + 6] * rowS[RS_S * (first + s) + 7]; s2 += t * t; }
            if (s2 > fn * fn) { real sc = fn / sqrt(s2); for (int s = 1; s < dim; s++) rowS[RS_S * (first + s) + 6] *= sc; }
        }
        GSYNC();
        // qacc = qacc_smooth + M^-1 J^T f : generalized force per dof first, then the per-tree inverse
        jt_force(fcon, nefc);
        for (int k = lane; k < ka->m.nv; k += G) {
            int t = dof_tree_()[k], a0 = tree_dofadr_()[t], kk = k - a0, n = tree_dofnum_()[t];
            real s = as[k];
            for (int j = 0; j < n; j++) s += Minv[64 * t + 8 * kk + j] * fcon[a0 + j];
            qacc[k] = s;
        }
        GSYNC();
        // Gauss-Seidel sweeps (+ noslip sweeps) in the register-resident wave kernel
        static_assert(G == 64, "the solver maps one env to one wavefront");
        pgs_groups<real>((LDS_PTR(real))rowS, (LDS_PTR(const int))rowI, (GLB_PTR(const real))rJ, (GLB_PTR(const real))rowsB_(), (LDS_PTR(real))qacc,
                         (LDS_PTR(const int))(ii + ka->lay.gI), (GLB_PTR(const real))coup_(), misc[5], pgs_iters, ka->m.noslip_iters, real(1e-6) / ka->m.nscale, NL_ARGS(noslip_lead()));
        }
        GSYNC();
        // qfrc_constraint = J^T f
        jt_force(fcon, nefc);
    }

    // what pgs_groups needs to relax the dry-friction rows of a noslip sweep per tree (lane 8 t + i: at most 8 trees)
    AVS_DEV bool lead_per_tree() const { return ka->m.ntree <= 8 && ka->m.noslip_iters > 0 && ka->m.noslip_per_tree != 0; }
    AVS_DEV NoslipLead<real> noslip_lead() const {
        NoslipLead<real> nl;
        nl.Minv = (LDS_PTR(const real))(r + ka->lay.Minv);
        nl.tadr = (LDS_PTR(const int))tree_dofadr_(); nl.tnum = (LDS_PTR(const int))tree_dofnum_(); nl.floss_dof = (LDS_PTR(const int))floss_dof_();
        nl.dmap = (LDS_PTR(int))(r + ka->lay.ng);       // the Newton gradient's words: dead once the primal solve has returned
        nl.ntree = ka->m.ntree; nl.nv = ka->m.nv; nl.neq = ka->m.neq; nl.nfloss = ka->m.nfloss; nl.tridiag = ka->m.qcqp_tridiag;
        nl.prof = profiling ? (LDS_PTR(int))(ii + ka->lay.nprof + 8) : (LDS_PTR(int))nullptr;
        nl.nlg = lead_per_tree() ? ((ii + ka->lay.misc)[4] + GRP_MAX - 1) / GRP_MAX : -1;
        return nl;
    }

    // MuJoCo's mj_checkPos / mj_checkVel [EXT]: a state with NaN / Inf / huge entries is unusable; MuJoCo warns and resets
    // the data, dm_control raises PhysicsError.  Batched: the env goes back to the state its episode started from (home pose,
    // the objects where avsim_reset put them), zero velocity, and the launch reports it through the NaN flag of avsim_get_diag;
    // its neighbours are not affected.
    __device__ void check_divergence() {
        LDS_BASES();
        real *qpos = r + ka->lay.qpos, *qvel = r + ka->lay.qvel, *warm = r + ka->lay.warm;
        bool bad = false;
        for (int i = lane; i < ka->m.nq; i += G) bad |= !(fabs(qpos[i]) < real(1e6));
        for (int i = lane; i < ka->m.nv; i += G) bad |= !(fabs(qvel[i]) < real(1e6));
        if (!__any(bad)) return;
        diverged = 1;
        for (int i = lane; i < ka->m.nq; i += G) qpos[i] = ka->m.qpos_home[i];
        GSYNC();
        for (int i = lane; i < ka->m.nobj * 7; i += G) qpos[ka->m.obj_qadr[i / 7] + i % 7] = (real)ka->m.obj_reset[((size_t)env * ka->m.nobj) * 7 + i];
        for (int i = lane; i < ka->m.nv; i += G) { qvel[i] = 0; warm[i] = 0; }
        if (lane == 0) (ii + ka->lay.misc)[7] = 0;      // the Verlet list is stale
        GSYNC();
    }

    // out = J^T f: one row per lane, scattered over the row's two dof windows with returnless LDS atomics
    __device__ __attribute__((always_inline)) void jt_force(real* out, int nefc) {
        LDS_BASES();
        AVS_ASSUME_LDS(out);
        (void)nefc;
        int* misc = ii + ka->lay.misc;
        for (int k = lane; k < ka->m.nv; k += G) out[k] = 0;
        GSYNC();
        rows_jt_force<real>((LDS_PTR(const real))(r + ka->lay.rowS), (LDS_PTR(const int))(ii + ka->lay.rowI), (LDS_PTR(const int))(ii + ka->lay.cefc),
                            (GLB_PTR(const real))rows_(), (LDS_PTR(real))out, misc[4], misc[0], lane, real(1));
        GSYNC();
    }

    // ---- P9 ------------------------------------------------------------------------------------
    __device__ void euler() {
        PHASE_BEGIN();
        real *qpos = r + ka->lay.qpos, *qvel = r + ka->lay.qvel, *warm = r + ka->lay.warm, *qacc = r + ka->lay.qacc, *fsm = r + ka->lay.fsm, *fcon = r + ka->lay.fcon;
        real *M = r + ka->lay.M, *L = r + ka->lay.L, *tmp = r + ka->lay.bias;
        real h = ka->m.timestep;
        for (int i = lane; i < ka->m.nv; i += G) { tmp[i] = fsm[i] + fcon[i]; warm[i] = qacc[i]; }
        GSYNC();
        // (M + h diag(damping)) qacc_d = qfrc: factor and solve all trees at once (M itself is rebuilt next substep)
        tree_chol(M, L, dof_damping_(), h);
        GSYNC();
        {
            const int t = lane >> 3, i = lane & 7;
            const bool mine = t < ka->m.ntree && i < tree_dofnum_()[t];
            const int k = mine ? tree_dofadr_()[t] + i : 0;
            const real x = tree_solve(L, mine ? tmp[k] : real(0));
            if (mine) qvel[k] += h * x;
        }
        GSYNC();
        for (int j = lane; j < ka->m.njnt; j += G) {
            int qa = jnt_qposadr_()[j], da = jnt_dofadr_()[j];
            if (jnt_type_()[j] == J_FREE) {
                for (int k = 0; k < 3; k++) qpos[qa + k] += h * qvel[da + k];
                real q[4] = {qpos[qa + 3], qpos[qa + 4], qpos[qa + 5], qpos[qa + 6]}, w[3] = {qvel[da + 3], qvel[da + 4], qvel[da + 5]};
                real wn = sqrt(dot3(w, w)), ang = h * wn;
                if (ang > 0) {
                    real s = sin(ang / 2) / wn, qr[4] = {cos(ang / 2), s * w[0], s * w[1], s * w[2]};
                    quatmul(q, qr, q);
                }
                quatnorm(q);
                for (int k = 0; k < 4; k++) qpos[qa + 3 + k] = q[k];
            } else {
                qpos[qa] += h * qvel[da];
            }
        }
        GSYNC();
    }

    __device__ int reward(int* latch) {
        PHASE_BEGIN();
        int *misc = ii + ka->lay.misc, *cpair = ii + ka->lay.cpair;
        int ncon = misc[0];
        int f = 0;
        const int t = ka->m.task_id;
        for (int c = lane; c < ncon; c += G) {
            int p = cpair[c];
            f |= reward_pair_flags(ka->m.geom_class[ka->m.pair_geom[2 * p]], ka->m.geom_class[ka->m.pair_geom[2 * p + 1]], t);
        }
        for (int o = 1; o < G; o <<= 1) f |= __shfl_xor(f, o, G);
        return reward_from_flags(f, t, latch);
    }
};

// Persistent workgroups: one per CU (two where two fit its LDS), up to MAXW wavefronts each, one env per wavefront at a time.  The
// block copies the hot model tables into LDS once; after that every wave works on its own: it takes the next env of the launch from
// a global counter (most expensive first when env_order is given), runs the env's whole step out of its own LDS record, and comes
// back for another.  A slow env therefore holds up one wave's slot, not its block's LDS (static block -> env maps made a block
// wait for the slowest of its eight).  No block barrier after the table copy.
#ifndef AVSIM_PHYS_MAXW
#define AVSIM_PHYS_MAXW 8
#endif
template <typename real, int G, int MAXW, bool RETRY>
#ifndef AVSIM_PHYS_ATTR
#ifdef AVSIM_TU_F64
#define AVSIM_PHYS_ATTR
#else
// two waves per SIMD (<= 256 VGPRs)
#define AVSIM_PHYS_ATTR __attribute__((amdgpu_waves_per_eu(2)))
#endif
#endif
__global__ void __launch_bounds__(64 * MAXW) AVSIM_PHYS_ATTR k_phys(KPtr<real> ka_small, const real* __restrict__ img_real, const int* __restrict__ img_int, int N, int nsub, int pgs_iters, const float* __restrict__ action,
                                             int want_reward, real* __restrict__ g_qpos, real* __restrict__ g_qvel, real* __restrict__ g_ctrl,
                                             real* __restrict__ g_warm, int* __restrict__ g_latch, double* __restrict__ o_agent,
                                             int* __restrict__ o_reward, unsigned char* __restrict__ o_success, int* __restrict__ o_ncon,
                                             int* __restrict__ o_cpairs, double* __restrict__ o_cdist, int* __restrict__ o_diag, int max_reward, int export_contacts, long long* __restrict__ o_prof, float* __restrict__ o_xpose,
                                             const int* __restrict__ env_order, int* __restrict__ o_cost, int* __restrict__ work_head, int* __restrict__ work_next,
                                             int* __restrict__ retry, int retry_mode, KPtr<real> ka_big) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // Two passes over a launch's envs (PhysHost::launch_t).  The first runs every env with the SMALL contact / row capacities -- the
    // LDS record that lets the most envs share a CU; an env that runs out of them in any substep is abandoned before anything of it is
    // written back and its index appended to retry_list.  The second pass (n_dev: the number of such envs, read here) steps those
    // envs again from their untouched state with the large capacities.  Results are those of the large capacities throughout.
    // retry[0], retry[1]: the list's length, used alternately by successive steps; retry + 2: the list; retry + 2 + N: one flag per
    // env, "needed the full capacities in its last step".  retry_mode & 7 = 1 / 3: first pass (appends, to counter 0 / 1); 2 / 4:
    // second pass (reads counter 0 / 1 as its number of envs and zeroes the other one for the next step).
    // In the first pass most such envs never reach the list (retry_mode & 8): the waves of a workgroup are paired, and a wave whose
    // env needs the full capacities -- predicted by the env's flag, or found out by running out of the small ones -- steps it in the
    // TWO adjacent records of the pair while its partner waits (handshake through LDS flags, below).  The list and the second pass
    // remain for what the pairs cannot take: the odd wave of a workgroup, two partners that want the pair at the same time.
    // RETRY = false: the one-pass kernel, none of this compiled in.
    int nwork = N;
    const bool pass2 = RETRY && ((retry_mode & 7) == 2 || (retry_mode & 7) == 4);
    const bool pass1 = RETRY && ((retry_mode & 7) == 1 || (retry_mode & 7) == 3);
    if (pass2) {
        nwork = retry[(retry_mode & 7) == 4];
        if (blockIdx.x == 0 && threadIdx.x == 0) { *work_next = 0; retry[(retry_mode & 7) == 2] = 0; }     // the next launch's counters
        if (nwork <= 0) return;
    }
    int* const bigflag = retry + 2 + N;
    // the first tier's capacities: the small layout's in the first pass; the second pass runs with ka_small == ka_big and gets them in the
    // upper bits of retry_mode (measured against the full ones, `beyond` was always false there and the pass cleared the prediction
    // flag of exactly the envs that need the full record: they were stepped twice in every step)
    const int cap1_con = pass2 ? (retry_mode >> 8) & 0x3ff : ka_small->lay.maxcon, cap1_efc = pass2 ? (retry_mode >> 18) & 0x3ff : ka_small->lay.maxefc;
    // (the pairs' flags: 2 x MAXW / 2 ints behind the tables in the dynamic LDS of a RETRY launch)
    int* const pair_want = reinterpret_cast<int*>(smem + (size_t)(blockDim.x >> 6) * ka_small->lay.bytes_per_env + (size_t)ka_small->mo.nreal * sizeof(real) + (size_t)ka_small->mo.nint * 4);
    int* const pair_grant = pair_want + MAXW / 2;
    if (RETRY && threadIdx.x < MAXW / 2) { pair_want[threadIdx.x] = 0; pair_grant[threadIdx.x] = 0; }      // (both arrays, each MAXW / 2 ints, by name)
    static_assert(G == 64, "one env per wavefront");
#ifdef AVSIM_NO_PROF
    o_prof = nullptr;
#endif
    const int wpb = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = 0;
    // hot model tables -> LDS, once per block
    real* lr = reinterpret_cast<real*>(smem + (size_t)wpb * ka_small->lay.bytes_per_env);
    int* li = reinterpret_cast<int*>(lr + ka_small->mo.nreal);
    for (int i = threadIdx.x; i < ka_small->mo.nreal; i += blockDim.x) lr[i] = img_real[i];
    for (int i = threadIdx.x; i < ka_small->mo.nint; i += blockDim.x) li[i] = img_int[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) *work_next = 0;      // the NEXT launch's counter (launches of a handle follow each other on its stream)
    __syncthreads();
    // ---- pairs (first pass of the two-tier capacities) ----
    const int pr = wave >> 1;
    const bool paired = pass1 && (retry_mode & 8) && (wave | 1) < wpb;
    auto lds_get = [&](int* p) { int v = 0; if (lane == 0) v = __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); return __builtin_amdgcn_readfirstlane(v); };
    auto lds_put = [&](int* p, int v) { if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto to_list = [&](int e) { if (lane == 0) retry[2 + atomicAdd(retry + ((retry_mode & 7) == 3), 1)] = e; };
    // the partner wants both records: wait here (this wave's record is dead between two envs) until it is done
    // A claim is (generation << 8) | (wave + 1), the grant is the claim it answers (PAIR_LEFT once a wave of the pair has left the
    // kernel): a grant left over from an earlier claim never matches a later one, and a claim renewed before the parked partner has
    // polled is granted like the first (the grant used to be a bare 1 set once per park: the renewed claim then waited out its
    // time-out, and a stale 1 could hand both records to a claimant whose partner was in the middle of an env).
    constexpr int PAIR_LEFT = -1;
    int claim_gen = 0, my_claim = 0;
    auto park = [&]() {
        for (;;) {
            const int w = lds_get(&pair_want[pr]);
            if (w == 0 || (w & 0xff) == wave + 1) break;
            lds_put(&pair_grant[pr], w);
            __builtin_amdgcn_s_sleep(32);
        }
    };
    auto claim = [&]() {
        int got = 0;
        const int c = (++claim_gen << 8) | (wave + 1);
        if (lane == 0) { int expect = 0; got = __hip_atomic_compare_exchange_strong(&pair_want[pr], &expect, c, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1 : 0; }
        my_claim = c;
        return __builtin_amdgcn_readfirstlane(got) != 0;
    };
    int redo_env = -1;      // the env this wave has just abandoned with the small record
  for (;;) {
    // wave -> env: next slot of the launch, in the order of the previous launch's cost (k_env_order) or in index order
    int env_ = 0;
    bool big = false;
    if (RETRY && redo_env >= 0) { env_ = redo_env; redo_env = -1; big = true; }
    else {
        if (RETRY && paired) park();
        int slot = 0;
        if (lane == 0) slot = atomicAdd(work_head, 1);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot >= nwork) break;
        env_ = env_order ? env_order[slot] : slot;
        if (RETRY && pass1 && (retry_mode & 8)) big = __builtin_amdgcn_readfirstlane(bigflag[env_]) != 0;
    }
    const int env = env_;
    if (RETRY && big) {
        // take the pair: the claim is an LDS compare-and-swap, the partner answers at its next env boundary (or has left: grant 2)
        bool mine = false;
        if (paired) mine = claim();
        if (!mine && paired) {
            // the partner has claimed the pair for an env of its own: let it (this wave's record is dead), then claim in turn -- the
            // partner answers at its next env boundary.  (The most expensive envs come first in a launch's order, and those are the
            // ones that need the full capacities: at the start of a launch both waves of a pair usually hold one.)
            for (int tries = 0; tries < 64 && !mine; tries++) {
                park();
                mine = claim();
            }
        }
        if (!mine) { to_list(env); continue; }      // (an unpaired wave; the partner's claim, if any, is answered by park() at the top)
        int polls = 0;
        for (;; polls++) {
            const int g = lds_get(&pair_grant[pr]);
            if (g == my_claim || g == PAIR_LEFT || polls >= (1 << 15)) break;
            __builtin_amdgcn_s_sleep(32);
        }
        if (polls >= (1 << 15)) { lds_put(&pair_want[pr], 0); to_list(env); continue; }      // (never seen: a partner's env-step is ~2000 polls; the claim is withdrawn first, its grant can never match another)
    }
    KPtr<real> ka = (RETRY && big) ? ka_big : ka_small;
    const long long t_launch = __builtin_readcyclecounter();
    real* r = reinterpret_cast<real*>(smem + (size_t)((RETRY && big) ? (wave & ~1) : wave) * ka_small->lay.bytes_per_env);
    int* ii = reinterpret_cast<int*>(r + ka->lay.nreal);
    bool beyond = false;      // this env needed more than the small capacities in some substep
    Env<real, G> E(ka, r, ii, lane, grp, lr, li);
    E.env = env;

    // ---- load state (coalesced: consecutive lanes read consecutive words of this env's record) ----
    for (int i = lane; i < ka->m.nq; i += G) r[ka->lay.qpos + i] = g_qpos[(size_t)env * ka->m.nq + i];
    for (int i = lane; i < ka->m.nv; i += G) { r[ka->lay.qvel + i] = g_qvel[(size_t)env * ka->m.nv + i]; r[ka->lay.warm + i] = g_warm[(size_t)env * ka->m.nv + i]; }
    for (int i = lane; i < ka->m.nu; i += G) r[ka->lay.ctrl + i] = g_ctrl[(size_t)env * ka->m.nu + i];
    if (lane == 0) for (int k = 0; k < 8; k++) { ii[ka->lay.misc + k] = 0; ii[ka->lay.nprof + k] = 0; ii[ka->lay.nprof + 8 + k] = 0; }
    if (lane == 0) ii[ka->lay.misc + 8] = 0;
    E.profiling = o_prof != nullptr;
    GSYNC();
    if (action) {
        // env.py:203-215: action -> ctrl, grippers un-normalised (env.py:156-161)
        const float* a = action + (size_t)env * ka->m.nj;
        for (int i = lane; i < ka->m.nj; i += G) {
            real v = (real)a[i];
            if (i == 6 || i == 13) v = v * (ka->m.grip_hi - ka->m.grip_lo) + ka->m.grip_lo;
            r[ka->lay.ctrl + i] = v;
        }
        GSYNC();
    }
    long long tp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PROF(k, stmt) do { if (o_prof) { long long t0_ = __builtin_readcyclecounter(); stmt; tp[k] += __builtin_readcyclecounter() - t0_; } else { stmt; } } while (0)
    for (int s = 0; s < nsub; s++) {
        // E is reached through `this` by the out-of-line phases, so it lives in scratch memory; the phases inlined here run on a
        // copy that never has its address taken (registers, and dead again before the calls: nothing extra to save around them)
#ifndef AVS_NO_SPLIT_PRE
        PROF(0, E.pre_phases());
#else
        {
            Env<real, G> e(E);
            PROF(0, e.kinematics());
            PROF(1, e.crb());
            PROF(2, e.rne_bias());
            PROF(3, e.smooth());
        }
#endif
        PROF(4, E.collide());
        PROF(5, E.make_constraints());
        if (RETRY) beyond = beyond || ii[ka->lay.misc + 0] > cap1_con || ii[ka->lay.misc + 1] > cap1_efc;
        PROF(6, E.solve(pgs_iters, ka->m.solver, ka->m.newton_iters, ka->m.newton_tol, ka->m.nscale));
#ifndef AVS_NO_SPLIT_PRE
        PROF(7, E.post_phases());
#else
        {
            Env<real, G> e(E);
            PROF(7, e.euler());
            e.check_divergence();
            E.diverged = e.diverged;
        }
#endif
    }
    if (o_prof && lane == 0) {
        for (int k = 0; k < 8; k++) o_prof[(size_t)env * PROF_W + k] = tp[k];
        o_prof[(size_t)env * PROF_W + 8] = E.t_broad; o_prof[(size_t)env * PROF_W + 9] = E.t_narrow;
        for (int k = 0; k < 16; k++) o_prof[(size_t)env * PROF_W + 10 + k] = ii[ka->lay.nprof + k];   // Newton: init, grad, hess, chol, search, final, noslip
    }
    // trailing refresh of the position-dependent quantities of the final state
    int nefc_last = ii[ka->lay.misc + 1];
    E.kinematics();
    if (o_xpose) {   // body poses for the depth renderer (avsim_render.hip.h)
        for (int b = lane; b < ka->m.nbody; b += G) {
            float* o = o_xpose + ((size_t)env * ka->m.nbody + b) * 12;
            for (int k = 0; k < 3; k++) o[k] = (float)r[ka->lay.xpos + 3 * b + k];
            for (int k = 0; k < 9; k++) o[3 + k] = (float)r[ka->lay.xmat + 9 * b + k];
        }
    }
    E.collide();
    // first pass: an env that ran out of contact slots or rows in some substep (sticky flags) is left to the second pass -- nothing of
    // it has been written back yet, so that pass steps it from the same state
    if (RETRY && pass1 && !big && (ii[ka->lay.misc + 2] & 3)) {
        if (paired) redo_env = env; else to_list(env);
        GSYNC();      // the record is reused by the wave's next env
        continue;
    }
    if (RETRY) {
        beyond = beyond || ii[ka->lay.misc + 0] > cap1_con;
        if (lane == 0 && (big || pass2 || (retry_mode & 8))) bigflag[env] = beyond ? 1 : 0;
    }

    // ---- write back ---------------------------------------------------------------------------
    if (nsub > 0) {
        for (int i = lane; i < ka->m.nq; i += G) g_qpos[(size_t)env * ka->m.nq + i] = r[ka->lay.qpos + i];
        for (int i = lane; i < ka->m.nv; i += G) { g_qvel[(size_t)env * ka->m.nv + i] = r[ka->lay.qvel + i]; g_warm[(size_t)env * ka->m.nv + i] = r[ka->lay.warm + i]; }
    }
    if (action) for (int i = lane; i < ka->m.nu; i += G) g_ctrl[(size_t)env * ka->m.nu + i] = r[ka->lay.ctrl + i];
    if (o_agent)
        for (int i = lane; i < ka->m.nj; i += G) o_agent[(size_t)env * ka->m.nj + i] = ((double)r[ka->lay.qpos + ka->m.obs_qposadr[i]] - (double)ka->m.obs_offset[i]) * (double)ka->m.obs_scale[i];
    int ncon = ii[ka->lay.misc + 0];
    if (want_reward) {
        int latch = g_latch[env];
        int rw = E.reward(&latch);
        if (lane == 0) {
            g_latch[env] = latch;
            if (o_reward) o_reward[env] = rw;
            if (o_success) o_success[env] = (rw == max_reward);
        }
    }
    if (export_contacts)
    {
    const int expcon = RETRY ? ka->lay.expcon : ka->lay.maxcon;      // (the export arrays have the full capacity's stride in both passes)
    for (int c = lane; c < expcon; c += G) {
        int p = c < ncon ? ii[ka->lay.cpair + c] : -1;
        o_cpairs[((size_t)env * expcon + c) * 2] = p >= 0 ? ka->m.pair_geom[2 * p] : -1;
        o_cpairs[((size_t)env * expcon + c) * 2 + 1] = p >= 0 ? ka->m.pair_geom[2 * p + 1] : -1;
        o_cdist[(size_t)env * expcon + c] = c < ncon ? (double)r[ka->lay.cdist + c] : 0.0;
    }
    }
    if (lane == 0 && o_cost && nsub > 0) o_cost[env] = (int)((__builtin_readcyclecounter() - t_launch) >> 6);    // this env's cost, for the next launch's order
    if (lane == 0 && !o_xpose) {   // the render path's pose-export pass leaves the step diagnostics alone
        o_ncon[env] = ncon;
        bool bad = E.diverged != 0;
        for (int i = 0; i < ka->m.nq; i++) bad |= !(fabs(r[ka->lay.qpos + i]) < real(1e6));
        o_diag[4 * env] = ncon; o_diag[4 * env + 1] = nefc_last; o_diag[4 * env + 2] = ii[ka->lay.misc + 2]; o_diag[4 * env + 3] = (bad ? 1 : 0) | ((ii[ka->lay.misc + 3] & 0xff) << 8) | ((E.nit_sum & 0xfff) << 16) | ((E.nit_max < 15 ? E.nit_max : 15) << 28);
    }
    GSYNC();      // the record is reused by the wave's next env
    if (RETRY && big && pass1) {      // give the pair back: the grant first (unless the partner has left), then the claim the partner waits on
        lds_put(&pair_want[pr], 0);           // (the grant stays: it is keyed to this claim)
    }
  }
    if (RETRY && paired) lds_put(&pair_grant[pr], PAIR_LEFT);      // this wave is leaving: its record is the partner's for the asking
}

// Launch order of the envs: by the cost (shader-clock cycles) of their last step, most expensive first.  A block holds its LDS
// until its slowest env is done and the blocks of a launch are dispatched in index order as CUs free up, so (a) envs of similar
// cost share a block and (b) the long blocks start first, the short ones fill in behind them (longest-processing-time-first).
// The state arrays stay indexed by env: only the wave -> env map changes, results do not depend on it (nor on the order inside a
// bucket, which the atomics leave open).  One block: counting sort into 256 cost buckets between the launch's minimum and maximum.
static __global__ void __launch_bounds__(1024) k_env_order(const int* __restrict__ cost, int* __restrict__ order, int N) {
    __shared__ int lo, hi, cnt[256], off[256];
    if (threadIdx.x == 0) { lo = 0x7fffffff; hi = 0; }
    if (threadIdx.x < 256) cnt[threadIdx.x] = 0;
    __syncthreads();
    int mn = 0x7fffffff, mx = 0;
    for (int i = threadIdx.x; i < N; i += 1024) { const int c = cost[i]; mn = c < mn ? c : mn; mx = c > mx ? c : mx; }
    atomicMin(&lo, mn);
    atomicMax(&hi, mx);
    __syncthreads();
    const int base = lo;
    const float scale = 255.0f / (float)(hi - lo + 1);
    auto bucket = [&](int c) { const int b = 255 - (int)((float)(c - base) * scale); return b < 0 ? 0 : (b > 255 ? 255 : b); };
    for (int i = threadIdx.x; i < N; i += 1024) atomicAdd(&cnt[bucket(cost[i])], 1);
    __syncthreads();
    if (threadIdx.x == 0) { int a = 0; for (int b = 0; b < 256; b++) { off[b] = a; a += cnt[b]; } }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 1024) order[atomicAdd(&off[bucket(cost[i])], 1)] = i;
}

// ------------------------------------------------------------------------------------------------
// host side: device model image, LDS layout, launch
// ------------------------------------------------------------------------------------------------
struct PhysHost;
// k_phys<double> (AVSIM_F64_PHYSICS, the parity mode): defined in avsim_phys_f64.hip, which is compiled with -ffp-contract=off so
// that the device evaluates every expression with the roundings of the oracle (oracle/Makefile: -ffp-contract=off) -- the
// narrow phase's support-vertex and clipping tie-breaks then fall the same way on both sides
int phys_launch_f64(PhysHost& ph, hipStream_t st, int nsub, const float* action, void* qpos, void* qvel, void* ctrl, void* warm, int* latch,
                    double* agent, int32_t* reward, uint8_t* success, std::string& err);

struct PhysHost {
    int maxcon = 48, maxefc = 144, pgs_iters = 20, group = 64, export_contacts = 1, force_reward = 0, wpb_override = 0;
    // Capacities in two tiers.  maxcon / maxefc are what an env can hold (the stride of the contact export, the size of the global row
    // scratch); maxcon1 / maxefc1 <= those size the LDS record of the FIRST pass of a launch, chosen so that the most envs share a CU.
    // An env that needs more in some substep is stepped again by a second pass with the full capacities (k_phys, retry_list): the
    // results are those of the full capacities, the common case runs at the residency of the small ones.  Equal tiers: one pass.
    int maxcon1 = 48, maxefc1 = 144;
    bool f64 = false;
    int N = 0, max_reward = 0;
    std::vector<void*> allocs;
    DevModel<float> mf;
    DevModel<double> md;
    Layout lay, lay2;               // first pass / second pass (lay2 == lay with one tier)
    bool two_pass() const { return maxcon1 < maxcon || maxefc1 < maxefc; }
    int pair_waves = 1;             // option "pair_waves": 0 = every env that needs the full capacities waits for the second pass
    int* d_retry = nullptr;         // two counters (used alternately, like d_head) + the list of envs for the second pass
    unsigned long long retry_parity = 0;
    void* d_kargs2 = nullptr;       // KArgs of the second pass
    double *d_qpos_home = nullptr, *d_ctrl_home = nullptr, *d_obj_reset = nullptr;
    int* d_obj_qadr = nullptr;
    int *d_ncon = nullptr, *d_cpairs = nullptr, *d_diag = nullptr;
    int *d_cost = nullptr, *d_order = nullptr;     // per-env cost of the last step and the launch order made from it (k_env_order)
    int order_envs = 1;                            // option "order_envs"
    bool have_cost = false;
    long long* d_prof = nullptr;   // optional per-env phase cycle counters (option "profile_phases")
    float* d_xpose = nullptr;      // when set, the launch also exports body poses float[N][nbody][12] (render path)
    double* d_cdist = nullptr;

    std::vector<int> img_int;
    std::vector<double> img_real;
    MOff moff;
    void* d_img_real = nullptr;
    int* d_img_int = nullptr;
    void* d_kargs = nullptr;        // KArgs<float|double> in device memory
    bool kargs_dirty = true;
    unsigned attr_done = 0;         // the kernel's LDS-size attribute, the CU count and the work counters are set up (once per handle)
    int num_cu = 256;               // CUs of the handle's device (persistent blocks: one per CU's LDS share)
    int persist_over = 1;           // option "persist_blocks": blocks launched per resident slot (1 = exactly what the CUs hold)
    int* d_head = nullptr;          // two work counters, used alternately: a launch takes envs from one and zeroes the other
    unsigned long long launch_count = 0;
    // device pointer that converts to the plain and to the global-address-space pointer types
    template <typename T>
    struct DevPtr {
        T* p;
        operator T*() const { return p; }
        operator GLB_PTR(T)() const { return (GLB_PTR(T))p; }
        operator GLB_PTR(const T)() const { return (GLB_PTR(const T))p; }
    };
    template <typename T>
    DevPtr<T> up(const std::vector<T>& v) {
        void* p = nullptr;
        size_t n = (v.size() ? v.size() : 1) * sizeof(T);
        if (hipMalloc(&p, n) != hipSuccess) throw std::runtime_error("hipMalloc failed while uploading the model");
        if (v.size() && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("hipMemcpy failed while uploading the model");
        allocs.push_back(p);
        return DevPtr<T>{(T*)p};
    }
    template <typename real>
    DevPtr<real> upr(const std::vector<double>& v) {
        std::vector<real> w(v.begin(), v.end());
        return up(w);
    }

    template <typename real>
    void build(const Blob& b, DevModel<real>& m) {
        auto I = [&](const char* n) { return b.i(n); };
        auto F = [&](const char* n) { return b.f(n); };
        m.nq = b.scalar("nq"); m.nv = b.scalar("nv"); m.nu = b.scalar("nu"); m.nbody = b.scalar("nbody"); m.njnt = b.scalar("njnt");
        m.ngeom = b.scalar("ngeom"); m.npair = b.scalar("npair"); m.ntree = b.scalar("ntree"); m.neq = b.scalar("neq");
        m.task_id = b.scalar("task_id");
        m.nj = b.scalar("num_arms") == 3 ? 21 : 14;
        auto opt = F("opt");
        m.timestep = (real)opt[0]; m.gravity[0] = (real)opt[1]; m.gravity[1] = (real)opt[2]; m.gravity[2] = (real)opt[3];
        m.impratio = (real)opt[4]; m.noslip_iters = (int)opt[5]; m.noslip_per_tree = 1; m.noslip_trees = 1; m.newton_component = 1; m.newton_early_exit = 1; m.qcqp_tridiag = sizeof(real) == 4 ? 2 : 0;
        m.solver = 1; m.newton_iters = 100; m.newton_tol = sizeof(real) == 8 ? (real)1e-8 : (real)1e-6; m.ls_tolerance = sizeof(real) == 8 ? (real)1e-10 : (real)1e-4; m.ls_iterations = 50;     // MuJoCo defaults: iterations 100, tolerance 1e-8
        m.nscale = (real)(1.0 / ((opt.size() > 7 && opt[7] > 0 ? opt[7] : 1.0) * std::max(1, m.nv)));
        auto gr = F("grip_range");
        m.grip_lo = (real)gr[0]; m.grip_hi = (real)gr[1];
        auto body_parent = I("body_parent"), body_dofadr = I("body_dofadr"), body_dofnum = I("body_dofnum"), body_tree = I("body_tree");
        auto dof_parent = I("dof_parent"), dof_tree = I("dof_tree"), tree_dofadr = I("tree_dofadr"), tree_dofnum = I("tree_dofnum");
        auto dof_body = I("dof_body");
        int nb = m.nbody, nv = m.nv, nt = m.ntree;
        for (int t = 0; t < nt; t++) if (tree_dofnum[t] > TREE_W) throw std::runtime_error("kinematic tree with more than 8 dofs");
        if (nb > 64) throw std::runtime_error("more than 64 bodies (kinematics maps one body per lane)");
        if (nt > 8) throw std::runtime_error("more than 8 kinematic trees (the per-tree solves map 8 lanes to a tree)");
        { auto jn = I("body_jntnum"); for (int bb = 0; bb < nb; bb++) if (jn[bb] > 1) throw std::runtime_error("body with more than one joint"); }
        // bodies of each tree, in id order; static poses
        std::vector<int> tba(nt + 1, 0), tbl;
        for (int t = 0; t < nt; t++) {
            tba[t] = (int)tbl.size();
            for (int bb = 1; bb < nb; bb++) if (body_tree[bb] == t) tbl.push_back(bb);
        }
        tba[nt] = (int)tbl.size();
        // static world poses from the blob's body tree at qpos0 (bodies welded to the world)
        auto bpos = F("body_pos"), bquat = F("body_quat");
        std::vector<double> sx(3 * nb, 0.0), sm(9 * nb, 0.0);
        sm[0] = sm[4] = sm[8] = 1;
        auto q2m = [](const double* q, double* R) {
            double w = q[0], x = q[1], y = q[2], z = q[3];
            R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
            R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
            R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
        };
        for (int bb = 1; bb < nb; bb++) {
            if (body_tree[bb] >= 0) { sm[9 * bb] = sm[9 * bb + 4] = sm[9 * bb + 8] = 1; continue; }
            int p = body_parent[bb];
            double Rl[9];
            q2m(&bquat[4 * bb], Rl);
            for (int i = 0; i < 3; i++) {
                sx[3 * bb + i] = sx[3 * p + i] + sm[9 * p + 3 * i] * bpos[3 * bb] + sm[9 * p + 3 * i + 1] * bpos[3 * bb + 1] + sm[9 * p + 3 * i + 2] * bpos[3 * bb + 2];
                for (int j = 0; j < 3; j++) sm[9 * bb + 3 * i + j] = sm[9 * p + 3 * i] * Rl[j] + sm[9 * p + 3 * i + 1] * Rl[3 + j] + sm[9 * p + 3 * i + 2] * Rl[6 + j];
            }
        }
        // dof masks: tree-local dofs that move each body
        std::vector<int> mask(nb, 0);
        for (int bb = 1; bb < nb; bb++) {
            int t = body_tree[bb];
            if (t < 0) continue;
            int c = bb;
            while (c > 0 && body_dofnum[c] == 0) c = body_parent[c];
            if (c == 0) continue;
            for (int d = body_dofadr[c] + body_dofnum[c] - 1; d >= 0; d = dof_parent[d]) mask[bb] |= 1 << (d - tree_dofadr[t]);
        }
        // mass-matrix entries (i, ancestor j) and per-tree block offsets
        std::vector<int> mi, mj, madr(nt, 0);
        int ms = 0;
        for (int t = 0; t < nt; t++) { madr[t] = ms; ms += tree_dofnum[t] * tree_dofnum[t]; }
        for (int i = 0; i < nv; i++) for (int j = i; j >= 0; j = dof_parent[j]) { mi.push_back(i); mj.push_back(j); }
        m.msize = ms;
        m.nment = (int)mi.size();
        std::vector<int> fl, lj;
        auto floss = F("dof_frictionloss");
        for (int i = 0; i < nv; i++) if (floss[i] > 0) fl.push_back(i);
        auto jl = I("jnt_limited");
        for (int j = 0; j < m.njnt; j++) if (jl[j]) lj.push_back(j);
        m.nfloss = (int)fl.size();
        m.nlimited = (int)lj.size();
        m.body_parent = up(body_parent); moff.body_parent = (int)img_int.size(); { auto v_ = body_parent; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.body_jntadr = up(I("body_jntadr")); moff.body_jntadr = (int)img_int.size(); { auto v_ = I("body_jntadr"); img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.body_jntnum = up(I("body_jntnum")); moff.body_jntnum = (int)img_int.size(); { auto v_ = I("body_jntnum"); img_int.insert(img_int.end(), v_.begin(), v_.end()); }
        m.body_dofadr = up(body_dofadr); moff.body_dofadr = (int)img_int.size(); { auto v_ = body_dofadr; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.body_dofnum = up(body_dofnum); moff.body_dofnum = (int)img_int.size(); { auto v_ = body_dofnum; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.body_tree = up(body_tree); moff.body_tree = (int)img_int.size(); { auto v_ = body_tree; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.body_dofmask = up(mask); moff.body_dofmask = (int)img_int.size(); { auto v_ = mask; img_int.insert(img_int.end(), v_.begin(), v_.end()); }
        {   // bodies are numbered depth first: the subtree of body p is the id range p .. body_last[p] (checked)
            std::vector<int> last(nb);
            for (int bb = 0; bb < nb; bb++) last[bb] = bb;
            for (int bb = nb - 1; bb > 0; bb--) { int pp = body_parent[bb]; if (last[bb] > last[pp]) last[pp] = last[bb]; }
            for (int bb = 1; bb < nb; bb++) { int pp = body_parent[bb]; if (!(pp < bb && bb <= last[pp])) throw std::runtime_error("bodies are not in depth-first order"); }
            for (int pp = 0; pp < nb; pp++) for (int d = pp + 1; d <= last[pp]; d++) { int a = d; while (a > pp) a = body_parent[a]; if (a != pp) throw std::runtime_error("bodies are not in depth-first order"); }
            m.body_last = up(last); moff.body_last = (int)img_int.size(); img_int.insert(img_int.end(), last.begin(), last.end());
        }
        m.body_pos = upr<real>(bpos); moff.body_pos = (int)img_real.size(); { auto v_ = bpos; img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.body_quat = upr<real>(bquat); moff.body_quat = (int)img_real.size(); { auto v_ = bquat; img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.body_mass = upr<real>(F("body_mass")); moff.body_mass = (int)img_real.size(); { auto v_ = F("body_mass"); img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.body_ipos = upr<real>(F("body_ipos")); moff.body_ipos = (int)img_real.size(); { auto v_ = F("body_ipos"); img_real.insert(img_real.end(), v_.begin(), v_.end()); }
        m.body_inertia = upr<real>(F("body_inertia")); moff.body_inertia = (int)img_real.size(); { auto v_ = F("body_inertia"); img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.body_invweight0 = upr<real>(F("body_invweight0")); moff.body_invweight0 = (int)img_real.size(); { auto v_ = F("body_invweight0"); img_real.insert(img_real.end(), v_.begin(), v_.end()); }
        m.static_xpos = upr<real>(sx); m.static_xmat = upr<real>(sm);
        m.tree_bodyadr = up(tba); moff.tree_bodyadr = (int)img_int.size(); { auto v_ = tba; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.tree_bodylist = up(tbl); moff.tree_bodylist = (int)img_int.size(); { auto v_ = tbl; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.tree_dofadr = up(tree_dofadr); moff.tree_dofadr = (int)img_int.size(); { auto v_ = tree_dofadr; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.tree_dofnum = up(tree_dofnum); moff.tree_dofnum = (int)img_int.size(); { auto v_ = tree_dofnum; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.tree_madr = up(madr); moff.tree_madr = (int)img_int.size(); { auto v_ = madr; img_int.insert(img_int.end(), v_.begin(), v_.end()); }
        m.jnt_type = up(I("jnt_type")); moff.jnt_type = (int)img_int.size(); { auto v_ = I("jnt_type"); img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.jnt_qposadr = up(I("jnt_qposadr")); moff.jnt_qposadr = (int)img_int.size(); { auto v_ = I("jnt_qposadr"); img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.jnt_dofadr = up(I("jnt_dofadr")); moff.jnt_dofadr = (int)img_int.size(); { auto v_ = I("jnt_dofadr"); img_int.insert(img_int.end(), v_.begin(), v_.end()); }
        m.jnt_actfrclimited = up(I("jnt_actfrclimited")); moff.jnt_actfrclimited = (int)img_int.size(); { auto v_ = I("jnt_actfrclimited"); img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.limited_jnt = up(lj); moff.limited_jnt = (int)img_int.size(); { auto v_ = lj; img_int.insert(img_int.end(), v_.begin(), v_.end()); }
        m.jnt_pos = upr<real>(F("jnt_pos")); moff.jnt_pos = (int)img_real.size(); { auto v_ = F("jnt_pos"); img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.jnt_axis = upr<real>(F("jnt_axis")); moff.jnt_axis = (int)img_real.size(); { auto v_ = F("jnt_axis"); img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.jnt_range = upr<real>(F("jnt_range")); moff.jnt_range = (int)img_real.size(); { auto v_ = F("jnt_range"); img_real.insert(img_real.end(), v_.begin(), v_.end()); }
        m.jnt_actfrcrange = upr<real>(F("jnt_actfrcrange")); moff.jnt_actfrcrange = (int)img_real.size(); { auto v_ = F("jnt_actfrcrange"); img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.jnt_solref = upr<real>(F("jnt_solref")); m.jnt_solimp = upr<real>(F("jnt_solimp"));
        m.jnt_margin = upr<real>(F("jnt_margin")); moff.jnt_margin = (int)img_real.size(); { auto v_ = F("jnt_margin"); img_real.insert(img_real.end(), v_.begin(), v_.end()); }
        m.dof_body = up(dof_body); moff.dof_body = (int)img_int.size(); { auto v_ = dof_body; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.dof_parent = up(dof_parent); moff.dof_parent = (int)img_int.size(); { auto v_ = dof_parent; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.dof_tree = up(dof_tree); moff.dof_tree = (int)img_int.size(); { auto v_ = dof_tree; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.dof_jnt = up(I("dof_jnt")); moff.dof_jnt = (int)img_int.size(); { auto v_ = I("dof_jnt"); img_int.insert(img_int.end(), v_.begin(), v_.end()); }
        m.floss_dof = up(fl); moff.floss_dof = (int)img_int.size(); { auto v_ = fl; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.ment_i = up(mi); moff.ment_i = (int)img_int.size(); { auto v_ = mi; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.ment_j = up(mj); moff.ment_j = (int)img_int.size(); { auto v_ = mj; img_int.insert(img_int.end(), v_.begin(), v_.end()); }
        m.dof_armature = upr<real>(F("dof_armature")); moff.dof_armature = (int)img_real.size(); { auto v_ = F("dof_armature"); img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.dof_damping = upr<real>(F("dof_damping")); moff.dof_damping = (int)img_real.size(); { auto v_ = F("dof_damping"); img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.dof_frictionloss = upr<real>(floss); moff.dof_frictionloss = (int)img_real.size(); { auto v_ = floss; img_real.insert(img_real.end(), v_.begin(), v_.end()); }
        m.dof_invweight0 = upr<real>(F("dof_invweight0")); moff.dof_invweight0 = (int)img_real.size(); { auto v_ = F("dof_invweight0"); img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.dof_solref = upr<real>(F("dof_solref")); m.dof_solimp = upr<real>(F("dof_solimp"));
        m.act_dof = up(I("act_dof")); moff.act_dof = (int)img_int.size(); { auto v_ = I("act_dof"); img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.act_qposadr = up(I("act_qposadr")); moff.act_qposadr = (int)img_int.size(); { auto v_ = I("act_qposadr"); img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.act_ctrllimited = up(I("act_ctrllimited")); moff.act_ctrllimited = (int)img_int.size(); { auto v_ = I("act_ctrllimited"); img_int.insert(img_int.end(), v_.begin(), v_.end()); }
        m.act_kp = upr<real>(F("act_kp")); moff.act_kp = (int)img_real.size(); { auto v_ = F("act_kp"); img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.act_kv = upr<real>(F("act_kv")); moff.act_kv = (int)img_real.size(); { auto v_ = F("act_kv"); img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.act_gear = upr<real>(F("act_gear")); moff.act_gear = (int)img_real.size(); { auto v_ = F("act_gear"); img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.act_ctrlrange = upr<real>(F("act_ctrlrange")); moff.act_ctrlrange = (int)img_real.size(); { auto v_ = F("act_ctrlrange"); img_real.insert(img_real.end(), v_.begin(), v_.end()); }
        m.eq_dof1 = up(I("eq_dof1")); m.eq_dof2 = up(I("eq_dof2")); m.eq_qpos1 = up(I("eq_qpos1")); m.eq_qpos2 = up(I("eq_qpos2"));
        m.eq_polycoef = upr<real>(F("eq_polycoef")); m.eq_solref = upr<real>(F("eq_solref")); m.eq_solimp = upr<real>(F("eq_solimp"));
        m.qpos0 = upr<real>(F("qpos0"));
        m.qpos_home = upr<real>(F("qpos_home"));
        m.obj_qadr = up(I("objects_qposadr")); m.nobj = (int)I("objects_qposadr").size();
        {   // the poses an episode starts with: the model's own until avsim_reset says otherwise
            auto home = F("qpos_home"); auto oa = I("objects_qposadr");
            std::vector<double> o((size_t)N * oa.size() * 7);
            for (int e = 0; e < N; e++) for (size_t k = 0; k < oa.size(); k++) for (int c = 0; c < 7; c++) o[((size_t)e * oa.size() + k) * 7 + c] = home[oa[k] + c];
            d_obj_reset = up(o); m.obj_reset = d_obj_reset;
        }
        // geoms: local rotation matrices, interior point in the body frame, world constants for static geoms
        auto gbody = I("geom_body");
        auto gpos = F("geom_pos"), gquat = F("geom_quat"), gbc = F("geom_bcenter");
        int ng = m.ngeom;
        std::vector<double> gmat(9 * ng), gcp(3 * ng), gx0(3 * ng, 0.0), gm0(9 * ng, 0.0), gc0(3 * ng, 0.0);
        std::vector<int> gstat(ng);
        std::vector<double> gaabb(6 * ng, 0.0), glbox(6 * ng, 0.0);   // world AABB of static geoms; local box (centre, half extents) of every geom
        auto ghull = I("geom_chull"); auto gtype = I("geom_type"); auto gsize = F("geom_size"); auto hv = F("chull_vert");      // the COLLISION hulls (hull_vert / geom_hull: the depth images' polyhedra)
        for (int g = 0; g < ng; g++) {
            q2m(&gquat[4 * g], &gmat[9 * g]);
            for (int i = 0; i < 3; i++)
                gcp[3 * g + i] = gpos[3 * g + i] + gmat[9 * g + 3 * i] * gbc[3 * g] + gmat[9 * g + 3 * i + 1] * gbc[3 * g + 1] + gmat[9 * g + 3 * i + 2] * gbc[3 * g + 2];
            {
                double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
                if (gtype[g] == G_MESH) {
                    for (int v = 0; v < ghull[2 * g + 1]; v++)
                        for (int k = 0; k < 3; k++) { double x = hv[3 * (ghull[2 * g] + v) + k]; if (x < lo[k]) lo[k] = x; if (x > hi[k]) hi[k] = x; }
                } else {
                    double ex[3] = {gsize[3 * g], gsize[3 * g + 1], gsize[3 * g + 2]};
                    if (gtype[g] == G_SPHERE) ex[1] = ex[2] = ex[0];
                    if (gtype[g] == G_CYLINDER) { ex[2] = ex[1]; ex[1] = ex[0]; }
                    for (int k = 0; k < 3; k++) { lo[k] = -ex[k]; hi[k] = ex[k]; }
                }
                for (int k = 0; k < 3; k++) { glbox[6 * g + k] = 0.5 * (lo[k] + hi[k]); glbox[6 * g + 3 + k] = 0.5 * (hi[k] - lo[k]); }
            }
            int bb = gbody[g];
            gstat[g] = body_tree[bb] < 0;
            if (gstat[g]) {
                for (int i = 0; i < 3; i++) {
                    gx0[3 * g + i] = sx[3 * bb + i] + sm[9 * bb + 3 * i] * gpos[3 * g] + sm[9 * bb + 3 * i + 1] * gpos[3 * g + 1] + sm[9 * bb + 3 * i + 2] * gpos[3 * g + 2];
                    gc0[3 * g + i] = sx[3 * bb + i] + sm[9 * bb + 3 * i] * gcp[3 * g] + sm[9 * bb + 3 * i + 1] * gcp[3 * g + 1] + sm[9 * bb + 3 * i + 2] * gcp[3 * g + 2];
                    for (int j = 0; j < 3; j++)
                        gm0[9 * g + 3 * i + j] = sm[9 * bb + 3 * i] * gmat[9 * g + j] + sm[9 * bb + 3 * i + 1] * gmat[9 * g + 3 + j] + sm[9 * bb + 3 * i + 2] * gmat[9 * g + 6 + j];
                }
                // world AABB of the static geom (hull vertices / box corners / bounding box of round shapes)
                std::vector<double> pts;
                if (gtype[g] == G_MESH) {
                    for (int v = 0; v < ghull[2 * g + 1]; v++) for (int k = 0; k < 3; k++) pts.push_back(hv[3 * (ghull[2 * g] + v) + k]);
                } else {
                    double ex[3] = {gsize[3 * g], gsize[3 * g + 1], gsize[3 * g + 2]};
                    if (gtype[g] == G_SPHERE) ex[1] = ex[2] = ex[0];
                    if (gtype[g] == G_CYLINDER) { ex[2] = ex[1]; ex[1] = ex[0]; }
                    for (int c = 0; c < 8; c++) for (int k = 0; k < 3; k++) pts.push_back(((c >> k) & 1) ? ex[k] : -ex[k]);
                }
                for (int k = 0; k < 3; k++) { gaabb[6 * g + k] = 1e30; gaabb[6 * g + 3 + k] = -1e30; }
                for (size_t v = 0; v < pts.size() / 3; v++)
                    for (int i = 0; i < 3; i++) {
                        double w = gx0[3 * g + i] + gm0[9 * g + 3 * i] * pts[3 * v] + gm0[9 * g + 3 * i + 1] * pts[3 * v + 1] + gm0[9 * g + 3 * i + 2] * pts[3 * v + 2];
                        if (w < gaabb[6 * g + i]) gaabb[6 * g + i] = w;
                        if (w > gaabb[6 * g + 3 + i]) gaabb[6 * g + 3 + i] = w;
                    }
            }
        }
        m.geom_type = up(I("geom_type")); moff.geom_type = (int)img_int.size(); { auto v_ = I("geom_type"); img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.geom_body = up(gbody); moff.geom_body = (int)img_int.size(); { auto v_ = gbody; img_int.insert(img_int.end(), v_.begin(), v_.end()); } m.geom_hull = up(I("geom_ctab")); m.geom_class = up(I("geom_class")); m.geom_static = up(gstat); moff.geom_static = (int)img_int.size(); { auto v_ = gstat; img_int.insert(img_int.end(), v_.begin(), v_.end()); }
        m.geom_pos = upr<real>(gpos); m.geom_mat = upr<real>(gmat); m.geom_size = upr<real>(F("geom_size")); m.geom_cpos = upr<real>(gcp); moff.geom_cpos = (int)img_real.size(); { auto v_ = gcp; img_real.insert(img_real.end(), v_.begin(), v_.end()); }
        m.geom_rbound = upr<real>(F("geom_rbound")); moff.geom_rbound = (int)img_real.size(); { auto v_ = F("geom_rbound"); img_real.insert(img_real.end(), v_.begin(), v_.end()); } m.geom_xpos0 = upr<real>(gx0); m.geom_xmat0 = upr<real>(gm0); m.geom_cen0 = upr<real>(gc0); m.geom_aabb0 = upr<real>(gaabb); m.geom_lbox = upr<real>(glbox);
        {   // support tables of the collision hulls (compiler/hull.py support_table).  The blob holds, per cube-map cell, a list of vertex
            // indices local to the cell's hull; the device gets one record of eight (x, y, z, w) entries per cell -- the first eight
            // candidates, a shorter list padded with its last vertex; w of entry 0 = the count, w of entry 1 = where the cell's further
            // candidates start in the overflow part that follows the records -- so that a support call is one round trip
            auto ctab = I("geom_ctab"), cells = I("chull_cells"), cand = I("chull_cand");
            const size_t ncell = cells.size();
            std::vector<double> tab(32 * ncell, 0.0), ovf;
            std::vector<char> done(ncell, 0);
            for (int g = 0; g < ng; g++) {
                if (gtype[g] != G_MESH) continue;
                const int cb = ctab[2 * g], R = ctab[2 * g + 1], adr = ghull[2 * g];
                for (int c = 0; c < 6 * R * R; c++) {
                    if (done[cb + c]) continue;
                    done[cb + c] = 1;
                    const int rec = cells[cb + c], off = rec >> 8, cnt = rec & 255;
                    if (cnt < 1) throw std::runtime_error("support table: empty cell");
                    double* o = &tab[32 * (size_t)(cb + c)];
                    for (int k = 0; k < 8; k++) {
                        const int v = cand[off + (k < cnt ? k : cnt - 1)];
                        if (v < 0 || v >= ghull[2 * g + 1]) throw std::runtime_error("support table: candidate index outside its hull");
                        for (int q = 0; q < 3; q++) o[4 * k + q] = hv[3 * (size_t)(adr + v) + q];
                        o[4 * k + 3] = (double)v;
                    }
                    o[3] = (double)cnt;
                    o[7] = (double)(ovf.size() / 4);
                    for (int k = 8; k < cnt; k++) {
                        const int v = cand[off + k];
                        if (v < 0 || v >= ghull[2 * g + 1]) throw std::runtime_error("support table: candidate index outside its hull");
                        for (int q = 0; q < 3; q++) ovf.push_back(hv[3 * (size_t)(adr + v) + q]);
                        ovf.push_back((double)v);
                    }
                }
            }
            if (ovf.size() / 4 >= (1u << 24)) throw std::runtime_error("support table: overflow part too large");
            m.hull_ovf = (int)(tab.size() / 4);
            tab.insert(tab.end(), ovf.begin(), ovf.end());
            if (tab.empty()) tab.assign(4, 0.0);
            m.hull_vert = upr<real>(tab);
        }
        m.pair_geom = up(I("pair_geom")); m.pair_condim = up(I("pair_condim"));
        m.pair_friction = upr<real>(F("pair_friction")); m.pair_solref = upr<real>(F("pair_solref")); m.pair_solimp = upr<real>(F("pair_solimp"));
        m.pair_margin = upr<real>(F("pair_margin")); m.pair_gap = upr<real>(F("pair_gap"));
        m.obs_qposadr = up(I("obs_qposadr")); m.obs_offset = upr<real>(F("obs_offset")); m.obs_scale = upr<real>(F("obs_scale"));
        // packed image of the hot tables: copied into LDS by every block (global latency is paid once per launch)
        while (img_real.size() % 4) img_real.push_back(0.0);
        while (img_int.size() % 4) img_int.push_back(0);
        moff.nreal = (int)img_real.size();
        moff.nint = (int)img_int.size();
        d_img_real = (void*)upr<real>(img_real);
        d_img_int = up(img_int);
    }

    void make_layout(int nq, int nv, int nu, int nb, int ng, int msize, int ntree) {
        if (maxcon1 > maxcon) maxcon1 = maxcon;
        if (maxefc1 > maxefc) maxefc1 = maxefc;
        make_layout_of(lay, maxcon1, maxefc1, nq, nv, nu, nb, ng, msize, ntree);
        make_layout_of(lay2, maxcon, maxefc, nq, nv, nu, nb, ng, msize, ntree);
    }
    void make_layout_of(Layout& L, const int maxcon, const int maxefc, int nq, int nv, int nu, int nb, int ng, int msize, int ntree) {
        kargs_dirty = true;
        int o = 0;
        auto R = [&](int n) { int a = o; o += n; return a; };
        L.qpos = R(nq); L.qvel = R(nv); L.ctrl = R(nu); L.warm = R(nv);
        L.xpos = R(3 * nb); L.xmat = R(9 * nb); L.xipos = R(3 * nb); L.cdof = R(6 * nv); L.gcen = R(3 * ng);
        L.M = R(msize); L.L = R(msize); o = (o + 3) & ~3; L.Minv = R(64 * ntree);
        L.bias = R(nv); L.fsm = R(nv); L.asm_ = R(nv); L.qacc = R(nv); L.fcon = R(nv);
        // Newton scratch (packed Hessian, gradient, direction, per-row J.dl) lives over xpos..gcen where it fits: every
        // position-derived quantity is dead between make_constraints and the next substep's kinematics
        {
            int nvh = nv * (nv + 1) / 2, need1 = nvh + 2 * nv, need2 = need1 + maxefc, avail = 15 * nb + 6 * nv + 3 * ng;
            int base = need1 <= avail ? L.xpos : R(need1);
            L.nH = base; L.ng = base + nvh; L.ndl = L.ng + nv;
            L.njv = need2 <= avail ? L.ndl + nv : R(maxefc);
        }
        L.U = o;
        int a = o;
        L.cinert = a; a += 10 * nb; L.binert = a; a += 10 * nb; L.cvel = a; a += 6 * nb; L.cacc = a; a += 6 * nb; L.cfrc = a; a += 6 * nb;
        int bq = o;
        L.cdist = bq; bq += maxcon; L.cpos = bq; bq += 3 * maxcon; L.cnrm = bq; bq += 3 * maxcon;
        bq = (bq + 3) & ~3; L.rowS = bq; L.scr = bq; bq += RS_S * maxefc;
        L.maxgrp = maxefc / 3 + 8;
        if (bq < L.scr + 64 * SLOT_W + 4 * 56) bq = L.scr + 64 * SLOT_W + 4 * 56;     // narrow phase: 64 result slots + 4 box work areas
        o = a > bq ? a : bq;
        L.nreal = (o + 3) & ~3;
        int io = 0;
        auto Iq = [&](int n) { int x = io; io += n; return x; };
        L.cand = Iq(ANC_MAX); L.cpair = Iq(maxcon); L.cefc = Iq(maxcon); L.rmeta = Iq(maxefc); L.rowI = Iq(maxefc); L.gI = Iq(maxefc / 3 + 8); L.misc = Iq(12); L.nprof = Iq(16);
        L.nint = (io + 3) & ~3;
        L.maxcon = maxcon;
        L.maxefc = maxefc;
        L.expcon = this->maxcon;
        L.gefc = this->maxefc;
        L.ggrp = this->maxefc / 3 + 8;
        size_t rs = f64 ? 8 : 4;
        L.bytes_per_env = (int)((L.nreal * rs + (size_t)L.nint * 4 + 15) & ~(size_t)15);
        if (getenv("AVSIM_DEBUG_LAYOUT"))
            fprintf(stderr, "avsim layout: %d reals + %d ints = %d B per env (U at %d: phase A %d, phase B %d words); tables %zu B\n", L.nreal, L.nint,
                    L.bytes_per_env, L.U, a - L.U, bq - L.U, (size_t)moff.nreal * rs + (size_t)moff.nint * 4);
    }

    int dims[7] = {0, 0, 0, 0, 0, 0, 0};
    size_t lds_bytes() const { return (size_t)lay.bytes_per_env + (size_t)moff.nreal * (f64 ? 8 : 4) + (size_t)moff.nint * 4; }   // WPB = 1 figure (first pass)
    void* d_rows = nullptr;
    void* d_coup = nullptr;
    int* d_near = nullptr;
    void* d_gref = nullptr;
    void alloc_contacts() {
        if (d_rows) (void)hipFree(d_rows);
        d_rows = nullptr;
        if (hipMalloc(&d_rows, (size_t)N * maxefc * ROW_S * (f64 ? 8 : 4) * 2) != hipSuccess) throw std::runtime_error("hipMalloc of the constraint row buffers failed");
        mf.rJ_glob = (float*)d_rows; md.rJ_glob = (double*)d_rows;
        mf.rB_glob = mf.rJ_glob + (size_t)N * maxefc * ROW_S; md.rB_glob = md.rJ_glob + (size_t)N * maxefc * ROW_S;
        if (d_coup) (void)hipFree(d_coup);
        if (d_near) (void)hipFree(d_near);
        if (d_gref) (void)hipFree(d_gref);
        d_coup = nullptr; d_near = nullptr; d_gref = nullptr;
        if (hipMalloc(&d_gref, ((size_t)N * dims[4] * 3 + (size_t)N * 64 * BOX_OVF_W) * (f64 ? 8 : 4)) != hipSuccess) throw std::runtime_error("hipMalloc of the Verlet reference buffer failed");
        mf.gref_glob = (float*)d_gref; md.gref_glob = (double*)d_gref;
        mf.bxo_glob = mf.gref_glob + (size_t)N * dims[4] * 3; md.bxo_glob = md.gref_glob + (size_t)N * dims[4] * 3;      // box-box overflow records behind the reference centres
        if (hipMalloc(&d_coup, (size_t)N * lay2.maxgrp * GA_W * (f64 ? 8 : 4)) != hipSuccess || hipMalloc((void**)&d_near, (size_t)N * (NEAR_MAX + CAND_MAX) * 4) != hipSuccess)
            throw std::runtime_error("hipMalloc of the coupling / neighbour buffers failed");
        mf.gA_glob = (float*)d_coup; md.gA_glob = (double*)d_coup; mf.near_glob = d_near; md.near_glob = d_near; mf.cand_glob = md.cand_glob = d_near + (size_t)N * NEAR_MAX;
        kargs_dirty = true;
        if (d_cpairs) (void)hipFree(d_cpairs);
        if (d_cdist) (void)hipFree(d_cdist);
        d_cpairs = nullptr; d_cdist = nullptr;
        if (hipMalloc((void**)&d_cpairs, (size_t)N * maxcon * 8) != hipSuccess || hipMalloc((void**)&d_cdist, (size_t)N * maxcon * 8) != hipSuccess)
            throw std::runtime_error("hipMalloc of the contact export buffers failed");
        (void)hipMemset(d_cpairs, 0xff, (size_t)N * maxcon * 8);
        (void)hipMemset(d_cdist, 0, (size_t)N * maxcon * 8);
    }
    bool init(const Blob& b, int N_, bool f64_, std::string& err) {
        N = N_;
        f64 = f64_;
        try {
            if (f64) build(b, md); else build(b, mf);
            static const int mx[5] = {4, 4, 5, 3, 4};
            max_reward = mx[b.scalar("task_id")];
            int ms = f64 ? md.msize : mf.msize;
            // row / contact capacities per task: every box of a compound object resting on the condim-6 table
            // contributes 4 contacts x 6 rows (SewNeedle 24 contacts / 128 rows, TubeTransfer 40 / 248 at rest)
            // Two tiers where the smaller first one lets more envs share a CU (SewNeedle: 7 instead of 6 -- the scripted grasp of
            // BASELINE config 3 reaches 194 rows / 35 contacts in most envs at once, so the first tier must hold that: with 176 rows,
            // 8 per CU, nearly every env needed the full record during the grasp; TubeTransfer): the second pass costs a launch and,
            // when its list is not empty, the latency of one env-step, so the other tasks keep one tier.
            static const int cap_efc[5] = {176, 176, 336, 480, 176}, cap_con[5] = {48, 48, 72, 96, 48};
            static const int cap_efc1[5] = {176, 176, 224, 288, 176}, cap_con1[5] = {48, 48, 56, 64, 48};
            maxefc = cap_efc[b.scalar("task_id")];
            maxcon = cap_con[b.scalar("task_id")];
            maxefc1 = cap_efc1[b.scalar("task_id")];
            maxcon1 = cap_con1[b.scalar("task_id")];
            dims[0] = b.scalar("nq"); dims[1] = b.scalar("nv"); dims[2] = b.scalar("nu"); dims[3] = b.scalar("nbody"); dims[4] = b.scalar("ngeom"); dims[5] = ms; dims[6] = b.scalar("ntree");
            make_layout(dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], dims[6]);
            d_qpos_home = up(b.f("qpos_home"));
            d_ctrl_home = up(b.f("ctrl_home"));
            d_obj_qadr = up(b.i("objects_qposadr"));
            d_ncon = up(std::vector<int>((size_t)N, 0));
            d_diag = up(std::vector<int>((size_t)N * 4, 0));
            d_cost = up(std::vector<int>((size_t)N, 0));
            d_order = up(std::vector<int>((size_t)N, 0));
            d_retry = up(std::vector<int>(2 * (size_t)N + 2, 0));      // two counters, the list, one flag per env
            alloc_contacts();
        } catch (const std::exception& e) {
            err = std::string("physics init: ") + e.what();
            return false;
        }
        return true;
    }
    void destroy() {
        for (void* p : allocs) (void)hipFree(p);
        allocs.clear();
        if (d_cpairs) (void)hipFree(d_cpairs);
        if (d_cdist) (void)hipFree(d_cdist);
        if (d_rows) (void)hipFree(d_rows);
        if (d_coup) (void)hipFree(d_coup);
        if (d_near) (void)hipFree(d_near);
        if (d_gref) (void)hipFree(d_gref);
        d_cpairs = nullptr; d_cdist = nullptr; d_rows = nullptr; d_coup = nullptr; d_near = nullptr; d_gref = nullptr;
    }
    bool set_option(const char* name, double v) {
        std::string n(name);
        kargs_dirty = true;
        if (n == "pgs_iters") { pgs_iters = (int)v; return true; }
        if (n == "solver") { if (v != 0 && v != 1) return false; mf.solver = md.solver = (int)v; return true; }
        if (n == "newton_iters") { if (v < 1 || v > 100) return false; mf.newton_iters = md.newton_iters = (int)v; return true; }
        if (n == "newton_tol") { if (v < 0) return false; mf.newton_tol = (float)v; md.newton_tol = v; return true; }
        if (n == "ls_tolerance") { if (!(v > 0) || v >= 1) return false; mf.ls_tolerance = (float)v; md.ls_tolerance = v; return true; }
        if (n == "ls_iterations") { if (v < 1 || v > 1000) return false; mf.ls_iterations = md.ls_iterations = (int)v; return true; }
        if (n == "export_contacts") { export_contacts = v != 0; return true; }
        if (n == "num_joints") { if (v != 14 && v != 21) return false; mf.nj = md.nj = (int)v; return true; }
        if (n == "order_envs") { order_envs = v != 0; return true; }
        if (n == "pair_waves") { pair_waves = v != 0; return true; }
        if (n == "qcqp_tridiag") { mf.qcqp_tridiag = md.qcqp_tridiag = v < 0 ? 0 : (v > 2 ? 2 : (int)v); return true; }
        if (n == "noslip_per_tree") { mf.noslip_per_tree = md.noslip_per_tree = v != 0; return true; }
        if (n == "noslip_trees") { mf.noslip_trees = md.noslip_trees = v != 0; return true; }
        if (n == "newton_component") { mf.newton_component = md.newton_component = v != 0; return true; }
        if (n == "newton_early_exit") { mf.newton_early_exit = md.newton_early_exit = v != 0; return true; }
        if (n == "persist_blocks") { int x = (int)v; if (x >= 1 && x <= 64) { persist_over = x; return true; } return false; }
        if (n == "waves_per_block") { int x = (int)v; if (x >= 0 && x <= AVSIM_PHYS_MAXW) { wpb_override = x; return true; } return false; }
        if (n == "profile_phases") {
            if (v != 0 && !d_prof) d_prof = up(std::vector<long long>((size_t)N * PROF_W, 0));
            if (v == 0) d_prof = nullptr;
            return true;
        }
        if (n == "maxefc" || n == "maxcon" || n == "maxefc_first" || n == "maxcon_first") {
            // "maxefc" / "maxcon" set both tiers (one pass with exactly this capacity); "*_first" the first tier alone
            int x = (int)v;
            if (x < 16 || x > 1000) return false;
            (void)hipDeviceSynchronize();
            if (n == "maxefc") maxefc = maxefc1 = x; else if (n == "maxcon") maxcon = maxcon1 = x;
            else if (n == "maxefc_first") maxefc1 = x < maxefc ? x : maxefc; else maxcon1 = x < maxcon ? x : maxcon;
            make_layout(dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], dims[6]);
            alloc_contacts();           // a failed hipMalloc throws: avsim_set_option reports it as AVSIM_EHIP, not as an unknown option
            return true;
        }
        return false;
    }

    // envs (wavefronts) per workgroup for a layout: as many as fit next to one copy of the tables in 160 KiB, at most MAXW
    template <typename real, int MAXW>
    int waves_per_block(const Layout& L) const {
        const size_t tables = (size_t)moff.nreal * sizeof(real) + (size_t)moff.nint * 4;
        int wpb = tables < 160 * 1024 ? (int)((160 * 1024 - tables) / (size_t)L.bytes_per_env) : 0;
        if (wpb > MAXW) wpb = MAXW;
        if (wpb_override > 0) wpb = wpb_override < MAXW ? wpb_override : MAXW;
        return wpb < 1 ? 1 : wpb;
    }

    template <typename real, int G, int MAXW>
    int launch_t(hipStream_t st, const DevModel<real>& m, int nsub, const float* action, void* qpos, void* qvel, void* ctrl, void* warm,
                 int* latch, double* agent, int32_t* reward, uint8_t* success, std::string& err) {
        const size_t tables = (size_t)moff.nreal * sizeof(real) + (size_t)moff.nint * 4;
        auto kern1 = k_phys<real, G, MAXW, false>;      // one pass
        auto kern2 = k_phys<real, G, MAXW, true>;       // the two passes of the two-tier capacities
        if (!attr_done) {    // once per handle (= per device): a second handle on another GPU of the same process sets its own
            hipError_t e = hipFuncSetAttribute((const void*)kern1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { err = std::string("hipFuncSetAttribute: ") + hipGetErrorString(e); return -3; }
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) num_cu = prop.multiProcessorCount;
            if (!d_head) {
                if (hipMalloc((void**)&d_head, 2 * sizeof(int)) != hipSuccess) { err = "hipMalloc(work counters) failed"; return -3; }
                allocs.push_back(d_head);
                (void)hipMemset(d_head, 0, 2 * sizeof(int));
            }
            attr_done = 1;
        }
        const bool two = two_pass();
        int wpb = waves_per_block<real, MAXW>(lay);
        const int wpb2 = waves_per_block<real, MAXW>(lay2);
        // Fewer envs than the chip has wave slots: spread them over ALL CUs instead of filling half of them with two waves per SIMD (a lone wave per
        // SIMD steps an env 10 % faster): 1024 envs = four waves on each of 256 CUs, 504 -> 538 k env-steps/s; 16 envs 1.80 -> 1.66 ms per launch
        // (round 6).  One-tier launches only (the wave pairs of the two-tier launch want their even / odd partners).
        if (!two && wpb_override <= 0 && num_cu > 0 && N < num_cu * wpb) { wpb = (N + num_cu - 1) / num_cu; if (wpb < 1) wpb = 1; }
        const size_t pairflags = two ? 2 * (MAXW / 2 > 0 ? MAXW / 2 : 1) * sizeof(int) : 0;
        const size_t shmem = (size_t)lay.bytes_per_env * wpb + tables + pairflags, shmem2 = (size_t)lay2.bytes_per_env * wpb2 + tables + pairflags;
        // pairs of waves take the envs that need the full capacities inside the first pass when the full record fits two small ones
        const int pairing = (two && wpb >= 2 && lay2.bytes_per_env <= 2 * lay.bytes_per_env && pair_waves) ? 8 : 0;
        if (shmem > 160 * 1024 || (two && shmem2 > 160 * 1024)) { err = "per-block LDS exceeds 160 KiB; lower maxefc/maxcon"; return -1; }
        // persistent blocks: as many as the CUs hold at once (LDS bound), never more than the envs need
        const int per_cu = (int)((160 * 1024) / shmem) > 0 ? (int)((160 * 1024) / shmem) : 1;
        int nblk = num_cu * per_cu * (persist_over > 0 ? persist_over : 1);
        if (nblk > (N + wpb - 1) / wpb) nblk = (N + wpb - 1) / wpb;
        if (kargs_dirty) {
            KArgs<real> ka{m, lay, moff}, ka2{m, lay2, moff};
            if (!d_kargs) { if (hipMalloc(&d_kargs, sizeof(KArgs<double>)) != hipSuccess) { err = "hipMalloc(kernel arguments) failed"; return -3; } allocs.push_back(d_kargs); }
            if (!d_kargs2) { if (hipMalloc(&d_kargs2, sizeof(KArgs<double>)) != hipSuccess) { err = "hipMalloc(kernel arguments) failed"; return -3; } allocs.push_back(d_kargs2); }
            (void)hipStreamSynchronize(st);
            if (hipMemcpy(d_kargs, &ka, sizeof(ka), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_kargs2, &ka2, sizeof(ka2), hipMemcpyHostToDevice) != hipSuccess) { err = "hipMemcpy(kernel arguments) failed"; return -3; }
            kargs_dirty = false;
        }
        // (checked BEFORE anything is enqueued: set_option admits 16 .. 1000, so this cannot fire through the API)
        // the second pass gets the first tier's capacities in two 10-bit fields of retry_mode, above the three mode bits and the pairing bit
        static_assert((8 | 7) < (1 << 8), "retry_mode: mode bits and pairing flag below bit 8");
        if (two && (maxcon1 >= 1024 || maxefc1 >= 1024 || maxcon1 < 0 || maxefc1 < 0)) { err = "first-tier capacities do not fit the 10-bit fields of retry_mode (maxcon_first / maxefc_first < 1024)"; return -1; }
        // launch order from the previous step's per-env cost
        const int* order = nullptr;
        if (order_envs && have_cost && nsub > 0 && N > wpb) {
            hipLaunchKernelGGL(k_env_order, dim3(1), dim3(1024), 0, st, (const int*)d_cost, d_order, N);
            order = d_order;
        }
        const int want_reward = (reward || success) && (nsub > 0 || force_reward) ? 1 : 0;
        const int par = (int)(retry_parity & 1);      // which of the two list counters this step uses (its second pass zeroes the other)
        int* const rlist = d_retry + 2;
        {
            int* head = d_head + (launch_count & 1);
            int* next = d_head + ((launch_count + 1) & 1);
            launch_count++;
            hipLaunchKernelGGL(two ? kern2 : kern1, dim3(nblk), dim3(64 * wpb), shmem, st, (KPtr<real>)d_kargs, (const real*)d_img_real, (const int*)d_img_int, N, nsub, pgs_iters, action, want_reward,
                               (real*)qpos, (real*)qvel, (real*)ctrl, (real*)warm, latch, agent, (int*)reward, (unsigned char*)success, d_ncon,
                               d_cpairs, d_cdist, d_diag, max_reward, export_contacts, d_prof, d_xpose, order, order_envs ? d_cost : (int*)nullptr, head, next,
                               d_retry, two ? (1 + 2 * par) | pairing : 0, (KPtr<real>)d_kargs2);
        }
        if (two) {
            // second pass: the envs the first one gave up on, with the full capacities; one workgroup per CU is plenty for the few there
            // are (the workgroups loop over the list), and a launch that finds the list empty returns at once
            int* head = d_head + (launch_count & 1);
            int* next = d_head + ((launch_count + 1) & 1);
            launch_count++;
            retry_parity++;
            int nblk2 = num_cu;
            if (nblk2 > (N + wpb2 - 1) / wpb2) nblk2 = (N + wpb2 - 1) / wpb2;
            hipLaunchKernelGGL(kern2, dim3(nblk2), dim3(64 * wpb2), shmem2, st, (KPtr<real>)d_kargs2, (const real*)d_img_real, (const int*)d_img_int, N, nsub, pgs_iters, action, want_reward,
                               (real*)qpos, (real*)qvel, (real*)ctrl, (real*)warm, latch, agent, (int*)reward, (unsigned char*)success, d_ncon,
                               d_cpairs, d_cdist, d_diag, max_reward, export_contacts, d_prof, d_xpose, (const int*)rlist, order_envs ? d_cost : (int*)nullptr, head, next,
                               d_retry, (2 + 2 * par) | (maxcon1 << 8) | (maxefc1 << 18), (KPtr<real>)d_kargs2);      // (+ the FIRST tier's capacities: what "needed the full record" is measured against)
        }
        if (nsub > 0) have_cost = true;
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { err = std::string("physics kernel launch: ") + hipGetErrorString(e); return -3; }
        return 0;
    }

#ifndef AVSIM_TU_F64
    int launch(hipStream_t st, int N_, int nsub, const float* action, int nj, void* qpos, void* qvel, void* ctrl, void* warm, int* latch,
               double* agent, int32_t* reward, uint8_t* success, std::string& err) {
        (void)N_; (void)nj;
        // the double-precision kernel lives in its own translation unit (avsim_phys_f64.hip), compiled without FMA contraction
        if (f64) return phys_launch_f64(*this, st, nsub, action, qpos, qvel, ctrl, warm, latch, agent, reward, success, err);
        // as many envs (wavefronts) per block as fit next to one copy of the tables in 160 KiB, at most 8 (two per SIMD).
        // Build option (round 6, profiles/r06_experiments.txt 3): -DAVSIM_PHYS_MAXW=12 "-DAVSIM_PHYS_ATTR=__attribute__((amdgpu_waves_per_eu(3)))" compiles
        // the f32 kernel for three waves per SIMD (168 VGPRs) and lets a workgroup hold up to twelve envs where the LDS record allows
        // (options maxefc_first / maxcon_first shrink it: 96 rows / 24 contacts = 14.6 KB = ten envs per CU for SlotInsertion)
        return launch_t<float, 64, AVSIM_PHYS_MAXW>(st, mf, nsub, action, qpos, qvel, ctrl, warm, latch, agent, reward, success, err);
    }
#endif
};

}  // namespace avs
