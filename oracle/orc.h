/* oracle/orc.h -- CPU restatement (f64, scalar, plain C) of the AV-ALOHA hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the
 * product (av_aloha_amd/, libavsim.so).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load liborc.so, and only as the checker / the timed CPU baseline.
 *
 * What is pinned and what is not:
 *  - IK / kinematics / SO(3) helpers / reward predicates follow the reference's own Python
 *    (data_collection_scripts/{transform_utils,kinematics,diff_ik,grad_ik}.py,
 *    gym_guided_vision/gym_guided_vision/env.py get_reward x5) and are checked against golden
 *    vectors produced by importing that Python (tests/golden/gen_golden.py).
 *  - The physics (env.py:218 -> MuJoCo mj_step, un-vendored third-party C library, mujoco ^3.2.2
 *    per gym_guided_vision/pyproject.toml:11) restates MuJoCo's *documented* pipeline with the
 *    Newton solver the reference runs (MuJoCo default) and the PGS solver BASELINE.json's
 *    north_star names (orc_data.solver), the options the reference's XML sets (aloha_sim.xml:4-5:
 *    elliptic cones, impratio 100, noslip_iterations 3 as mj_solNoSlip's per-contact QCQP,
 *    multiccd perturbation contacts) and MuJoCo's defaults for the rest.  MuJoCo cannot be
 *    imported or built here, the reference ships no golden trajectories: PARITY UNPINNED at the
 *    MuJoCo boundary.  tests/golden/gen_mujoco_traj.py records such trajectories where MuJoCo is
 *    installed; tests/test_mujoco_pin.py compares this oracle and the device with them.
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* flop-counting build (tools/count_flops.py, oracle/count/cdouble.h): the phase the counted operations are booked under */
enum { ORC_PH_OTHER = 0, ORC_PH_KINEMATICS, ORC_PH_CRB, ORC_PH_COLLIDE, ORC_PH_RNE, ORC_PH_SMOOTH, ORC_PH_ROWS, ORC_PH_NEWTON, ORC_PH_NOSLIP,
       ORC_PH_EULER, ORC_PH_IK, ORC_PH_COUNT };
#ifdef ORC_COUNT_FLOPS
#define ORC_PHASE(k) (orc_phase = (k))
#else
#define ORC_PHASE(k) ((void)0)
#endif

#define ORC_MAXCON 128   /* (64 / 400 until round 4: box-box manifolds of up to 8 points need more) */
#define ORC_MAXEFC 640
#define ORC_MAXNV 48

enum { ORC_FREE = 0, ORC_BALL = 1, ORC_SLIDE = 2, ORC_HINGE = 3 };
enum { ORC_SPHERE = 2, ORC_CYLINDER = 5, ORC_BOX = 6, ORC_MESH = 7 };
enum { ORC_EQ = 0, ORC_FLOSS = 1, ORC_LIMIT = 2, ORC_CONTACT = 3 };

typedef struct {
    /* sizes */
    int nq, nv, nu, nbody, njnt, ngeom, npair, ntree, neq, task_id, num_arms, nhullvert;
    double timestep, gravity[3], impratio, meaninertia;
    int noslip_iterations, cone_elliptic;
    /* bodies */
    const int *body_parent, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum, *body_weldid, *body_tree;
    const double *body_pos, *body_quat, *body_mass, *body_ipos, *body_inertia, *body_invweight0;
    /* joints / dofs */
    const int *jnt_type, *jnt_body, *jnt_qposadr, *jnt_dofadr, *jnt_limited, *jnt_actfrclimited;
    const double *jnt_pos, *jnt_axis, *jnt_range, *jnt_actfrcrange, *jnt_solref, *jnt_solimp, *jnt_margin;
    const int *dof_body, *dof_jnt, *dof_parent, *dof_tree, *tree_dofadr, *tree_dofnum;
    const double *dof_armature, *dof_damping, *dof_frictionloss, *dof_invweight0, *dof_solref, *dof_solimp;
    /* actuators, equalities */
    const int *act_dof, *act_qposadr, *act_ctrllimited;
    const double *act_kp, *act_kv, *act_gear, *act_ctrlrange;
    const int *eq_dof1, *eq_dof2, *eq_qpos1, *eq_qpos2;
    const double *eq_polycoef, *eq_solref, *eq_solimp;
    /* geoms, hulls, pairs */
    const int *geom_type, *geom_body, *geom_hull, *geom_class;
    const double *geom_pos, *geom_quat, *geom_size, *geom_bcenter, *geom_rbound, *hull_vert;
    const int *pair_geom, *pair_condim;
    const double *pair_friction, *pair_solref, *pair_solimp, *pair_margin, *pair_gap;
    /* poses, obs, IK */
    const double *qpos0, *qpos_home, *ctrl_home, *obs_offset, *obs_scale, *grip_range;
    const int *obs_qposadr, *obs_dofadr, *objects_qposadr;
    int nobj;
    const int *ik_n, *ik_qadr;
    const double *ik_w0, *ik_p0, *ik_site0, *ik_range;
    void* blob; /* owned copy */
    void* hull_override; /* owned: orc_model_set_hulls */
    /* depth render (orc_render.c): hull half-spaces, visibility, cameras */
    int ncam;
    const int *geom_hplane, *geom_visible, *cam_body;
    const double *hull_plane, *cam_pos, *cam_quat, *cam_fovy, *cam_clip;
    const double *geom_rgba, *render_light; /* colour render: material colours, lights + sky (compile.py) */
} orc_model;

typedef struct {
    double dist, pos[3], frame[9]; /* frame rows: normal (geom1->geom2), tangent1, tangent2 */
    int geom1, geom2, pair, dim, efc_adr; /* efc_adr = -1 when the contact makes no rows (dist >= margin-gap) */
    double friction[5], solref[2], solimp[5], includemargin;
} orc_contact;

typedef struct {
    const orc_model* m;
    /* state */
    double *qpos, *qvel, *ctrl, *qacc_warmstart;
    double time;
    int threaded; /* SewNeedle latch (env.py:602, 686) */
    /* position-dependent */
    double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *cdof; /* cdof: nv x 6 [ang; lin] about world origin */
    double *geom_xpos, *geom_xmat;
    double *M, *L;            /* dense nv x nv mass matrix and its Cholesky factor (block diagonal) */
    /* velocity-dependent and forces */
    double *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint, *qacc;
    /* contacts and constraints */
    int ncon, nefc;
    orc_contact contact[ORC_MAXCON];
    double *efc_J;            /* nefc x nv dense */
    double *efc_B;            /* nefc x nv: rows of J M^-1 */
    double efc_pos[ORC_MAXEFC], efc_margin[ORC_MAXEFC], efc_aref[ORC_MAXEFC], efc_R[ORC_MAXEFC],
        efc_D[ORC_MAXEFC], efc_force[ORC_MAXEFC], efc_diag[ORC_MAXEFC], efc_floss[ORC_MAXEFC], efc_KBIP[ORC_MAXEFC * 4];
    int efc_type[ORC_MAXEFC], efc_id[ORC_MAXEFC];
    int pgs_iters;
    double pgs_tol, pgs_scale; /* early-termination tolerance (0 = fixed sweep count) and 1/(meaninertia*nv) */
    int stat_sweeps;
    int solver;       /* 0 = PGS (north_star), 1 = Newton (the reference's MuJoCo default) */
    int newton_iters; double newton_tol;
    int overflow; /* set when ORC_MAXCON / ORC_MAXEFC was hit */
    long stat_narrow; /* narrow-phase calls, for the flop/pair accounting */
    int stat_noslip;  /* noslip sweeps used by the last solve */
    /* Newton's exact line search: stop when |phi'(alpha)| < ls_tol |phi'(0)|, at most ls_iters evaluations after phi'(0).  MuJoCo's own
     * search [EXT mjOption.ls_tolerance = 0.01, ls_iterations = 50] stops at a derivative of tolerance x ls_tolerance x |search vector| /
     * scale; this restatement searches to 1e-10 RELATIVE by default (both sides then take the same step to rounding, which the whole-episode
     * parity tests need) -- a listed deviation; the device's twin options are "ls_tolerance" / "ls_iterations" */
    double ls_tol; int ls_iters;
} orc_data;

/* model / data */
orc_model* orc_model_load(const void* blob, size_t nbytes);
void orc_model_free(orc_model* m);
orc_data* orc_data_new(const orc_model* m);
int orc_capacity(int which);   /* 0: ORC_MAXCON, 1: ORC_MAXEFC, 2: sizeof(orc_data) -- for the Python mirror of the struct (tests/orc_env.py) */
void orc_data_free(orc_data* d);

/* depth image of camera `cam` at the current state (positions must be fresh: orc_forward / orc_step): float32 metres along
 * the optical axis, out[H][W] row 0 = top; returns the number of pixels that hit a geom */
int orc_render_depth(const orc_data* d, int cam, int H, int W, float* out);
int orc_render_depth_rows(const orc_data* d, int cam, int H, int W, int row0, int row_step, int nrows, float* out);
int orc_render_rgb(const orc_data* d, int cam, int H, int W, unsigned char* out, float* depth);
/* colour image of the visual meshes (orc_vis.c): brute-force ray caster over the expanded scene of compiler/vismesh.py */
int orc_vis_render(const orc_data* d, int cam, int nvert, const double* vert, const int* vbody, int ntri, const int* tri, const double* rgb,
                   const double* uv, const int* tex, const int* texel, int texn, int H, int W, unsigned char* out, int* tri_out, double* depth_out);

int orc_vis_render_ex(const orc_data* d, int cam, int nvert, const double* vert, const int* vbody, int ntri, const int* tri, const double* rgb,
                      const double* uv, const int* tex, const int* texel, int texn, int H, int W, int ss, int shadows, unsigned char* out, int* tri_out, double* depth_out);
/* the same with per-corner lighting normals [ntri][9] (body frame): smooth shading; tnorm NULL = orc_vis_render_ex */
int orc_vis_render_sm(const orc_data* d, int cam, int nvert, const double* vert, const int* vbody, int ntri, const int* tri, const double* rgb,
                      const double* uv, const int* tex, const int* texel, int texn, const double* tnorm, int H, int W, int ss, int shadows, unsigned char* out, int* tri_out, double* depth_out);

/* env-level (env.py:203-249): reset to home pose with given object free-joint poses (nobj x 7) */
void orc_reset(orc_data* d, const double* obj_qpos);
void orc_set_qpos(orc_data* d, const double* qpos);                 /* env.py:251-253 */
void orc_forward(orc_data* d);                                       /* mj_forward */
void orc_step(orc_data* d, int nsub);                                /* nsub x mj_step, then refresh kinematics+collision */
void orc_env_step(orc_data* d, const double* action, int nsub, double* agent_pos, int* reward, int* success);
void orc_agent_pos(const orc_data* d, double* agent_pos);            /* env.py:168-178 */
int orc_reward(orc_data* d);                                         /* env.py get_reward, per task */
int orc_reward_from_pairs(const orc_model* m, const int* geom_pairs, int ncon, int* threaded_latch);
int orc_max_reward(const orc_model* m);

/* stages, exposed for unit tests */
void orc_kinematics(orc_data* d);
void orc_crb(orc_data* d);
void orc_rne_bias(orc_data* d);
void orc_smooth(orc_data* d);   /* passive + actuator forces, smooth acceleration (needs crb + rne_bias) */
void orc_collide(orc_data* d);
void orc_make_constraints(orc_data* d);
void orc_solve(orc_data* d);
void orc_solve_newton(orc_data* d);
void orc_noslip(orc_data* d);

/* box-box manifolds: 8 (default; every clipped vertex, as MuJoCo's mjc_BoxBox [EXT]) or 4 (the reduced manifold of rounds 1-4) */
void orc_set_boxbox_maxpoints(int n);
int orc_get_boxbox_maxpoints(void);
/* replace the mesh geoms' hull vertices (the blob's decimated hulls) by other ones -- the FULL convex hulls of the STL files, as MuJoCo
 * collides them [EXT]: vert double[nvert][3] in the geoms' frames, geom_hull int[ngeom][2] = (first vertex, count) per geom (0, 0 for
 * non-mesh geoms), rbound double[ngeom] bounding radii about geom_bcenter.  Copies are owned by the model.  Collision only: the depth
 * images keep the blob's hull planes. */
void orc_model_set_hulls(orc_model* m, const double* vert, int nvert, const int* geom_hull, const double* rbound);

/* narrow phase entry for tests: geometry types as ORC_*; returns number of contacts (<=8) */
int orc_narrow(int t1, const double* size1, const double* pos1, const double* mat1, const double* hull1, int nh1,
               int t2, const double* size2, const double* pos2, const double* mat2, const double* hull2, int nh2,
               double* dist, double* pos, double* normal);

/* IK path (kinematics.py, diff_ik.py, grad_ik.py, transform_utils.py) */
void orc_quat2mat(const double q_xyzw[4], double R[9]);
void orc_mat2quat(const double R[9], double q_xyzw[4]);
void orc_quat2axisangle(const double q_xyzw[4], double aa[3]);
void orc_axisangle2quat(const double aa[3], double q_xyzw[4]);
void orc_angular_error(const double Rd[9], const double Rc[9], double e[3]);
void orc_exp2mat(const double w[3], const double v[3], double th, double T[16]);
void orc_adjoint(const double T[16], double A[36]);
void orc_limit_pose(const double cp[3], const double cR[9], const double tp[3], const double tR[9], double maxp,
                    double maxr, double op[3], double oR[9]);
void orc_fk(const orc_model* m, int arm, const double* q, double T[16]);
void orc_jac(const orc_model* m, int arm, const double* q, double* J /* 6 x n */);
void orc_diffik(const orc_model* m, int arm, const double* q, const double pos[3], const double quat_wxyz[4],
                double k_pos, double k_ori, double damping, const double* k_null, const double* q0, double max_angvel,
                double dt, int iterations, double* q_out);
void orc_gradik(const orc_model* m, int arm, const double* q, const double pos[3], const double quat_wxyz[4],
                double* q_out);
void orc_diffik_R(const orc_model* m, int arm, const double* q, const double pos[3], const double Rt[9], double k_pos,
                  double k_ori, double damping, const double* k_null, const double* q0, double max_angvel, double dt,
                  int iterations, double* q_out);
void orc_gradik_R(const orc_model* m, int arm, const double* q, const double pos[3], const double Rt[9], double* q_out);
/* sim_env.py:277-301 mapping: 23-D Cartesian action -> 21 ctrl (mode 0: GradIK,GradIK,DiffIK as the reference;
   mode 1: DiffIK on all three arms as north_star) */
void orc_cart_to_ctrl(const orc_data* d, const double* action23, int mode, double* action21);

#ifdef __cplusplus
}
#endif
#endif
