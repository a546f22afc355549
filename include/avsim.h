/* include/avsim.h -- C-ABI of libavsim.so: the MI355X-native batched replacement for the hot path
 * of the AV-ALOHA gym_guided_vision environments.
 *
 * Boundary it replaces (reference file:line, all under /root/reference):
 *   gym_guided_vision/gym_guided_vision/env.py:36-166   GuidedVisionEnv.__init__  -> avsim_create
 *   env.py:228-249 (+ task overrides :474-501,:513-543,:604-637,:705-735,:792-818) reset -> avsim_reset
 *   env.py:203-226 step / :255-269 step_action (20 x MuJoCo mj_step, env.py:218)   -> avsim_step
 *   env.py:168-178 get_obs agent_pos, :425-863 get_reward x5, :224 is_success       -> outputs of avsim_step
 *   env.py:251-253 set_qpos                                                          -> avsim_set_qpos
 *   data_collection_scripts/sim_env.py:277-312 step with IK (GradIK/DiffIK)          -> avsim_step_cartesian
 *   data_collection_scripts/diff_ik.py:89-90 DiffIK.run, grad_ik.py:150-166 GradIK.run -> avsim_ik
 *
 * Conventions: every entry point returns 0 on success or a negative AVSIM_E* code; the message is
 * available from avsim_last_error().  All bulk pointers are HOST pointers unless the handle was
 * created with AVSIM_IO_DEVICE, in which case they are device pointers on the handle's device and
 * work is enqueued on the handle's stream without synchronising (call avsim_sync).  The library
 * never keeps a caller pointer past the call.  One host thread per handle; handles are independent.
 * There is NO CPU fallback: creation fails if no gfx950 device is usable.
 */
#ifndef AVSIM_H
#define AVSIM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct avsim avsim_t;

enum {
    AVSIM_OK = 0,
    AVSIM_EINVAL = -1,  /* bad argument (python facade: AssertionError / ValueError) */
    AVSIM_ENODEV = -2,  /* no usable HIP device */
    AVSIM_EHIP = -3,    /* HIP runtime error, text in avsim_last_error */
    AVSIM_EMODEL = -4,  /* malformed model blob */
    AVSIM_ENOTIMPL = -5 /* python facade: NotImplementedError (env.py:30) */
};

enum {
    AVSIM_IO_DEVICE = 1u << 0, /* bulk I/O pointers are device pointers */
    AVSIM_F64_PHYSICS = 1u << 1 /* debug: run the physics kernels in double precision */
};

/* IK controller selection for avsim_step_cartesian / avsim_ik */
enum {
    AVSIM_IK_REFERENCE = 0, /* left/right GradIK, middle DiffIK (sim_env.py:89-138) */
    AVSIM_IK_DLS = 1        /* damped least squares on all three arms (BASELINE.json north_star) */
};

/* dims[]: 0 nq, 1 nv, 2 nu (actuators, 21), 3 num_joints of the action/agent_pos (14|21), 4 nobj (free objects),
 * 5 max_reward, 6 num_envs, 7 task id, 8 ncon capacity, 9 nefc capacity (the full tier), 10 LDS bytes per block (one env per
 * wavefront with the first tier's record + the hot model tables), 11 blocks that fit one CU's 160 KiB */
#define AVSIM_NDIMS 12

/* Build a batched simulation from a compiled model blob (av_aloha_amd/compiler, replaces env.py:53-56).
 * num_arms is encoded in the blob (2-arm blobs carry the hidden middle arm of env.py:394-395). */
int avsim_create(const void* model_blob, size_t nbytes, int num_envs, int device, uint32_t flags, avsim_t** out);
void avsim_destroy(avsim_t* h);
const char* avsim_last_error(const avsim_t* h); /* h may be NULL for creation errors */
int avsim_dims(const avsim_t* h, int32_t dims[AVSIM_NDIMS]);

/* Solver / capacity / debug knobs (returns AVSIM_EINVAL for an unknown name or a value out of range):
 *   "solver"            0 PGS (BASELINE north_star), 1 Newton (MuJoCo's default, what the reference runs; default)
 *   "pgs_iters"         Gauss-Seidel sweeps of the PGS solver (default 20); "newton_iters" cap (default 100 = MuJoCo), "newton_tol" (1e-8 in f64 = MuJoCo, 1e-6 in f32)
 *   "ls_tolerance", "ls_iterations"   Newton's exact line search: stop when |phi'(alpha)| < ls_tolerance x |phi'(0)|, at most ls_iterations evaluations
 *                       after the one at alpha = 0.  Defaults 1e-10 (f64) / 1e-4 (f32) and 50.  MuJoCo's mjOption.ls_iterations is 50 and its
 *                       ls_tolerance 0.01, applied to a differently scaled derivative [EXT]: the default here searches much further than MuJoCo
 *                       does (the whole-episode parity tests need both sides to take the same step to rounding) -- a listed deviation, DESIGN.md 2
 *   "maxefc", "maxcon"  constraint rows / contacts an env can hold (per-task defaults 176-480 / 48-96): the stride of the contact export
 *                       and of the global row scratch; setting one makes it the capacity of a single tier (one pass)
 *   "maxefc_first", "maxcon_first"   the FIRST tier of the two-tier capacities (defaults: SewNeedle 224 / 56 of 336 / 72, TubeTransfer
 *                       288 / 64 of 480 / 96; the other tasks have one tier): a launch steps every env with the LDS record of the first
 *                       tier -- more envs per CU --, and an env that needs more in some substep is stepped again from its untouched
 *                       state with the full capacities: by its wave in the two adjacent records of a wave pair ("pair_waves" 1, the
 *                       default, when the full record fits two small ones), else by a second pass over the list of such envs.
 *                       Results are those of one pass with the full capacities, bit for bit
 *   "qcqp_tridiag"      multiplier iteration of a sliding contact's noslip QCQP: 0 MuJoCo's Cholesky per iterate (f64 default), 1 the
 *                       same iterates on the Householder-tridiagonal form of the friction block, 2 tridiagonal form + secular-equation
 *                       steps (Newton on 1/r - 1/|y|; f32 default): the same multiplier within the iteration's own thresholds
 *   "num_joints"        14 | 21: width of the action / agent_pos rows, whatever the blob's arm count (a 3-arm env whose camera
 *                       arm was parked by hide_middle_arm, env.py:394-395, keeps its 21-D action on the 2-arm model)
 *   "waves_per_block"   envs per workgroup, 0 = as many as fit in 160 KiB of LDS (<= 8)
 *   "noslip_per_tree"   1 (default): the dry-friction rows of the noslip pass are relaxed per kinematic tree, all trees at once
 *                       (models with <= 8 trees); 0 = six rows at a time through the Gauss-Seidel groups (what models with more
 *                       trees get); same results to rounding
 *   "noslip_trees"      1 (default): a noslip pass whose contacts all touch one kinematic tree runs per tree, octet t of the wave on tree t,
 *                       the trees' contact chains side by side; 0 = always the wave-wide Gauss-Seidel groups (same results to rounding)
 *   "newton_component"  1 (default): in a scene where some contact couples two kinematic trees (a needle in a gripper) Newton's dense
 *                       factorisation and substitutions run over the dofs of the coupled trees only, the other trees in their lane
 *                       octets; 0 = over all nv columns (the same bits: the entries in between are zeros)
 *   "newton_early_exit" 1 (default): when a Newton step ends in the active set it started from and no contact of either end is in the cone's
 *                       middle zone, the cost was one quadratic along the step and the point is its minimiser: the solver returns without
 *                       evaluating the gradient that would confirm it (a third of an iteration; same qacc, same forces); 0 = always evaluate
 *   "order_envs"        1 (default): workgroups take the envs in the order of their cost in the previous step, most expensive
 *                       first (results do not depend on it); 0 = in index order
 *   "export_contacts"   0 skips the per-step contact export (avsim_get_contacts); "kernel_timing" 1 brackets every physics
 *                       launch AND every image kernel of avsim_render_depth / the proxy mode of avsim_render_rgb with HIP events
 *                       (avsim_kernel_time, avsim_render_kernel_time; at most 1024 event pairs are kept per list, older ones are folded
 *                       into a running sum); "profile_phases" 1 enables avsim_get_phase_cycles */
int avsim_set_option(avsim_t* h, const char* name, double value);

/* env.py:228-249 + task reset: envs with mask[i]!=0 (NULL = all) go to the home pose, zero velocity,
 * home ctrl; their free objects get obj_qpos[i][nobj][7] = [x y z qw qx qy qz]. Derived quantities are
 * refreshed (mj_forward).  mask: uint8[N]; obj_qpos: double[N][nobj*7]. */
int avsim_reset(avsim_t* h, const uint8_t* mask, const double* obj_qpos);

/* env.py:203-226: action float[N][num_joints] -> ctrl, nsub physics substeps, then
 * agent_pos double[N][num_joints], reward int32[N], success uint8[N]. Any output may be NULL. */
int avsim_step(avsim_t* h, const float* action, int nsub, double* agent_pos, int32_t* reward, uint8_t* success);

/* `physics.step(nstep)` with the control vector as it stands (env.py:218 after env.py:203-215 has written physics.data.ctrl; dm_control's
 * Physics.step): nsub substeps driven by the handle's ctrl -- what avsim_set_state / an earlier step left there --, outputs as avsim_step.
 * A replay of recorded actuator commands and substep-by-substep debugging go through this. */
int avsim_step_ctrl(avsim_t* h, int nsub, double* agent_pos, int32_t* reward, uint8_t* success);

/* sim_env.py:277-312: action double[N][23] Cartesian targets; IK on measured qpos -> ctrl; physics.
 * Outputs as avsim_step (agent_pos has 21 entries per env here). */
int avsim_step_cartesian(avsim_t* h, const double* action23, int ik_mode, int nsub, double* agent_pos,
                         int32_t* reward, uint8_t* success);

/* Stand-alone batched IK (diff_ik.py / grad_ik.py `run`): arm 0 left, 1 right, 2 middle;
 * q double[n][nj], pos double[n][3], quat_wxyz double[n][4] -> q_out double[n][nj] (nj = 6,6,7).
 * controller: 0 DiffIK, 1 GradIK; max_iters <= 0 keeps the reference's iteration count (10 / 50). */
int avsim_ik(avsim_t* h, int arm, int controller, int max_iters, int n, const double* q, const double* pos,
             const double* quat_wxyz, double* q_out);
/* kinematics.py:17-24 / :35-50 batched: T double[n][16], J double[n][6][nj] (either may be NULL) */
int avsim_fk_jac(avsim_t* h, int arm, int n, const double* q, double* T, double* J);

/* env.py:168-178 get_obs (agent_pos) and env.py get_reward / :224 is_success of the CURRENT state, no time
 * stepping; any output may be NULL.  Evaluating the reward advances SewNeedle's latch as env.py:686-689 does. */
int avsim_observe(avsim_t* h, double* agent_pos, int32_t* reward, uint8_t* success);

/* env.py:251-253 set_qpos (all envs, double[N][nq]) followed by forward kinematics + collision */
int avsim_set_qpos(avsim_t* h, const double* qpos);
/* full state for checkpoint / tests: qpos[N][nq], qvel[N][nv], ctrl[N][nu], warmstart[N][nv]; NULL = skip */
int avsim_get_state(avsim_t* h, double* qpos, double* qvel, double* ctrl, double* warmstart);
int avsim_set_state(avsim_t* h, const double* qpos, const double* qvel, const double* ctrl, const double* warmstart);
/* the object poses an env whose state diverged is put back to (what avsim_reset was given; the model's default poses before the
 * first reset): double[N][nobj][7].  Not touched by avsim_set_state / avsim_set_qpos -- a handle that continues another handle's
 * episode carries them over with this pair, next to avsim_get_latch / avsim_set_latch */
int avsim_get_reset_poses(avsim_t* h, double* obj_qpos);
int avsim_set_reset_poses(avsim_t* h, const double* obj_qpos);
/* the per-env reward latch int32[N] (SewNeedle's _threaded_needle, env.py:602, :631, :673, :686-689; 0 for the other tasks): part
 * of an env's state next to qpos / qvel / ctrl -- a checkpoint or a move of the env to another handle carries it along */
int avsim_get_latch(avsim_t* h, int32_t* latch);
int avsim_set_latch(avsim_t* h, const int32_t* latch);
/* contacts of the current state, env.py:436-441 view: ncon int32[N], geom pairs int32[N][cap][2], dist double[N][cap] */
int avsim_get_contacts(avsim_t* h, int32_t* ncon, int32_t* geom_pairs, double* dist);
/* per-env diagnostics of the last step: int32[N][4] = {ncon, nefc, overflow flags, packed}; packed = divergence flag (bit 0: the state became NaN / Inf / > 1e6 during the step and the env was put back to the
 * state its episode started from -- home pose, the objects where the last avsim_reset put them, zero velocity --, as MuJoCo resets
 * its data on a bad state) |
 * broad-phase survivors (bits 8-15) | Newton iterations summed over the substeps (bits 16-27) | their maximum, saturated at 15 (bits 28-31) */
int avsim_get_diag(avsim_t* h, int32_t* diag);

/* debug: shader-clock cycles each env's wave spent in the 8 phases (kinematics, CRB, RNE, smooth, collide, rows,
 * solve, integrate, + broad / narrow phase incl. the trailing refresh) during the last launch; needs
 * avsim_set_option("profile_phases", 1); int64[N][26]: 8 phases, broad, narrow, then the Newton solver's
 * init / gradient / Hessian / factorisation / line search / final forces / noslip / back-substitution cycles and 8 probe slots
 * (noslip: group steps, dry-friction passes, their cycles, look-ahead set-up, entry set-up, whole call; two free) */
int avsim_get_phase_cycles(avsim_t* h, int64_t* out);

/* Replaces the camera part of get_obs / render (gym_guided_vision/gym_guided_vision/env.py:180-188, :195-200; MuJoCo OpenGL
 * renderer) by depth images (BASELINE config 5): out = float32[N][ncam][height][width], metres along the optical axis of
 * camera cam_ids[c] (index into the model's camera table, manifest "camera_names"; avsim_camera_count entries), row 0 = top,
 * pixels that see nothing = far plane (30 m).  Drawn are the collision proxies of the current state.  `out` is a host or a
 * device pointer according to AVSIM_IO_DEVICE; cam_ids is always a host pointer.  Batches of more than 4096 envs (option "render_chunk")
 * go through the kernels in chunks, so that the per-view scratch (~100 KB) is bounded by the chunk. */
int avsim_render_depth(avsim_t* h, const int32_t* cam_ids, int ncam, int height, int width, float* out);

/* The same cameras as colour images, the layout of the reference's "pixels" observation and of render()
 * (env.py:180-188, :195-200): out = uint8[N][ncam][height][width][3] (RGB).  Once avsim_load_visual has run (the Python facades
 * do that on first use) the VISUAL scene is rasterised: the decimated visual meshes of the robots, the frame and camera mounts,
 * the textured table, the task objects (k_vis_render; flat Lambert shading under the scene's headlight scene.xml:9 and directional
 * light :48, table texture, skybox gradient :34).  Without a loaded visual scene, or with option "render_proxies" 1, the collision
 * proxies are drawn in their flat material colours instead (the depth rasteriser's colour variant).  Options of the visual image (round 5):
 * "render_shadows" 1 -- the scene's directional light (scene.xml:48) casts shadows inside its shadow box (<statistic center extent>,
 * scene.xml:6), from a 512 x 512 depth map rendered from the light per env ("render_shadow_size" 1024 | 2048: a finer one, 4 / 16 MB per env); "render_samples" 4 -- 2 x 2 supersampling (MuJoCo's offscreen
 * buffer is multisampled, offsamples default 4 [EXT]).  Both are off at the C-ABI and on in the gym facades.  "render_cam_major" 1 -- out is
 * uint8[ncam][N][height][width][3] (every camera's batch contiguous: the facades hand out one array per camera without copying).  The directional light's specular
 * term (MJCF defaults: light 0.3 x material 0.5, exponent 64) is part of the shade.  "render_smooth" 1 (round 6; on in the gym / Cartesian facades, off
 * at the C-ABI) -- the three corners of a triangle are lit with their own normals (the library's lib_tnorm: area-weighted means over the faces
 * within the crease angle, as MuJoCo generates vertex normals with smoothnormal="false" [EXT]) and the shade is interpolated perspective-correctly
 * over the triangle, as fixed-function GL lights per vertex; 0: one shade per triangle.  No transparency: a stand-in for MuJoCo's OpenGL output,
 * not a pixel match.  A view that runs out of triangle
 * records or tile-list entries sets the overflow flags of avsim_visual_info (the image then lacks triangles).  Pointer conventions
 * as avsim_render_depth. */
int avsim_render_rgb(avsim_t* h, const int32_t* cam_ids, int ncam, int height, int width, uint8_t* out);
/* The visual scene of avsim_render_rgb (env.py:180-188, :195-200 draw the visual meshes of aloha_sim.xml class
 * "visual", the frame and the textured table of scene.xml, the task objects): library_blob = models/visual_meshes.avv, the decimated
 * mesh library of av_aloha_amd/compiler/vismesh.py; the model blob given to avsim_create carries the instances (vis_inst_*).  After
 * it avsim_render_rgb rasterises those triangles (flat Lambert shading, table texture) instead of the collision proxies; option
 * "render_proxies" 1 switches back.  AVSIM_EMODEL when the model was compiled without instances or the library lacks a mesh. */
int avsim_load_visual(avsim_t* h, const void* library_blob, size_t nbytes);
/* info = {triangles, vertices of the loaded visual scene (0: none), overflow flags of the last visual render (bit 0: a view ran out
 * of triangle records, bit 1: of tile-list entries), instances in the model blob}; synchronises the stream */
int avsim_visual_info(avsim_t* h, int32_t info[4]);
/* debug: per-view records of the last visual render, int32[nviews][8] = {overflow bits, shader-clock cycles / 1024 of the stages
 * (vertex transform, triangle set-up, tile count, tile fill, tiles), triangle records, tile-list entries} */
int avsim_visual_profile(avsim_t* h, int32_t* out, int nviews);
int avsim_camera_count(const avsim_t* h);

/* get_reward of the handle's task (gym_guided_vision/gym_guided_vision/env.py:425-863, five subclasses) evaluated on
 * caller-supplied contact lists instead of the simulator's own contacts: geom_pairs = int32[nsets][cap][2], ids into the
 * model's collision geom table (manifest "geom_names"), a slot with a negative id is empty.  The kernel applies the same
 * predicate as the step kernel.  latch (may be NULL = all zero) is int32[nsets], read and updated in place (SewNeedle's
 * threaded_needle, env.py:596); reward = int32[nsets].  All pointers are host pointers. */
int avsim_reward_from_pairs(avsim_t* h, const int32_t* geom_pairs, int nsets, int cap, int32_t* latch, int32_t* reward);

/* stream / timing helpers (HIP events on the stream the kernels are launched on) */
int avsim_sync(avsim_t* h);
int avsim_set_stream(avsim_t* h, void* hip_stream);
int avsim_event_record(avsim_t* h, int slot);                            /* slot in [0,16) */
int avsim_event_elapsed_ms(avsim_t* h, int slot_a, int slot_b, float* ms); /* synchronises on slot_b */
/* accumulated device time of the physics kernel since the last call with reset!=0, measured with
 * HIP events around every launch (on the launch stream, no per-launch synchronisation) when enabled via
 * avsim_set_option("kernel_timing", 1); this call synchronises on the recorded events */
int avsim_kernel_time(avsim_t* h, int reset, double* total_ms, int64_t* launches);
/* the same for the image kernel of avsim_render_depth / the proxy mode of avsim_render_rgb (k_render_depth alone: the pose pass
 * and the per-view set-up kernel in front of it are not in the figure) */
int avsim_render_kernel_time(avsim_t* h, int reset, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif
