"""Cartesian-action environments: the counterpart of the reference's data_collection_scripts/sim_env.py.

`GuidedVisionEnv.step(action[23])` (sim_env.py:277-312) takes end-effector targets [left xyz + quat wxyz (7), left trigger,
right (7), right trigger, middle (7)], runs the reference's controllers on the MEASURED joint angles (GradIK for the two
manipulators, DiffIK for the camera arm, sim_env.py:89-138), writes the result to ctrl with the grippers at
unnorm(1 - trigger) (:300-301) and steps 20 substeps; `step_joints(action[21])` (:252-273) is the joint-space variant.
`get_obs()` returns the reference's dictionary (:160-218): joints.position / joints.velocity (fingers normalised by the
gripper ctrl range), qpos, control, poses.{left,right,middle} = FK of the *commanded* joints as [xyz, quat wxyz], images.
IK, FK and physics run in libavsim (avsim_step_cartesian, avsim_fk_jac); this file is host glue only.  Rewards are 0 here
exactly as in the reference (:307).  `images` follow the reference's naming and sizes (:187-203: zed_cam = left | right 720 x 720
side by side, the others 480 x 640), drawn by the library's rasteriser over the visual meshes (avsim_load_visual + avsim_render_rgb)."""
from __future__ import annotations

import numpy as np

from . import _ffi
from .constants import SIM_PHYSICS_ENV_STEP_RATIO
from .env import _TASK_OF_SUBSTRING, PhysicsError, sample_object_poses
from .sim import BatchedSim


def mat2quat_xyzw(R):
    """transform_utils.py:9-49 (robosuite's eigen-decomposition form), batched: rotation matrices [..., 3, 3] -> (x, y, z, w)
    with w >= 0."""
    R = np.asarray(R, dtype=np.float64)
    K = np.zeros(R.shape[:-2] + (4, 4))
    m = lambda i, j: R[..., i, j]
    K[..., 0, 0] = m(0, 0) - m(1, 1) - m(2, 2)
    K[..., 1, 0] = m(0, 1) + m(1, 0); K[..., 1, 1] = m(1, 1) - m(0, 0) - m(2, 2)
    K[..., 2, 0] = m(0, 2) + m(2, 0); K[..., 2, 1] = m(1, 2) + m(2, 1); K[..., 2, 2] = m(2, 2) - m(0, 0) - m(1, 1)
    K[..., 3, 0] = m(2, 1) - m(1, 2); K[..., 3, 1] = m(0, 2) - m(2, 0); K[..., 3, 2] = m(1, 0) - m(0, 1)
    K[..., 3, 3] = m(0, 0) + m(1, 1) + m(2, 2)
    K /= 3.0
    w, V = np.linalg.eigh(K)                      # uses the lower triangle, as numpy's default UPLO='L' does in the reference
    q = np.take_along_axis(V, np.argmax(w, axis=-1)[..., None, None], axis=-1)[..., 0]     # (x, y, z, w)
    q1 = q[..., [3, 0, 1, 2]]
    q1 = np.where(q1[..., :1] < 0.0, -q1, q1)
    return q1[..., [1, 2, 3, 0]]


CAMERAS = ["zed_cam", "cam_left_wrist", "cam_right_wrist", "cam_high", "cam_low"]      # sim_env.py:22
_CAMERA_IDS = {"cam_left_wrist": "wrist_cam_left", "cam_right_wrist": "wrist_cam_right", "cam_high": "overhead_cam", "cam_low": "worms_eye_cam"}


class GuidedVisionEnv:
    """One env (reference shapes) or a batch (leading axis num_envs)."""

    task = None

    def __init__(self, cameras=CAMERAS, num_envs: int = 1, device: int = 0, f64: bool = False, options: dict | None = None,
                 variant: str = "data_collection"):
        for camera in cameras:
            assert camera in CAMERAS, f"Invalid camera name: {camera}"
        self._cameras = list(cameras)
        if self.task is None:
            raise NotImplementedError("use one of the task classes or make_sim_env()")
        self.num_envs = int(num_envs)
        # the reference's sim_env.py loads data_collection_scripts/assets (constants.py:5), not the gym package's assets: the needle
        # and the peg have MuJoCo's default solref there (task_sew_needle.xml:17, task_insert_peg.xml:7) and the ZED cameras fovy 90
        # (aloha_sim.xml:357-358); variant="gym" runs the Cartesian env on the gym assets' model instead
        self.variant = variant
        options = {"render_shadows": 1, "render_samples": 4, "render_smooth": 1, **(options or {})}       # (MuJoCo's defaults: shadows on, 4 offscreen samples [EXT])
        self.sim = BatchedSim(self.task, 3, self.num_envs, device=device, f64=f64, options=options, variant=variant)
        from .compiler.compile import read_blob
        from .constants import MODEL_DIR
        from .sim import VARIANT_PREFIX
        import os
        md = read_blob(os.path.join(MODEL_DIR, f"{VARIANT_PREFIX[variant]}{self.task}_3arms.avm"))
        self._qadr = md["obs_qposadr"].astype(np.int64)          # LEFT(6+left_left_finger), RIGHT(6+right_right_finger), MIDDLE(7)
        self._dadr = md["obs_dofadr"].astype(np.int64)
        lo, hi = md["grip_range"]
        self._grip_lo, self._grip_span = float(lo), float(hi - lo)

    # ---- helpers -----------------------------------------------------------------------------------
    def _sq(self, a):
        return a[0] if self.num_envs == 1 else a

    def _fk_pose(self, arm, q):
        n = q.shape[0]
        q = np.ascontiguousarray(q, dtype=np.float64)
        T = np.empty((n, 16))
        h = self.sim.h
        h.check(h.L.avsim_fk_jac(h.h, arm, n, q.ctypes.data, T.ctypes.data, None))
        T = T.reshape(n, 4, 4)
        quat = mat2quat_xyzw(T[:, :3, :3])                       # mat2pose (:153-157) then xyzw_to_wxyz (:159-161)
        return np.concatenate([T[:, :3, 3], quat[:, [3, 0, 1, 2]]], axis=1)

    def get_obs(self):
        qpos, qvel, ctrl, _ = self.sim.get_state()
        pos = qpos[:, self._qadr].copy()
        vel = qvel[:, self._dadr].copy()
        con = ctrl.copy()                                          # actuator order = LEFT(7) RIGHT(7) MIDDLE(7)
        for k in (6, 13):
            pos[:, k] = (pos[:, k] - self._grip_lo) / self._grip_span
            vel[:, k] = vel[:, k] / self._grip_span
            con[:, k] = (con[:, k] - self._grip_lo) / self._grip_span
        poses = {"left": self._fk_pose(0, ctrl[:, 0:6]), "right": self._fk_pose(1, ctrl[:, 7:13]), "middle": self._fk_pose(2, ctrl[:, 14:21])}
        return {
            "joints": {"position": self._sq(pos), "velocity": self._sq(vel)},
            "qpos": self._sq(qpos),
            "control": self._sq(con),
            "poses": {k: self._sq(v) for k, v in poses.items()},
            "images": self._images(),
        }

    def _images(self):
        images = {}
        for camera in self._cameras:                               # sim_env.py:187-203
            if camera == "zed_cam":
                lr = self.sim.render_rgb(["zed_cam_left", "zed_cam_right"], 720, 720)
                images["zed_cam"] = self._sq(np.concatenate([lr[:, 0], lr[:, 1]], axis=2))
            else:
                images[camera] = self._sq(self.sim.render_rgb([_CAMERA_IDS[camera]], 480, 640)[:, 0])       # (a view of the call's own array: one camera, contiguous)
        return images

    # ---- gym-style surface (sim_env.py:220-312) ------------------------------------------------------
    def reset(self, seed=None):
        # sim_env.py:220-250: `seed` only reaches gym's np_random, which nothing reads; the object poses come from the GLOBAL
        # numpy RNG (env.py:482 ...) and therefore do not depend on it -- kept that way
        poses = np.stack([sample_object_poses(self.task) for _ in range(self.num_envs)])
        self.sim.reset(poses)
        return self.get_obs(), "Resetting arms..."

    def set_qpos(self, qpos):
        self.sim.set_qpos(np.asarray(qpos, dtype=np.float64).reshape(self.num_envs, self.sim.nq))

    def _truncated(self):
        """Divergence flags of the last step: a single env raises (dm_control's PhysicsError in the reference), a batch returns
        them as its `truncated` array."""
        div = (self.sim.diag()[:, 3] & 1).astype(bool)
        if self.num_envs == 1:
            if div[0]:
                raise PhysicsError("the simulation state diverged during the step; the env was put back to the home pose")
            return False
        return div

    def step_joints(self, action):
        a = np.asarray(action, dtype=np.float32).reshape(self.num_envs, 21)
        self.sim.step(a, SIM_PHYSICS_ENV_STEP_RATIO, want_reward=False)
        return self.get_obs(), 0, False, self._truncated(), ""

    def step(self, action):
        a = np.asarray(action, dtype=np.float64).reshape(self.num_envs, 23)
        self.sim.step_cartesian(a, _ffi.IK_REFERENCE, SIM_PHYSICS_ENV_STEP_RATIO)
        return self.get_obs(), 0, False, self._truncated(), ""

    def hide_middle_arm(self):
        """replay_sim_episode.py:59 calls this on the Cartesian env; the reference class does not define it there either (the gym
        flavour does, env.py:394-395).  This env always simulates three arms."""
        raise NotImplementedError("the Cartesian-action env drives three arms; use the gym flavour with num_arms=2 for a parked camera arm")

    def close(self):
        if getattr(self, "sim", None) is not None:
            self.sim.close()
            self.sim = None


def _task_class(key):
    return type("".join(p.capitalize() for p in key.split("_")) + "Env", (GuidedVisionEnv,), {"task": key})


InsertPegEnv, SlotInsertionEnv, SewNeedleEnv, TubeTransferEnv, HookPackageEnv = (
    _task_class(k) for k in ("insert_peg", "slot_insertion", "sew_needle", "tube_transfer", "hook_package"))
_CLASSES = {"insert_peg": InsertPegEnv, "slot_insertion": SlotInsertionEnv, "sew_needle": SewNeedleEnv,
            "tube_transfer": TubeTransferEnv, "hook_package": HookPackageEnv}


def make_sim_env(task_name, cameras=CAMERAS, **kw):
    """sim_env.py:18-35: keyed on the same substrings, NotImplementedError otherwise."""
    for sub, key in _TASK_OF_SUBSTRING:
        if sub in task_name:
            return _CLASSES[key](cameras=cameras, **kw)
    raise NotImplementedError
