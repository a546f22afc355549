"""Gym-style environments over the MI355X batched simulator.

Mirrors the public surface of the reference's gym_guided_vision/gym_guided_vision/env.py
(`make_sim_env` :18-30, `GuidedVisionEnv` :32-406, the five task classes :412-863) and the registry
of gym_guided_vision/gym_guided_vision/__init__.py:4-101, so callers written against the reference
(`env.reset()`, `env.step(a)`, `env.set_qpos`, `env.step_action`, `env.get_reward`, `max_reward`,
`num_arms`) keep working.  Physics, IK, reward and observation gathering all run in libavsim's HIP
kernels; this file only holds host-side glue (action/observation packing, object-pose sampling).

Differences a caller can see (DESIGN.md lists them): the `pixels` observations and `render()` are drawn by the library's
own triangle rasteriser over the decimated visual meshes (robot, frame, textured table, task objects; flat shading per triangle with the
scene's lights, the directional light's shadows and specular term, 2 x 2 multisampling), not by MuJoCo's OpenGL renderer; `render_depth()` returns float32 depth images of the same cameras (BASELINE config 5); envs can be batched
with `num_envs > 1` (every array gains a leading axis).
"""
from __future__ import annotations

import numpy as np

from . import _ffi
from .constants import CAMERAS, RENDER_CAMERA, SIM_DT, SIM_PHYSICS_ENV_STEP_RATIO
from .sim import BatchedSim

try:  # gymnasium is optional: the API below does not depend on it
    import gymnasium as gym
    from gymnasium import spaces
    _EnvBase = gym.Env
except Exception:  # pragma: no cover - gymnasium is absent in the build image
    gym = None
    spaces = None
    _EnvBase = object


class PhysicsError(RuntimeError):
    """The simulation state of an env became NaN / Inf / larger than 1e6 during a step: the counterpart of
    dm_control.rl.control.PhysicsError, which the reference's physics.step (env.py:218) raises on MuJoCo's bad-state
    warnings.  Raised by single-env facades; batches report per-env flags instead (info["diverged"], truncated)."""


class _Box:
    """Minimal stand-in for gymnasium.spaces.Box when gymnasium is not installed."""

    def __init__(self, low, high, shape, dtype):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

    def sample(self):
        return np.zeros(self.shape, dtype=self.dtype)

    def contains(self, x):
        return np.asarray(x).shape == self.shape


# ---- object placement (task `reset` overrides, env.py:474-501, 513-543, 604-637, 705-735, 792-818) ----
def _uniform(lo, hi):
    # the reference draws from the GLOBAL numpy RNG (env.py:482 ...), not from the gym-seeded one
    return np.random.uniform(np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64))


def sample_object_poses(task: str) -> np.ndarray:
    """One draw of the free objects' [x y z qw qx qy qz] in qpos order, consuming the global numpy RNG in
    exactly the reference's order (including the discarded draws of env.py:525 and :616)."""
    ident = [1.0, 0.0, 0.0, 0.0]
    if task == "insert_peg":          # qpos order: peg, hole
        peg = _uniform([0.1, -0.1, 0.01], [0.2, 0.1, 0.01])
        hole = _uniform([-0.1, -0.1, 0.021], [-0.2, 0.1, 0.021])
        objs = [peg, hole]
    elif task == "slot_insertion":    # slot, stick
        slot = _uniform([-0.05, 0.1, 0.0], [0.05, 0.15, 0.0])
        _uniform([-0.05, 0.1, 0.0], [0.05, 0.15, 0.0])            # discarded "peg_position"
        stick = _uniform([-0.08, -0.1, 0.0], [0.08, 0.0, 0.0])
        objs = [slot, stick]
    elif task == "sew_needle":        # wall, needle (draw order: needle, discarded, wall)
        needle = _uniform([0.15, -0.025, 0.0], [0.2, 0.1, 0.0])
        _uniform([0.15, -0.025, 0.0], [0.2, 0.1, 0.0])
        wall = _uniform([-0.025, -0.025, 0.0], [0.025, 0.1, 0.0])
        objs = [wall, needle]
    elif task == "tube_transfer":     # ball, tube1, tube2 (ball shares tube1's draw)
        ball = _uniform([0.05, -0.05, 0.0], [0.1, 0.05, 0.0])
        tube2 = _uniform([-0.1, -0.05, 0.0], [-0.05, 0.05, 0.0])
        objs = [ball, ball, tube2]
    elif task == "hook_package":      # hook, package
        hook = _uniform([-0.1, 0.3, 0.2], [0.1, 0.3, 0.3])
        package = _uniform([-0.1, 0.0, 0.0], [0.1, 0.15, 0.0])
        objs = [hook, package]
    else:
        raise NotImplementedError(task)
    return np.array([np.concatenate([p, ident]) for p in objs])


_TASK_OF_SUBSTRING = (("sim_insert_peg", "insert_peg"), ("sim_slot_insertion", "slot_insertion"),
                      ("sim_sew_needle", "sew_needle"), ("sim_tube_transfer", "tube_transfer"),
                      ("sim_hook_package", "hook_package"))


class GuidedVisionEnv(_EnvBase):
    """Single- or multi-env facade.  With num_envs == 1 (default) every method has the reference's
    shapes; with num_envs > 1 arrays gain a leading env axis."""

    metadata = {"render_modes": ["rgb_array"], "render_fps": 1 / SIM_DT}
    task = None
    max_reward = 0

    def __init__(self, num_arms: int = 3, cameras=CAMERAS, observation_height: int = 480, observation_width: int = 640,
                 num_envs: int = 1, device: int = 0, f64: bool = False, options: dict | None = None):
        if _EnvBase is not object:
            super().__init__()
        assert num_arms in [2, 3], f"Invalid number of arms: {num_arms}"
        assert all([camera in CAMERAS for camera in cameras]), f"Invalid camera names: {cameras}"
        if self.task is None:
            raise NotImplementedError("use one of the task classes or make_sim_env()")
        self.cameras = list(cameras)
        self.num_arms = num_arms
        self.num_envs = int(num_envs)
        self.observation_height = observation_height
        self.observation_width = observation_width
        self.num_joints = 14 if num_arms == 2 else 21
        # the colour images as MuJoCo's renderer makes them by default [EXT]: the directional light casts shadows, the offscreen buffer is
        # multisampled (4 samples); options={"render_shadows": 0, "render_samples": 1} gives the plain image
        options = {"render_shadows": 1, "render_samples": 4, "render_smooth": 1, **(options or {})}
        self._device, self._f64, self._options, self._model_arms = device, f64, options, num_arms
        self.sim = BatchedSim(self.task, num_arms, self.num_envs, device=device, f64=f64, options=options)
        self.max_reward = self.sim.max_reward
        box = spaces.Box if spaces is not None else _Box
        shape = (self.num_joints,) if self.num_envs == 1 else (self.num_envs, self.num_joints)
        self.action_space = box(low=-np.inf, high=np.inf, shape=shape, dtype=np.float32)
        agent_space = box(low=-np.inf, high=np.inf, shape=shape, dtype=np.float64)
        ishape = (observation_height, observation_width, 3)
        ishape = ishape if self.num_envs == 1 else (self.num_envs,) + ishape
        pix = {c: box(low=0, high=255, shape=ishape, dtype=np.uint8) for c in self.cameras}
        if spaces is not None:
            self.observation_space = spaces.Dict({"pixels": spaces.Dict(pix), "agent_pos": agent_space})
        else:
            self.observation_space = {"pixels": pix, "agent_pos": agent_space}
        self._agent_pos = np.zeros((self.num_envs, self.num_joints))
        self._reward = np.zeros(self.num_envs, dtype=np.int32)

    # -- helpers ---------------------------------------------------------------------------------
    def _squeeze(self, a):
        return a[0] if self.num_envs == 1 else a

    def _pixels(self):
        if not self.cameras:
            return {}
        # camera-major from the library: one contiguous array per camera, no second copy of the batch's pixels on the host
        img = self.sim.render_rgb(self.cameras, self.observation_height, self.observation_width, cam_major=True)
        return {c: self._squeeze(img[i]) for i, c in enumerate(self.cameras)}

    def _obs(self):
        return {"pixels": self._pixels(), "agent_pos": self._squeeze(self._agent_pos).copy()}

    def _check_diverged(self):
        """Divergence flag of the last step (bit 0 of diag[3], include/avsim.h): the library has put such an env back to the state
        its episode started from (home pose, the objects where reset() put them, zero velocity)."""
        div = (self.sim.diag()[:, 3] & 1).astype(bool)
        if self.num_envs == 1 and div[0]:
            raise PhysicsError("the simulation state diverged (NaN / Inf / > 1e6) during the step; the env was put back to the "
                               "home pose -- call reset()")
        return div

    def _refresh_agent_pos(self):
        ap = np.empty((self.num_envs, self.num_joints))
        h = self.sim.h
        h.check(h.L.avsim_observe(h.h, ap.ctypes.data, None, None))
        self._agent_pos = ap

    # -- gym API ---------------------------------------------------------------------------------
    def reset(self, seed=None, options=None):
        if _EnvBase is not object:
            super().reset(seed=seed, options=options)    # seeds self.np_random, unused afterwards (env.py:229)
        poses = np.stack([sample_object_poses(self.task) for _ in range(self.num_envs)])
        self.sim.reset(poses)
        self._refresh_agent_pos()
        self._reward[:] = 0
        # env.py:249: a plain False for one env; a batch gets one flag per env
        return self._obs(), {"is_success": False if self.num_envs == 1 else np.zeros(self.num_envs, dtype=bool)}

    def step(self, action):
        a = np.asarray(action, dtype=np.float32).reshape(self.num_envs, self.num_joints)
        self._agent_pos, self._reward, success = self.sim.step(a, SIM_PHYSICS_ENV_STEP_RATIO)
        diverged = self._check_diverged()
        reward = self._squeeze(self._reward)
        if self.num_envs == 1:
            return self._obs(), int(reward), False, False, {"is_success": bool(success[0])}
        # batch: a diverged env is truncated (its episode cannot continue) and flagged; the others are unaffected
        return self._obs(), reward, np.zeros(self.num_envs, dtype=bool), diverged, {"is_success": success, "diverged": diverged}

    def step_action(self, action):
        """env.py:255-269: apply the action and advance the physics without computing obs / reward."""
        a = np.asarray(action, dtype=np.float32).reshape(self.num_envs, self.num_joints)
        self._agent_pos, _, _ = self.sim.step(a, SIM_PHYSICS_ENV_STEP_RATIO, want_reward=False)
        self._check_diverged()

    def get_obs(self):
        return self._obs()

    def get_reward(self):
        """Reward of the CURRENT contact set (env.py get_reward): re-evaluated on the device through a zero-substep
        launch, so SewNeedle's latch (env.py:686-689) advances exactly as the reference's does."""
        rw = np.empty(self.num_envs, dtype=np.int32)
        su = np.empty(self.num_envs, dtype=np.uint8)
        h = self.sim.h
        h.check(h.L.avsim_observe(h.h, None, rw.ctypes.data, su.ctypes.data))
        self._reward = rw
        r = self._squeeze(rw)
        return int(r) if self.num_envs == 1 else r

    def set_qpos(self, qpos):
        self.sim.set_qpos(np.asarray(qpos, dtype=np.float64).reshape(self.num_envs, self.sim.nq))
        self._refresh_agent_pos()

    def render(self):
        """env.py:195-200: uint8 [225, 300, 3] image of the overhead camera (leading num_envs axis for a batch)."""
        return self._squeeze(self.sim.render_rgb([RENDER_CAMERA], 225, 300)[:, 0]).copy()

    def render_depth(self, cameras=("zed_cam_left", "zed_cam_right", "wrist_cam_left", "wrist_cam_right"), height=None, width=None):
        """Depth images (float32 metres along the optical axis, far plane 30 m) of the named cameras at the current state:
        {camera: [H, W]} for one env, {camera: [num_envs, H, W]} for a batch.  Stands where the reference's get_obs renders
        `pixels` (env.py:180-188); sizes default to observation_height x observation_width."""
        assert all(c in CAMERAS for c in cameras), f"Invalid camera names: {cameras}"
        h = self.observation_height if height is None else height
        w = self.observation_width if width is None else width
        img = self.sim.render_depth(list(cameras), h, w)
        return {c: self._squeeze(img[:, i]) for i, c in enumerate(cameras)}

    def _swap_model(self, arms):
        """The base position of the camera arm is a constant of the compiled model, so moving it (env.py:394-398) means
        continuing on the other blob of the same task: identical state layout, the arm parked at (0, -2.4, -0.4) or in place.
        The state carries over; action / agent_pos keep this env's width."""
        if self._model_arms == arms:
            return
        q, v, c, w = self.sim.get_state()
        latch = self.sim.get_latch()                 # SewNeedle's threaded_needle stage carries over (env.py:596, :686-689)
        fallback = self.sim.get_reset_poses()        # ... and so do the object poses a diverged env is put back to
        old = self.sim
        self.sim = BatchedSim(self.task, arms, self.num_envs, device=self._device, f64=self._f64, options=self._options)
        self.sim.set_option("num_joints", self.num_joints)
        self.sim.nj = self.sim.h.nj = self.num_joints
        self.sim.set_state(q, v, c, w)
        self.sim.set_latch(latch)
        self.sim.set_reset_poses(fallback)
        old.close()
        self._model_arms = arms
        self._refresh_agent_pos()

    def hide_middle_arm(self):
        """env.py:394-395: park the camera arm's base at (0, -2.4, -0.4)."""
        self._swap_model(2)

    def show_middle_arm(self):
        """env.py:397-398: back to its place."""
        self._swap_model(3)

    def close(self):
        if getattr(self, "sim", None) is not None:
            self.sim.close()
            self.sim = None


class InsertPegEnv(GuidedVisionEnv):
    task = "insert_peg"


class SlotInsertionEnv(GuidedVisionEnv):
    task = "slot_insertion"


class SewNeedleEnv(GuidedVisionEnv):
    task = "sew_needle"


class TubeTransferEnv(GuidedVisionEnv):
    task = "tube_transfer"


class HookPackageEnv(GuidedVisionEnv):
    task = "hook_package"


_CLASS_OF_TASK = {c.task: c for c in (InsertPegEnv, SlotInsertionEnv, SewNeedleEnv, TubeTransferEnv, HookPackageEnv)}


def make_sim_env(task_name, **kwargs):
    """env.py:18-30: substring dispatch, NotImplementedError otherwise."""
    for sub, task in _TASK_OF_SUBSTRING:
        if sub in task_name:
            return _CLASS_OF_TASK[task](**kwargs)
    raise NotImplementedError


# registry of gym ids, identical to gym_guided_vision/__init__.py:4-86
_CAMS3 = ["zed_cam_left", "zed_cam_right", "wrist_cam_left", "wrist_cam_right", "overhead_cam", "worms_eye_cam"]
_CAMS2 = ["overhead_cam", "worms_eye_cam", "wrist_cam_left", "wrist_cam_right"]
ENVS = {}
for _name, _cls in (("InsertPeg", "InsertPegEnv"), ("SlotInsertion", "SlotInsertionEnv"), ("SewNeedle", "SewNeedleEnv"),
                    ("TubeTransfer", "TubeTransferEnv"), ("HookPackage", "HookPackageEnv")):
    for _n, _c in ((3, _CAMS3), (2, _CAMS2)):
        ENVS[f"gym_guided_vision/{_name}-{_n}Arms-v0"] = {
            "env": _cls, "num_arms": _n, "cameras": list(_c), "observation_height": 480, "observation_width": 640}


def make(env_id: str, **overrides):
    """gymnasium-free equivalent of gym.make(id, **kwargs) over the same registry."""
    spec = ENVS[env_id]
    kw = {k: v for k, v in spec.items() if k != "env"}
    kw.update(overrides)
    return globals()[spec["env"]](**kw)


def register_with_gymnasium(entry_module="gym_guided_vision.env"):
    if gym is None:
        return False
    from gymnasium.envs.registration import register, registry
    for env_id, kw in ENVS.items():
        if env_id in registry:
            continue
        register(id=env_id, entry_point=f"{entry_module}:{kw['env']}", nondeterministic=True,
                 kwargs={k: v for k, v in kw.items() if k != "env"})
    return True
