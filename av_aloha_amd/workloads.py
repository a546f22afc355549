"""Synthetic workloads of the measurement configurations (SURVEY.md 8(d) configs 2-5 = BASELINE.json configs[1..4]).

Host-side generators only (numpy): object poses keyed by the GLOBAL env id, and the per-step action tensors.  bench.py
uploads them to HBM before its timed region; the parity tests step the same inputs through the oracle.  Nothing here
touches the device or the oracle.

  config 2  SlotInsertion-3Arms, 23-D Cartesian sinusoid targets -> DLS IK on three arms (north_star), seeds 1000 + i
  config 3  SewNeedle-3Arms, scripted reach - grasp - lift of the needle by the right arm (targets derived from the sampled
            needle pose) through the reference's controllers (GradIK, GradIK, DiffIK; sim_env.py:89-138), the camera arm
            sways around its home pose; seeds 2000 + i; contact-rich
  config 4  HookPackage-2Arms, 14-D joint-space smooth random walk (default_rng(3000 + i), sigma 0.02 rad per step, clipped
            to ctrlrange); sharded over ranks, results independent of the sharding
  config 5  config 2 + depth images of zed_cam_left/right + wrist_cam_left/right every step
"""
from __future__ import annotations

import math

import numpy as np

from .env import sample_object_poses

# per configuration: task key, arms, action kind, seed base, episode length (data_collection_scripts/constants.py:23-58)
CONFIGS = {
    2: dict(task="slot_insertion", arms=3, action="cartesian_dls", seed=1000, episode_len=300, gym_id="gym_guided_vision/SlotInsertion-3Arms-v0"),
    3: dict(task="sew_needle", arms=3, action="cartesian_reference", seed=2000, episode_len=250, gym_id="gym_guided_vision/SewNeedle-3Arms-v0"),
    4: dict(task="hook_package", arms=2, action="joint", seed=3000, episode_len=300, gym_id="gym_guided_vision/HookPackage-2Arms-v0"),
    5: dict(task="slot_insertion", arms=3, action="cartesian_dls", seed=1000, episode_len=300, gym_id="gym_guided_vision/SlotInsertion-3Arms-v0"),
}
RENDER_CAMERAS = ("zed_cam_left", "zed_cam_right", "wrist_cam_left", "wrist_cam_right")

# the control site sits at the wrist (aloha_sim.xml:249 right_gripper_control), the pinch point 0.13 m further along the
# gripper (:248 right_gripper): a top-down grasp of a 2 cm bar lying on the table holds the site this far above its centre
GRASP_HEIGHT = 0.14


def object_poses(task: str, global_ids, seed0: int) -> np.ndarray:
    """[n, nobj, 7]: per env np.random.seed(seed0 + global id), then the reference's draw order (env.py:474-818)."""
    out = []
    for g in global_ids:
        np.random.seed(seed0 + int(g))
        out.append(sample_object_poses(task))
    return np.stack(out)


def mat2quat_wxyz(R) -> np.ndarray:
    """Rotation matrix -> unit quaternion (w, x, y, z), w >= 0 (closed form; used for the constant home orientations)."""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0.0] * 4
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q) * (1 if q[0] >= 0 else -1)


def home_poses(T_home) -> dict:
    """{'left','right','middle'} -> [7] = xyz + quat wxyz of the eef sites at the home joints, from their 4x4 FK matrices
    (avsim_fk_jac on ctrl_home; known answers in SURVEY.md Appendix A)."""
    out = {}
    for name, T in zip(("left", "right", "middle"), T_home):
        T = np.asarray(T, dtype=np.float64).reshape(4, 4)
        out[name] = np.concatenate([T[:3, 3], mat2quat_wxyz(T[:3, :3])])
    return out


def sinusoid_actions(home: dict, global_ids, n_total: int, t: int) -> np.ndarray:
    """Config 2/5, env-step t: FK(home) + 3 cm / 0.5 Hz sinusoid with per-env phase 2 pi i / N on all three arms, triggers
    square wave every 50 steps.  [n, 23] in the sim_env.py:278-282 layout."""
    n = len(global_ids)
    ph = 2 * np.pi * np.asarray(global_ids, dtype=np.float64) / n_total
    w = 2 * np.pi * 0.5 * 0.04 * t
    a = np.zeros((n, 23))
    trig = 1.0 if (t // 50) % 2 == 1 else 0.0
    for name, off in (("left", 0), ("right", 8), ("middle", 16)):
        p = home[name]
        a[:, off + 0] = p[0] + 0.03 * np.sin(w + ph)
        a[:, off + 1] = p[1] + 0.03 * np.cos(w + ph)
        a[:, off + 2] = p[2] + 0.03 * np.sin(2 * w + ph)
        a[:, off + 3:off + 7] = p[3:7]
        if off < 16:
            a[:, off + 7] = trig
    return a


def qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def grasp_lift_targets(home, obj_xyz, T=(70, 50, 30, 60, 40), lift=0.12, sway=0.0):
    """Config 3.  home: {'left','right','middle'} -> [n, 7] (or [7]) eef poses at reset; obj_xyz [n, 3]: centre of the bar.
    Yields the [n, 23] action of every step: move above, descend, close, lift, hold (sum(T) steps).  The right gripper
    points straight down: the home orientation turned by -90 degrees about the world y axis (the right arm faces -x).
    sway > 0 moves the camera arm's target on a slow circle of that radius (its DiffIK then has work to do)."""
    n = obj_xyz.shape[0]
    home = {k: np.broadcast_to(np.asarray(v, dtype=np.float64), (n, 7)) for k, v in home.items()}
    ry = np.array([np.cos(-np.pi / 4), 0.0, np.sin(-np.pi / 4), 0.0])
    down = np.stack([qmul(ry, home["right"][i, 3:]) for i in range(n)])
    grasp = obj_xyz + np.array([0.0, 0.0, GRASP_HEIGHT])
    above = grasp + np.array([0.0, 0.0, 0.10])
    up = np.array([0.0, 0.0, lift])
    step = [0]

    def act(rpos, grip):
        a = np.zeros((n, 23))
        a[:, 0:7] = home["left"]
        a[:, 8:11] = rpos
        a[:, 11:15] = down
        a[:, 15] = grip
        a[:, 16:23] = home["middle"]
        if sway:
            w = 2 * np.pi * 0.25 * 0.04 * step[0]
            a[:, 16] += sway * np.sin(w)
            a[:, 18] += sway * (np.cos(w) - 1.0)
        step[0] += 1
        return a
    for t in range(T[0]):
        yield act(above, 0.0)
    for t in range(T[1]):
        yield act(above + (grasp - above) * min(1.0, (t + 1) / (0.7 * T[1])), 0.0)
    for t in range(T[2]):
        yield act(grasp, min(1.0, (t + 1) / (0.5 * T[2])))
    for t in range(T[3]):
        yield act(grasp + up * min(1.0, (t + 1) / (0.67 * T[3])), 1.0)
    for t in range(T[4]):
        yield act(grasp + up, 1.0)


def walk_actions(qpos_home, act_ctrlrange, global_ids, T: int, nj: int, seed0: int, close_grippers: bool = False) -> np.ndarray:
    """Config 4's smooth random walk, float32 [T, n, nj]: default_rng(seed0 + global env id), sigma 0.02 per step, clipped to
    ctrlrange (the grippers to [0, 1]).  The stream of env i is the one T sequential rng.normal(size=nj) calls give."""
    h = np.asarray(qpos_home, dtype=np.float64)
    home = np.concatenate([h[:6], [1.0], h[8:14], [1.0], h[16:23]])[:nj]
    lo, hi = np.asarray(act_ctrlrange, dtype=np.float64).reshape(-1, 2)[:nj].T.copy()
    lo[[6, 13]], hi[[6, 13]] = 0.0, 1.0
    n = len(global_ids)
    noise = np.empty((T, n, nj))
    for k, g in enumerate(global_ids):
        noise[:, k] = np.random.default_rng(seed0 + int(g)).normal(scale=0.02, size=(T, nj))
    acts = np.empty((T, n, nj), dtype=np.float32)
    a = np.repeat(home[None], n, 0)
    for t in range(T):
        a = np.clip(a + noise[t], lo, hi)
        if close_grippers:
            a[:, 6] = a[:, 13] = 0.0 if t >= 2 else 1.0
        acts[t] = a
    return acts
