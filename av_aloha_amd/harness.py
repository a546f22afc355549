"""Rollout and recording harnesses: the counterparts of eval_scripts/eval.py and data_collection_scripts/record_sim_episodes.py
(SURVEY 8f ranks 1-2), host glue over the environments of env.py / sim_env.py.

* `preprocess_observation` (eval.py:23-66): gym observation -> LeRobot-style tensors (`observation.images.<cam>` float32 CHW in
  [0, 1] resized to 480 x 640, `observation.state` float32 with a leading batch axis).
* `rollout` (eval.py:96-124): reset / select_action / step loop for `num_episodes` x `episode_len`, frames of zed_cam_left when
  the env returns pixels, plus the success / return bookkeeping the reference leaves out.  Works on one env (reference shapes)
  and on a batch (`num_envs > 1`).
* `record_episode` / `save_episode` / `load_episode` (record_sim_episodes.py:83-128, :155-212): a scripted Cartesian action
  sequence replaces the VR headset; the episode holds T = len(actions) + 1 time steps with `/observations/qpos` (T, 21),
  `/observations/qvel` (T, 21), `/observations/all_qpos` (T, nq), `/action` (T, 21: the joint-space command with
  normalised grippers, i.e. obs['control']) as float32, `/observations/images/<cam>` uint8 (T, H, W, 3) for the env's cameras
  and the attribute sim = True.  Written as HDF5 either way: through h5py when it is importable, otherwise through the
  package's own minimal writer (av_aloha_amd/hdf5min.py: same groups, names, dtypes, image chunking (1, H, W, 3) and attribute, in
  the structures libhdf5 writes by default); `load_episode` reads both, and the older .npz files.
* `replay_episode` (gym_guided_vision/scripts/replay_sim_episode.py:221-262): set_qpos through `/observations/all_qpos`.
"""
from __future__ import annotations

import os

import numpy as np

IMAGE_SIZE = (480, 640)        # eval.py:20 RESIZE


def preprocess_observation(observations: dict) -> dict:
    import torch
    import torch.nn.functional as F
    out = {}
    if "pixels" in observations and observations["pixels"] is not None:
        px = observations["pixels"]
        imgs = {f"observation.images.{k}": v for k, v in px.items()} if isinstance(px, dict) else {"observation.image": px}
        for key, img in imgs.items():
            t = torch.from_numpy(np.ascontiguousarray(img).copy())
            if t.ndim == 3:
                t = t.unsqueeze(0)
            _, h, w, c = t.shape
            assert c < h and c < w, f"expect channel last images, but instead got {tuple(t.shape)}"
            assert t.dtype == torch.uint8, f"expect torch.uint8, but instead {t.dtype}"
            t = t.permute(0, 3, 1, 2).contiguous().to(torch.float32) / 255
            if (h, w) != IMAGE_SIZE:
                t = F.interpolate(t, size=IMAGE_SIZE, mode="bilinear", antialias=True, align_corners=False)
            out[key] = t
    if "environment_state" in observations:
        out["observation.environment_state"] = torch.from_numpy(np.asarray(observations["environment_state"])).float()
    state = torch.from_numpy(np.asarray(observations["agent_pos"])).float()
    out["observation.state"] = state.unsqueeze(0) if state.ndim == 1 else state
    return out


def rollout(env, select_action, episode_len: int, num_episodes: int = 1, reset_policy=None):
    """select_action(dict of tensors) -> array-like (batch, action_dim).  Returns per-episode dicts with 'return' (sum of the
    rewards), 'success' (is_success seen at any step), 'max_reward' and the captured 'frames'."""
    results = []
    batched = getattr(env, "num_envs", 1) > 1
    for _ in range(num_episodes):
        if reset_policy is not None:
            reset_policy()
        observation, info = env.reset()
        ret = 0
        success = np.zeros(getattr(env, "num_envs", 1), dtype=bool)
        frames = []
        for _ in range(episode_len):
            action = np.asarray(select_action(preprocess_observation(observation)))
            assert action.ndim == 2, "Action dimensions should be (batch, action_dim)"
            observation, reward, terminated, truncated, info = env.step(action if batched else action[0])
            ret = ret + np.asarray(reward)
            success |= np.asarray(info["is_success"], dtype=bool).reshape(-1)
            px = observation.get("pixels") or {}
            if "zed_cam_left" in px:
                frames.append(px["zed_cam_left"])
        results.append({"return": ret, "success": success if batched else bool(success[0]), "max_reward": env.max_reward, "frames": frames})
    return results


def record_episode(env, actions23) -> dict:
    """Steps a Cartesian-action env (av_aloha_amd.sim_env) through `actions23` [T-1, 23] and returns the episode arrays."""
    ts = env.get_obs()
    steps = [ts]
    for a in np.asarray(actions23, dtype=np.float64):
        ts, _, _, _, _ = env.step(a)
        steps.append(ts)
    stack = lambda f: np.stack([np.asarray(f(s)) for s in steps]).astype(np.float32)
    data = {"/observations/qpos": stack(lambda s: s["joints"]["position"]), "/observations/qvel": stack(lambda s: s["joints"]["velocity"]),
            "/observations/all_qpos": stack(lambda s: s["qpos"]), "/action": stack(lambda s: s["control"])}
    for cam in steps[0].get("images", {}):                     # record_sim_episodes.py:197-200: uint8 (T, H, W, 3) per camera
        data[f"/observations/images/{cam}"] = np.stack([s["images"][cam] for s in steps])
    return data


def save_episode(data: dict, dataset_dir: str, episode_idx: int, use_h5py: bool | None = None) -> str:
    """episode_<idx>.hdf5 with the reference's layout (record_sim_episodes.py:186-206): through h5py when it is importable
    (use_h5py None / True), else through av_aloha_amd.hdf5min."""
    os.makedirs(dataset_dir, exist_ok=True)
    base = os.path.join(dataset_dir, f"episode_{episode_idx}")
    h5py = None
    if use_h5py is not False:
        try:
            import h5py
        except ImportError:
            if use_h5py:
                raise
    if h5py is None:
        from . import hdf5min
        chunks = {k: (1, *v.shape[1:]) for k, v in data.items() if "/images/" in k}
        hdf5min.write(base + ".hdf5", data, attrs={"sim": np.bool_(True)}, chunks=chunks)
        return base + ".hdf5"
    with h5py.File(base + ".hdf5", "w", rdcc_nbytes=1024 ** 2 * 2) as root:
        root.attrs["sim"] = True
        for name, array in data.items():
            chunks = (1, *array.shape[1:]) if "/images/" in name else None
            root.create_dataset(name, data=array, chunks=chunks)
    return base + ".hdf5"


def load_episode(path: str) -> dict:
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files if k != "sim"}
    try:
        import h5py
    except ImportError:
        from . import hdf5min
        return hdf5min.read(path)[0]
    out = {}
    with h5py.File(path, "r") as root:
        root.visititems(lambda n, o: out.__setitem__("/" + n, o[()]) if hasattr(o, "shape") else None)
    return out


def replay_episode(env, data: dict):
    """Drives `env` (gym flavour, av_aloha_amd.env) through the recorded full states; returns the observations and rewards."""
    env.reset()
    obs, rewards = [], []
    na = 14 if env.num_arms == 2 else 21
    for q in data["/observations/all_qpos"]:
        env.set_qpos(np.asarray(q, dtype=np.float64))
        obs.append(env.get_obs()["agent_pos"][..., :na])
        rewards.append(env.get_reward())
    return np.stack(obs), np.asarray(rewards)
