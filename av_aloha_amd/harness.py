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


def record_scripted(task_name: str, num_episodes: int, cameras=(), seed: int | None = None, device: int = 0, only_success: bool = False, **script_kw):
    """The counterpart of record_sim_episodes.py:68-212 with a scripted teleoperator in the headset's place (av_aloha_amd/scripted.py):
    `num_episodes` episodes of `task_name` ("sim_insert_peg", ...) are run SIDE BY SIDE on the device -- one env each, object poses from the
    task's own reset sampling (global numpy RNG, `seed` seeds it) -- through the Cartesian-action env (sim_env.py:277-312), and come back as
    a list of episode dicts in the layout of record_sim_episodes.py:155-212 (`/observations/{qpos,qvel,all_qpos}`, `/action` = the joint-space
    control with normalised grippers, `/observations/images/<cam>` u8 (T, H, W, 3); T = steps + 1, float32) next to per-episode
    {"max_reward_reached", "success", "rewards"}.  only_success: keep the episodes that end at max_reward (check_dataset_reward.py's criterion)."""
    from . import scripted
    from .sim_env import make_sim_env, _TASK_OF_SUBSTRING
    task = next(key for sub, key in _TASK_OF_SUBSTRING if sub in task_name)
    n = int(num_episodes)
    if seed is not None:
        np.random.seed(seed)
    env = make_sim_env(task_name, cameras=list(cameras), num_envs=n, device=device)
    obs, _ = env.reset()
    b = (lambda a: np.asarray(a)[None]) if n == 1 else np.asarray          # batch axis for a single env
    home = {k: b(obs["poses"][k]).copy() for k in ("left", "right", "middle")}
    script = scripted.make_script(scripted.SCRIPT_OF_TASK[task], home, b(obs["qpos"]), **script_kw)
    steps, rewards = [obs], []
    max_reward = env.sim.max_reward
    for _ in range(script.steps()):
        _, rw, _ = env.sim.step_cartesian(script.action(b(steps[-1]["qpos"])))
        rewards.append(rw.copy())
        steps.append(env.get_obs())
    env.close()
    rewards = np.stack(rewards)                                           # [T - 1, n]
    stack = lambda f: np.stack([b(f(s)) for s in steps]).astype(np.float32)   # [T, n, ...]
    arrays = {"/observations/qpos": stack(lambda s: s["joints"]["position"]), "/observations/qvel": stack(lambda s: s["joints"]["velocity"]),
              "/observations/all_qpos": stack(lambda s: s["qpos"]), "/action": stack(lambda s: s["control"])}
    images = {cam: np.stack([b(s["images"][cam]) for s in steps]) for cam in steps[0].get("images", {})}
    episodes = []
    for k in range(n):
        ok = bool(rewards[-1, k] == max_reward)
        if only_success and not ok:
            continue
        data = {name: np.ascontiguousarray(a[:, k]) for name, a in arrays.items()}
        for cam, img in images.items():
            data[f"/observations/images/{cam}"] = np.ascontiguousarray(img[:, k])
        episodes.append({"data": data, "success": ok, "max_reward_reached": int(rewards[:, k].max()), "rewards": rewards[:, k].copy(), "max_reward": int(max_reward)})
    return episodes


def check_dataset_reward(gym_id: str, episodes: list, device: int = 0):
    """gym_guided_vision/scripts/check_dataset_reward.py:15-63 for a list of episode dicts (or loaded files): the gym env of the task is put
    into the episode's first recorded state (`set_qpos(all_qpos[0])`) and steps the recorded `/action` sequence OPEN LOOP through `step_action`
    -- on the gym assets' model, whose peg / needle contacts are stiffer than those of the data-collection assets the episode was recorded on
    (task_insert_peg.xml:7 "HACK: modified solref different from data collection") --, `get_reward()` after every step; an episode passes when
    its largest reward is `max_reward`.  All episodes are replayed side by side in one batched env.  -> (passed bool [n], rewards int [T, n])"""
    from .env import make
    n = len(episodes)
    q0 = np.stack([np.asarray(e["/observations/all_qpos"][0], dtype=np.float64) for e in episodes])
    acts = np.stack([np.asarray(e["/action"], dtype=np.float32) for e in episodes], axis=1)          # [T, n, 21]
    env = make(gym_id, cameras=[], num_envs=n, device=device)
    env.reset()
    env.set_qpos(q0)
    rewards = np.zeros((acts.shape[0], n), dtype=np.int32)
    for t in range(acts.shape[0]):
        env.step_action(acts[t][:, :env.num_joints])
        rewards[t] = np.asarray(env.get_reward()).reshape(n)
    ok = rewards.max(axis=0) == env.max_reward
    env.close()
    return ok, rewards


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
