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
* `rerender_episode` / `rerender_dataset` (gym_guided_vision/scripts/replay_sim_episode.py:47-113): how the reference makes its
  per-camera-configuration training sets -- a recorded episode's full states are put back frame by frame (`set_qpos`), the gym env
  of the wanted camera configuration renders its registered cameras (`get_obs()["pixels"]`), and a new HDF5 is written with those
  images next to the recorded qpos / qvel / action (the first 14 columns for a 2-arm env).  Here the T frames of an episode are T
  envs of ONE batched handle: one set_qpos, one render call per chunk of frames.
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


def _image_bytes_per_step(cameras) -> int:
    """u8 bytes of one env's images per time step in the Cartesian env's sizes (sim_env.py:187-203)."""
    return sum(720 * 1440 * 3 if c == "zed_cam" else 480 * 640 * 3 for c in cameras)


def record_scripted(task_name: str, num_episodes: int, cameras=(), seed: int | None = None, device: int = 0, only_success: bool = False,
                    image_budget_bytes: int = 8 << 30, keep_diverged: bool = False, sink=None, stream_dir: str | None = None, max_batch: int = 256,
                    **script_kw):
    """The counterpart of record_sim_episodes.py:68-212 with a scripted teleoperator in the headset's place (av_aloha_amd/scripted.py):
    `num_episodes` episodes of `task_name` ("sim_insert_peg", ...) are run SIDE BY SIDE on the device -- one env each, object poses from the
    task's own reset sampling (global numpy RNG, `seed` seeds it) -- through the Cartesian-action env (sim_env.py:277-312), and come back as
    a list of episode dicts in the layout of record_sim_episodes.py:155-212 (`/observations/{qpos,qvel,all_qpos}`, `/action` = the joint-space
    control with normalised grippers, `/observations/images/<cam>` u8 (T, H, W, 3); T = steps + 1, float32) next to per-episode
    {"max_reward_reached", "success", "rewards", "diverged"}.

    With cameras the episodes are run in batches sized so that one batch's images stay below `image_budget_bytes` of host memory (every
    step's images are written straight into the per-episode arrays: one copy, not three); the object poses are drawn for ALL episodes first,
    in episode order, so the data set does not depend on the batching.  An env whose state diverged during some step (avsim_get_diag, the
    flag the env facades turn into PhysicsError / `truncated`) was put back to its reset state mid-episode: such an episode is dropped, as the
    reference drops an episode whose physics raised (unless keep_diverged; it is flagged either way).  only_success: keep the episodes whose
    LARGEST reward over time is max_reward -- check_dataset_reward.py:52-58's criterion; "success" is that flag, "final_success" says
    whether the episode also ENDS at max_reward.  sink(episode): called for every kept episode as its batch finishes INSTEAD of collecting
    them (a recorder that writes and forgets keeps one batch in memory; the function then returns the episodes' summaries without "data").
    stream_dir: the episodes are written as they are recorded -- every step's images go straight into `<stream_dir>/episode_<i>.hdf5`
    (hdf5min.StreamWriter: a chunk per frame, as save_episode lays the image stacks out), the tables follow at the end, a dropped episode's file
    is removed, the kept ones are numbered consecutively from the files already there.  No image stays in memory, so the batches are not sized by
    the image budget but by `max_batch`: a step of 32 envs costs what a step of 3 costs (one wave each), which made the budgeted batches the
    slow part of a recording with cameras.  The summaries then carry "path" instead of "data"."""
    from . import scripted
    from .env import sample_object_poses
    from .sim_env import make_sim_env, _TASK_OF_SUBSTRING
    task = next(key for sub, key in _TASK_OF_SUBSTRING if sub in task_name)
    n_all = int(num_episodes)
    if seed is not None:
        np.random.seed(seed)
    poses_all = np.stack([sample_object_poses(task) for _ in range(n_all)])           # the draws of n_all sequential env.reset() calls
    cameras = list(cameras)
    b = lambda a, n: np.asarray(a)[None] if n == 1 else np.asarray(a)                 # batch axis for a single env
    episodes = []
    start = 0
    while start < n_all:
        # batch size from the image budget: T is only known once the script exists, so size it with the longest script (600 steps)
        per_ep = _image_bytes_per_step(cameras) * 601
        n = min(n_all - start, max_batch) if (per_ep == 0 or stream_dir is not None) else max(1, min(n_all - start, int(image_budget_bytes // per_ep)))
        env = make_sim_env(task_name, cameras=cameras, num_envs=n, device=device)
        env.sim.reset(poses_all[start:start + n])
        obs = env.get_obs()
        home = {k: b(obs["poses"][k], n).copy() for k in ("left", "right", "middle")}
        script = scripted.make_script(scripted.SCRIPT_OF_TASK[task], home, b(obs["qpos"], n), **script_kw)
        T = script.steps() + 1
        max_reward = env.sim.max_reward
        fields = {"/observations/qpos": lambda s: s["joints"]["position"], "/observations/qvel": lambda s: s["joints"]["velocity"],
                  "/observations/all_qpos": lambda s: s["qpos"], "/action": lambda s: s["control"]}
        # per-episode arrays, written step by step
        data = [{name: np.empty((T,) + b(f(obs), n).shape[1:], dtype=np.float32) for name, f in fields.items()} for _ in range(n)]
        writers = None
        if stream_dir is not None:
            from . import hdf5min
            os.makedirs(stream_dir, exist_ok=True)
            writers = [hdf5min.StreamWriter(os.path.join(stream_dir, f".recording_{start + k}.part")) for k in range(n)]
        else:
            for k in range(n):
                for cam in obs.get("images", {}):
                    data[k][f"/observations/images/{cam}"] = np.empty((T,) + b(obs["images"][cam], n).shape[1:], dtype=np.uint8)
        rewards = np.zeros((T - 1, n), dtype=np.int32)
        diverged = np.zeros(n, dtype=bool)

        def put(t, o):
            for name, f in fields.items():
                v = b(f(o), n)
                for k in range(n):
                    data[k][name][t] = v[k]
            for cam, img in o.get("images", {}).items():
                img = b(img, n)
                for k in range(n):
                    if writers is not None:
                        writers[k].append(f"/observations/images/{cam}", img[k])
                    else:
                        data[k][f"/observations/images/{cam}"][t] = img[k]
        try:
            put(0, obs)
            for t in range(1, T):
                _, rw, _ = env.sim.step_cartesian(script.action(b(obs["qpos"], n)))
                diverged |= (env.sim.diag()[:, 3] & 1).astype(bool)
                rewards[t - 1] = rw
                obs = env.get_obs()
                put(t, obs)
        except BaseException:
            for w in writers or []:
                w.abort()                       # (no half-written episode files behind a failed recording)
            raise
        finally:
            env.close()
        for k in range(n):
            reached = int(rewards[:, k].max())
            ok = reached == max_reward
            if (only_success and not ok) or (diverged[k] and not keep_diverged):
                if writers is not None:
                    writers[k].abort()
                continue
            if writers is not None:
                writers[k].finish(data[k], attrs={"sim": np.bool_(True)})
                i = 0
                while os.path.exists(os.path.join(stream_dir, f"episode_{i}.hdf5")):
                    i += 1
                path = os.path.join(stream_dir, f"episode_{i}.hdf5")
                os.replace(writers[k].path, path)
                episodes.append({"path": path, "success": bool(ok), "final_success": bool(rewards[-1, k] == max_reward), "max_reward_reached": reached,
                                 "rewards": rewards[:, k].copy(), "max_reward": int(max_reward), "diverged": bool(diverged[k]), "episode_index": start + k, "steps": T})
                continue
            ep = {"data": data[k], "success": bool(ok), "final_success": bool(rewards[-1, k] == max_reward), "max_reward_reached": reached,
                             "rewards": rewards[:, k].copy(), "max_reward": int(max_reward), "diverged": bool(diverged[k]), "episode_index": start + k}
            if sink is not None:
                sink(ep)
                ep = {k2: v for k2, v in ep.items() if k2 != "data"}
                data[k] = None
            episodes.append(ep)
        start += n
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


def rerender_episode(data, env_id: str, save_path: str | None = None, device: int = 0, frames_per_batch: int = 128, env=None, use_h5py: bool | None = None):
    """gym_guided_vision/scripts/replay_sim_episode.py:47-89 `replay_episode`: `data` is an episode (a path or a loaded dict with
    `/observations/{qpos,qvel,all_qpos}` and `/action`); `env_id` names the gym env whose camera configuration the new data set is for
    (`gym_guided_vision/<Task>-{2,3}Arms-v0`: 6 cameras for 3 arms, 4 for 2, 480 x 640, __init__.py:6-19).  Every recorded full state
    `all_qpos[t]` is put back (`set_qpos`, env.py:251-253) and the env's cameras are rendered (`get_obs()["pixels"]`, env.py:180-188); the
    result holds `/observations/qpos`, `/observations/qvel`, `/action` -- the first 14 columns for a 2-arm env, all 21 otherwise (:62-73) --
    and `/observations/images/<cam>` u8 (T, H, W, 3) per camera; `all_qpos` is not carried over (the reference's data_dict does not have it).
    The frames are independent, so they are T envs of one batched handle, `frames_per_batch` at a time (one set_qpos + one render call each).
    save_path: also written there in the reference's layout (replay_sim_episode.py:11-44).  env: a batched gym env of `env_id` with
    num_envs == frames_per_batch to reuse (rerender_dataset passes one)."""
    from .env import ENVS, make
    if isinstance(data, str):
        data = load_episode(data)
    spec = ENVS[env_id]
    all_qpos = np.asarray(data["/observations/all_qpos"], dtype=np.float64)
    T = all_qpos.shape[0]
    na = 14 if spec["num_arms"] == 2 else 21
    out = {"/observations/qpos": np.ascontiguousarray(np.asarray(data["/observations/qpos"])[:, :na]),
           "/observations/qvel": np.ascontiguousarray(np.asarray(data["/observations/qvel"])[:, :na]),
           "/action": np.ascontiguousarray(np.asarray(data["/action"])[:, :na])}
    own = env is None
    B = max(1, min(int(frames_per_batch), T)) if own else env.num_envs
    if own:
        env = make(env_id, num_envs=B, device=device)
    assert env.num_arms == spec["num_arms"] and list(env.cameras) == list(spec["cameras"]), "env does not have the camera configuration of env_id"
    H, W = env.observation_height, env.observation_width
    for cam in env.cameras:
        out[f"/observations/images/{cam}"] = np.empty((T, H, W, 3), dtype=np.uint8)
    nq = env.sim.nq
    assert all_qpos.shape[1] == nq, f"the episode's all_qpos has {all_qpos.shape[1]} columns, the model of {env_id} {nq}"
    for t0 in range(0, T, B):
        q = all_qpos[t0:t0 + B]
        n = q.shape[0]
        if n < B:                                   # the last chunk: the spare envs repeat its last frame
            q = np.concatenate([q, np.repeat(q[-1:], B - n, 0)])
        env.sim.set_qpos(q)
        if n == B:      # a full chunk: every camera's frames straight from the device into their place in the episode's array
            for cam in env.cameras:
                env.sim.render_rgb([cam], H, W, cam_major=True, out=out[f"/observations/images/{cam}"][t0:t0 + B])
        else:
            img = env.sim.render_rgb(env.cameras, H, W, cam_major=True)   # [ncam, B, H, W, 3]: a camera's frames are one contiguous block
            for ci, cam in enumerate(env.cameras):
                out[f"/observations/images/{cam}"][t0:t0 + n] = img[ci, :n]
    if own:
        env.close()
    if save_path is not None:
        os.makedirs(os.path.dirname(os.path.abspath(save_path)), exist_ok=True)
        d, f = os.path.split(os.path.abspath(save_path))
        assert f.startswith("episode_") and f.endswith(".hdf5"), "save_path is <dir>/episode_<i>.hdf5 (replay_sim_episode.py:104-107)"
        save_episode(out, d, int(f[len("episode_"):-len(".hdf5")]), use_h5py=use_h5py)
    return out


def rerender_dataset(dataset_dir: str, env_id: str, episode_idx: int | None = None, device: int = 0, frames_per_batch: int = 128):
    """replay_sim_episode.py:92-113 `main`: every `episode_*.hdf5` of `dataset_dir` (or the one with `episode_idx`) re-rendered for
    `env_id` into `<dataset_dir>/<EnvName>/episode_<i>.hdf5` (EnvName = the id without its namespace).  Returns the written paths and
    the frames per second over the whole run."""
    import glob
    import time
    from .env import make
    pat = "episode_*.hdf5" if episode_idx is None else f"episode_{episode_idx}.hdf5"
    paths = sorted(glob.glob(os.path.join(dataset_dir, pat)))
    env = make(env_id, num_envs=frames_per_batch, device=device) if paths else None
    written, frames, t0 = [], 0, time.time()
    for p in paths:
        save_path = os.path.join(dataset_dir, env_id.split("/")[-1], os.path.basename(p))
        out = rerender_episode(p, env_id, save_path, env=env)
        frames += out["/action"].shape[0]
        written.append(save_path)
    if env is not None:
        env.close()
    dt = time.time() - t0
    return written, (frames / dt if dt > 0 else 0.0)
