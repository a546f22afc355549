"""Frames per second of the re-render replay (harness.rerender_dataset = gym_guided_vision/scripts/replay_sim_episode.py): scripted
SlotInsertion episodes are recorded without images, then re-rendered for the 3-arm (6 cameras) and the 2-arm (4 cameras) camera
configuration at 480 x 640, with and without writing the HDF5 files.   usage: python tools/prof_rerender.py [episodes]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from av_aloha_amd import harness
from av_aloha_amd.env import make

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
d = tempfile.mkdtemp()
eps = harness.record_scripted("sim_slot_insertion", n, seed=1)
for i, e in enumerate(eps):
    harness.save_episode(e["data"], d, i)
T = eps[0]["data"]["/action"].shape[0]
print(f"{len(eps)} scripted SlotInsertion episodes of {T} frames recorded (no images), {sum(e['success'] for e in eps)} reach max_reward")
for env_id in ("gym_guided_vision/SlotInsertion-3Arms-v0", "gym_guided_vision/SlotInsertion-2Arms-v0"):
    ncam = 6 if "3Arms" in env_id else 4
    for fpb in (128, 351):
        env = make(env_id, num_envs=fpb)
        harness.rerender_episode(eps[0]["data"], env_id, env=env)            # warm-up: visual scene upload, scratch
        t0 = time.time()
        for e in eps:
            out = harness.rerender_episode(e["data"], env_id, env=env)
        dt = time.time() - t0
        env.close()
        print(f"{env_id}: {len(eps) * T / dt:7.0f} frames/s = {len(eps) * T * ncam / dt:8.0f} images/s of 480 x 640 x 3 ({ncam} cameras per frame, {fpb} frames per batch; "
              f"host copies included, no file)")
    t0 = time.time()
    written, fps = harness.rerender_dataset(d, env_id)
    print(f"{env_id}: {fps:7.0f} frames/s with the HDF5 files written ({len(written)} files of {os.path.getsize(written[0]) / 1e6:.0f} MB, pure-Python writer unless h5py is installed)")
