#!/usr/bin/env python
"""Re-render a recorded data set for another camera configuration: the counterpart of gym_guided_vision/scripts/replay_sim_episode.py
(same flags).  Every episode_*.hdf5 of --dataset_dir is loaded, its recorded full states are put back frame by frame and the cameras
registered for --env are rendered (T frames = T envs of one batched handle); the result goes to <dataset_dir>/<EnvName>/episode_<i>.hdf5
with qpos / qvel / action cut to 14 columns for a 2-arm env.

    python tools/replay_sim_episode.py --env gym_guided_vision/InsertPeg-2Arms-v0 --dataset_dir data/sim_insert_peg [--episode_idx 3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from av_aloha_amd import harness

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", required=True, help="gym id, e.g. gym_guided_vision/SlotInsertion-3Arms-v0")
    ap.add_argument("--dataset_dir", required=True)
    ap.add_argument("--episode_idx", type=int, default=None)
    ap.add_argument("--frames_per_batch", type=int, default=128)
    a = ap.parse_args()
    written, fps = harness.rerender_dataset(a.dataset_dir, a.env, a.episode_idx, frames_per_batch=a.frames_per_batch)
    print(f"{len(written)} episodes re-rendered for {a.env} at {fps:.0f} frames/s (all registered cameras per frame, incl. HDF5 writing) -> "
          f"{os.path.join(a.dataset_dir, a.env.split('/')[-1])}")
