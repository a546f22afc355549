#!/usr/bin/env python
"""Data-set recording with a scripted teleoperator: the counterpart of data_collection_scripts/record_sim_episodes.py (the reference
drives the sim with a VR headset and writes episode_<i>.hdf5, :155-212).  All episodes run side by side on the device.

    python tools/record_scripted_episodes.py --task_name sim_insert_peg --num_episodes 64 --dataset_dir data/sim_insert_peg \
        [--cameras zed_cam,cam_left_wrist] [--seed 0] [--only_success] [--check]

--check replays every saved episode on the task's gym env both ways the reference has: its recorded full states through set_qpos
(replay_sim_episode.py:221-262) and its recorded actions open loop through step_action from the first state
(gym_guided_vision/scripts/check_dataset_reward.py), and reports how many reach max_reward."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from av_aloha_amd import harness

GYM_ID = {"sim_insert_peg": "InsertPeg", "sim_slot_insertion": "SlotInsertion", "sim_sew_needle": "SewNeedle", "sim_tube_transfer": "TubeTransfer",
          "sim_hook_package": "HookPackage"}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--task_name", required=True, help="sim_insert_peg | sim_slot_insertion | sim_sew_needle | sim_tube_transfer | sim_hook_package (record_sim_episodes.py:33)")
    ap.add_argument("--num_episodes", type=int, default=16)
    ap.add_argument("--dataset_dir", required=True)
    ap.add_argument("--cameras", default="", help="comma-separated camera names of the Cartesian env (sim_env.py:22: zed_cam = the stereo pair 720 x 1440; cam_left_wrist, cam_right_wrist, cam_high, cam_low 480 x 640); none by default")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--only_success", action="store_true")
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    cams = [c for c in args.cameras.split(",") if c]
    t0 = time.time()
    # the episodes are written while they are recorded (harness.record_scripted stream_dir: no image kept in memory, all episodes side by side)
    eps = harness.record_scripted(args.task_name, args.num_episodes, cameras=cams, seed=args.seed, only_success=args.only_success, stream_dir=args.dataset_dir)
    paths = [e["path"] for e in eps]
    T = [eps[0]["steps"] if eps else 0]
    t1 = time.time()
    print(f"{args.task_name}: {len(eps)} episodes of {T[0]} steps recorded and saved to {args.dataset_dir} in {t1 - t0:.1f} s (written while recording; "
          f"{args.num_episodes} run side by side), {sum(e['success'] for e in eps)} reach max_reward {eps[0]['max_reward'] if eps else '-'} "
          f"({sum(e['final_success'] for e in eps)} end there), {args.num_episodes - len(eps)} dropped (diverged{' / unsuccessful' if args.only_success else ''})")
    if args.check and eps:
        from av_aloha_amd.env import make
        key = next(v for k, v in GYM_ID.items() if k in args.task_name)
        env = make(f"gym_guided_vision/{key}-3Arms-v0", cameras=[])
        ok = 0
        for p, e in zip(paths, eps):
            _, rewards = harness.replay_episode(env, harness.load_episode(p))
            ok += int(rewards.max() == env.max_reward)
        env.close()
        print(f"state replay: {ok} / {len(paths)} episodes reach max_reward {env.max_reward} when their recorded states are replayed through set_qpos (replay_sim_episode.py)")
        passed, _ = harness.check_dataset_reward(f"gym_guided_vision/{key}-3Arms-v0", [harness.load_episode(p) for p in paths])
        print(f"check_dataset_reward: {int(passed.sum())} / {len(paths)} episodes reach max_reward when their recorded ACTIONS are stepped open loop on the gym env "
              f"(gym_guided_vision/scripts/check_dataset_reward.py)")
