"""Whole-episode parity table, device vs CPU oracle (tests/episode_util.py; run on the MI355X box):
    python tools/report_episode_parity.py [out.json]
f64 device mode and the f32 product mode of the scripted episodes of the four task families north_star names -- SlotInsertion (grasp -
carry - insert, 350 env-steps), InsertPeg (two pitched grasps, peg into the tube, 350), SewNeedle (all five reward stages: grasp,
thread through the wall's window, hand over to the left gripper, 535; and BASELINE config 3's reach - grasp - lift, 250), HookPackage
(two-arm carry onto the hook, release, 410) -- against the oracle stepping the same ctrl sequences: per-step reward agreement, final
is_success (env.py:224) on both sides, position distance over the episode."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import episode_util as U

out = {}
only = [a for a in sys.argv[2:]]
for task, n64, n32 in (("slot_insertion", 16, 128), ("insert_peg", 16, 128), ("sew_needle_thread", 16, 128), ("hook_package", 16, 128), ("tube_transfer", 16, 128), ("sew_needle", 16, 128)):
    if only and task not in only:
        continue
    for mode, n in (("f64", n64), ("f32", n32)):
        dev = U.device_episode(task, n, f64=mode == "f64", record_state=True)
        rows = U.compare_with_replay(task, dev)
        e = np.array([r["max_qpos_err"] for r in rows])
        mism = [r for r in rows if r["dev_success"] != r["orc_success"]]
        fin = [r for r in rows if r["dev_final_reward"] != r["orc_final_reward"]]
        out[f"{task}_{mode}"] = {
            "envs": n, "steps": int(dev["ctrl"].shape[0]), "seeds": f"{U.TASK_SEED[task]} + i",
            "device_success_rate": float(np.mean([r["dev_success"] for r in rows])), "oracle_success_rate": float(np.mean([r["orc_success"] for r in rows])),
            "device_final_reward_hist": np.bincount([r["dev_final_reward"] for r in rows], minlength=6).tolist(),
            "oracle_final_reward_hist": np.bincount([r["orc_final_reward"] for r in rows], minlength=6).tolist(),
            "success_flag_mismatches": len(mism), "final_reward_mismatches": len(fin),
            "envs_with_identical_reward_sequence": int(sum(r["first_reward_diff"] == -1 for r in rows)),
            "reward_steps_differing_p50_p90_max": [float(x) for x in np.percentile([r["n_reward_diff"] for r in rows], [50, 90, 100])],
            "device_max_reward_hist": np.bincount([r["dev_max_reward"] for r in rows], minlength=6).tolist(),
            "envs_with_identical_contact_counts": int(sum(r["ncon_diff_steps"] == 0 for r in rows)),
            "max_qpos_err_p50_p90_max": [float(np.percentile(e, 50)), float(np.percentile(e, 90)), float(e.max())],
            "held_phase": {"what": "the steps before the script's release / pour phase (tests/episode_util.py release_step): objects in the grippers or at rest",
                           "steps": rows[0]["held_steps"], "reward_diff_steps_total": int(sum(r["held_reward_diff"] for r in rows)),
                           "max_qpos_err_p50_p90_max": [float(x) for x in np.percentile([r["held_max_qpos_err"] for r in rows], [50, 90, 100])],
                           "ncon_diff_steps_p50_p90_max": [float(x) for x in np.percentile([r["held_ncon_diff_steps"] for r in rows], [50, 90, 100])]},
            "arm_joints_max_qpos_err_p50_p90_max": [float(x) for x in np.percentile([r["arm_max_qpos_err"] for r in rows], [50, 90, 100])],
            "ncon_diff_steps_p50_p90_max": [float(x) for x in np.percentile([r["ncon_diff_steps"] for r in rows], [50, 90, 100])],
            "diverged_envs": int(dev["diverged"].sum()), "capped_envs": int(dev["capped"].sum()),
            "mismatching_envs": mism[:16], "reward_diff_envs": [r for r in rows if r["first_reward_diff"] != -1][:16],
        }
        if True:
            ls = U.compare_lockstep(task, dev)
            tot = sum(r["steps"] for r in ls)
            out[f"{task}_{mode}"]["lockstep"] = {
                "what": "the oracle put into the device's state at the start of every env-step and stepped once with the device's ctrl (teacher-forced): per-step flags without the divergence of two chaotic trajectories",
                "env_steps": tot, "reward_diff_steps": int(sum(r["reward_diff_steps"] for r in ls)), "success_flag_diff_steps": int(sum(r["success_diff_steps"] for r in ls)),
                "final_success_flag_mismatches": int(sum(r["dev_success"] != r["orc_success"] for r in ls)),
                "envs_with_identical_reward_sequence": int(sum(r["reward_diff_steps"] == 0 for r in ls)),
                "ncon_diff_steps": int(sum(r["ncon_diff_steps"] for r in ls)),
                "max_one_step_qpos_err_p50_p99_max": [float(x) for x in np.percentile([r["max_step_err"] for r in ls], [50, 99, 100])],
                "median_one_step_qpos_err_max_over_envs": float(max(r["median_step_err"] for r in ls)),
                "steps_with_one_step_err_above_1e-6": int(sum(r["steps_err_above_1e6"] for r in ls))}
        print(task, mode, json.dumps({k: v for k, v in out[f"{task}_{mode}"].items() if not k.endswith("_envs") or k in ("diverged_envs", "capped_envs")}), flush=True)
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05_episode_parity.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
