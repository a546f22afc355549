#!/usr/bin/env python
"""Device against the FAITHFUL oracle on the five scripted episodes -> profiles/r05_fidelity.json.

The CPU oracle has no LDS or register budget, so since round 5 it can run what MuJoCo runs where the device takes a shortcut:
  * mesh geoms collide as the FULL convex hulls of their STL files (18 032 vertices over 27 meshes; models/oracle_full_hulls.*,
    compile.py --oracle-hulls full) instead of the device's collision hulls (<= 128 vertices per mesh behind a support table since round 5:
    at most 0.3 mm of a mesh sticks out, 0.02 mm for the fingers; 20 / 32 vertices and up to 8.9 mm in rounds 1-4:
    profiles/r05_fidelity_decimated20_32.json is this report for those) (aloha_sim.xml:106-111 class "collision");
  * box-box manifolds keep every clipped vertex, up to 8 (the device does too since round 5; the four-point reduction of rounds 1-4 is
    kept in the oracle as a switch so that its effect is a number here).
The scripted policy of every task runs closed loop on the device (f64 physics), its ctrl sequence and per-step states are recorded, and
three oracles look at them: "matched" (the device's hulls, 8 points: the parity oracle), "faithful" (full hulls, 8 points), "r04"
(the device's hulls, 4 points).  Two comparisons each:
  * teacher-forced: at every env-step the oracle starts from the device's state and steps the device's ctrl once -- contact counts,
    rewards, success flags and the one-step position difference without the divergence of two chaotic trajectories in between;
  * open-loop replay of the whole ctrl sequence: final success flags, largest reward reached, and the grasp: distance between the carried
    object and the gripper that holds it, device against oracle, over the carry phase (slip).

    python tools/fidelity.py [--envs 16] [--out profiles/r05_fidelity.json]        (needs the GPU; the oracle runs on the host cores)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import episode_util as U

TASKS = ["slot_insertion", "insert_peg", "sew_needle_thread", "hook_package", "tube_transfer"]
MODES = {"matched": dict(hulls="model", boxbox_points=8), "faithful": dict(hulls="full", boxbox_points=8), "r04_four_points": dict(hulls="model", boxbox_points=4)}
# (task: qpos slice of the object's position that a gripper carries, the arm whose wrist joints define "the hand": left 0..5, right 8..13)
CARRIED = {"slot_insertion": (slice(30, 33), "stick"), "insert_peg": (slice(23, 26), "peg"), "sew_needle_thread": (slice(30, 33), "needle"),
           "hook_package": (slice(30, 33), "package"), "tube_transfer": (slice(23, 26), "tube1")}


def pct(x):
    x = np.asarray(x, dtype=np.float64)
    return [float(v) for v in np.percentile(x, [50, 90, 100])] if x.size else [0.0, 0.0, 0.0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_fidelity.json"))
    ap.add_argument("--tasks", nargs="*", default=TASKS)
    a = ap.parse_args()
    out = {"note": __doc__.split("\n\n")[0] + "  See tools/fidelity.py for what every field is.", "envs_per_task": a.envs, "device": "f64 physics (AVSIM_F64_PHYSICS), collision hulls of <= 128 vertices (support tables), 8-point box-box",
           "modes": MODES, "tasks": {}}
    for task in a.tasks:
        dev = U.device_episode(task, a.envs, f64=True, record_state=True)
        T = dev["ctrl"].shape[0]
        row = {"steps": T, "device_success_rate": float(dev["success"][-1].mean()), "device_max_reward_hist": np.bincount(dev["reward"].max(0), minlength=6).tolist(),
               "device_ncon_mean_max": [float(dev["ncon"].mean()), int(dev["ncon"].max())], "device_diverged": int(dev["diverged"].sum()), "device_capped": int(dev["capped"].sum())}
        sl, name = CARRIED[task]
        for mode, kw in MODES.items():
            U.ORACLE_MODE.update(kw)
            ls = U.compare_lockstep(task, dev)
            rp_rows = U.pool_map(U.replay_worker, [(task, dev["poses"][k], np.ascontiguousarray(dev["ctrl"][:, k])) for k in range(a.envs)])
            steps = sum(r["steps"] for r in ls)
            # open-loop replay
            succ_mis = sum(bool(dev["success"][-1, k]) != bool(su[-1]) for k, (rw, su, qs, nc) in enumerate(rp_rows))
            maxr_mis = sum(int(dev["reward"][:, k].max()) != int(rw.max()) for k, (rw, su, qs, nc) in enumerate(rp_rows))
            rdiff = [int((rw != dev["reward"][:, k]).sum()) for k, (rw, su, qs, nc) in enumerate(rp_rows)]
            # the carried object's position, oracle against device, while the device holds it off the table (reward >= 2 on the device)
            slip = []
            for k, (rw, su, qs, nc) in enumerate(rp_rows):
                held = dev["reward"][:, k] >= 2
                if held.any():
                    slip.append(float(np.linalg.norm(qs[held][:, sl] - dev["qpos"][held, k][:, sl], axis=1).max()))
            row[mode] = {
                "teacher_forced": {"env_steps": steps, "reward_diff_steps": sum(r["reward_diff_steps"] for r in ls), "success_flag_diff_steps": sum(r["success_diff_steps"] for r in ls),
                                   "ncon_diff_steps": sum(r["ncon_diff_steps"] for r in ls), "ncon_diff_fraction": sum(r["ncon_diff_steps"] for r in ls) / steps,
                                   "one_step_max_qpos_err_p50_p90_max_over_envs": pct([r["max_step_err"] for r in ls])},
                "open_loop_replay": {"final_success_mismatches": int(succ_mis), "max_reward_mismatches": int(maxr_mis), "oracle_success_rate": float(np.mean([su[-1] for rw, su, qs, nc in rp_rows])),
                                     "reward_diff_steps_p50_p90_max_over_envs": pct(rdiff), "oracle_ncon_mean_max": [float(np.mean([nc.mean() for rw, su, qs, nc in rp_rows])), int(max(nc.max() for rw, su, qs, nc in rp_rows))],
                                     f"carried_{name}_position_difference_while_held_m_p50_p90_max_over_envs": pct(slip)}}
            print(task, mode, json.dumps(row[mode]), flush=True)
        out["tasks"][task] = row
    U.ORACLE_MODE.update(MODES["matched"])
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
