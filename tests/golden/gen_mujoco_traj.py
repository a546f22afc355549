"""MuJoCo pinning trajectories.

Runs the REFERENCE's own gym environments (real MuJoCo + dm_control, gym_guided_vision/gym_guided_vision/env.py:203-249) through
the action script of tests/mj_actions.py and records, per env-step, the full state and what the reward was computed from:
qpos, qvel, ctrl, ncon, the contact list (geom names, dist), reward, is_success.  Output: tests/golden/mujoco_traj_<task>_<n>arms.npz.
tests/test_mujoco_pin.py compares the oracle and the device against these files whenever they exist.

It needs `mujoco`, `dm_control`, `gymnasium` and /root/reference -- none of the first three is installed in the build image of
rounds 1-2, so the files could not be produced yet and the physics half of the oracle stays "parity unpinned".  The first machine
that has them turns the pin on:

    pip install mujoco dm_control gymnasium   # where a network exists
    PYTHONPATH=/root/reference/gym_guided_vision python tests/golden/gen_mujoco_traj.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    try:
        import mujoco  # noqa: F401
        import gymnasium as gym
    except ImportError as e:      # the documented state of the build image
        raise SystemExit(f"gen_mujoco_traj.py needs real MuJoCo ({e}); nothing written")
    sys.path.insert(0, "/root/reference/gym_guided_vision")
    import gym_guided_vision  # noqa: F401  (registers the ids, __init__.py:88-101)
    import mj_actions as A
    for gym_name, key in A.TASKS:
        for arms in (2, 3):
            env = gym.make(f"gym_guided_vision/{gym_name}-{arms}Arms-v0", cameras=[])
            np.random.seed(A.SEED)                       # the reference samples object poses from the global RNG (env.py:482 ...)
            obs, info = env.reset()
            u = env.unwrapped
            ph = u._physics
            rec = {k: [] for k in ("qpos", "qvel", "ctrl", "ncon", "reward", "success", "agent_pos", "con_dist", "con_geom1", "con_geom2")}
            names = [ph.model.id2name(i, "geom") for i in range(ph.model.ngeom)]
            q0 = ph.data.qpos.copy()
            for a in A.actions(arms):
                obs, reward, term, trunc, info = env.step(a)
                rec["qpos"].append(ph.data.qpos.copy()); rec["qvel"].append(ph.data.qvel.copy()); rec["ctrl"].append(ph.data.ctrl.copy())
                nc = int(ph.data.ncon)
                rec["ncon"].append(nc); rec["reward"].append(int(reward)); rec["success"].append(bool(info["is_success"]))
                rec["agent_pos"].append(np.asarray(obs["agent_pos"]).copy())
                d = np.full(64, np.nan); g1 = -np.ones(64, dtype=np.int32); g2 = -np.ones(64, dtype=np.int32)
                for i in range(min(nc, 64)):
                    c = ph.data.contact[i]
                    d[i], g1[i], g2[i] = c.dist, c.geom1, c.geom2
                rec["con_dist"].append(d); rec["con_geom1"].append(g1); rec["con_geom2"].append(g2)
            out = {k: np.array(v) for k, v in rec.items()}
            out["qpos0"] = q0
            out["geom_names"] = np.array(names)
            out["mujoco_version"] = np.array(mujoco.__version__)
            np.savez_compressed(os.path.join(HERE, f"mujoco_traj_{key}_{arms}arms.npz"), **out)
            print("wrote", key, arms, "final reward", rec["reward"][-1])
            env.close()


if __name__ == "__main__":
    main()
