"""The action script of the MuJoCo pinning trajectories (tests/golden/gen_mujoco_traj.py writes them with real MuJoCo, tests/
test_mujoco_pin.py replays them on the oracle and on the device): the arms swing around the home pose, the grippers toggle, the
camera arm sways -- the same family as actions_wiggle of tests/test_gpu_physics.py.  float32, as the env API takes them."""
import numpy as np

from av_aloha_amd.constants import LEFT_ARM_POSE, MIDDLE_ARM_POSE, RIGHT_ARM_POSE

TASKS = (("InsertPeg", "insert_peg"), ("SlotInsertion", "slot_insertion"), ("SewNeedle", "sew_needle"), ("TubeTransfer", "tube_transfer"),
         ("HookPackage", "hook_package"))
T_STEPS = 40
SEED = 12345


def actions(num_arms: int, T: int = T_STEPS) -> np.ndarray:
    nj = 21 if num_arms == 3 else 14
    a0 = np.concatenate([LEFT_ARM_POSE[:6], [1.0], RIGHT_ARM_POSE[:6], [1.0], MIDDLE_ARM_POSE])[:nj]
    out = np.empty((T, nj), dtype=np.float32)
    for t in range(T):
        a = a0.copy()
        a[0] += 0.2 * np.sin(0.3 * t)
        a[1] += 0.1 * np.sin(0.2 * t)
        a[7] -= 0.2 * np.sin(0.25 * t)
        a[6] = 1.0 if (t // 5) % 2 == 0 else 0.0
        a[13] = 0.0 if (t // 7) % 2 == 0 else 1.0
        if nj == 21:
            a[14] += 0.3 * np.sin(0.15 * t)
            a[18] += 0.2 * np.cos(0.2 * t)
        out[t] = a
    return out
