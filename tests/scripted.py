"""Scripted Cartesian targets for the contact-rich configurations (BASELINE config 3: reach - grasp - lift of the needle by
the right arm, targets derived from the sampled needle pose).  The 23-D actions are those of sim_env.py:277-312:
[left pos 3, quat wxyz 4, trigger, right pos 3, quat 4, trigger, middle pos 3, quat 4], trigger 1 = closed."""
import numpy as np

from av_aloha_amd.workloads import GRASP_HEIGHT, grasp_lift_targets, qmul  # noqa: F401  (config 3 lives with the bench workloads)


class SlotInsertionScript:
    """One manipulator grasps the stick top-down at its centre, lifts it, carries it over the slot, lowers it until its
    underside is 1.5 cm above the slot walls (4 cm high: lower, and the opening fingers press on the walls and keep the stick pinched) and lets go (the arm nearer to the stick does it) (task_slot_insertion.xml:5-16; clearance 4 mm a side): the pins
    touch when the stick has dropped in (env.py:584-587, reward 4).  Closed loop on the measured stick and slot poses
    (qpos[23:30] slot, [30:37] stick): the carry / lower phases add the integrated xy error to the hand target."""
    T = (60, 40, 25, 40, 90, 50, 10, 15, 20)

    def __init__(self, home, qpos, drop=0.055, clip=0.05, gain=0.15, yaw_gain=0.15, yaw_clip=1.2, side=0.04):
        n = qpos.shape[0]
        self.n = n
        self.home = home
        self.drop, self.clip, self.gain = drop, clip, gain
        c, s = np.cos(np.pi / 4), np.sin(np.pi / 4)
        self.down_r = np.stack([qmul(np.array([c, 0.0, -s, 0.0]), home["right"][i, 3:]) for i in range(n)])
        self.down_l = np.stack([qmul(np.array([c, 0.0, s, 0.0]), home["left"][i, 3:]) for i in range(n)])
        self.stick0 = qpos[:, 30:33].copy()
        self.use_left = self.stick0[:, 0] < 0.0          # the nearer arm carries (top-down reach ends near the far side)
        # grasp `side` metres off the stick's centre, towards the carrying arm (the stick is 34 cm long): that much less reach
        self.off = np.zeros((n, 2))
        self.off[:, 0] = np.where(self.use_left, -side, side)
        self.corr = np.zeros((n, 2))
        # the pinched stick follows the hand's rotation about the vertical, and the IK trades some of the commanded orientation
        # for its joint-centring terms on the way to the slot: the commanded hand yaw integrates the measured stick / slot yaw error
        self.yaw_gain, self.yaw_clip = yaw_gain, yaw_clip
        self.yaw = np.zeros(n)
        self.t = 0

    def phase(self):
        t = self.t
        for k, d in enumerate(self.T):
            if t < d:
                return k, (t + 1) / d
            t -= d
        return len(self.T) - 1, 1.0

    def steps(self):
        return sum(self.T)

    def action(self, qpos):
        n = self.n
        k, f = self.phase()
        slot, stick = qpos[:, 23:26], qpos[:, 30:33]
        zc = 0.02 + GRASP_HEIGHT                      # site height that pinches the stick at mid height on the table
        hi = zc + 0.10
        base = self.stick0[:, :2] + self.off
        grip = 0.0
        if k == 0:
            z = hi
        elif k == 1:
            z = hi + (zc - hi) * min(1.0, f / 0.8)
        elif k == 2:
            z, grip = zc, min(1.0, f / 0.6)
        elif k == 3:
            z, grip = zc + (hi - zc) * min(1.0, f / 0.8), 1.0
        else:
            g = min(1.0, f / 0.7) if k == 4 else 1.0
            if (k == 4 and f > 0.7) or k in (5, 6):
                self.corr = np.clip(self.corr + self.gain * (slot[:, :2] - stick[:, :2]), -self.clip, self.clip)
            if k in (4, 5, 6):
                yaw_of = lambda qq: 2.0 * np.arctan2(qq[:, 3], qq[:, 0])
                err = yaw_of(qpos[:, 26:30]) - yaw_of(qpos[:, 33:37])
                err = (err + np.pi / 2) % np.pi - np.pi / 2                  # the stick fits either way round
                self.yaw = np.clip(self.yaw + self.yaw_gain * err, -self.yaw_clip, self.yaw_clip)
            base = self.stick0[:, :2] + self.off + g * (slot[:, :2] - self.stick0[:, :2]) + self.corr
            zr = zc + self.drop
            z = hi if k == 4 else (hi + (zr - hi) * min(1.0, f / 0.8) if k == 5 else zr)
            grip = 1.0 if k <= 6 else (max(0.0, 1.0 - f / 0.5) if k == 7 else 0.0)
        a = np.zeros((n, 23))
        a[:, 0:7] = self.home["left"]
        a[:, 8:15] = self.home["right"]
        L, R = self.use_left, ~self.use_left
        qz = np.stack([np.cos(self.yaw / 2), np.zeros(n), np.zeros(n), np.sin(self.yaw / 2)], axis=1)
        dl = np.stack([qmul(qz[i], self.down_l[i]) for i in range(n)])
        dr = np.stack([qmul(qz[i], self.down_r[i]) for i in range(n)])
        a[L, 0:2] = base[L]; a[L, 2] = z; a[L, 3:7] = dl[L]; a[L, 7] = grip
        a[R, 8:10] = base[R]; a[R, 10] = z; a[R, 11:15] = dr[R]; a[R, 15] = grip
        a[:, 16:23] = self.home["middle"]
        self.t += 1
        return a
