"""Scripted Cartesian targets for the contact-rich configurations (BASELINE config 3: reach - grasp - lift of the needle by
the right arm, targets derived from the sampled needle pose).  The 23-D actions are those of sim_env.py:277-312:
[left pos 3, quat wxyz 4, trigger, right pos 3, quat 4, trigger, middle pos 3, quat 4], trigger 1 = closed."""
import numpy as np

# the control site sits at the wrist (aloha_sim.xml:249 right_gripper_control), the pinch point 0.13 m further along the
# gripper (:248 right_gripper): a top-down grasp of a 2 cm bar lying on the table holds the site this far above its centre
GRASP_HEIGHT = 0.14


def qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def grasp_lift_targets(home, obj_xyz, T=(70, 50, 30, 60, 40), lift=0.12):
    """home: {'left','right','middle'} -> [N, 7] eef poses at reset (obs['poses']); obj_xyz [N, 3]: centre of the bar.
    Yields the [N, 23] action of every step: move above, descend, close, lift, hold.  The right gripper points straight
    down: the home orientation turned by -90 degrees about the world y axis (the right arm faces -x)."""
    n = obj_xyz.shape[0]
    ry = np.array([np.cos(-np.pi / 4), 0.0, np.sin(-np.pi / 4), 0.0])
    down = np.stack([qmul(ry, home["right"][i, 3:]) for i in range(n)])
    grasp = obj_xyz + np.array([0.0, 0.0, GRASP_HEIGHT])
    above = grasp + np.array([0.0, 0.0, 0.10])
    up = np.array([0.0, 0.0, lift])

    def act(rpos, grip):
        a = np.zeros((n, 23))
        a[:, 0:7] = home["left"]
        a[:, 8:11] = rpos
        a[:, 11:15] = down
        a[:, 15] = grip
        a[:, 16:23] = home["middle"]
        return a
    for t in range(T[0]):
        yield act(above, 0.0)
    for t in range(T[1]):
        yield act(above + (grasp - above) * min(1.0, (t + 1) / (0.7 * T[1])), 0.0)
    for t in range(T[2]):
        yield act(grasp, min(1.0, (t + 1) / (0.5 * T[2])))
    for t in range(T[3]):
        yield act(grasp + up * min(1.0, (t + 1) / (0.67 * T[3])), 1.0)
    for t in range(T[4]):
        yield act(grasp + up, 1.0)
