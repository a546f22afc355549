"""Scripted teleoperators: Cartesian policies that solve the five tasks through the path a headset operator's actions take
(data_collection_scripts/sim_env.py:277-312: 23-D action = [left pos 3, quat wxyz 4, trigger, right pos 3, quat 4, trigger, middle pos
3, quat 4], trigger 1 = closed -> GradIK x2 + DiffIK on the measured joints -> 20 physics substeps).  The reference records its data
sets with a VR headset (record_sim_episodes.py:68-153); these stand in for the operator: `av_aloha_amd.harness.record_scripted` /
`tools/record_scripted_episodes.py` write their episodes in the reference's HDF5 layout, and the whole-episode parity tests
(tests/test_gpu_episode_parity.py) run them on the device and on the CPU oracle.  Closed loop on the measured object poses (the free
joints' qpos) and on the hands' measured positions (numpy product-of-exponentials FK).  Every script: .steps(), .action(qpos [n, nq])
-> [n, 23]; `make_script(name, home, qpos0)` builds one by name."""
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


# ---- the other task families (north_star: success flags for InsertPeg, SewNeedle, HookPackage as well) -----------------------------
def quat_rot(q, v):
    """Rotate v [n, 3] by unit quaternions q [n, 4] (w x y z)."""
    w, u = q[:, :1], q[:, 1:]
    t = 2.0 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


def yaw_quat(yaw):
    n = len(yaw)
    return np.stack([np.cos(yaw / 2), np.zeros(n), np.zeros(n), np.sin(yaw / 2)], axis=1)


def make_fk(task):
    """-> fk(qpos [n, nq], arm) -> 4x4 [n, 4, 4]: pose of the arm's control site (product of exponentials over the model's joint screws,
    kinematics.py:7-26; ik_* tables of the compiled data-collection model)."""
    import os
    from .compiler.compile import read_blob
    from .constants import MODEL_DIR
    md = read_blob(os.path.join(MODEL_DIR, f"dc_{task}_3arms.avm"))
    w0, p0, site0, qadr, nj = (np.asarray(md[k]) for k in ("ik_w0", "ik_p0", "ik_site0", "ik_qadr", "ik_n"))

    def fk(qpos, arm):
        n = qpos.shape[0]
        T = np.tile(np.eye(4), (n, 1, 1))
        for i in range(int(nj[arm])):
            th = qpos[:, int(qadr[arm, i])]
            w, p = w0[arm, i], p0[arm, i]
            K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
            R = np.eye(3)[None] + np.sin(th)[:, None, None] * K[None] + (1 - np.cos(th))[:, None, None] * (K @ K)[None]
            Ti = np.tile(np.eye(4), (n, 1, 1))
            Ti[:, :3, :3] = R
            Ti[:, :3, 3] = p[None] - R @ p
            T = T @ Ti
        return T @ site0[arm][None]
    return fk


class HandServo:
    """Integral correction of a hand's position target: GradIK settles a few millimetres (centimetres at the edge of the workspace) off
    the commanded point because its cost trades the pose against joint-centring terms (grad_ik.py:168-198); the command becomes target +
    the accumulated (target - measured control-site position), measured through `fk` on the joints.  It accumulates only while the hand is
    within `near` of its target (not while it is still travelling)."""

    def __init__(self, fk, arm, n, gain=0.3, clip=0.04, near=0.03):
        self.fk, self.arm, self.gain, self.clip, self.near = fk, arm, gain, clip, near
        self.acc = np.zeros((n, 3))

    def __call__(self, target, qpos):
        err = target - self.fk(qpos, self.arm)[:, :3, 3]
        close = np.linalg.norm(err, axis=1) < self.near
        self.acc[close] = np.clip(self.acc[close] + self.gain * err[close], -self.clip, self.clip)
        return target + self.acc


class _Phases:
    """Phase clock shared by the scripts: T = step counts; phase() -> (index, fraction done in (0, 1])."""
    T = ()

    def phase(self):
        t = self.t
        for k, d in enumerate(self.T):
            if t < d:
                return k, (t + 1) / d
            t -= d
        return len(self.T) - 1, 1.0

    def steps(self):
        return sum(self.T)

    def _down(self, home):
        n = self.n
        c, s = np.cos(np.pi / 4), np.sin(np.pi / 4)
        self.down_r = np.stack([qmul(np.array([c, 0.0, -s, 0.0]), home["right"][i, 3:]) for i in range(n)])
        self.down_l = np.stack([qmul(np.array([c, 0.0, s, 0.0]), home["left"][i, 3:]) for i in range(n)])

    def _assemble(self, lpos, lquat, lgrip, rpos, rquat, rgrip):
        a = np.zeros((self.n, 23))
        a[:, 0:3], a[:, 3:7], a[:, 7] = lpos, lquat, lgrip
        a[:, 8:11], a[:, 11:15], a[:, 15] = rpos, rquat, rgrip
        a[:, 16:23] = self.home["middle"]
        self.t += 1
        return a


def ramp(f, frac=0.8):
    return min(1.0, f / frac)


PINCH = 0.135          # control site (wrist, aloha_sim.xml:249) -> pinch point between the finger pads (:248: 0.13 along the gripper)
BASE_X, BASE_Y = 0.469, 0.032     # the manipulators' bases (aloha_sim.xml:119, :208): left at -BASE_X facing +x, right at +BASE_X facing -x


def radial_hand(P, pitch, home_quat, right):
    """Wrist target for a gripper whose pinch point is at P [n, 3], pitched `pitch` radians below the horizontal and heading along the
    line from its arm's base through P (the heading a 6-dof arm gives for free: the waist turns, nothing else has to; a heading held
    parallel to x costs forearm roll, which GradIK's joint-centring terms trade against the position: 3-4 cm off at |y| = 8 cm).
    -> (site position [n, 3], quaternion wxyz [n, 4], heading angle [n])"""
    n = P.shape[0]
    if right:
        psi = np.arctan2(BASE_Y - P[:, 1], BASE_X - P[:, 0])
        a = np.stack([-np.cos(pitch) * np.cos(psi), -np.cos(pitch) * np.sin(psi), -np.sin(pitch) * np.ones(n)], axis=1)
        qp = np.array([np.cos(pitch / 2), 0.0, -np.sin(pitch / 2), 0.0])
    else:
        psi = np.arctan2(P[:, 1] - BASE_Y, P[:, 0] + BASE_X)
        a = np.stack([np.cos(pitch) * np.cos(psi), np.cos(pitch) * np.sin(psi), -np.sin(pitch) * np.ones(n)], axis=1)
        qp = np.array([np.cos(pitch / 2), 0.0, np.sin(pitch / 2), 0.0])
    qz = yaw_quat(psi)
    quat = np.stack([qmul(qz[i], qmul(qp, home_quat[i])) for i in range(n)])
    return P - PINCH * a, quat, psi


class InsertPegScript(_Phases):
    """InsertPeg (task_insert_peg.xml; reward stages env.py:453-462): the right arm grasps the peg (12 x 2 x 2 cm, lying along x)
    `side` metres off its centre towards its own base, the left arm the square tube (`hole`, 12 cm long, 3.6 cm clear inside) as far off
    its centre, both with the gripper pitched `pitch` radians below the horizontal (pointing straight down, the wrist-camera mounts on
    the grippers' backs face each other and meet 18 cm apart) and heading radially from the arm's base (`radial_hand`: the objects turn
    into the pads when the fingers close and turn back as they are carried to the line between the bases); both lift (reward 2), the
    tube is carried to a fixed place above the table on that line, the peg in front of its mouth, and the peg's free end is pushed
    `depth` metres into it (peg touches the tube: 3; peg overlaps the `pin` box that fills the tube's middle 8 cm: 4 = success).
    Closed loop on the measured poses (qpos[23:30] peg, [30:37] hole): while aligning and inserting, the right hand's target integrates
    the error between the peg's free end and the point on the tube's axis it should be at, so that the sag of the off-centre grasps and
    the IK's residual do not matter."""
    T = (50, 40, 25, 45, 70, 40, 60, 20)

    def __init__(self, home, qpos, side=0.02, carry=0.08, depth=0.035, gain=0.15, clip=0.06, pitch=1.0):
        self.n = n = qpos.shape[0]
        self.home = home
        self.pitch = pitch
        self.peg0, self.hole0 = qpos[:, 23:26].copy(), qpos[:, 30:33].copy()
        self.side, self.carry, self.depth, self.gain, self.clip = side, carry, depth, gain, clip
        self.corr = np.zeros((n, 3))
        fk = make_fk("insert_peg")
        self.servo_l, self.servo_r = HandServo(fk, 0, n), HandServo(fk, 1, n)
        self.t = 0

    def action(self, qpos):
        n = self.n
        k, f = self.phase()
        peg, pegq, hole, holeq = qpos[:, 23:26], qpos[:, 26:30], qpos[:, 30:33], qpos[:, 33:37]
        ex = np.tile([1.0, 0.0, 0.0], (n, 1))
        # pinch points: the peg a little above mid height, the tube above its axis (finger tips clear of the table)
        pr = self.peg0 + np.array([self.side, 0.0, 0.004])
        pl = self.hole0 + np.array([-self.side, 0.0, 0.008])
        up = np.array([0.0, 0.0, 1.0])
        meet = np.array([-0.06, BASE_Y])                            # where the tube is held: xy of its centre, on the line between the bases
        if k == 0:
            pr, pl, g = pr + 0.10 * up, pl + 0.10 * up, 0.0
        elif k == 1:
            pr, pl, g = pr + 0.10 * (1 - ramp(f)) * up, pl + 0.10 * (1 - ramp(f)) * up, 0.0
        elif k == 2:
            g = ramp(f, 0.6)
        elif k == 3:
            pr, pl, g = pr + self.carry * ramp(f) * up, pl + self.carry * ramp(f) * up, 1.0
        else:
            g = 1.0
            s = ramp(f) if k == 4 else 1.0
            # the peg's free end `gap` metres in front of the tube's mouth, then `depth` inside
            gap = 0.02 if k <= 5 else (0.02 - (0.02 + self.depth) * ramp(f) if k == 6 else -self.depth)
            dl = np.concatenate([meet - self.hole0[:, :2], np.zeros((n, 1))], axis=1)
            dr = np.concatenate([meet + np.array([0.12 + gap, 0.0]) - self.peg0[:, :2], np.reshape(self.hole0[:, 2] - self.peg0[:, 2], (n, 1))], axis=1)
            pl = pl + self.carry * up + s * dl
            pr = pr + self.carry * up + s * dr
            if k >= 5:
                tip = peg - 0.06 * quat_rot(pegq, ex)
                goal = hole + (0.06 + gap) * quat_rot(holeq, ex)      # where the peg's free end should be on the tube's axis
                self.corr = np.clip(self.corr + self.gain * (goal - tip), -self.clip, self.clip)
            pr = pr + self.corr
        sl, ql, _ = radial_hand(pl, self.pitch, self.home["left"][:, 3:], False)
        sr, qr, _ = radial_hand(pr, self.pitch, self.home["right"][:, 3:], True)
        return self._assemble(self.servo_l(sl, qpos), ql, g, self.servo_r(sr, qpos), qr, g)


class HookPackageScript(_Phases):
    """HookPackage (task_hook_package.xml; reward stages env.py:851-862): both arms take the package from its sides, grippers horizontal
    (the home orientation: the left arm faces +x, the right arm -x, fingers open along y) pinching the 3 cm thick body 1.5 cm in from its
    edges at mid height; both lift (reward 2), carry it in front of the hook's tip with the loop on top of the package (2 x 2 cm clear)
    on the hook's axis, and push it `along` metres up the hook towards the wall (package touches the hook: 3; the pin box in the loop
    overlaps the pin cylinder inside the hook: 4 = success); then both let go and back off, and the package hangs on the hook.
    Closed loop on the measured poses (qpos[23:30] hook, [30:37] package): from the approach on, both hands' targets integrate the
    error between the loop's centre and the point of the hook's axis it should be at."""
    T = (30, 35, 40, 25, 50, 60, 50, 60, 20, 40)
    AXIS = np.array([0.0, -np.sin(1.3), np.cos(1.3)])     # hook cylinder's axis in the hook body's frame (euler="1.3 0 0"), towards the tip
    LOOP = np.array([0.0, 0.0, 0.11])                      # centre of the loop's opening in the package frame

    def __init__(self, home, qpos, inset=0.015, along=0.05, gain=0.15, clip=0.06, lift=0.04):
        self.n = n = qpos.shape[0]
        self.home = home
        self.pkg0 = qpos[:, 30:33].copy()
        self.inset, self.along, self.gain, self.clip, self.lift = inset, along, gain, clip, lift
        self.corr = np.zeros((n, 3))
        fk = make_fk("hook_package")
        self.servo_l, self.servo_r = HandServo(fk, 0, n), HandServo(fk, 1, n)
        self.t = 0

    def action(self, qpos):
        n = self.n
        k, f = self.phase()
        hook, hookq, pkg, pkgq = qpos[:, 23:26], qpos[:, 26:30], qpos[:, 30:33], qpos[:, 33:37]
        ax = quat_rot(hookq, np.tile(self.AXIS, (n, 1)))
        loop = pkg + quat_rot(pkgq, np.tile(self.LOOP, (n, 1)))
        ex = np.array([1.0, 0.0, 0.0])
        body = self.pkg0 + np.array([0.0, -0.01, 0.05])             # centre of the package's body (package-1) at reset
        pl, pr = body - (0.05 - self.inset) * ex, body + (0.05 - self.inset) * ex
        g = 0.0
        k -= 1
        if k < 0:         # at the height of the home pose to the sides of the package (the home pose's pinch points can be right above it)
            pl, pr = pl - 0.05 * ex, pr + 0.05 * ex
            pl[:, 2] = pr[:, 2] = self.home["left"][:, 2]
        elif k == 0:      # down beside the package, clear of its edges
            pl, pr = pl - 0.05 * ex, pr + 0.05 * ex
            pl[:, 2] = pr[:, 2] = self.home["left"][:, 2] + ramp(f) * (pl[:, 2] - self.home["left"][:, 2])
        elif k == 1:      # move in
            pl, pr = pl - 0.05 * (1 - ramp(f)) * ex, pr + 0.05 * (1 - ramp(f)) * ex
        elif k == 2:
            g = ramp(f, 0.6)
        elif k <= 6:
            g = 1.0
            # where the loop's centre should be: s metres from the hook's centre along its axis (tip at 0.1), `lift` above it until it is in front of the tip
            if k == 3:      # straight up to the height of the hook's tip
                want = self.pkg0 + self.LOOP
                want[:, 2] = want[:, 2] + ramp(f) * (hook[:, 2] + 0.13 * ax[:, 2] + self.lift - want[:, 2])
                goal = None
            else:
                s = 0.13 if k == 4 else (0.13 if k == 5 else 0.13 - (0.13 - self.along) * ramp(f))
                goal = hook + s * ax
                if k == 4:
                    goal = goal + np.array([0.0, 0.0, self.lift])
                start = self.pkg0 + self.LOOP
                start[:, 2] = hook[:, 2] + 0.13 * ax[:, 2] + self.lift
                want = start + (ramp(f) if k == 4 else 1.0) * (goal - start)
                if k >= 5 or f > 0.8:
                    self.corr = np.clip(self.corr + self.gain * (goal - loop), -self.clip, self.clip)
            d = want - (self.pkg0 + self.LOOP) + self.corr
            pl, pr = pl + d, pr + d
            self.last = (pl.copy(), pr.copy())
        else:             # let go (7), back off sideways (8)
            pl, pr = self.last
            if k == 8:
                pl, pr = pl - 0.06 * ramp(f) * ex, pr + 0.06 * ramp(f) * ex
        site = PINCH * ex
        return self._assemble(self.servo_l(pl - site, qpos), self.home["left"][:, 3:], g, self.servo_r(pr + site, qpos), self.home["right"][:, 3:], g)


class SewNeedleThreadScript(_Phases):
    """SewNeedle through all its reward stages (task_sew_needle.xml; env.py:676-689).  The right arm grasps the needle (10 x 2 x 2 cm, along
    x) top-down `side` metres off its centre towards its own base and lifts it (1, 2), holds it in front of the wall's window (3 x 3 cm
    clear, 2 cm deep, centre 5 cm above the wall body's origin) and pushes it through until the needle's centre is `past` metres beyond
    the wall's mid-plane (needle touches the wall: 3; pin-needle meets pin-wall: the threading latch env.py:673, reward 4 from then on);
    it lets go -- the needle rests in the window, its centre of mass over the sill -- and backs off upwards; only then does the left
    arm come down on the end that sticks out on the other side (top-down both: with the two hands at the wall together the wrist-camera
    mounts on the grippers' backs would meet), closes and pulls the needle `pull` metres out of the window: held by the left gripper
    alone, clear of the table and of pin-wall, latch set = 5 = success (env.py:686-689).  Closed loop on the measured poses
    (qpos[23:30] wall, [30:37] needle): in front of the window and while pushing, the right hand's target integrates the error of the
    needle's leading end against the window's axis; the left hand is placed on the measured needle."""
    T = (50, 40, 25, 40, 60, 30, 60, 15, 70, 40, 25, 60, 20)

    def __init__(self, home, qpos, side=0.035, past=0.005, pull=0.11, gain=0.15, clip=0.05):
        self.n = n = qpos.shape[0]
        self.home = home
        self._down(home)
        self.needle0 = qpos[:, 30:33].copy()
        self.side, self.past, self.pull, self.gain, self.clip = side, past, pull, gain, clip
        self.corr = np.zeros((n, 3))
        self.lgrasp = None
        fk = make_fk("sew_needle")
        self.servo_l, self.servo_r = HandServo(fk, 0, n), HandServo(fk, 1, n)
        self.t = 0

    def action(self, qpos):
        n = self.n
        k, f = self.phase()
        wall, wallq, ndl, ndlq = qpos[:, 23:26], qpos[:, 26:30], qpos[:, 30:33], qpos[:, 33:37]
        ex, up = np.array([1.0, 0.0, 0.0]), np.array([0.0, 0.0, 1.0])
        exn = np.tile(ex, (n, 1))
        window = wall + quat_rot(wallq, np.tile([0.0, 0.0, 0.05], (n, 1)))          # centre of the window
        wax = quat_rot(wallq, exn)                                                     # its axis (the wall's x)
        c0 = self.needle0 + np.array([0.0, 0.0, 0.01])                                 # needle's centre at reset (the body's origin is its underside)
        site = np.array([0.0, 0.0, PINCH + 0.005])                                     # top-down: the wrist above the pinch point, the pads a little above the needle's axis
        pr = c0 + self.side * ex
        gr, gl = 0.0, 0.0
        # the left hand waits further out than its home pose (there its fingers are 6 cm from the centre line, where the right
        # gripper's camera mount arrives when the needle goes through the wall)
        park_l = self.home["left"][:, :3] + np.array([-0.10, 0.0, 0.04])
        pl_site = park_l.copy()
        lquat = self.home["left"][:, 3:]
        if k == 0:
            pr = pr + 0.10 * up
        elif k == 1:
            pr = pr + 0.10 * (1 - ramp(f)) * up
        elif k == 2:
            gr = ramp(f, 0.6)
        elif k == 3:
            gr = 1.0
            pr = pr + ramp(f) * (window[:, 2:3] - c0[:, 2:3]) * up
        elif k <= 7:
            gr = 1.0 if k < 7 else max(0.0, 1.0 - f / 0.6)
            # the needle's centre on the window's axis: its leading end `gap` in front of the wall's face, then through until the centre is `past` beyond the mid-plane
            gap = 0.02
            x_front = 0.01 + gap + 0.05
            s = x_front if k <= 5 else (x_front - (x_front + self.past) * ramp(f) if k == 6 else -self.past)
            goal_c = window + s * wax
            start = c0.copy(); start[:, 2] = window[:, 2]
            want = start + (ramp(f) if k == 4 else 1.0) * (goal_c - start)
            if k in (5, 6):
                lead = ndl + quat_rot(ndlq, np.tile([-0.05, 0.0, 0.01], (n, 1)))      # the needle's leading end (its -x face's centre)
                self.corr = np.clip(self.corr + self.gain * ((goal_c - 0.05 * wax) - lead), -self.clip, self.clip)
            pr = want + self.side * ex + self.corr
            self.rlast = pr.copy()
        else:
            # the right hand straight up and back to its side; the left hand above the end that sticks out, down, close, pull
            r8 = ramp(f, 0.5) if k == 8 else 1.0
            back = np.maximum(0.0, 0.30 - (self.rlast[:, :1] + 0.0))                      # ... until its wrist is 30 cm from the centre line
            pr = self.rlast + 0.07 * min(1.0, 2 * r8) * up + r8 * back * ex
            if self.lgrasp is None or k == 8:
                cn = ndl + quat_rot(ndlq, np.tile([0.0, 0.0, 0.01], (n, 1)))
                self.lgrasp = cn - self.side * quat_rot(ndlq, exn)
            pl = self.lgrasp.copy()
            lquat = self.down_l
            if k == 8:
                h = park_l - site
                fl = max(0.0, (f - 0.4) / 0.6)                                             # once the right hand is out of the way
                pl = h + ramp(fl) * (pl + 0.10 * up - h)
                lquat = self.down_l if fl > 0.2 else self.home["left"][:, 3:]
            elif k == 9:
                pl = pl + 0.10 * (1 - ramp(f)) * up
            elif k == 10:
                gl = ramp(f, 0.6)
            else:           # pull it out along the window's axis, then a little up: clear of the sill, of pin-wall and of the base plate
                gl = 1.0
                pl = pl - self.pull * (ramp(f, 0.7) if k == 11 else 1.0) * ex + 0.02 * (max(0.0, (f - 0.7) / 0.3) if k == 11 else 1.0) * up
            pl_site = pl + site
        return self._assemble(self.servo_l(pl_site, qpos), lquat, gl, self.servo_r(pr + site, qpos), self.down_r, gr)


def rot_x(th, v):
    c, s = np.cos(th), np.sin(th)
    return np.array([v[0], c * v[1] - s * v[2], s * v[1] + c * v[2]])


class TubeTransferScript(_Phases):
    """TubeTransfer (task_tube_transfer.xml; reward stages env.py:771-778): the right arm takes tube1 (square tube, 10 cm tall, 2.3 cm
    clear, the 1 cm ball inside, friction 1e-5) from its side, the left arm tube2, grippers horizontal (home orientation) pinching the
    tubes at mid height; both lift (reward 2); the hands roll about their own axes (the world's x: wrist_rotate) until the tubes lie
    along y mouth to mouth on one line that slopes `slope` radians down from tube1 to tube2 -- tube1 first stops `slope` short of the
    horizontal (mouth up, the ball stays at its bottom) and is tipped over, about its mouth, only when the mouths are `gap` metres
    apart: the ball rolls along the aligned inner walls into tube2 and meets the `pin` box inside it (3 = success).  Closed loop on the
    measured tube poses (qpos[30:37] tube1, [37:44] tube2): the right hand's target integrates the error of tube1's mouth against the
    point in front of tube2's mouth."""
    T = (40, 40, 40, 25, 40, 90, 50, 60, 130)

    def __init__(self, home, qpos, slope=0.5, gap=0.006, height=0.20, gain=0.08, clip=0.05):
        self.n = n = qpos.shape[0]
        self.home = home
        self.t1, self.t2 = qpos[:, 30:33].copy(), qpos[:, 37:40].copy()
        self.slope, self.gap, self.height, self.gain, self.clip = slope, gap, height, gain, clip
        self.corr = np.zeros((n, 3))
        fk = make_fk("tube_transfer")
        self.servo_l, self.servo_r = HandServo(fk, 0, n), HandServo(fk, 1, n)
        self.t = 0

    def action(self, qpos):
        n = self.n
        k, f = self.phase()
        tube1, q1, tube2, q2 = qpos[:, 30:33], qpos[:, 33:37], qpos[:, 37:40], qpos[:, 40:44]
        ex, up = np.array([1.0, 0.0, 0.0]), np.array([0.0, 0.0, 1.0])
        mid = np.array([0.0, 0.0, 0.05])
        pr, pl = self.t1 + mid, self.t2 + mid                       # pinch points: the tubes' centres
        th1 = th2 = 0.0
        g = 0.0
        hz = self.home["left"][:, 2]
        if k == 0:
            pr, pl = pr + 0.07 * ex, pl - 0.07 * ex
            pr[:, 2] = pl[:, 2] = hz
        elif k == 1:
            pr, pl = pr + 0.07 * ex, pl - 0.07 * ex
            pr[:, 2] = pl[:, 2] = hz + ramp(f) * (0.05 - hz)
        elif k == 2:
            pr, pl = pr + 0.07 * (1 - ramp(f)) * ex, pl - 0.07 * (1 - ramp(f)) * ex
        elif k == 3:
            g = ramp(f, 0.6)
        elif k == 4:
            g = 1.0
            pr, pl = pr + ramp(f) * (self.height - 0.05) * up, pl + ramp(f) * (self.height - 0.05) * up
        else:
            g = 1.0
            s = ramp(f) if k == 5 else 1.0
            half = np.pi / 2
            th2 = -s * (half - self.slope)                          # tube2: mouth towards +y, `slope` above the horizontal
            tip = ramp(f) if k == 7 else (1.0 if k == 8 else 0.0)
            th1 = s * (half - self.slope) + tip * 2 * self.slope    # tube1: mouth towards -y, first above, then `slope` below the horizontal
            a1, a2 = rot_x(th1, up), rot_x(th2, up)                 # the tubes' axes, bottom -> mouth
            gp = 0.03 if k == 5 else (0.03 + (self.gap - 0.03) * ramp(f) if k == 6 else self.gap)
            J = np.array([0.0, BASE_Y, self.height])
            u = rot_x(-(half - self.slope), up)                     # final direction of the common line, from tube2's bottom up to tube1's bottom: tube2's axis reversed ... (0, cos(slope), sin(slope))
            u = np.array([0.0, np.cos(self.slope), np.sin(self.slope)])
            m2 = J - 0.5 * gp * u                                   # tube2's mouth, tube1's mouth
            m1 = J + 0.5 * gp * u
            c2, c1 = m2 - 0.05 * a2, m1 - 0.05 * a1                 # their centres = the pinch points
            start_r, start_l = self.t1 + mid + (self.height - 0.05) * up, self.t2 + mid + (self.height - 0.05) * up
            pr, pl = start_r + s * (c1 - start_r), start_l + s * (c2 - start_l)
            if k >= 6:
                mouth1 = tube1 + quat_rot(q1, np.tile([0.0, 0.0, 0.10], (n, 1)))
                mouth2 = tube2 + quat_rot(q2, np.tile([0.0, 0.0, 0.10], (n, 1)))
                ax2 = quat_rot(q2, np.tile(up, (n, 1)))
                goal = mouth2 + gp * ax2                              # tube1's mouth in front of tube2's, on tube2's axis
                self.corr = np.clip(self.corr + self.gain * (goal - mouth1), -self.clip, self.clip)
            pr = pr + self.corr
        qr = np.stack([qmul(np.array([np.cos(th1 / 2), np.sin(th1 / 2), 0.0, 0.0]), self.home["right"][i, 3:]) for i in range(n)])
        ql = np.stack([qmul(np.array([np.cos(th2 / 2), np.sin(th2 / 2), 0.0, 0.0]), self.home["left"][i, 3:]) for i in range(n)])
        site = PINCH * ex
        return self._assemble(self.servo_l(pl - site, qpos), ql, g, self.servo_r(pr + site, qpos), qr, g)


# script name -> (class, task model, episode seed base of the parity tests)
SCRIPTS = {"slot_insertion": (SlotInsertionScript, "slot_insertion"), "insert_peg": (InsertPegScript, "insert_peg"),
           "sew_needle_thread": (SewNeedleThreadScript, "sew_needle"), "hook_package": (HookPackageScript, "hook_package"),
           "tube_transfer": (TubeTransferScript, "tube_transfer")}
SCRIPT_OF_TASK = {"slot_insertion": "slot_insertion", "insert_peg": "insert_peg", "sew_needle": "sew_needle_thread", "hook_package": "hook_package",
                  "tube_transfer": "tube_transfer"}


def make_script(name, home, qpos0, **kw):
    """home: {'left','right','middle'} -> [n, 7] (or [7]) eef poses at reset (the env's obs['poses']); qpos0 [n, nq]."""
    n = qpos0.shape[0]
    home = {k: np.broadcast_to(np.asarray(v, dtype=np.float64), (n, 7)).copy() for k, v in home.items()}
    return SCRIPTS[name][0](home, qpos0, **kw)
