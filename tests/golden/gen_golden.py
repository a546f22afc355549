"""Generate golden vectors by importing the reference's own Python (IK, kinematics,
transform helpers, reward predicates) under import shims.  Runs ONLY in the build
container (needs /root/reference); the .npz files it writes are committed and are
the only artefacts of the reference that travel.

Shims: `numba` -> identity decorators, `mujoco` -> no-op
`mj_kinematics`, `dm_control`/`gymnasium` -> empty stand-ins, a fake `physics`
whose `bind()` returns the zero-pose screw data of OUR compiled model.

    python tests/golden/gen_golden.py
"""
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def install_shims():
    d = tempfile.mkdtemp(prefix="avshim_")
    os.makedirs(f"{d}/numba")
    open(f"{d}/numba/__init__.py", "w").write(
        "class _D:\n"
        "    def __getitem__(self, k): return self\n"
        "    def __call__(self, *a, **k): return self\n"
        "float64 = boolean = _D()\n"
        "prange = range\n"
        "def jit(*a, **k):\n"
        "    if len(a) == 1 and callable(a[0]) and not isinstance(a[0], _D) and not k:\n"
        "        return a[0]\n"
        "    return lambda f: f\n")
    open(f"{d}/numba/types.py", "w").write("from . import _D\ndef UniTuple(*a): return _D()\n")
    os.makedirs(f"{d}/mujoco")
    open(f"{d}/mujoco/__init__.py", "w").write("def mj_kinematics(m, d): pass\n")
    open(f"{d}/mujoco/viewer.py", "w").write("")
    os.makedirs(f"{d}/dm_control")
    open(f"{d}/dm_control/__init__.py", "w").write("")
    open(f"{d}/dm_control/mjcf.py", "w").write("")
    os.makedirs(f"{d}/gymnasium")
    open(f"{d}/gymnasium/__init__.py", "w").write(
        "from . import spaces\nclass Env:\n    def reset(self, seed=None, options=None): pass\n")
    open(f"{d}/gymnasium/spaces.py", "w").write(
        "class Box:\n    def __init__(self,*a,**k): pass\nclass Dict:\n    def __init__(self,*a,**k): pass\n")
    sys.path.insert(0, d)
    sys.path.insert(0, f"{REF}/data_collection_scripts")


class _Bound:
    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.qpos = None


class FakePhysics:
    """bind(joints) -> xaxis/xanchor/range; bind(site) -> xmat/xpos (kinematics.py:8-15)."""

    class _P:
        ptr = None

    def __init__(self, w0, p0, rng, site0):
        self.model = self.data = self._P()
        self.w0, self.p0, self.rng, self.site0 = w0, p0, rng, site0

    def bind(self, what):
        if what == "site":
            return _Bound(xmat=self.site0[:3, :3].reshape(-1).copy(), xpos=self.site0[:3, 3].copy())
        return _Bound(xaxis=self.w0.copy(), xanchor=self.p0.copy(), range=self.rng.copy())


def main():
    install_shims()
    import transform_utils as T
    import kinematics as K
    from diff_ik import DiffIK
    from grad_ik import GradIK
    from av_aloha_amd.compiler.compile import read_blob
    from av_aloha_amd.constants import LEFT_ARM_POSE, RIGHT_ARM_POSE, MIDDLE_ARM_POSE

    md = read_blob(os.path.join(ROOT, "models", "slot_insertion_3arms.avm"))
    rng = np.random.default_rng(20241022)
    out = {}

    # ---- SO(3)/SE(3) helpers (transform_utils.py) ----
    n = 256
    quats = rng.normal(size=(n, 4))
    quats /= np.linalg.norm(quats, axis=1, keepdims=True)          # xyzw
    out["q_xyzw"] = quats
    out["quat2mat"] = np.stack([np.asarray(T.quat2mat(q.copy()), dtype=np.float64) for q in quats])
    mats = out["quat2mat"]
    out["mat2quat"] = np.stack([T.mat2quat(np.ascontiguousarray(m)) for m in mats])
    out["quat2axisangle"] = np.stack([T.quat2axisangle(q.copy()) for q in out["mat2quat"]])
    vecs = rng.normal(size=(n, 3))
    out["aa_in"] = vecs
    out["axisangle2quat"] = np.stack([T.axisangle2quat(v) for v in vecs])
    out["angular_error"] = np.stack([T.angular_error(mats[i], mats[(i + 1) % n]) for i in range(n)])
    w = rng.normal(size=(n, 3))
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    v = rng.normal(size=(n, 3))
    th = rng.uniform(-3, 3, size=n)
    out["exp_w"], out["exp_v"], out["exp_th"] = w, v, th
    out["exp2mat"] = np.stack([T.exp2mat(w[i], v[i], th[i]) for i in range(n)])
    Ts = out["exp2mat"]
    out["adjoint"] = np.stack([T.adjoint(t) for t in Ts])
    lp_pos, lp_mat = [], []
    for i in range(n):
        a, b = Ts[i], Ts[(i + 7) % n]
        p, m_ = T.limit_pose(a[:3, 3] * 0.05, np.ascontiguousarray(a[:3, :3]), b[:3, 3] * 0.05,
                             np.ascontiguousarray(b[:3, :3]), 0.1, 0.3)
        lp_pos.append(p)
        lp_mat.append(np.asarray(m_, dtype=np.float64))
    out["limit_pose_pos"], out["limit_pose_mat"] = np.stack(lp_pos), np.stack(lp_mat)
    np.savez_compressed(os.path.join(HERE, "so3_helpers.npz"), **out)

    # ---- FK / Jacobian / IK per arm ----
    homes = [np.array(LEFT_ARM_POSE[:6]), np.array(RIGHT_ARM_POSE[:6]), np.array(MIDDLE_ARM_POSE)]
    names = ["left", "right", "middle"]
    for a in range(3):
        nj = int(md["ik_n"][a])
        w0, p0 = md["ik_w0"][a, :nj], md["ik_p0"][a, :nj]
        jr = md["ik_range"][a, :nj]
        site0 = md["ik_site0"][a]
        ph = FakePhysics(w0, p0, jr, site0)
        joints = list(range(nj))
        fk = K.create_fk_fn(ph, joints, "site")
        jac = K.create_jac_fn(ph, joints)
        N = 512
        q = rng.uniform(jr[:, 0], jr[:, 1], size=(N, nj))
        q[0] = homes[a]
        o = {"q": q, "w0": w0, "p0": p0, "site0": site0, "range": jr}
        o["fk"] = np.stack([fk(x) for x in q])
        o["jac"] = np.stack([jac(x) for x in q])
        np.savez_compressed(os.path.join(HERE, f"fk_jac_{names[a]}.npz"), **o)

        # reachable targets: FK of a perturbed configuration; start: perturbed again
        M = 256
        qs = np.clip(homes[a] + rng.normal(scale=0.4, size=(M, nj)), jr[:, 0], jr[:, 1])
        qt = np.clip(qs + rng.normal(scale=0.15, size=(M, nj)), jr[:, 0], jr[:, 1])
        qs[0] = homes[a]
        Tt = np.stack([fk(x) for x in qt])
        tpos = Tt[:, :3, 3].copy()
        tquat = np.stack([T.xyzw_to_wxyz(T.mat2quat(np.ascontiguousarray(t[:3, :3]))) for t in Tt])
        # a few far / unreachable targets as well
        tpos[-16:] += rng.normal(scale=0.3, size=(16, 3))
        if a == 2:
            k_null = np.array([10.0, 10.0, 10.0, 10.0, 5.0, 5.0, 5.0])
        else:
            k_null = np.array([10.0, 10.0, 10.0, 10.0, 5.0, 5.0])
        ctl = DiffIK(physics=ph, joints=joints, actuators=None, eef_site="site", k_pos=0.9, k_ori=0.9,
                     damping=1.0e-4, k_null=k_null, q0=homes[a].copy(), max_angvel=3.14,
                     integration_dt=0.04, iterations=10)
        # the float32-rounded target matrix the reference computes internally (transform_utils.py:66)
        tmat = np.stack([np.asarray(T.quat2mat(T.wxyz_to_xyzw(x)), dtype=np.float64) for x in tquat])
        o = {"q": qs, "target_pos": tpos, "target_quat_wxyz": tquat, "target_mat": tmat, "k_null": k_null, "q0": homes[a]}
        o["q_out"] = np.stack([ctl.run(qs[i].copy(), tpos[i].copy(), tquat[i].copy()) for i in range(M)])
        np.savez_compressed(os.path.join(HERE, f"diffik_{names[a]}.npz"), **o)

        if a < 2:
            def mk(max_it):
                return GradIK(physics=ph, joints=joints, actuators=None, eef_site="site", step_size=0.0001,
                              min_cost_delta=1.0e-12, max_iterations=max_it, position_weight=500.0,
                              rotation_weight=100.0, joint_center_weight=np.array([10.0, 10.0, 1.0, 50.0, 1.0, 1.0]),
                              joint_displacement_weight=np.array(6 * [50.0]), position_threshold=0.001,
                              rotation_threshold=0.001, max_pos_diff=0.1, max_rot_diff=0.3, joint_p=0.9)
            o = {"q": qs, "target_pos": tpos, "target_quat_wxyz": tquat, "target_mat": tmat}
            # the secant descent amplifies rounding noise ~x3-10 per iteration (it is chaotic by iteration
            # ~30), so the algorithm is pinned on truncated runs and the full 50-iteration run is kept
            # for a statistical check only
            for max_it in (1, 4, 8, 50):
                g = mk(max_it)
                o[f"q_out_it{max_it}"] = np.stack(
                    [g.run(qs[i].copy(), tpos[i].copy(), tquat[i].copy()) for i in range(M)])
            o["q_out"] = o["q_out_it50"]
            np.savez_compressed(os.path.join(HERE, f"gradik_{names[a]}.npz"), **o)

    # ---- reward truth tables (env.py get_reward x5) ----
    pkg = types.ModuleType("gym_guided_vision")
    pkg.__path__ = [f"{REF}/gym_guided_vision/gym_guided_vision"]
    sys.modules["gym_guided_vision"] = pkg
    import gym_guided_vision.env as E
    import json

    class FakeP:
        def __init__(self, names, pairs):
            self.model = types.SimpleNamespace(id2name=lambda i, kind: names[i])
            self.data = types.SimpleNamespace(
                ncon=len(pairs), contact=[types.SimpleNamespace(geom1=a, geom2=b) for a, b in pairs])

    tasks = {"insert_peg": E.InsertPegEnv, "slot_insertion": E.SlotInsertionEnv, "sew_needle": E.SewNeedleEnv,
             "tube_transfer": E.TubeTransferEnv, "hook_package": E.HookPackageEnv}
    rw = {}
    for t, cls in tasks.items():
        man = json.load(open(os.path.join(ROOT, "models", f"{t}_3arms.json")))
        names = man["geom_names"]
        ng = len(names)
        named = [i for i, nm in enumerate(names) if nm != ""]
        fingers = [i for i in named if names[i].startswith(("left", "right"))]
        objs = [i for i in named if i not in fingers and names[i] != "table"]
        table = names.index("table")
        env = object.__new__(cls)
        seqs, rewards = [], []
        S, L, C = 96, 12, 10     # sequences x steps x max contacts; Sew latches across a sequence
        for s in range(S):
            env._threaded_needle = False
            p_table = rng.uniform(0, 0.3)
            for l in range(L):
                nc = int(rng.integers(0, C + 1))
                pairs = []
                for _ in range(nc):
                    u = rng.random()
                    if u < 0.45:
                        a_, b_ = rng.choice(objs), rng.choice(fingers)
                    elif u < 0.45 + p_table:
                        a_, b_ = table, rng.choice(objs)
                    elif u < 0.9:
                        a_, b_ = rng.choice(objs, size=2, replace=True)
                    else:
                        a_, b_ = rng.integers(0, ng, size=2)
                    if rng.random() < 0.5:
                        a_, b_ = b_, a_
                    pairs.append((int(a_), int(b_)))
                env._physics = FakeP(names, pairs)
                r = env.get_reward()
                row = -np.ones((C, 2), dtype=np.int32)
                for k, pr in enumerate(pairs):
                    row[k] = pr
                seqs.append(row)
                rewards.append(r)
        rw[f"{t}_pairs"] = np.array(seqs).reshape(S, L, C, 2)
        rw[f"{t}_reward"] = np.array(rewards, dtype=np.int32).reshape(S, L)
    np.savez_compressed(os.path.join(HERE, "reward_tables.npz"), **rw)

    # ---- reset sampling (task `reset` overrides, env.py:474-501, 513-543, 604-637, 705-735, 792-818) ----
    # run the reference's own reset() against a recording stand-in for dm_control's Physics: every
    # `physics.bind(elem).qpos = value` is captured, so the draw order (incl. the discarded draws) is pinned
    class Rec:
        def __init__(self, log, elem):
            object.__setattr__(self, "_log", log)
            object.__setattr__(self, "_elem", elem)

        def __setattr__(self, k, v):
            self._log.append((self._elem, k, np.array(v, dtype=np.float64).copy()))

    class RecPhysics:
        def __init__(self):
            self.log = []

        def bind(self, elem):
            return Rec(self.log, elem if isinstance(elem, str) else tuple(elem))

        def reset(self):
            pass

        def forward(self):
            pass

    rs = {}
    joints = {"insert_peg": ("_peg_joint", "_hole_joint"), "slot_insertion": ("_slot_joint", "_stick_joint"),
              "sew_needle": ("_needle_joint", "_wall_joint"), "tube_transfer": ("_ball_joint", "_tube1_joint", "_tube2_joint"),
              "hook_package": ("_hook_joint", "_package_joint")}
    for t, cls in tasks.items():
        env = object.__new__(cls)
        env._physics = RecPhysics()
        env.get_obs = lambda: None
        for nm in ("_left_joints", "_right_joints", "_middle_joints", "_left_actuators", "_right_actuators", "_middle_actuators",
                   "_left_gripper_joints", "_right_gripper_joints"):
            setattr(env, nm, [nm + str(i) for i in range(7)])
        env.left_gripper_unnorm_fn = env.right_gripper_unnorm_fn = lambda x: x * 0.035 + 0.002
        for jn in joints[t]:
            setattr(env, jn, jn)
        out = []
        for seed in range(16):
            env._physics.log.clear()
            np.random.seed(seed)
            env.reset(seed=seed)
            got = {e: v for e, k, v in env._physics.log if k == "qpos" and isinstance(e, str) and e in joints[t]}
            out.append(np.stack([got[jn] for jn in joints[t]]))
        rs[t] = np.stack(out)                      # [seed][object in the order of `joints[t]`][7]
        rs[t + "_order"] = np.array(joints[t])
    np.savez_compressed(os.path.join(HERE, "reset_samples.npz"), **rs)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
