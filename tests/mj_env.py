"""Real MuJoCo stepping the build's OWN model (test / bench infrastructure; needs `import mujoco`, which neither the build image nor the
GPU box has: profiles/r06_mujoco_probe.txt).

`MjEnv(task, arms)` loads the MJCF that av_aloha_amd/compiler/emit_mjcf.py writes from models/<task>_<n>arms.{avm,json} -- no reference file
is read -- and gives it the reference env's semantics (gym_guided_vision/gym_guided_vision/env.py): reset = home pose and home ctrl with
the object free joints at the given poses, `mj_forward` (:228-249 + the task resets); step = action -> ctrl with the gripper un-normalised
(:203-215), `mj_step` x 20 (:218), agent_pos (:168-178), reward from the contact list's geom-name pairs (:425-863, evaluated by the pinned
oracle/orc_reward.c through orc_reward_from_pairs), is_success = reward == max_reward (:224)."""
import ctypes as C
import json
import os

import numpy as np

from orc_ffi import ROOT, ip, lib, load_model


def mujoco_importable():
    try:
        import mujoco  # noqa: F401
        return True
    except Exception:
        return False


class MjEnv:
    def __init__(self, task="slot_insertion", num_arms=3, variant="gym", hulls="device"):
        import mujoco
        import sys
        sys.path.insert(0, ROOT)
        from av_aloha_amd.compiler import emit_mjcf
        from av_aloha_amd.compiler.compile import read_blob
        self.mj = mujoco
        prefix = "dc_" if variant == "data_collection" else ""
        base = os.path.join(ROOT, "models", f"{prefix}{task}_{num_arms}arms")
        self.blob = read_blob(base + ".avm")
        self.man = json.load(open(base + ".json"))
        self.xml = emit_mjcf.emit_files(os.path.join(ROOT, "models"), task, num_arms, prefix=prefix, hulls=hulls)
        self.model = mujoco.MjModel.from_xml_string(self.xml)
        self.data = mujoco.MjData(self.model)
        b = self.blob
        assert (self.model.nq, self.model.nv, self.model.nu, self.model.nbody) == (int(b["nq"][0]), int(b["nv"][0]), int(b["nu"][0]), int(b["nbody"][0]))
        self.nj = 21 if num_arms == 3 else 14
        # MuJoCo numbers geoms body by body, in the order they appear inside a body [EXT]; the blob numbers them in document order and most of them
        # carry no name: MuJoCo's geom g is the g-th blob geom in a stable sort by body (emit_mjcf.py keeps a body's geoms in the blob's order)
        self.geom_of = np.argsort(np.asarray(b["geom_body"]), kind="stable").astype(np.int32)
        assert self.model.ngeom == len(self.geom_of)
        names = self.man["geom_names"]
        for g in range(self.model.ngeom):
            nm = mujoco.mj_id2name(self.model, mujoco.mjtObj.mjOBJ_GEOM, g)
            assert (nm or "") == names[self.geom_of[g]], (g, nm, names[self.geom_of[g]])
        self.orc_model = load_model(task, num_arms, variant)
        self.latch = C.c_int(0)
        self.max_reward = lib().orc_max_reward(self.orc_model)

    def reset(self, obj_qpos):
        mj, b = self.mj, self.blob
        mj.mj_resetData(self.model, self.data)
        self.data.qpos[:] = b["qpos_home"]
        for adr, p in zip(b["objects_qposadr"], np.asarray(obj_qpos, dtype=np.float64).reshape(-1, 7)):
            self.data.qpos[adr:adr + 7] = p
        self.data.ctrl[:] = b["ctrl_home"]
        self.latch = C.c_int(0)
        mj.mj_forward(self.model, self.data)

    def agent_pos(self):
        b = self.blob
        q = self.data.qpos[b["obs_qposadr"][:self.nj]]
        return (q - b["obs_offset"][:self.nj]) * b["obs_scale"][:self.nj]

    def contact_pairs(self):
        d = self.data
        out = np.zeros((max(int(d.ncon), 1), 2), dtype=np.int32)
        for i in range(int(d.ncon)):
            c = d.contact[i]
            g = c.geom if hasattr(c, "geom") else (c.geom1, c.geom2)
            out[i] = self.geom_of[int(g[0])], self.geom_of[int(g[1])]
        return out

    def reward(self):
        pairs = np.ascontiguousarray(self.contact_pairs())
        return int(lib().orc_reward_from_pairs(self.orc_model, ip(pairs), int(self.data.ncon), C.byref(self.latch)))

    def step(self, action, nsub=20):
        b = self.blob
        a = np.asarray(action, dtype=np.float64)
        lo, hi = b["grip_range"]
        ctrl = np.array(self.data.ctrl)
        ctrl[:self.nj] = a[:self.nj]
        for k in (6, 13):                                   # env.py:156-161, 209-212: unnorm(a) = lo + a (hi - lo)
            ctrl[k] = lo + a[k] * (hi - lo)
        self.data.ctrl[:] = ctrl
        self.mj.mj_step(self.model, self.data, nstep=nsub)
        # dm_control's Physics.step (legacy mode, the reference's: env.py:218) ends with mj_step1, so that the contact list the reward
        # reads belongs to the state AFTER the last substep [EXT]; the oracle's orc_step refreshes kinematics + collision the same way
        self.mj.mj_step1(self.model, self.data)
        r = self.reward()
        return self.agent_pos(), r, r == self.max_reward

    @property
    def qpos(self):
        return np.array(self.data.qpos)

    @property
    def ncon(self):
        return int(self.data.ncon)
