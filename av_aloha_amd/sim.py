"""BatchedSim: numpy-facing wrapper of a libavsim handle (host-pointer I/O mode).

This is the batched engine under the gym-style environments in env.py; every method is a direct call
into the C-ABI (include/avsim.h).  There is no CPU fallback."""
import ctypes as C
import json
import os

import numpy as np

from . import _ffi
from .constants import MODEL_DIR, SIM_PHYSICS_ENV_STEP_RATIO

TASK_KEYS = ("insert_peg", "slot_insertion", "sew_needle", "tube_transfer", "hook_package")


VARIANT_PREFIX = {"gym": "", "data_collection": "dc_"}


def load_blob(task, num_arms, variant="gym"):
    """variant "gym": compiled from gym_guided_vision/gym_guided_vision/assets (the gym envs); "data_collection": from
    data_collection_scripts/assets, which sim_env.py loads (data_collection_scripts/constants.py:5): default solref on the needle
    and the peg, ZED fovy 90 (compiler --variant)."""
    base = os.path.join(MODEL_DIR, f"{VARIANT_PREFIX[variant]}{task}_{num_arms}arms")
    with open(base + ".avm", "rb") as f:
        blob = f.read()
    with open(base + ".json") as f:
        manifest = json.load(f)
    return blob, manifest


class BatchedSim:
    def __init__(self, task, num_arms=3, num_envs=1, device=0, f64=False, options=None, variant="gym", blob=None):
        assert task in TASK_KEYS, task
        file_blob, self.manifest = load_blob(task, num_arms, variant)
        blob = file_blob if blob is None else blob        # (tests: the task's model with an edited constant, e.g. gravity)
        self.h = _ffi.Handle(blob, num_envs, device, _ffi.AVSIM_F64_PHYSICS if f64 else 0)
        self.N = num_envs
        for k in ("nq", "nv", "nu", "nj", "nobj", "max_reward", "maxcon", "maxefc"):
            setattr(self, k, getattr(self.h, k))
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def set_option(self, name, value):
        self.h.check(self.h.L.avsim_set_option(self.h.h, name.encode(), float(value)))
        if name in ("maxcon", "maxefc"):       # the capacities size the contact export
            d = np.zeros(_ffi.NDIMS, dtype=np.int32)
            self.h.check(self.h.L.avsim_dims(self.h.h, d.ctypes.data))
            self.maxcon, self.maxefc = int(d[8]), int(d[9])

    def reset(self, obj_qpos, mask=None):
        obj = np.ascontiguousarray(obj_qpos, dtype=np.float64).reshape(self.N, self.nobj * 7)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.h.check(self.h.L.avsim_reset(self.h.h, _ffi.ptr(m), obj.ctypes.data))

    def step(self, action, nsub=SIM_PHYSICS_ENV_STEP_RATIO, want_reward=True):
        a = np.ascontiguousarray(action, dtype=np.float32).reshape(self.N, self.nj)
        ap = np.empty((self.N, self.nj))
        rw = np.empty(self.N, dtype=np.int32) if want_reward else None
        su = np.empty(self.N, dtype=np.uint8) if want_reward else None
        self.h.check(self.h.L.avsim_step(self.h.h, a.ctypes.data, nsub, ap.ctypes.data, _ffi.ptr(rw), _ffi.ptr(su)))
        return ap, rw, (su.astype(bool) if su is not None else None)

    def step_ctrl(self, nsub=SIM_PHYSICS_ENV_STEP_RATIO):
        """nsub substeps driven by the ctrl vector as it stands (set_state / an earlier step): dm_control's physics.step(nsub)."""
        ap = np.empty((self.N, self.nj))
        rw = np.empty(self.N, dtype=np.int32)
        su = np.empty(self.N, dtype=np.uint8)
        self.h.check(self.h.L.avsim_step_ctrl(self.h.h, nsub, ap.ctypes.data, rw.ctypes.data, su.ctypes.data))
        return ap, rw, su.astype(bool)

    def step_cartesian(self, action23, ik_mode=_ffi.IK_REFERENCE, nsub=SIM_PHYSICS_ENV_STEP_RATIO):
        a = np.ascontiguousarray(action23, dtype=np.float64).reshape(self.N, 23)
        ap = np.empty((self.N, 21))
        rw = np.empty(self.N, dtype=np.int32)
        su = np.empty(self.N, dtype=np.uint8)
        self.h.check(self.h.L.avsim_step_cartesian(self.h.h, a.ctypes.data, ik_mode, nsub, ap.ctypes.data, rw.ctypes.data, su.ctypes.data))
        return ap, rw, su.astype(bool)

    def get_state(self):
        qpos, qvel = np.empty((self.N, self.nq)), np.empty((self.N, self.nv))
        ctrl, warm = np.empty((self.N, self.nu)), np.empty((self.N, self.nv))
        self.h.check(self.h.L.avsim_get_state(self.h.h, qpos.ctypes.data, qvel.ctypes.data, ctrl.ctypes.data, warm.ctypes.data))
        return qpos, qvel, ctrl, warm

    def set_state(self, qpos=None, qvel=None, ctrl=None, warm=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (qpos, qvel, ctrl, warm)]
        self.h.check(self.h.L.avsim_set_state(self.h.h, *[_ffi.ptr(a) for a in arrs]))

    def get_latch(self):
        """Per-env reward latch int32 [N] (SewNeedle's _threaded_needle, env.py:602): state next to qpos / qvel / ctrl."""
        l = np.empty(self.N, dtype=np.int32)
        self.h.check(self.h.L.avsim_get_latch(self.h.h, l.ctypes.data))
        return l

    def set_latch(self, latch):
        l = np.ascontiguousarray(latch, dtype=np.int32).reshape(self.N)
        self.h.check(self.h.L.avsim_set_latch(self.h.h, l.ctypes.data))

    def get_reset_poses(self):
        """[N, nobj, 7] object poses a diverged env is put back to (those of the episode's reset)."""
        o = np.empty((self.N, self.nobj, 7))
        self.h.check(self.h.L.avsim_get_reset_poses(self.h.h, o.ctypes.data))
        return o

    def set_reset_poses(self, poses):
        o = np.ascontiguousarray(poses, dtype=np.float64).reshape(self.N, self.nobj, 7)
        self.h.check(self.h.L.avsim_set_reset_poses(self.h.h, o.ctypes.data))

    def set_qpos(self, qpos):
        q = np.ascontiguousarray(qpos, dtype=np.float64).reshape(self.N, self.nq)
        self.h.check(self.h.L.avsim_set_qpos(self.h.h, q.ctypes.data))

    def contacts(self):
        ncon = np.empty(self.N, dtype=np.int32)
        pairs = np.empty((self.N, self.maxcon, 2), dtype=np.int32)
        dist = np.empty((self.N, self.maxcon))
        self.h.check(self.h.L.avsim_get_contacts(self.h.h, ncon.ctypes.data, pairs.ctypes.data, dist.ctypes.data))
        return ncon, pairs, dist

    def render_depth(self, cameras, height, width):
        """Depth images float32 [N, len(cameras), height, width] (metres along the optical axis) of the named cameras
        (names from the manifest's camera table, or indices) at the current state."""
        names = self.manifest["camera_names"]
        ids = np.array([names.index(c) if isinstance(c, str) else int(c) for c in cameras], dtype=np.int32)
        out = np.empty((self.N, len(ids), height, width), dtype=np.float32)
        self.h.check(self.h.L.avsim_render_depth(self.h.h, ids.ctypes.data, len(ids), height, width, out.ctypes.data))
        return out

    def load_visual(self, path=None):
        """The visual scene of render_rgb: the decimated mesh library models/visual_meshes.avv (compiler/vismesh.py) against the
        instances in the model blob; from then on render_rgb rasterises the visual meshes (robot, frame, textured table, task
        objects) instead of the collision proxies."""
        path = path or os.path.join(MODEL_DIR, "visual_meshes.avv")
        with open(path, "rb") as f:
            lib = f.read()
        self.h.check(self.h.L.avsim_load_visual(self.h.h, lib, len(lib)))
        self._visual = True

    def visual_info(self):
        """{triangles, vertices, overflow flags of the last visual render, instances in the model}."""
        info = np.zeros(4, dtype=np.int32)
        self.h.check(self.h.L.avsim_visual_info(self.h.h, info.ctypes.data))
        return dict(zip(("triangles", "vertices", "overflow", "instances"), (int(x) for x in info)))

    def render_rgb(self, cameras, height, width, visual=True, cam_major=False, out=None):
        """Colour images uint8 [N, len(cameras), height, width, 3] of the named cameras at the current state (the layout
        of the reference's "pixels" observation, env.py:180-188).  visual: the scene's visual meshes (loaded on first use from
        models/visual_meshes.avv); False: the collision proxies in flat colours (the depth renderer's geometry)."""
        if visual and not getattr(self, "_visual", False):
            self.load_visual()
        self.set_option("render_proxies", 0 if visual else 1)
        self.set_option("render_cam_major", 1 if (cam_major and visual) else 0)      # [len(cameras), N, height, width, 3]: every camera's batch contiguous
        names = self.manifest["camera_names"]
        ids = np.array([names.index(c) if isinstance(c, str) else int(c) for c in cameras], dtype=np.int32)
        shape = (len(ids), self.N, height, width, 3) if (cam_major and visual) else (self.N, len(ids), height, width, 3)
        if out is None:
            out = np.empty(shape, dtype=np.uint8)
        else:           # the caller's own memory (a slice of a data set's image array): no copy after the one from the device
            assert out.dtype == np.uint8 and out.flags.c_contiguous and out.size == int(np.prod(shape)), "render_rgb(out=...): a C-contiguous uint8 array of the result's size"
        self.h.check(self.h.L.avsim_render_rgb(self.h.h, ids.ctypes.data, len(ids), height, width, out.ctypes.data))
        if visual:
            ov = self.visual_info()["overflow"]
            if ov:          # the image lacks triangles: say so instead of handing a policy a silently incomplete observation
                import warnings
                warnings.warn(f"avsim_render_rgb: a view ran out of {'triangle records' if ov & 1 else ''}{' and ' if ov == 3 else ''}{'tile-list entries' if ov & 2 else ''} "
                              f"at {height}x{width}: triangles were dropped from the image", RuntimeWarning, stacklevel=2)
        return out

    def reward_from_pairs(self, geom_pairs, latch=None):
        """The task's get_reward (env.py:425-863) on explicit contact lists: geom_pairs int [nsets, cap, 2] (collision
        geom ids, negative = empty slot); latch int32 [nsets] is updated in place.  Returns int32 [nsets]."""
        p = np.ascontiguousarray(geom_pairs, dtype=np.int32)
        assert p.ndim == 3 and p.shape[2] == 2
        if latch is not None:
            assert latch.dtype == np.int32 and latch.shape == (p.shape[0],) and latch.flags.c_contiguous
        rw = np.empty(p.shape[0], dtype=np.int32)
        self.h.check(self.h.L.avsim_reward_from_pairs(self.h.h, p.ctypes.data, p.shape[0], p.shape[1],
                                                      latch.ctypes.data if latch is not None else None, rw.ctypes.data))
        return rw

    def diag(self):
        d = np.empty((self.N, 4), dtype=np.int32)
        self.h.check(self.h.L.avsim_get_diag(self.h.h, d.ctypes.data))
        return d

    def close(self):
        self.h.close()
