"""Small Python view of the CPU oracle's orc_data (test infrastructure only)."""
import ctypes as C
import json
import os

import numpy as np

from orc_ffi import ROOT, dp, ip, lib, load_model

MAXCON, MAXEFC = 128, 640        # ORC_MAXCON / ORC_MAXEFC (oracle/orc.h), checked against orc_capacity() in OrcEnv.__init__


class Contact(C.Structure):
    _fields_ = [("dist", C.c_double), ("pos", C.c_double * 3), ("frame", C.c_double * 9),
                ("geom1", C.c_int), ("geom2", C.c_int), ("pair", C.c_int), ("dim", C.c_int), ("efc_adr", C.c_int),
                ("friction", C.c_double * 5), ("solref", C.c_double * 2), ("solimp", C.c_double * 5),
                ("includemargin", C.c_double)]


class Data(C.Structure):
    _fields_ = ([("m", C.c_void_p)] +
                [(n, C.POINTER(C.c_double)) for n in ("qpos", "qvel", "ctrl", "qacc_warmstart")] +
                [("time", C.c_double), ("threaded", C.c_int)] +
                [(n, C.POINTER(C.c_double)) for n in ("xpos", "xquat", "xmat", "xipos", "ximat", "xanchor", "xaxis", "cdof",
                                                      "geom_xpos", "geom_xmat", "M", "L", "qfrc_bias", "qfrc_passive",
                                                      "qfrc_actuator", "qfrc_smooth", "qacc_smooth", "qfrc_constraint", "qacc")] +
                [("ncon", C.c_int), ("nefc", C.c_int), ("contact", Contact * MAXCON),
                 ("efc_J", C.POINTER(C.c_double)), ("efc_B", C.POINTER(C.c_double))] +
                [(n, C.c_double * MAXEFC) for n in ("efc_pos", "efc_margin", "efc_aref", "efc_R", "efc_D", "efc_force",
                                                    "efc_diag", "efc_floss")] +
                [("efc_KBIP", C.c_double * (4 * MAXEFC)), ("efc_type", C.c_int * MAXEFC), ("efc_id", C.c_int * MAXEFC),
                 ("pgs_iters", C.c_int), ("pgs_tol", C.c_double), ("pgs_scale", C.c_double), ("stat_sweeps", C.c_int),
                 ("solver", C.c_int), ("newton_iters", C.c_int), ("newton_tol", C.c_double),
                 ("overflow", C.c_int), ("stat_narrow", C.c_long), ("stat_noslip", C.c_int), ("ls_tol", C.c_double), ("ls_iters", C.c_int)])


class OrcEnv:
    def __init__(self, task="slot_insertion", num_arms=3, variant="gym", hulls="model"):
        self.L = lib()
        assert (self.L.orc_capacity(0), self.L.orc_capacity(1), self.L.orc_capacity(2)) == (MAXCON, MAXEFC, C.sizeof(Data)), "tests/orc_env.py Data is out of step with oracle/orc.h orc_data"
        self.m = load_model(task, num_arms, variant, hulls=hulls)
        self.man = json.load(open(os.path.join(ROOT, "models", f"{'dc_' if variant == 'data_collection' else ''}{task}_{num_arms}arms.json")))
        self.nq, self.nv, self.nu = self.man["nq"], self.man["nv"], self.man["nu"]
        self.nj = 21 if num_arms == 3 else 14
        self.L.orc_data_new.restype = C.c_void_p
        self.dptr = C.c_void_p(self.L.orc_data_new(self.m))
        self.d = C.cast(self.dptr, C.POINTER(Data)).contents

    def arr(self, name, n):
        return np.ctypeslib.as_array(getattr(self.d, name), shape=(n,))

    @property
    def qpos(self):
        return self.arr("qpos", self.nq)

    @property
    def qvel(self):
        return self.arr("qvel", self.nv)

    @property
    def ctrl(self):
        return self.arr("ctrl", self.nu)

    def reset(self, obj_qpos):
        o = np.ascontiguousarray(obj_qpos, dtype=np.float64).reshape(-1)
        self.L.orc_reset(self.dptr, dp(o))

    def set_qpos(self, qpos):
        q = np.ascontiguousarray(qpos, dtype=np.float64).reshape(self.nq)
        self.L.orc_set_qpos(self.dptr, dp(q))

    def step(self, nsub=20):
        self.L.orc_step(self.dptr, nsub)

    def env_step(self, action, nsub=20):
        a = np.ascontiguousarray(action, dtype=np.float64)
        ap = np.zeros(self.nj)
        r, s = C.c_int(0), C.c_int(0)
        self.L.orc_env_step(self.dptr, dp(a), nsub, dp(ap), C.byref(r), C.byref(s))
        return ap, r.value, bool(s.value)

    def contacts(self):
        names = self.man["geom_names"]
        return [(names[c.geom1], names[c.geom2], c.dist, c.efc_adr) for c in list(self.d.contact)[: self.d.ncon]]

    def render_depth(self, cam, H, W):
        """cam: name or index into the manifest's camera table; float32 [H, W] metres along the optical axis."""
        ci = self.man["camera_names"].index(cam) if isinstance(cam, str) else int(cam)
        out = np.empty((H, W), dtype=np.float32)
        self.L.orc_render_depth.restype = C.c_int
        hits = self.L.orc_render_depth(self.dptr, ci, H, W, out.ctypes.data_as(C.c_void_p))
        assert hits >= 0
        return out

    def render_depth_rows(self, cam, H, W, row0, row_step, nrows):
        """rows row0, row0 + row_step, ... of the H x W depth image: float32 [nrows, W] (the same rays as render_depth)."""
        ci = self.man["camera_names"].index(cam) if isinstance(cam, str) else int(cam)
        out = np.empty((nrows, W), dtype=np.float32)
        self.L.orc_render_depth_rows.restype = C.c_int
        hits = self.L.orc_render_depth_rows(self.dptr, ci, H, W, row0, row_step, nrows, out.ctypes.data_as(C.c_void_p))
        assert hits >= 0
        return out

    def render_rgb(self, cam, H, W):
        """uint8 [H, W, 3] colour image and float32 [H, W] depth of the same rays."""
        ci = self.man["camera_names"].index(cam) if isinstance(cam, str) else int(cam)
        out = np.empty((H, W, 3), dtype=np.uint8)
        dep = np.empty((H, W), dtype=np.float32)
        self.L.orc_render_rgb.restype = C.c_int
        hits = self.L.orc_render_rgb(self.dptr, ci, H, W, out.ctypes.data_as(C.c_void_p), dep.ctypes.data_as(C.c_void_p))
        assert hits >= 0
        return out, dep

    def render_visual(self, cam, H, W, scene, ss=1, shadows=False, smooth=False):
        """Colour image uint8 [H, W, 3] of the visual scene (scene = vismesh.expand_instances(library, model blob) + (texels,)), the index of the
        triangle every pixel sees (-1: sky) and its depth: oracle/orc_vis.c, one ray per pixel against every triangle.  smooth (the device's
        option render_smooth, on in the gym / Cartesian facades): the corners lit with their own normals (the scene's tnorm) and interpolated; False: one shade per triangle."""
        ci = self.man["camera_names"].index(cam) if isinstance(cam, str) else int(cam)
        if len(scene) == 8:
            vert, vbody, tri, rgb, uv, tex, tnorm, texel = scene
        else:                                   # (a scene without corner normals: flat shading)
            vert, vbody, tri, rgb, uv, tex, texel = scene
            tnorm = None
        vert = np.ascontiguousarray(vert, dtype=np.float64); vbody = np.ascontiguousarray(vbody, dtype=np.int32)
        tri = np.ascontiguousarray(tri, dtype=np.int32); rgb = np.ascontiguousarray(rgb, dtype=np.float64)
        uv = np.ascontiguousarray(uv, dtype=np.float64); tex = np.ascontiguousarray(tex, dtype=np.int32)
        texel = np.ascontiguousarray(texel, dtype=np.int32)
        tn = np.ascontiguousarray(tnorm, dtype=np.float64) if (tnorm is not None and smooth) else None
        out = np.empty((H, W, 3), dtype=np.uint8)
        tid = np.empty((H, W), dtype=np.int32)
        dep = np.empty((H, W), dtype=np.float64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        # ss = 2: the four samples 1/4 pixel off the centre averaged; shadows: an exact ray towards the scene's directional light per sample
        self.L.orc_vis_render_sm.restype = C.c_int
        hits = self.L.orc_vis_render_sm(self.dptr, ci, len(vbody), vp(vert), vp(vbody), len(tex), vp(tri), vp(rgb), vp(uv), vp(tex), vp(texel),
                                        int(round(len(texel) ** 0.5)), vp(tn), H, W, int(ss), int(bool(shadows)), vp(out), vp(tid), vp(dep))
        assert hits >= 0
        return out, tid, dep

    def close(self):
        self.L.orc_data_free(self.dptr)
