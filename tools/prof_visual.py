"""Timing of the visual-mesh colour renderer (avsim_load_visual + avsim_render_rgb) with device-resident output (run on the GPU box):
python tools/prof_visual.py [num_envs] [HxW]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from av_aloha_amd import _ffi
from av_aloha_amd.sim import load_blob
from av_aloha_amd.constants import MODEL_DIR
from test_oracle_physics import OBJ

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H, W = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "480x640").split("x"))
blob, man = load_blob("slot_insertion", 3)
h = _ffi.Handle(blob, N, 0, _ffi.AVSIM_IO_DEVICE)
L = h.L
lib = open(os.path.join(MODEL_DIR, "visual_meshes.avv"), "rb").read()
h.check(L.avsim_load_visual(h.h, lib, len(lib)))
for opt, env in (("render_shadows", "SHADOWS"), ("render_samples", "SAMPLES"), ("render_shadow_size", "SHSIZE"), ("render_smooth", "SMOOTH")):      # SHADOWS=1 SAMPLES=4 SMOOTH=1: the facades' defaults
    if os.environ.get(env):
        h.check(L.avsim_set_option(h.h, opt.encode(), float(os.environ[env])))
obj = torch.tensor(np.repeat(OBJ[None], N, 0).reshape(N, -1), dtype=torch.float64, device="cuda")
h.check(L.avsim_reset(h.h, None, obj.data_ptr()))
cams = ["zed_cam_left", "wrist_cam_left", "wrist_cam_right", "overhead_cam"]
ids = np.array([man["camera_names"].index(c) for c in cams], dtype=np.int32)
out = torch.empty((N, len(ids), H, W, 3), dtype=torch.uint8, device="cuda")
for rep in range(3):
    h.check(L.avsim_reset(h.h, None, obj.data_ptr()))        # (a new state version: the call below renders its shadow maps, as a call after a step does)
    h.check(L.avsim_sync(h.h))
    torch.cuda.synchronize(); t = time.time()
    h.check(L.avsim_render_rgb(h.h, ids.ctypes.data, len(ids), H, W, out.data_ptr()))
    h.check(L.avsim_sync(h.h)); torch.cuda.synchronize()
    dt = time.time() - t
    print(f"{N} envs x {len(ids)} cameras x {H} x {W}: {dt * 1e3:.2f} ms = {dt / (N * len(ids)) * 1e6:.1f} us per view, {out.numel() / dt / 1e9:.1f} GB/s written")
info = np.zeros(4, dtype=np.int32); h.check(L.avsim_visual_info(h.h, info.ctypes.data)); print("scene", info)
prof = np.zeros((N * len(ids), 8), dtype=np.int32); h.check(L.avsim_visual_profile(h.h, prof.ctypes.data, len(prof)))
prof = prof.reshape(N, len(ids), 8)
for ci, c in enumerate(cams):
    p = prof[:, ci].mean(0)
    print(f"{c:16s} kcycles: transform {p[1]:.0f} set-up {p[2]:.0f} count {p[3]:.0f} fill {p[4]:.0f} tiles {p[5]:.0f}; records {p[6]:.0f}, list entries {p[7]:.0f}")
