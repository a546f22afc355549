"""Writes gpurun_out/r05_visual_<cam>_shadows_ss.png: the device's colour image with shadows and 2 x 2 supersampling (480 x 640), and the plain one next to it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, time
import png
from av_aloha_amd.sim import BatchedSim
from test_oracle_physics import OBJ
sim = BatchedSim("slot_insertion", 3, 1)
sim.reset(OBJ[None])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for cam in ("overhead_cam", "zed_cam_left"):
    a = sim.render_rgb([cam], 480, 640)[0, 0]
    sim.set_option("render_shadows", 1); sim.set_option("render_samples", 4)
    b = sim.render_rgb([cam], 480, 640)[0, 0]
    sim.set_option("render_shadows", 0); sim.set_option("render_samples", 1)
    png.write_png(os.path.join(ROOT, "gpurun_out", f"r05_visual_{cam}_shadows_ss.png"), np.concatenate([a, b], axis=1))
    print(cam, "shadowed / changed pixels", (np.abs(a.astype(int) - b.astype(int)).sum(-1) > 12).mean(), sim.visual_info())
sim.close()
N = 1024
sim = BatchedSim("slot_insertion", 3, N)
sim.reset(np.repeat(OBJ[None], N, 0))
import ctypes
for opts in ((0, 1), (1, 1), (0, 4), (1, 4)):
    sim.set_option("render_shadows", opts[0]); sim.set_option("render_samples", opts[1])
    sim.render_rgb(["zed_cam_left", "zed_cam_right", "wrist_cam_left", "overhead_cam"], 480, 640)
    t0 = time.time()
    for _ in range(3):
        sim.render_rgb(["zed_cam_left", "zed_cam_right", "wrist_cam_left", "overhead_cam"], 480, 640)
    print(f"shadows {opts[0]} samples {opts[1]}: {(time.time() - t0) / 3 * 1e3:.1f} ms per call of {N} envs x 4 cameras x 480 x 640 (host copy of {N * 4 * 480 * 640 * 3 / 1e9:.1f} GB included)")
