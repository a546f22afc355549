"""Writes gpurun_out/r06_visual_<cam>_flat_smooth.png: the device's colour image (480 x 640, shadows + 4 samples) with one shade per triangle
(option render_smooth 0: rounds 3-5) next to smooth shading (the corners lit with their own normals, interpolated: round 6), and times both."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import png
from av_aloha_amd.sim import BatchedSim
from test_oracle_physics import OBJ
sim = BatchedSim("slot_insertion", 3, 1)
sim.reset(OBJ[None])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
sim.set_option("render_shadows", 1); sim.set_option("render_samples", 4)
for cam in ("zed_cam_left", "wrist_cam_left"):
    sim.set_option("render_smooth", 0)
    a = sim.render_rgb([cam], 480, 640)[0, 0]
    sim.set_option("render_smooth", 1)
    b = sim.render_rgb([cam], 480, 640)[0, 0]
    png.write_png(os.path.join(ROOT, "gpurun_out", f"r06_visual_{cam}_flat_smooth.png"), np.concatenate([a, b], axis=1))
    print(cam, "pixels that differ by more than 4 levels:", (np.abs(a.astype(int) - b.astype(int)).max(-1) > 4).mean(), sim.visual_info())
sim.close()
