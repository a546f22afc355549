"""Timing of the depth renderer at BASELINE config 5 sizes (debug aid; run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from av_aloha_amd import _ffi
from av_aloha_amd.sim import load_blob
from test_oracle_physics import OBJ
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 640)
cams = ["zed_cam_left", "zed_cam_right", "wrist_cam_left", "wrist_cam_right"]
blob, man = load_blob("slot_insertion", 3)
h = _ffi.Handle(blob, N, 0, _ffi.AVSIM_IO_DEVICE)
L = h.L
dev = torch.device("cuda:0")
h.check(L.avsim_set_stream(h.h, torch.cuda.current_stream().cuda_stream))
obj = torch.tensor(np.repeat(OBJ[None], N, 0).reshape(N, -1), device=dev)
h.check(L.avsim_reset(h.h, None, obj.data_ptr()))
ids = np.array([man["camera_names"].index(c) for c in cams], dtype=np.int32)
out = torch.empty((N, len(ids), H, W), dtype=torch.float32, device=dev)
for _ in range(2):
    h.check(L.avsim_render_depth(h.h, ids.ctypes.data, len(ids), H, W, out.data_ptr()))
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    h.check(L.avsim_render_depth(h.h, ids.ctypes.data, len(ids), H, W, out.data_ptr()))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
gb = out.numel() * 4 / 1e9
print(f"N={N} {len(ids)} cams {H}x{W}: {dt * 1e3:.2f} ms per render call, {gb:.2f} GB out -> {gb / dt:.1f} GB/s write; hit fraction {(out < 30).float().mean().item():.3f}")
rgb = torch.empty((N, len(ids), H, W, 3), dtype=torch.uint8, device=dev)
for _ in range(2):
    h.check(L.avsim_render_rgb(h.h, ids.ctypes.data, len(ids), H, W, rgb.data_ptr()))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    h.check(L.avsim_render_rgb(h.h, ids.ctypes.data, len(ids), H, W, rgb.data_ptr()))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"colour: {dt * 1e3:.2f} ms per render call, {rgb.numel() / 1e9:.2f} GB out -> {rgb.numel() / 1e9 / dt:.1f} GB/s write; mean level {rgb.float().mean().item():.1f}")
