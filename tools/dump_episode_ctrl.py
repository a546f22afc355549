"""ctrl sequence, rewards and contact counts of the scripted SewNeedle episode on the device (tests/episode_util.py) with the library of the
current tree; two builds with the same arithmetic give the same file (the GradIK descent is chaotic: a rounding-level difference anywhere in the
step shows within a few steps).  usage (GPU box): python tools/dump_episode_ctrl.py out.npz [f64]      compare: --compare a.npz b.npz"""
import os, sys
import numpy as np
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = np.abs(a["ctrl"] - b["ctrl"]).max(axis=(1, 2)); nz = np.nonzero(d)[0]
    print("steps", len(d), "first differing steps", nz[:3], "max |ctrl difference|", d.max(), "contact counts equal", np.array_equal(a["ncon"], b["ncon"]), "max ncon", a["ncon"].max())
    sys.exit(0)
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import episode_util as U
dev = U.device_episode("sew_needle", 8, f64=len(sys.argv) > 2)
np.savez(sys.argv[1], ctrl=dev["ctrl"], reward=dev["reward"], ncon=dev["ncon"])
