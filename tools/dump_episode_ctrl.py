"""ctrl sequence of the scripted SewNeedle episode on the device (tests/episode_util.py) with the library of the current tree; two
builds with the same arithmetic give the same file.  usage (GPU box): python tools/dump_episode_ctrl.py out.npz [f64]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import episode_util as U
dev = U.device_episode("sew_needle", 8, f64=len(sys.argv) > 2)
np.savez(sys.argv[1], ctrl=dev["ctrl"], reward=dev["reward"], ncon=dev["ncon"])
