import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import episode_util as U
for f64 in (True, False):
    a = U.device_episode("slot_insertion", 8, f64=f64)
    b = U.device_episode("slot_insertion", 8, f64=f64, options={"waves_per_block": 4})
    c = U.device_episode("slot_insertion", 8, f64=f64)
    print("f64", f64, "default vs wpb4: ctrl equal", np.array_equal(a["ctrl"], b["ctrl"]), "qpos max diff", np.abs(a["qpos"] - b["qpos"]).max(), "| default twice equal", np.array_equal(a["qpos"], c["qpos"]))
    if not np.array_equal(a["qpos"], b["qpos"]):
        d = np.abs(a["qpos"] - b["qpos"]).max(axis=(1, 2)); print("  first differing step", int(np.argmax(d > 0)), "per-env first", [(int(np.argmax(np.abs(a['qpos'][:, k] - b['qpos'][:, k]).max(1) > 0))) for k in range(8)])
