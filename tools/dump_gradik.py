"""GradIK outputs of the library in the current directory on the golden inputs, all 50 iterations (chaotic: equal outputs of two builds
mean equal arithmetic).  usage (GPU box): python tools/dump_gradik.py out.npy, from the tree whose av_aloha_amd/libavsim.so is meant."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from avsim_test_util import blob
from av_aloha_amd._ffi import Handle
G = os.path.join(os.getcwd(), "tests", "golden")
h = Handle(blob(), 8)
outs = []
for ai, arm in enumerate(("left", "right")):
    d = np.load(os.path.join(G, f"gradik_{arm}.npz"))
    q = np.ascontiguousarray(d["q"]); pos = np.ascontiguousarray(d["target_pos"]); quat = np.ascontiguousarray(d["target_quat_wxyz"])
    for K in (1, 8, 0):
        out = np.zeros(q.shape)
        h.check(h.L.avsim_ik(h.h, ai, 1, K, q.shape[0], q.ctypes.data, pos.ctypes.data, quat.ctypes.data, out.ctypes.data))
        outs.append(out)
np.save(sys.argv[1], np.concatenate(outs))
h.close()
