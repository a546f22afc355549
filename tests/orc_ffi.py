"""ctypes access to the CPU oracle (oracle/liborc.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


def dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_dp)


def ip(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_ip)


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("ORC_LIB") or os.path.join(ROOT, "oracle", "liborc.so")      # ORC_LIB: the sanitizer build (tests/test_oracle_sanitizers.py)
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _LIB = C.CDLL(so)
        _LIB.orc_model_load.restype = C.c_void_p
        for name, rt in (("orc_data_new", C.c_void_p), ("orc_reward", C.c_int)):
            if hasattr(_LIB, name):
                getattr(_LIB, name).restype = rt
    return _LIB


def load_model(task="slot_insertion", num_arms=3, variant="gym"):
    path = os.path.join(ROOT, "models", f"{'dc_' if variant == 'data_collection' else ''}{task}_{num_arms}arms.avm")
    b = open(path, "rb").read()
    m = lib().orc_model_load(b, C.c_size_t(len(b)))
    assert m
    return C.c_void_p(m)
