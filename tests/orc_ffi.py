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


def load_model(task="slot_insertion", num_arms=3, variant="gym", hulls="model"):
    """hulls: "model" = the collision hulls of the model blob (<= 128 vertices per mesh, what the device collides); "full" = the FULL convex hulls of the STL files
    (models/oracle_full_hulls.*, compile.py --oracle-hulls full), as MuJoCo collides mesh geoms [EXT] -- the oracle's faithful mode."""
    import json
    base = os.path.join(ROOT, "models", f"{'dc_' if variant == 'data_collection' else ''}{task}_{num_arms}arms")
    b = open(base + ".avm", "rb").read()
    m = lib().orc_model_load(b, C.c_size_t(len(b)))
    assert m
    m = C.c_void_p(m)
    if hulls == "full":
        import sys
        sys.path.insert(0, ROOT)
        from av_aloha_amd.compiler.compile import read_blob
        md = read_blob(base + ".avm")
        man = json.load(open(base + ".json"))
        fh = read_blob(os.path.join(ROOT, "models", "oracle_full_hulls.avh"))
        fnames = json.load(open(os.path.join(ROOT, "models", "oracle_full_hulls.json")))["mesh_names"]
        # the blob's hulls in the order the compiler met them = the order of the manifest's report: first vertex -> mesh name
        adr, name_of = 0, {}
        for name, info in man["hulls"].items():
            name_of[adr] = name
            adr += info["collision_nvert"]
        gh = np.asarray(md["geom_chull"], dtype=np.int32).reshape(-1, 2)
        bc = np.asarray(md["geom_bcenter"], dtype=np.float64).reshape(-1, 3)
        rb = np.asarray(md["geom_rbound"], dtype=np.float64).copy()
        new = np.zeros_like(gh)
        for g in range(len(gh)):
            if gh[g, 1] > 0:
                k = fnames.index(name_of[int(gh[g, 0])])
                new[g] = (fh["full_adr"][k], fh["full_num"][k])
                v = fh["full_vert"][new[g, 0]:new[g, 0] + new[g, 1]]
                rb[g] = np.sqrt(((v - bc[g]) ** 2).sum(1).max())
        vert = np.ascontiguousarray(fh["full_vert"], dtype=np.float64)
        new = np.ascontiguousarray(new, dtype=np.int32)
        lib().orc_model_set_hulls(m, dp(vert), C.c_int(len(vert)), ip(new), dp(np.ascontiguousarray(rb)))
    else:
        assert hulls == "model", hulls
    return m
