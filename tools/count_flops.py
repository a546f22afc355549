#!/usr/bin/env python
"""Counted floating-point operations per env-step of the measurement configurations ("to be replaced by counted
values from the CPU restatement's instrumented build").

Builds the oracle a second time as C++ with `#define double cdouble` (oracle/count/cdouble.h: a double whose +, -, *, / and sqrt
count one flop each and whose libm calls count eight), runs the workloads of av_aloha_amd/workloads.py on a few envs and writes
the per-phase counts to profiles/flop_counts.json, which bench.py reads for its `valu_*` fields.  These are the flops of the
ALGORITHM as the scalar oracle executes it (dense nv x nv Jacobian rows and all); the kernel's sparse row windows do fewer.
CPU only; test / measurement infrastructure."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
PHASES = ["other", "kinematics", "crb", "collide", "rne", "smooth", "rows", "newton", "noslip", "euler", "ik"]


def build(tmp):
    objs = []
    for f in sorted(os.listdir(os.path.join(ROOT, "oracle"))):
        if not (f.startswith("orc_") and f.endswith(".c")) or f == "orc_render.c":
            continue
        w = os.path.join(tmp, "w_" + f[:-2] + ".cpp")
        with open(w, "w") as fh:
            fh.write("#include <math.h>\n#include <stdlib.h>\n#include <string.h>\n#include <stdio.h>\n#include <stdint.h>\n#include <stddef.h>\n"
                     f'#include "{ROOT}/oracle/count/cdouble.h"\n#define double cdouble\n#define ORC_COUNT_FLOPS 1\n#include "{ROOT}/oracle/{f}"\n')
        o = w[:-4] + ".o"
        subprocess.check_call(["g++", "-std=c++17", "-fpermissive", "-w", "-O1", "-fPIC", "-ffp-contract=off", "-c", w, "-o", o])
        objs.append(o)
    g = os.path.join(tmp, "globals.cpp")
    with open(g, "w") as fh:
        fh.write('extern "C" { long long orc_flops[16]; int orc_phase; }\n')
    so = os.path.join(tmp, "liborc_count.so")
    subprocess.check_call(["g++", "-shared", "-fPIC", "-o", so, g] + objs)
    return so


def main():
    import orc_ffi
    from av_aloha_amd import workloads as W
    from av_aloha_amd.compiler.compile import read_blob
    tmp = tempfile.mkdtemp(prefix="orc_count_")
    so = build(tmp)
    lib = C.CDLL(so)
    lib.orc_model_load.restype = C.c_void_p
    lib.orc_data_new.restype = C.c_void_p
    orc_ffi._LIB = lib                       # OrcEnv picks the counting build up
    from orc_env import OrcEnv
    from orc_ffi import dp
    flops = (C.c_longlong * 16).in_dll(lib, "orc_flops")
    nenv = int(os.environ.get("COUNT_ENVS", "4"))
    out = {"note": "flops per env-step (20 substeps + controller) of the scalar f64 oracle, counted by oracle/count/cdouble.h: + - * / sqrt = 1, "
                   "libm transcendental = 8; mean over the envs and steps sampled; tools/count_flops.py",
           "phases": PHASES}
    home = {k: np.load(os.path.join(ROOT, "tests", "golden", f"fk_jac_{k}.npz"))["fk"][0] for k in ("left", "right", "middle")}
    home = W.home_poses([home["left"], home["right"], home["middle"]])
    for cfg_id, steps in ((2, 40), (3, 240), (4, 40)):
        cfg = W.CONFIGS[cfg_id]
        ids = np.arange(nenv)
        poses = W.object_poses(cfg["task"], ids, cfg["seed"])
        md = read_blob(os.path.join(ROOT, "models", f"{cfg['task']}_{cfg['arms']}arms.avm"))
        envs = []
        for k in ids:
            e = OrcEnv(cfg["task"], cfg["arms"])
            e.d.solver = 1
            e.reset(poses[k])
            envs.append(e)
        if cfg["action"] == "cartesian_reference":
            acts = list(W.grasp_lift_targets(home, poses[:, 1, :3] + np.array([0.0, 0.0, 0.01]), sway=0.02))[:steps]
        elif cfg["action"] == "joint":
            acts = W.walk_actions(md["qpos_home"], md["act_ctrlrange"], ids, steps, 14, cfg["seed"]).astype(np.float64)
        for i in range(16):
            flops[i] = 0
        a21 = np.zeros(21)
        ncon = 0
        for t in range(steps):
            if cfg["action"] == "cartesian_dls":
                a = W.sinusoid_actions(home, ids, 4096, t)
            elif cfg["action"] == "cartesian_reference":
                a = acts[t]
            for k, e in enumerate(envs):
                if cfg["action"] == "joint":
                    e.env_step(acts[t, k])
                else:
                    e.L.orc_cart_to_ctrl(e.dptr, dp(np.ascontiguousarray(a[k])), 1 if cfg["action"] == "cartesian_dls" else 0, dp(a21))
                    e.env_step(a21)
                ncon += e.d.ncon
        per = np.array([flops[i] for i in range(len(PHASES))], dtype=np.float64) / (nenv * steps)
        out[f"config{cfg_id}"] = {"flops_per_env_step": float(per.sum()), "by_phase": {p: float(v) for p, v in zip(PHASES, per)},
                                 "envs": nenv, "env_steps_each": steps, "mean_ncon": ncon / (nenv * steps), "workload": cfg["gym_id"] + " / " + cfg["action"]}
        print(f"config {cfg_id}: {per.sum() / 1e6:.2f} Mflop per env-step;", ", ".join(f"{p} {v / 1e3:.0f}k" for p, v in zip(PHASES, per)), flush=True)
        for e in envs:
            e.close()
    json.dump(out, open(os.path.join(ROOT, "profiles", "flop_counts.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
