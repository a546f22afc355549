#!/usr/bin/env python
"""CPU-only experiment: how many hull vertices does the device need?  The oracle runs a scripted episode closed loop with the FULL hulls
(the faithful mode), recording its state at every env-step; oracles with hulls decimated to various vertex budgets are then teacher-forced
along that trajectory (same state, same ctrl, one env-step): fraction of steps whose contact count differs, one-step position difference.

    python tools/exp_hull_budget.py hook_package 8 "20,32" "32,64" "48,64" "64,64"      # budgets "links,hand" (hand = gripper parts, wrist, camera mounts, fingers)
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import episode_util as U
from av_aloha_amd import workloads as W
from av_aloha_amd.compiler import hull as H
from av_aloha_amd.compiler.compile import read_blob
from av_aloha_amd.compiler.mjcf import parse
from orc_env import OrcEnv
from orc_ffi import dp, ip, lib

HAND = ("gripper", "d405", "finger", "wrist", "zedm")
_cache = {}


def hulls_for(task, budget_links, budget_hand, budget_finger=None):
    """(vert [n, 3], geom_hull [ngeom, 2], rbound [ngeom]) for the data-collection model of `task` with the given vertex budgets; budget 0 = full hull."""
    base = os.path.join(ROOT, "models", f"dc_{task}_3arms")
    md, man = read_blob(base + ".avm"), json.load(open(base + ".json"))
    m = parse(os.path.join("/root/reference/data_collection_scripts/assets", {"hook_package": "task_hook_package.xml", "slot_insertion": "task_slot_insertion.xml",
              "insert_peg": "task_insert_peg.xml", "sew_needle": "task_sew_needle.xml", "tube_transfer": "task_tube_transfer.xml"}[task]))
    adr, name_of = 0, {}
    for name, info in man["hulls"].items():
        name_of[adr] = name
        adr += info["nvert"]
    gh = np.asarray(md["geom_hull"], dtype=np.int32).reshape(-1, 2)
    bc = np.asarray(md["geom_bcenter"]).reshape(-1, 3)
    rb = np.asarray(md["geom_rbound"], dtype=np.float64).copy()
    verts, where, new = [], {}, np.zeros_like(gh)
    for g in range(len(gh)):
        if gh[g, 1] == 0:
            continue
        name = name_of[int(gh[g, 0])]
        k = budget_hand if any(s in name for s in HAND) else (budget_links if name.startswith("vx300s") else 20)
        if budget_finger is not None and "finger" in name:
            k = budget_finger
        if (name, k) not in _cache:
            me = m.meshes[name]
            pts = H.read_stl(me["file"]) * me["scale"]
            from scipy.spatial import ConvexHull
            _cache[(name, k)] = pts[ConvexHull(pts).vertices] if k == 0 else H.decimate_hull(pts, k)[0]
        if name not in where:
            where[name] = (sum(len(v) for v in verts), len(_cache[(name, k)]))
            verts.append(_cache[(name, k)])
        new[g] = where[name]
        v = _cache[(name, k)]
        rb[g] = max(rb[g], np.sqrt(((v - bc[g]) ** 2).sum(1).max()))
    return np.ascontiguousarray(np.concatenate(verts)), np.ascontiguousarray(new, dtype=np.int32), np.ascontiguousarray(rb)


def new_env(task, pose, hulls):
    e = OrcEnv(U.MODEL_OF.get(task, task), 3, U.VARIANT)
    vert, gh, rb = hulls
    e.L.orc_model_set_hulls(e.m, dp(vert), C.c_int(len(vert)), ip(gh), dp(rb))
    e.d.solver = 1
    e.reset(pose)
    return e


def closed_loop(args):
    task, pose, home, hulls = args
    e = new_env(task, pose, hulls)
    script = U.make_script(task, home, np.array(e.qpos)[None])
    T = script.steps()
    q0, v0, w0, l0, cs, q1, nc, rw = np.zeros((T, e.nq)), np.zeros((T, e.nv)), np.zeros((T, e.nv)), np.zeros(T, np.int32), np.zeros((T, e.nu)), np.zeros((T, e.nq)), np.zeros(T, np.int32), np.zeros(T, np.int32)
    a21 = np.zeros(21)
    lo, hi = U.GRIP_RANGE
    warm = e.arr("qacc_warmstart", e.nv)
    for t in range(T):
        q0[t], v0[t], w0[t], l0[t] = e.qpos, e.qvel, warm, e.d.threaded
        a = np.ascontiguousarray(script.action(np.array(e.qpos)[None])[0])
        e.L.orc_cart_to_ctrl(e.dptr, dp(a), 0, dp(a21))
        c = a21.copy()
        for k in (6, 13):
            c[k] = a21[k] * (hi - lo) + lo
        cs[t] = c
        rw[t], _ = U._step_ctrl(e, c)
        q1[t], nc[t] = e.qpos, e.d.ncon
    e.close()
    return q0, v0, w0, l0, cs, q1, nc, rw


def forced(args):
    task, pose, hulls, q0, v0, w0, l0, cs, q1, nc = args
    e = new_env(task, pose, hulls)
    warm = e.arr("qacc_warmstart", e.nv)
    T = len(cs)
    er, nd, rws = np.zeros(T), 0, np.zeros(T, np.int32)
    for t in range(T):
        e.qpos[:] = q0[t]; e.qvel[:] = v0[t]; warm[:] = w0[t]
        e.d.threaded = int(l0[t])
        rws[t], _ = U._step_ctrl(e, cs[t])
        er[t] = np.abs(np.array(e.qpos) - q1[t]).max()
        nd += int(e.d.ncon != nc[t])
    e.close()
    return er, nd, rws


if __name__ == "__main__":
    task, n = sys.argv[1], int(sys.argv[2])
    budgets = [tuple(int(x) for x in b.split(",")) for b in sys.argv[3:]] or [(20, 32), (32, 64), (48, 64), (64, 64)]
    model = U.MODEL_OF.get(task, task)
    poses = W.object_poses(model, np.arange(n), U.TASK_SEED[task])
    home = U.oracle_home(task)
    full = hulls_for(model, 0, 0)
    ref = U.pool_map(closed_loop, [(task, poses[k], home, full) for k in range(n)])
    T = len(ref[0][4])
    print(f"{task}: {n} envs x {T} steps closed loop on the full-hull oracle; max reward reached {[int(r[7].max()) for r in ref]}")
    for b in budgets:
        bl, bh = b[0], b[1]
        hz = hulls_for(model, *b)
        res = U.pool_map(forced, [(task, poses[k], hz) + tuple(ref[k][:7]) for k in range(n)])
        nd = sum(r[1] for r in res)
        rd = sum(int((r[2] != ref[k][7]).sum()) for k, r in enumerate(res))
        e = np.array([r[0].max() for r in res])
        em = np.concatenate([r[0] for r in res])
        print(f"  {b}: links {bl:3d} / hand {bh:3d} vertices ({len(hz[0])} in all): ncon differs in {nd} of {n * T} steps ({nd / (n * T):.4f}), reward in {rd}; one-step |dq| per env max p50 / p90 / max "
              f"{np.percentile(e, 50):.2e} / {np.percentile(e, 90):.2e} / {e.max():.2e}; per step p50 / p99 {np.percentile(em, 50):.2e} / {np.percentile(em, 99):.2e}", flush=True)
