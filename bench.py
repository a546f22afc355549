#!/usr/bin/env python
"""Headline benchmark: env-steps/sec of the batched SlotInsertion-3Arms simulation (BASELINE.json metric).

One "step" = one env step of every env on the rank: Cartesian 23-D action -> damped-least-squares IK on the
three arms (k_cart_ctrl) -> 20 physics substeps + agent_pos + reward/success (k_phys).  Inputs (the action
tensor) and all state are resident in HBM before the timed region starts; nothing crosses PCIe inside it.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit/..., plus
  "roofline":     algorithmic HBM bytes of the dominant kernel (k_phys) / its mean launch time (HIP events on the
                  launch stream), against the 8 TB/s HBM peak -- this path keeps its state in LDS for 20 substeps,
                  so the fraction is tiny by design; the VALU view is reported next to it.
  "cpu_baseline": the CPU oracle (oracle/liborc.so, a scalar f64 C restatement) timed on this host's cores on a
                  bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
EPISODE_LEN = 300                      # data_collection_scripts/constants.py:23-58 (slot insertion)
ALGO_BYTES_PER_ENV_STEP = 1040         # fp32 state in/out + 23-D action + agent_pos/reward/success
ALGO_FLOPS_PER_ENV_STEP = 2.0e6        # SURVEY.md 8(d) estimate (0.8-3 Mflop physics + 3 x 70 kflop DLS IK)
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md
VALU_PEAK_TFLOPS = 157.3


def object_poses(global_ids):
    """per env np.random.seed(1000+i), then the reference's draw order."""
    from av_aloha_amd.env import sample_object_poses
    out = np.zeros((len(global_ids), 2, 7))
    for k, i in enumerate(global_ids):
        np.random.seed(1000 + int(i))
        out[k] = sample_object_poses("slot_insertion")
    return out


def home_targets():
    """FK(home) poses of the three eef sites = the centre of the scripted Cartesian motion (known answers of
    SURVEY.md Appendix A, recomputed from the committed IK golden fixture)."""
    T = []
    for arm in ("left", "right", "middle"):
        d = np.load(os.path.join(ROOT, "tests", "golden", f"fk_jac_{arm}.npz"))
        T.append(d["fk"][0])
    return T


def mat2quat_wxyz(R):
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0.0] * 4
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q) * (1 if q[0] >= 0 else -1)


def scripted_actions(global_ids, n_total, t):
    """23-D Cartesian action of env-step t: FK(home) + 3 cm / 0.5 Hz sinusoid with per-env phase 2*pi*i/N,
    trigger square wave every 50 steps.  sim_env.py:278-282 layout."""
    Th = home_targets()
    n = len(global_ids)
    ph = 2 * np.pi * np.asarray(global_ids, dtype=np.float64) / n_total
    w = 2 * np.pi * 0.5 * 0.04 * t
    a = np.zeros((n, 23))
    trig = 1.0 if (t // 50) % 2 == 1 else 0.0
    for arm, off in ((0, 0), (1, 8), (2, 16)):
        p = Th[arm][:3, 3]
        a[:, off + 0] = p[0] + 0.03 * np.sin(w + ph)
        a[:, off + 1] = p[1] + 0.03 * np.cos(w + ph)
        a[:, off + 2] = p[2] + 0.03 * np.sin(2 * w + ph)
        a[:, off + 3:off + 7] = mat2quat_wxyz(Th[arm][:3, :3])
        if arm < 2:
            a[:, off + 7] = trig
    return a


def cpu_baseline_worker(args):
    ids, n_total, steps, solver = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from orc_env import OrcEnv
    from orc_ffi import dp
    poses = object_poses(ids)
    envs = []
    for k in range(len(ids)):
        e = OrcEnv("slot_insertion", 3)
        e.d.pgs_iters = 20
        e.d.solver = solver
        e.reset(poses[k])
        envs.append(e)
    a21 = np.zeros(21)
    t0 = time.perf_counter()
    for t in range(steps):
        acts = scripted_actions(ids, n_total, t)
        for k, e in enumerate(envs):
            e.L.orc_cart_to_ctrl(e.dptr, dp(np.ascontiguousarray(acts[k])), 1, dp(a21))
            e.env_step(a21)
    return time.perf_counter() - t0


def cpu_baseline(n_total, solver=1):
    """Oracle (kind 'port') on the host cores: every core steps its own envs of the same workload."""
    import multiprocessing as mp
    from av_aloha_amd.build import build_oracle
    build_oracle()
    cores = max(1, min(os.cpu_count() or 1, 64))
    per, steps = 4, 150                     # ~10-25 s per core: 4 envs x 150 env-steps at ~40-60 env-steps/s/core
    jobs = [(list(range(c * per, (c + 1) * per)), n_total, steps, solver) for c in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(cpu_baseline_worker, jobs)
    wall = time.perf_counter() - t0
    return {"value": cores * per * steps / wall, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{cores * per} envs x {steps} env-steps of the same workload (oracle/liborc.so, scalar f64 C, "
                      f"one process per core, solver={'newton' if solver else 'pgs-20'}), wall {wall:.1f} s incl. process start"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--pgs-iters", type=int, default=20)
    ap.add_argument("--solver", choices=["pgs", "newton"], default="newton")
    ap.add_argument("--newton-iters", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only for testing the multi-rank path on a one-GPU box together with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="testing aid: every rank uses cuda:0")
    ap.add_argument("--render", default="", help="HxW: also render depth images of the 4 zed/wrist cameras every step (BASELINE configs[4]); off by default")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the simulation path has no CPU fallback")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    from av_aloha_amd import _ffi
    from av_aloha_amd.build import build_hip
    from av_aloha_amd.sim import load_blob
    build_hip()
    N = args.envs_per_gpu
    n_total = N * world
    from av_aloha_amd.dist import gather_episode_stats, shard_ids
    ids = shard_ids(rank, world, N)                     # contiguous shard, global env ids
    blob, _ = load_blob("slot_insertion", 3)
    h = _ffi.Handle(blob, N, local, _ffi.AVSIM_IO_DEVICE)
    L = h.L
    stream = torch.cuda.current_stream()
    h.check(L.avsim_set_stream(h.h, stream.cuda_stream))
    for name, v in (("pgs_iters", args.pgs_iters), ("solver", 1 if args.solver == "newton" else 0), ("newton_iters", args.newton_iters),
                    ("export_contacts", 0), ("kernel_timing", 1)):
        h.check(L.avsim_set_option(h.h, name.encode(), float(v)))

    dev = torch.device("cuda", local)
    total = args.warmup + args.steps
    period = min(total, EPISODE_LEN)
    # synthetic inputs, resident in HBM before timing: one action tensor per step of an episode
    acts = torch.empty((period, N, 23), dtype=torch.float64, device=dev)
    for t in range(period):
        acts[t] = torch.from_numpy(scripted_actions(ids, n_total, t)).to(dev)
    obj = torch.from_numpy(object_poses(ids).reshape(N, 14)).to(dev)
    agent = torch.empty((N, 21), dtype=torch.float64, device=dev)
    reward = torch.empty((N,), dtype=torch.int32, device=dev)
    success = torch.empty((N,), dtype=torch.uint8, device=dev)
    ret = torch.zeros((N,), dtype=torch.float32, device=dev)
    succ_any = torch.zeros((N,), dtype=torch.int32, device=dev)

    rH = rW = 0
    depth = None
    r_events = []
    if args.render:
        rH, rW = (int(x) for x in args.render.lower().split("x"))
        _, man = load_blob("slot_insertion", 3)
        cam_ids = np.array([man["camera_names"].index(c) for c in ("zed_cam_left", "zed_cam_right", "wrist_cam_left", "wrist_cam_right")], dtype=np.int32)
        depth = torch.empty((N, 4, rH, rW), dtype=torch.float32, device=dev)

    def do_render(timed):
        if depth is None:
            return
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        h.check(L.avsim_render_depth(h.h, cam_ids.ctypes.data, 4, rH, rW, depth.data_ptr()))
        if timed:
            e1.record()
            r_events.append((e0, e1))

    def do_step(t):
        k = t % EPISODE_LEN
        if k == 0:
            h.check(L.avsim_reset(h.h, None, obj.data_ptr()))
            ret.zero_()
            succ_any.zero_()
        h.check(L.avsim_step_cartesian(h.h, acts[k % period].data_ptr(), _ffi.IK_DLS, 20, agent.data_ptr(),
                                       reward.data_ptr(), success.data_ptr()))
        ret.add_(reward.to(torch.float32))
        torch.maximum(succ_any, success.to(torch.int32), out=succ_any)

    for t in range(args.warmup):
        do_step(t)
        do_render(False)
    h.check(L.avsim_kernel_time(h.h, 1, None, None))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.warmup, total):
        do_step(t)
        do_render(True)
    # end-of-rollout exchange: one all-gather (RCCL) of (return f32, success i32) per env
    all_ret, all_succ = gather_episode_stats(ret, succ_any, dist)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k_ms, k_n = C.c_double(0), C.c_int64(0)
    h.check(L.avsim_kernel_time(h.h, 1, C.byref(k_ms), C.byref(k_n)))
    diag = torch.empty((N, 4), dtype=torch.int32, device=dev)
    h.check(L.avsim_get_diag(h.h, diag.data_ptr()))
    torch.cuda.synchronize()
    diag = diag.cpu().numpy()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total * args.steps / elapsed
        k_avg_s = (k_ms.value / max(1, k_n.value)) * 1e-3
        achieved = ALGO_BYTES_PER_ENV_STEP * N / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(f"k_phys_bytes_per_launch_N{N}")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec (whole node) at 4096 parallel envs, SlotInsertion-3Arms",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "gym_guided_vision/SlotInsertion-3Arms-v0 (BASELINE configs[1] at the metric's 4096 envs): "
                                   "23-D Cartesian action -> DLS IK on 3 arms -> 20 substeps (dt 0.002) + agent_pos + reward/success, no render",
                       "envs_per_gpu": N, "envs_total": n_total, "substeps_per_step": 20, "solver": args.solver, "pgs_iters": args.pgs_iters, "newton_iters": args.newton_iters,
                       "noslip_iters": 3, "lanes_per_env": 64, "episode_len": EPISODE_LEN,
                       "physics_substeps_per_s": value * 20,
                       "overflow_envs": int((diag[:, 2] != 0).sum()), "nan_envs": int((diag[:, 3] & 1).sum()),
                       "newton_iters_per_substep": float(((diag[:, 3] >> 16) & 0xfff).mean()) / 20.0, "newton_iters_max": int(((diag[:, 3] >> 28) & 0xf).max()),
                       "mean_ncon": float(diag[:, 0].mean()), "mean_nefc": float(diag[:, 1].mean()),
                       "gathered_envs": int(all_ret.numel()), "mean_return": float(all_ret.mean().item()),
                       "success_rate": float(all_succ.to(torch.float32).mean().item())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_phys<float>", "kernel_avg_ms": k_avg_s * 1e3, "kernel_launches": int(k_n.value),
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * N,
                         "valu_achieved_tflops_est": ALGO_FLOPS_PER_ENV_STEP * N / k_avg_s / 1e12 if k_avg_s > 0 else 0.0,
                         "valu_peak_tflops": VALU_PEAK_TFLOPS,
                         "note": "state stays in LDS across the 20 substeps, so HBM sees ~1 KB per env-step; the kernel is "
                                 "VALU/LDS-latency bound (SURVEY 8d), the HBM fraction is reported because the contract asks for it"},
        }
        if depth is not None:
            r_ms = sum(a.elapsed_time(b) for a, b in r_events) / max(1, len(r_events))
            r_bytes = depth.numel() * 4
            out["config"]["workload"] = out["config"]["workload"].replace("BASELINE configs[1]", "BASELINE configs[4]").replace(
                "no render", f"depth render of zed_cam_left/right + wrist_cam_left/right at {rH}x{rW} f32 every step")
            out["roofline_physics"] = out["roofline"]
            out["roofline"] = {"bound": "hbm", "achieved": r_bytes / (r_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": r_bytes / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "k_render_depth (+ k_render_geoms + forward k_phys)", "kernel_avg_ms": r_ms, "kernel_launches": len(r_events),
                               "algorithmic_bytes_per_launch": r_bytes,
                               "note": "4 B per pixel written once; the ray casting against the convex hulls is VALU work, so the kernel sits "
                                       "far below the HBM roof (see DESIGN.md)"}
            out["config"]["hit_fraction"] = float((depth < 29.9).float().mean().item())
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n_total, 1 if args.solver == "newton" else 0)
        elif world > 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    h.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
