#!/usr/bin/env python
"""Headline benchmark: env-steps/sec of the batched gym_guided_vision simulation (BASELINE.json metric).

One "step" = one env step of every env on the rank.  `--config` selects the workload (SURVEY.md 8(d) numbering; the
generators live in av_aloha_amd/workloads.py):

  2 (default) 4096 SlotInsertion-3Arms envs: 23-D Cartesian action -> damped-least-squares IK on the three arms
              (k_cart_ctrl) -> 20 physics substeps + agent_pos + reward/success (k_phys).  This is BASELINE configs[1] at the
              metric's env count, the configuration the metric is quoted on.
  3           4096 SewNeedle-3Arms envs, contact-rich: scripted reach - grasp - lift of the needle through the reference's
              controllers (GradIK, GradIK, DiffIK) (BASELINE configs[2])
  4           4096 HookPackage-2Arms envs per GPU, 14-D joint-space random walk, sharded over ranks (BASELINE configs[3])
  5           config 2 + depth images of the two ZED and the two wrist cameras at 480x640 every step (BASELINE configs[4])

Inputs (the action tensors) and all state are resident in HBM before the timed region starts; nothing crosses PCIe inside it.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python bench.py --gpus 8 ...            (no launcher: starts the 8 ranks itself through torch.distributed.run, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit/..., plus
  "roofline":     algorithmic HBM bytes of the dominant kernel / its mean launch time (HIP events on the launch stream),
                  against the 8 TB/s HBM peak.  The physics kernel keeps its state in LDS for 20 substeps, so the fraction is
                  tiny by design; the VALU view (counted flops of the oracle's instrumented build, profiles/flop_counts.json)
                  is reported next to it.
  "cpu_baseline": the CPU oracle (oracle/liborc.so, a scalar f64 C restatement) timed on this host's cores on a bounded
                  sample of the same workload.
The default line (one GPU, config 2, f32) also carries short side runs on the same GPU, outside the timed region of `value`:
  "f64_value"         the same workload with AVSIM_F64_PHYSICS (the reference's arithmetic: MuJoCo computes in double)
  "value_episode300"  reset + one whole 300-step episode of config 2
  "config3_value", "config4_value"   the contact-rich configurations (whole 250-step grasp script; 100 steps of the random walk)
  "config5_value" / "config5"        config 2 + depth images of four cameras at 480x640 every step (BASELINE configs[4]): env-steps/s, the
                                     image kernel's ms per launch (HIP events, avsim_render_kernel_time) and its fraction of the HBM write roof
`roofline.bound` is "valu+latency" for the physics kernel: `frac` = its own floating-point work (profiles/kernel_flops.json) per launch time against
the f32 vector peak; the HBM view of the contract (algorithmic bytes per launch / launch time against 8 TB/s) is kept as hbm_achieved / hbm_frac.
`scaling` is "weak" with --envs-per-gpu (default 4096 on every GPU) and "strong" with --envs-total; `n_ranks_seen` is
torch.distributed's world size as the process group reports it.
"""
import argparse
import ctypes as C
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md
VALU_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}
RENDER_DEFAULT = "480x640"


def algorithmic_bytes(nq, nv, nj_action, action_words, f64):
    """state resident on chip across the 20 substeps: read qpos + qvel + warmstart + action, write qpos + qvel +
    warmstart + agent_pos + reward + success; 4-byte words (state terms 8-byte in the f64 variant)."""
    sw = 8 if f64 else 4
    state = nq + nv + nv
    return state * sw + action_words * 4 + state * sw + (nj_action + 2) * 4


def flop_counts():
    """Counted flops per env-step of the oracle's instrumented build (tools/count_flops.py -> profiles/flop_counts.json)."""
    p = os.path.join(ROOT, "profiles", "flop_counts.json")
    try:
        return json.load(open(p))
    except Exception:
        return {}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return platform.processor() or "unknown"



def kernel_flops(cfg_id, f64):
    """The kernel's own floating-point work per env-step (SQ instruction counters, tools/prof_flops.sh -> profiles/kernel_flops.json)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "kernel_flops.json"))).get(f"config{2 if cfg_id == 5 else cfg_id}" + ("_f64" if f64 else ""), {})
    except Exception:
        return {}


N_SIMD, CLOCK_HZ = 1024, 2.4e9          # 256 CUs x 4 SIMDs, peak engine clock (MI355X_MICROARCH.md)


def valu_issue_frac(cfg_id, f64, k_avg_ms):
    """How busy the VALU issue ports are: VALU wave-instructions per launch (SQ_INSTS_VALU, profiles/kernel_flops.json) x 2 cycles per
    wave64 instruction on a 32-lane-wide SIMD [guide: cdna_hip_programming.md] / (SIMDs x launch time x clock).  1 - this is what
    better latency hiding could still return; only recorded for the configuration whose instruction mix was profiled (config 2, f32)."""
    if f64 or cfg_id not in (2, 5) or k_avg_ms <= 0:
        return None
    try:
        mix = json.load(open(os.path.join(ROOT, "profiles", "kernel_flops.json"))).get("config2_instruction_mix_per_launch", {})
        return mix["SQ_INSTS_VALU"] * 2.0 / (N_SIMD * k_avg_ms * 1e-3 * CLOCK_HZ)
    except Exception:
        return None


def mujoco_row(cfg_id, home, budget_s=20.0):
    """SURVEY 8(d)(ii) "MuJoCo CPU (optional)": if and only if `import mujoco` works on this host, real mj_step x 20 per env-step on the
    model rebuilt from the build's own blob (av_aloha_amd/compiler/emit_mjcf.py -> tests/mj_env.py; no reference file), one thread, the
    joint-space home action, a bounded sample.  -> (importable, row or None)."""
    try:
        import mujoco  # noqa: F401
    except Exception:
        return False, None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from av_aloha_amd import workloads as W
        from mj_env import MjEnv
        cfg = W.CONFIGS[cfg_id]
        e = MjEnv(cfg["task"], cfg["arms"], hulls="full")
        e.reset(W.object_poses(cfg["task"], [0], cfg["seed"])[0])
        a = np.concatenate([e.blob["ctrl_home"][:6], [1.0], e.blob["ctrl_home"][7:13], [1.0], e.blob["ctrl_home"][14:21]])
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s and n < 3000:
            a[0] = e.blob["ctrl_home"][0] + 0.1 * np.sin(0.05 * n)
            e.step(a)
            n += 1
        wall = time.perf_counter() - t0
        return True, {"value": n / wall, "unit": "env-steps/s", "cores": 1, "kind": "reference-engine", "mujoco_version": mujoco.__version__,
                      "sample": f"{n} env-steps of ONE env ({cfg['task']}, full mesh hulls, joint-space sway of one joint), mj_step x 20 + mj_step1 per env-step, {wall:.1f} s",
                      "note": "real MuJoCo on the MJCF emitted from the build's own model blob; x host cores for a process-per-core figure"}
    except Exception as ex:            # importable but the emitted model did not load / step: say so, do not hide it
        return True, {"error": repr(ex)[:300]}


def valu_roofline(cfg_id, f64, N, k_avg_ms, k_n, abytes):
    """roofline block of a k_phys run: the kernel's own flops per launch / mean launch time against the vector peak of the dtype, the
    HBM view next to it; `frac` stays numeric (the HBM fraction) when profiles/kernel_flops.json lacks the configuration."""
    dtype = "f64" if f64 else "f32"
    kf = kernel_flops(cfg_id, f64)
    k_s = k_avg_ms * 1e-3
    kflops = kf.get("flops_per_env_step")
    tf = kflops * N / k_s / 1e12 if (kflops and k_s > 0) else None
    hbm = abytes * N / k_s / 1e9 if k_s > 0 else 0.0
    peak = VALU_PEAK_TFLOPS[dtype]
    if tf is None:
        return {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS, "traffic": None,
                "kernel": f"k_phys<{'double' if f64 else 'float'}>", "kernel_avg_ms": k_avg_ms, "kernel_launches": int(k_n),
                "algorithmic_bytes_per_launch": abytes * N, "note": "no kernel flop count for this configuration in profiles/kernel_flops.json: HBM view only"}
    return {"bound": "valu+latency", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
            "hbm_achieved": hbm, "hbm_peak": HBM_PEAK_GBS, "hbm_unit": "GB/s", "hbm_frac": hbm / HBM_PEAK_GBS,
            "kernel": f"k_phys<{'double' if f64 else 'float'}>", "kernel_avg_ms": k_avg_ms, "kernel_launches": int(k_n),
            "algorithmic_bytes_per_launch": abytes * N, "valu_flops_per_env_step_kernel": kflops,
            "valu_lane_utilisation": kf.get("lane_utilisation"), "valu_issue_frac": valu_issue_frac(cfg_id, f64, k_avg_ms), "kernel_flops_source": kf.get("source"),
            "peak_source": "FP64 vector 78.6 TF / FP32 vector 157.3 TF (AMD MI355X specification; MI355X_MICROARCH.md lists the matrix peaks only)"}


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle, kind "port"): every core steps its own envs of the same workload
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline_worker(args):
    cfg_id, ids, n_total, steps, solver, home = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from av_aloha_amd import workloads as W
    from orc_env import OrcEnv
    from orc_ffi import dp
    cfg = W.CONFIGS[cfg_id]
    poses = W.object_poses(cfg["task"], ids, cfg["seed"])
    envs = []
    for k in range(len(ids)):
        e = OrcEnv(cfg["task"], cfg["arms"])
        e.d.pgs_iters = 20
        e.d.solver = solver
        e.reset(poses[k])
        envs.append(e)
    if cfg["action"] == "cartesian_reference":
        acts = list(W.grasp_lift_targets(home, poses[:, 1, :3] + np.array([0.0, 0.0, 0.01]), sway=0.02))[:steps]
    elif cfg["action"] == "joint":
        from av_aloha_amd.compiler.compile import read_blob
        md = read_blob(os.path.join(ROOT, "models", f"{cfg['task']}_{cfg['arms']}arms.avm"))
        acts = W.walk_actions(md["qpos_home"], md["act_ctrlrange"], ids, steps, 14, cfg["seed"]).astype(np.float64)
    a21 = np.zeros(21)
    t0 = time.perf_counter()
    for t in range(steps):
        if cfg["action"] == "cartesian_dls":
            a = W.sinusoid_actions(home, ids, n_total, t)
        elif cfg["action"] == "cartesian_reference":
            a = acts[t]
        for k, e in enumerate(envs):
            if cfg["action"] == "joint":
                e.env_step(acts[t, k])
            else:
                e.L.orc_cart_to_ctrl(e.dptr, dp(np.ascontiguousarray(a[k])), 1 if cfg["action"] == "cartesian_dls" else 0, dp(a21))
                e.env_step(a21)
    return time.perf_counter() - t0


def cpu_baseline(cfg_id, n_total, home, solver=1, budget_s=None):
    """The oracle on all host cores (kind "port").  Sample: SURVEY 8(d)'s CPU-baseline size -- min(N, 1024) envs x one whole 300-step
    episode -- when the host finishes it within `budget_s` (default 120 s of wall clock, AVSIM_CPU_BUDGET_S; estimated from the
    oracle's measured ~60 env-steps/s per core on config 2), otherwise the largest whole number of envs per core x 300 steps that does;
    the line says which."""
    import multiprocessing as mp
    from av_aloha_amd.build import build_oracle
    build_oracle()
    cores = max(1, min(os.cpu_count() or 1, 64))
    budget_s = float(os.environ.get("AVSIM_CPU_BUDGET_S", "120")) if budget_s is None else budget_s
    rate_core = {2: 55.0, 5: 55.0, 3: 14.0, 4: 45.0}.get(cfg_id, 40.0)        # env-steps/s per core observed in rounds 2-4 (EPYC 9575F)
    steps = 300 if cfg_id != 3 else 250                                       # one whole episode / config 3's whole script
    want = min(n_total, 1024)
    per = max(1, min((want + cores - 1) // cores, int(budget_s * rate_core / steps)))
    envs = min(want, per * cores)
    # contiguous global env ids, the GPU run's first `envs`
    jobs = [(cfg_id, list(range(c * per, min(envs, (c + 1) * per))), n_total, steps, solver, home) for c in range(cores) if c * per < envs]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(len(jobs)) as pool:
        busy = pool.map(cpu_baseline_worker, jobs)
    wall = time.perf_counter() - t0
    full = envs == want
    return {"value": envs * steps / wall, "unit": "env-steps/s", "cores": len(jobs), "kind": "port", "cpu_model": cpu_model(),
            "value_per_core": envs * steps / sum(busy) if sum(busy) > 0 else None,
            "sample": f"{envs} envs x {steps} env-steps of the same workload (config {cfg_id}: global env ids 0..{envs - 1}, the GPU run's seeds and actions; "
                      f"{'SURVEY 8(d) size min(N, 1024) x one whole episode' if full else f'bounded below SURVEY 8(d) min(N, 1024) = {want} envs by the {budget_s:.0f} s budget'}; "
                      f"oracle/liborc.so, scalar f64 C, one process per core on {len(jobs)} cores, solver={'newton' if solver else 'pgs-20'}), wall {wall:.1f} s incl. process start",
            "note": "a straightforward scalar restatement (dense rows, no sparsity, no SIMD): a reported baseline, not a tuned engine; the GPU / CPU ratio says nothing about kernel quality"}


class Workload:
    """One measurement configuration on this rank: handle, synthetic inputs resident in HBM, the per-step driver."""

    def __init__(self, args, cfg_id, N, rank, world, local, f64, total_steps, torch, render=""):
        from av_aloha_amd import _ffi
        from av_aloha_amd import workloads as W
        from av_aloha_amd.compiler.compile import read_blob
        from av_aloha_amd.dist import shard_ids
        from av_aloha_amd.sim import load_blob
        self.torch, self.cfg_id, self.N, self.f64 = torch, cfg_id, N, f64
        cfg = self.cfg = W.CONFIGS[cfg_id]
        n_total = self.n_total = N * world
        ids = shard_ids(rank, world, N)                     # contiguous shard, global env ids
        blob, man = load_blob(cfg["task"], cfg["arms"])
        h = self.h = _ffi.Handle(blob, N, local, _ffi.AVSIM_IO_DEVICE | (_ffi.AVSIM_F64_PHYSICS if f64 else 0))
        L = self.L = h.L
        h.check(L.avsim_set_stream(h.h, torch.cuda.current_stream().cuda_stream))
        opts = [("pgs_iters", args.pgs_iters), ("solver", 1 if args.solver == "newton" else 0), ("export_contacts", 0), ("kernel_timing", 1)]
        if args.newton_iters:
            opts.append(("newton_iters", args.newton_iters))
        for o in args.option:
            k, v = o.split("=")
            opts.append((k, float(v)))
        for name, v in opts:
            h.check(L.avsim_set_option(h.h, name.encode(), float(v)))
        dev = self.dev = torch.device("cuda", local)
        self.EPISODE_LEN = cfg["episode_len"]
        period = self.period = min(total_steps, self.EPISODE_LEN)
        # eef poses at the home joints (centre of the scripted Cartesian motions): FK on the device
        md = read_blob(os.path.join(ROOT, "models", f"{cfg['task']}_{cfg['arms']}arms.avm"))
        ch = np.asarray(md["ctrl_home"], dtype=np.float64)
        T_home = []
        for arm, sl in ((0, slice(0, 6)), (1, slice(7, 13)), (2, slice(14, 21))):
            q = torch.from_numpy(np.ascontiguousarray(ch[sl])[None]).to(dev)
            T = torch.empty((1, 16), dtype=torch.float64, device=dev)
            h.check(L.avsim_fk_jac(h.h, arm, 1, q.data_ptr(), T.data_ptr(), None))
            torch.cuda.synchronize()
            T_home.append(T.cpu().numpy())
        home = self.home = W.home_poses(T_home)
        # synthetic inputs, resident in HBM before timing: one action tensor per step of an episode
        poses = W.object_poses(cfg["task"], ids, cfg["seed"])
        self.obj = torch.from_numpy(poses.reshape(N, poses.shape[1] * 7)).to(dev)
        if cfg["action"] == "cartesian_dls":
            acts = torch.empty((period, N, 23), dtype=torch.float64, device=dev)
            for t in range(period):
                acts[t] = torch.from_numpy(W.sinusoid_actions(home, ids, n_total, t)).to(dev)
            self.ik_mode, self.nj = _ffi.IK_DLS, 21
        elif cfg["action"] == "cartesian_reference":
            gen = W.grasp_lift_targets(home, poses[:, 1, :3] + np.array([0.0, 0.0, 0.01]), sway=0.02)      # qpos order: wall, needle
            acts = torch.empty((period, N, 23), dtype=torch.float64, device=dev)
            for t, a in zip(range(period), gen):
                acts[t] = torch.from_numpy(a).to(dev)
            self.ik_mode, self.nj = _ffi.IK_REFERENCE, 21
        else:
            acts = torch.from_numpy(W.walk_actions(md["qpos_home"], md["act_ctrlrange"], ids, period, 14, cfg["seed"])).to(dev)
            self.ik_mode, self.nj = None, 14
        self.acts = acts
        nj = self.nj
        self.agent = torch.empty((N, nj), dtype=torch.float64, device=dev)
        self.reward = torch.empty((N,), dtype=torch.int32, device=dev)
        self.success = torch.empty((N,), dtype=torch.uint8, device=dev)
        self.ret = torch.zeros((N,), dtype=torch.float32, device=dev)
        self.succ_any = torch.zeros((N,), dtype=torch.uint8, device=dev)
        self.diverged = torch.zeros((N,), dtype=torch.int32, device=dev)
        self.diag = torch.empty((N, 4), dtype=torch.int32, device=dev)
        self.ncon_sum = torch.zeros((N,), dtype=torch.float64, device=dev)
        self.rich = torch.zeros((N,), dtype=torch.int32, device=dev)
        self.resets = 0
        self.rH = self.rW = 0
        self.depth = None
        self.r_events = []
        if render:
            self.rH, self.rW = (int(x) for x in render.lower().split("x"))
            self.cam_ids = np.array([man["camera_names"].index(c) for c in W.RENDER_CAMERAS], dtype=np.int32)
            self.depth = torch.empty((N, 4, self.rH, self.rW), dtype=torch.float32, device=dev)

    def render(self, timed):
        if self.depth is None:
            return
        torch = self.torch
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.h.check(self.L.avsim_render_depth(self.h.h, self.cam_ids.ctypes.data, 4, self.rH, self.rW, self.depth.data_ptr()))
        if timed:
            e1.record()
            self.r_events.append((e0, e1))

    def step(self, t):
        h, L, torch = self.h, self.L, self.torch
        k = t % self.EPISODE_LEN
        if k == 0:                                   # auto-reset at the episode boundary
            h.check(L.avsim_reset(h.h, None, self.obj.data_ptr()))
            self.ret.zero_()
            self.succ_any.zero_()
            self.resets += 1
        a = self.acts[k % self.period]
        if self.ik_mode is None:
            h.check(L.avsim_step(h.h, a.data_ptr(), 20, self.agent.data_ptr(), self.reward.data_ptr(), self.success.data_ptr()))
        else:
            h.check(L.avsim_step_cartesian(h.h, a.data_ptr(), self.ik_mode, 20, self.agent.data_ptr(), self.reward.data_ptr(), self.success.data_ptr()))
        # one elementwise kernel each (type promotion inside add_ / maximum, no temporaries)
        self.ret.add_(self.reward)
        torch.maximum(self.succ_any, self.success, out=self.succ_any)
        # per-step diagnostics stay on the device: divergence flags, contact counts (a few tiny elementwise kernels)
        h.check(L.avsim_get_diag(h.h, self.diag.data_ptr()))
        self.diverged.bitwise_or_(self.diag[:, 3])          # (bit 0 = diverged; masked where it is read)
        if self.cfg_id == 3:
            self.ncon_sum.add_(self.diag[:, 0].to(torch.float64))
            self.rich.add_((self.diag[:, 0] >= 8).to(torch.int32))

    def begin_timed(self):
        self.h.check(self.L.avsim_kernel_time(self.h.h, 1, None, None))
        self.diverged.zero_(); self.ncon_sum.zero_(); self.rich.zero_()
        self.resets = 0

    def kernel_time(self):
        k_ms, k_n = C.c_double(0), C.c_int64(0)
        self.h.check(self.L.avsim_kernel_time(self.h.h, 1, C.byref(k_ms), C.byref(k_n)))
        return k_ms.value, int(k_n.value)

    def close(self):
        self.h.close()


def side_run(args, torch, cfg_id, N, local, f64, warmup, steps, render=""):
    """A short run of another configuration / precision on this GPU, for the extra fields of the bench line (single rank):
    -> (env-steps/s, ms per step, mean k_phys launch ms, diagnostics; with `render` the depth images' time and write rate too)."""
    w = Workload(args, cfg_id, N, 0, 1, local, f64, warmup + steps, torch, render)
    for t in range(warmup):
        w.step(t)
        w.render(False)
    w.begin_timed()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(warmup, warmup + steps):
        w.step(t)
        w.render(True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    k_ms, k_n = w.kernel_time()
    dg = w.diag.cpu().numpy()
    info = {"value": N * steps / el, "ms_per_step": el / steps * 1e3, "steps": steps, "warmup": warmup, "kernel_avg_ms": k_ms / max(1, k_n),
            "mean_ncon": float(dg[:, 0].mean()), "mean_nefc": float(dg[:, 1].mean()), "nan_envs": int((w.diverged & 1).sum().item()),
            "resets_in_timed_region": w.resets, "mean_return": float(w.ret.mean().item()), "success_rate": float(w.succ_any.float().mean().item())}
    info["roofline"] = valu_roofline(cfg_id, f64, N, k_ms / max(1, k_n), k_n,
                                     algorithmic_bytes(w.h.nq, w.h.nv, w.nj, 23 if w.ik_mode is not None else 14, f64))
    if w.depth is not None:
        r_ms = sum(a.elapsed_time(b) for a, b in w.r_events) / max(1, len(w.r_events))
        r_bytes = w.depth.numel() * 4
        k_rd_ms, k_rd_n = C.c_double(0), C.c_int64(0)
        w.h.check(w.L.avsim_render_kernel_time(w.h.h, 1, C.byref(k_rd_ms), C.byref(k_rd_n)))
        info.update({"render": f"{w.rH}x{w.rW} f32 x 4 cameras", "render_call_ms": r_ms, "render_bytes_per_step": r_bytes,
                     "render_call_frac_of_hbm_write_roof": r_bytes / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "k_render_depth_ms": k_rd_ms.value / max(1, k_rd_n.value) if k_rd_n.value else None,
                     "k_render_depth_frac_of_hbm_write_roof": (r_bytes / (k_rd_ms.value / k_rd_n.value * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_rd_n.value else None,
                     "hit_fraction": float((w.depth < 29.9).float().mean().item())})
    w.close()
    return info


def launch_ranks(n, share_gpu):
    """Re-run this command line as n ranks of one node through torch.distributed.run (rendezvous on 127.0.0.1, a free port), the
    form the driver itself uses for N > 1.  Fails loudly when the node has fewer GPUs than ranks (unless --share-gpu)."""
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and not share_gpu:
        print(f"bench.py: --gpus {n} needs {n} GPUs on this node, {have} visible (--share-gpu runs every rank on cuda:0: a testing aid, not a measurement)", file=sys.stderr)
        return 2
    # --standalone: torch.distributed.run's own rendezvous store on a free port of 127.0.0.1 (no port picked here and raced for)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n}",
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="SURVEY.md 8(d) workload (2 = the metric's configuration)")
    ap.add_argument("--f64", action="store_true", help="run the physics in double precision (AVSIM_F64_PHYSICS): the precision trade next to the f32 product mode")
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU, help="weak scaling: this many envs on every GPU")
    ap.add_argument("--envs-total", type=int, default=0, help="strong scaling: this many envs in all, split evenly over the GPUs (overrides --envs-per-gpu)")
    ap.add_argument("--pgs-iters", type=int, default=20)
    ap.add_argument("--solver", choices=["pgs", "newton"], default="newton")
    ap.add_argument("--newton-iters", type=int, default=0, help="0 = the library default")
    ap.add_argument("--option", action="append", default=[], help="name=value passed to avsim_set_option (experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side runs (f64, whole episode, configs 3 and 4) that the default single-GPU config-2 line carries")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only for testing the multi-rank path on a one-GPU box together with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="testing aid: every rank uses cuda:0")
    ap.add_argument("--dist-always", action="store_true", help="testing aid: initialise torch.distributed and run the collectives for a world of ONE rank too (RCCL init, barrier, all-reduce and all-gather on a one-GPU box)")
    ap.add_argument("--render", default="", help="HxW: render depth images of the 4 zed/wrist cameras every step (config 5 default 480x640)")
    ap.add_argument("--dump", default="", help="testing aid: rank 0 saves the gathered per-env returns / successes to this .npz")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU) and pass their
        # exit code on; rank 0 of the child world prints the line
        raise SystemExit(launch_ranks(args.gpus, args.share_gpu))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the simulation path has no CPU fallback")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.share_gpu:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} (LOCAL_RANK {local}) has no GPU: {torch.cuda.device_count()} visible; --share-gpu puts every rank on cuda:0 (testing aid)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or args.dist_always:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    from av_aloha_amd.build import build_hip
    from av_aloha_amd.dist import gather_episode_stats
    # the library travels prebuilt; should its sources look newer on this box, ONE rank rebuilds it and the others wait
    if rank == 0:
        build_hip()
    if dist is not None:
        dist.barrier()
    if args.config == 5 and not args.render:
        args.render = RENDER_DEFAULT
    if args.envs_total:
        assert args.envs_total % world == 0, "--envs-total must be divisible by the number of GPUs"
        N, scaling = args.envs_total // world, "strong"
    else:
        N, scaling = args.envs_per_gpu, "weak"
    total = args.warmup + args.steps
    w = Workload(args, args.config, N, rank, world, local, args.f64, total, torch, args.render)
    cfg, h, n_total, nj, ik_mode, dev = w.cfg, w.h, w.n_total, w.nj, w.ik_mode, w.dev
    EPISODE_LEN = w.EPISODE_LEN

    for t in range(args.warmup):
        w.step(t)
        w.render(False)
    w.begin_timed()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.warmup, total):
        w.step(t)
        w.render(True)
    # end-of-rollout exchange: one all-gather (RCCL) of (return f32, success i32) per env; events around it on the
    # current stream (the collective is enqueued there), and this rank's own time up to it (before the barrier evens the ranks out)
    ev_g0, ev_g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev_g0.record()
    t_g0 = time.perf_counter()
    all_ret, all_succ = gather_episode_stats(w.ret, w.succ_any, dist, always=args.dist_always)
    ev_g1.record()
    torch.cuda.synchronize()
    t_own = time.perf_counter() - t0
    allgather_host_ms = (time.perf_counter() - t_g0) * 1e3
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    allgather_ms = ev_g0.elapsed_time(ev_g1)
    all_agent = None
    if args.dump:               # testing aid, outside the timed region: a per-env checksum of the final joint positions
        from av_aloha_amd.dist import _all_gather
        all_agent = w.agent.sum(dim=1).contiguous()
        if dist is not None:
            all_agent = _all_gather(all_agent, dist)
    k_ms, k_n = w.kernel_time()
    torch.cuda.synchronize()
    dg = w.diag.cpu().numpy()
    n_ranks_seen = 1
    rank_ms = [t_own / args.steps * 1e3]
    rank_kernel_ms = [k_ms / max(1, k_n)]
    rank_allgather_ms = [allgather_ms]
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        n_ranks_seen = dist.get_world_size()
        # per-rank figures so that a multi-GPU line explains itself: each rank's own ms per step (without the final barrier), its mean
        # k_phys launch, and the time of the all-gather as its events saw it
        from av_aloha_amd.dist import _all_gather
        pr = _all_gather(torch.tensor([[t_own / args.steps * 1e3, k_ms / max(1, k_n), allgather_ms]], dtype=torch.float64,
                                      device=dev if args.backend == "nccl" else "cpu"), dist).cpu().numpy()
        rank_ms, rank_kernel_ms, rank_allgather_ms = pr[:, 0].tolist(), pr[:, 1].tolist(), pr[:, 2].tolist()

    assert n_ranks_seen == world == args.gpus, (n_ranks_seen, world, args.gpus)
    if rank == 0:
        if args.dump:
            np.savez(args.dump, ret=all_ret.cpu().numpy(), succ=all_succ.cpu().numpy(), agent_sum=all_agent.cpu().numpy())
        dtype = "f64" if args.f64 else "f32"
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total * args.steps / elapsed
        k_avg_s = (k_ms / max(1, k_n)) * 1e-3
        # 1040 B (3 arms, Cartesian action), 1032 B (joint action); config 4 is stated there as 808 B for a model
        # without the parked camera arm -- this build simulates it as the reference does (env.py:394-395), so the formula's
        # figure for nq 37 / nv 35 is used
        abytes = algorithmic_bytes(h.nq, h.nv, nj, 23 if ik_mode is not None else 14, args.f64)
        achieved = abytes * N / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
        traffic, traffic_source = None, None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                traffic = tj.get(f"k_phys_bytes_per_launch_N{N}" + ("" if args.config in (2, 5) and not args.f64 else f"_config{args.config}{'_f64' if args.f64 else ''}"))
                if traffic is not None:
                    traffic_source = "profiles/hbm_traffic.json: " + tj.get("source", "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this command (tools/prof_traffic.sh), not collected in this run")
            except Exception:
                traffic = None
        ckey = f"config{2 if args.config == 5 else args.config}"
        fc = flop_counts().get(ckey, {})
        flops_step = fc.get("flops_per_env_step")
        kf = {}
        try:
            kf = json.load(open(os.path.join(ROOT, "profiles", "kernel_flops.json"))).get(ckey + ("_f64" if args.f64 else ""), {})
        except Exception:
            pass
        kflops_step = kf.get("flops_per_env_step")
        valu_k_tflops = kflops_step * N / k_avg_s / 1e12 if (kflops_step and k_avg_s > 0) else None
        headline = args.config == 2
        workload = {
            2: f"{cfg['gym_id']} (BASELINE configs[1] at the metric's 4096 envs): 23-D Cartesian action -> DLS IK on 3 arms -> 20 substeps (dt 0.002) + agent_pos + reward/success, no render",
            3: f"{cfg['gym_id']} (BASELINE configs[2]): scripted reach-grasp-lift of the needle, 23-D Cartesian action -> GradIK x2 + DiffIK -> 20 substeps + agent_pos + reward/success, contact-rich, no render",
            4: f"{cfg['gym_id']} (BASELINE configs[3], {N} envs per GPU): 14-D joint-space random walk -> 20 substeps + agent_pos + reward/success, RCCL all-gather of episode returns",
            5: f"{cfg['gym_id']} (BASELINE configs[4]): config 2 + depth render of zed_cam_left/right + wrist_cam_left/right at {w.rH}x{w.rW} f32 every step"}[args.config]
        out = {
            # the BASELINE metric string belongs to its own configuration only; the other workloads say what they are
            "metric": "env-steps/sec (whole node) at 4096 parallel envs, SlotInsertion-3Arms" if headline else
                      f"env-steps/sec (whole node), {cfg['gym_id'].split('/')[1]} workload of SURVEY 8(d) config {args.config} (not the headline metric)",
            "is_headline_metric": headline,
            "value": value, "value_per_gpu": value / world, "unit": "env-steps/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "multi_rank": {"allgather_ms": max(rank_allgather_ms), "allgather_ms_per_rank": rank_allgather_ms, "allgather_host_ms_rank0_incl_stream_drain": allgather_host_ms,
                           "allgather_bytes_per_rank": int(N * 8), "backend": (args.backend if dist is not None else None),
                           "ms_per_step_rank_min": min(rank_ms), "ms_per_step_rank_max": max(rank_ms), "ms_per_step_per_rank": rank_ms,
                           "k_phys_ms_per_rank": rank_kernel_ms,
                           "note": "per-rank wall clock per step up to the end of the rank's own all-gather (before the closing barrier); allgather_ms = HIP "
                                   "events around gather_episode_stats on the rank's stream (two collectives: f32 returns, i32 successes); world of one rank: a device copy"},
            "config": {"workload": workload,
                       # the f32 product mode is admissible through north_star's "stated FP tolerance": per-step joint trajectories against the f64
                       # oracle (the reference's arithmetic: MuJoCo mjtNum = double), asserted in tests/test_gpu_physics.py / test_gpu_bench_path.py
                       "fp_tolerance": ("f64 device mode: 1e-10 rad per env-step vs the f64 oracle (tests/test_gpu_physics.py)" if args.f64 else
                                        "f32 state and arithmetic: joint positions within 2e-5 rad of the f64 oracle over 25-30 env-steps (600 substeps) of this workload at 4096 envs "
                                        "(tests/test_gpu_physics.py, tests/test_gpu_bench_path.py); rewards and success flags exact there; the like-for-like f64 figure is f64_value"),
                       "survey_config": args.config, "envs_per_gpu": N, "envs_total": n_total, "substeps_per_step": 20, "solver": args.solver,
                       "pgs_iters": args.pgs_iters, "noslip_iters": 3, "lanes_per_env": 64, "episode_len": EPISODE_LEN,
                       "resets_in_timed_region": w.resets,
                       "physics_substeps_per_s": value * 20,
                       "overflow_envs": int((dg[:, 2] != 0).sum()), "nan_envs": int((w.diverged & 1).sum().item()),
                       "newton_iters_per_substep": float(((dg[:, 3] >> 16) & 0xfff).mean()) / 20.0, "newton_iters_max": int(((dg[:, 3] >> 28) & 0xf).max()),
                       "mean_ncon": float(w.ncon_sum.mean().item()) / args.steps if args.config == 3 else float(dg[:, 0].mean()),
                       "mean_nefc": float(dg[:, 1].mean()),
                       "nefc_percentiles_50_90_99_100": [float(x) for x in np.percentile(dg[:, 1], [50, 90, 99, 100])],
                       "ncon_percentiles_50_90_99_100": [float(x) for x in np.percentile(dg[:, 0], [50, 90, 99, 100])],
                       "envs_with_8_contacts_for_100_steps": float((w.rich >= 100).float().mean().item()) if args.config == 3 else None,
                       "gathered_envs": int(all_ret.numel()), "mean_return": float(all_ret.mean().item()),
                       "success_rate": float(all_succ.to(torch.float32).mean().item())},
            # what bounds k_phys is the issue rate and latency of one wave's dependent VALU / LDS chain (neither HBM nor
            # MFMA): `frac` is the kernel's own floating-point work against the vector peak; the HBM view the contract asks for
            # (algorithmic bytes per launch / launch time against 8 TB/s) is kept next to it as hbm_*
            "roofline": {**valu_roofline(args.config, args.f64, N, k_avg_s * 1e3, k_n, abytes),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_over_algorithmic": traffic / (abytes * N) if traffic else None,
                         "valu_flops_per_env_step_counted": flops_step,
                         "valu_achieved_tflops": flops_step * N / k_avg_s / 1e12 if (flops_step and k_avg_s > 0) else None,
                         "valu_peak_tflops": VALU_PEAK_TFLOPS[dtype],
                         "valu_frac": flops_step * N / k_avg_s / 1e12 / VALU_PEAK_TFLOPS[dtype] if (flops_step and k_avg_s > 0) else None,
                         # the kernel's OWN arithmetic (its sparse row windows and per-tree solves, not the oracle's dense rows):
                         # floating-point VALU instructions the SQ counted for k_phys x lanes active, per env-step
                         "valu_frac_kernel": valu_k_tflops / VALU_PEAK_TFLOPS[dtype] if valu_k_tflops is not None else None,
                         "note": "state stays in LDS across the 20 substeps, so HBM sees ~1 KB per env-step; the kernel is "
                                 "VALU/LDS-latency bound (SURVEY 8d): frac = valu_frac_kernel = achieved / peak with achieved from kernel_flops_source (a profile of this "
                                 "command, not a count made in this run) / this run's HIP-event launch time; hbm_frac is reported because the contract asks for it; "
                                 "valu_frac uses the flops counted in the oracle's instrumented dense build (profiles/flop_counts.json); "
                                 "valu_issue_frac = VALU wave-instructions x 2 cycles / (1024 SIMDs x launch time x 2.4 GHz): the share of issue slots in use"},
        }
        if w.depth is not None:
            r_ms = sum(a.elapsed_time(b) for a, b in w.r_events) / max(1, len(w.r_events))
            r_bytes = w.depth.numel() * 4
            out["roofline_physics"] = out["roofline"]
            out["roofline"] = {"bound": "hbm", "achieved": r_bytes / (r_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": r_bytes / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "k_render_depth (+ k_render_geoms + forward k_phys)", "kernel_avg_ms": r_ms, "kernel_launches": len(w.r_events),
                               "algorithmic_bytes_per_launch": r_bytes,
                               "note": "4 B per pixel written once (SURVEY 8d: config 5 is HBM-write-bound)"}
            out["config"]["hit_fraction"] = float((w.depth < 29.9).float().mean().item())
    home = w.home
    w.close()
    if rank == 0:
        # Side runs next to the headline line (single GPU, default configuration): the same workload at the reference's precision,
        # one whole 300-step episode with its auto-reset, and the contact-rich configurations 3 and 4
        if headline and world == 1 and not args.f64 and not args.no_extras:
            f = side_run(args, torch, 2, N, local, True, 5, 20)
            out["f64_value"] = f["value"]
            out["f64"] = {**f, "dtype": "f64", "note": "AVSIM_F64_PHYSICS: the arithmetic of the reference (MuJoCo mjtNum = double), same workload, same flags; "
                                                       "its own roofline block (k_phys<double> against the FP64 vector peak)"}
            out["roofline_f64"] = f["roofline"]
            # small-N behaviour: BASELINE configs[1] as written (1024 envs: half a round of the 2048 resident wave slots) and the latency of
            # ONE env-step of ONE env (the reference's documented use is 1 - 10 envs, README.md:163-169: below ~2048 envs throughput is N / latency)
            v1k = side_run(args, torch, 2, 1024, local, False, 5, 20)
            out["value_1024"] = v1k["value"]
            out["config2_1024"] = {**v1k, "note": "BASELINE configs[1] as written: 1024 parallel SlotInsertion-3Arms envs, physics + diff-IK, no render"}
            one = side_run(args, torch, 2, 1, local, False, 5, 40)
            out["latency_ms_1env"] = one["ms_per_step"]
            out["config2_1env"] = {**one, "note": "ONE env: wall clock of one env-step (IK launch + 20 substeps in one k_phys launch + outputs), the latency floor of the design"}
            e = side_run(args, torch, 2, N, local, False, 0, 300)
            out["value_episode300"] = e["value"]
            out["episode300"] = {**e, "note": "steps 0..299 of config 2: reset + one whole 300-step episode (data_collection_scripts/constants.py:23-58)"}
            c3 = side_run(args, torch, 3, N, local, False, 0, 250)
            out["config3_value"] = c3["value"]
            out["config3"] = {**c3, "note": "SewNeedle-3Arms scripted reach-grasp-lift, GradIK x2 + DiffIK, one whole 250-step script (BASELINE configs[2])"}
            c4 = side_run(args, torch, 4, N, local, False, 10, 100)
            out["config4_value"] = c4["value"]
            out["config4"] = {**c4, "note": "HookPackage-2Arms 14-D joint random walk, this GPU's 4096-env shard (BASELINE configs[3])"}
            c5 = side_run(args, torch, 5, N, local, False, 2, 10, RENDER_DEFAULT)
            out["config5_value"] = c5["value"]
            out["config5"] = {**c5, "note": "config 2 + depth images of zed_cam_left/right + wrist_cam_left/right at 480x640 f32 every step (BASELINE configs[4]); 4.9 MB of depth per env-step: HBM-write bound by construction (SURVEY 8d)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.config, n_total, home, 1 if args.solver == "newton" else 0)
        elif world > 1:
            out["cpu_baseline"] = None
        importable, mj = mujoco_row(args.config, home) if world == 1 else (None, None)
        out["mujoco_importable"] = importable
        if mj is not None:
            out["mujoco_cpu"] = mj
        # the driver keeps the line's last 2000 characters: the figures that matter once more, compactly, as the LAST key
        r4 = lambda x: None if x is None else float(f"{x:.4g}")
        out["summary"] = {k: r4(out.get(k)) for k in ("value", "f64_value", "value_1024", "latency_ms_1env", "value_episode300", "config3_value", "config4_value", "config5_value")}
        out["summary"].update({"k_phys_ms": r4(out.get("roofline_physics", out["roofline"]).get("kernel_avg_ms")), "frac": r4(out.get("roofline_physics", out["roofline"]).get("frac")),
                               "valu_issue_frac": r4(out.get("roofline_physics", out["roofline"]).get("valu_issue_frac")),
                               "k_render_depth_ms": r4((out.get("config5") or {}).get("k_render_depth_ms")),
                               "allgather_ms": r4(out["multi_rank"]["allgather_ms"]), "cpu_baseline": r4((out.get("cpu_baseline") or {}).get("value")),
                               "mujoco_importable": importable, "mujoco_cpu": r4((mj or {}).get("value"))})
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
