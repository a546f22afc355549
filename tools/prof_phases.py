"""Per-phase cycle breakdown of the physics kernel (debug aid; run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim import BatchedSim
from test_oracle_physics import OBJ, home_action, model_dict
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
opts = {"pgs_iters": 20, "profile_phases": 1, "export_contacts": 0}
task, arms = os.environ.get("TASK", "slot_insertion"), int(os.environ.get("ARMS", "3"))
for a in sys.argv[2:]:
    k, v = a.split("="); opts[k] = float(v)
sim = BatchedSim(task, arms, N, options=opts, f64=bool(int(os.environ.get("F64", "0"))))
md = model_dict(task, arms)
if task == "slot_insertion":
    sim.reset(np.repeat(OBJ[None], N, 0))
    a = np.repeat(home_action(md)[None], N, 0)
    for _ in range(3):
        sim.step(a)
else:       # the random-walk workload of bench.py --config 4 (av_aloha_amd/workloads.py), 30 steps in
    from av_aloha_amd import workloads as W
    seed = {"hook_package": 3000, "sew_needle": 2000}.get(task, 1000)
    sim.reset(W.object_poses(task, np.arange(N), seed))
    nj = 21 if arms == 3 else 14
    acts = W.walk_actions(md["qpos_home"], md["act_ctrlrange"], np.arange(N), 30, nj, seed)
    for t in range(30):
        sim.step(acts[t])
out = np.zeros((N, 26), dtype=np.int64)
sim.h.check(sim.h.L.avsim_get_phase_cycles(sim.h.h, out.ctypes.data))
names = ["kinematics", "crb", "rne", "smooth", "collide", "rows", "solve", "euler"]
print("broad/narrow per collide call:", out[:, 8].mean() / 21, out[:, 9].mean() / 21)
m = out[:, :8].mean(0) / 20
if m[1] == 0 and m[2] == 0:      # the default build runs kinematics .. smooth as ONE out-of-line function (AVS_NO_SPLIT_PRE builds them apart)
    names[0] = "kin..smooth"
    print("(kinematics, CRB, RNE and the smooth forces are one out-of-line function in this build: their sum is in the first line; build with AVSIM_EXTRA_FLAGS=-DAVS_NO_SPLIT_PRE for the four)")
print("cycles per substep per wave (mean over envs):")
for n, v in zip(names, m):
    print(f"  {n:10s} {v:10.0f}  {100 * v / m.sum():5.1f}%")
print(f"  total      {m.sum():10.0f}  -> {m.sum() * 20 / 2.4e6:.2f} ms per env-step per wave at 2.4 GHz;  diag {sim.diag()[0]}")
nn = ["init", "grad", "hess", "chol", "search", "final", "noslip", "backsub"]
mn = out[:, 10:18].mean(0) / 20
print("inside solve (Newton):", "  ".join(f"{n} {v:.0f}" for n, v in zip(nn, mn)), f"  iterations/substep {((sim.diag()[:, 3] >> 16) & 0xfff).mean() / 20:.2f}")
tot = out[:, :8].sum(1) / 20
print("per-env total cycles/substep percentiles 50/90/99/max:", np.percentile(tot, [50, 90, 99, 100]).round(0), " noslip 50/90/99/max:", np.percentile(out[:, 16] / 20, [50, 90, 99, 100]).round(0),
      " narrow 50/90/99/max:", np.percentile(out[:, 9] / 21, [50, 90, 99, 100]).round(0))
# (when the noslip pass runs per tree -- noslip_trees, the default where every contact touches one tree -- slots 1 / 2 / 3 count passes refused at
# entry / given up on a sliding contact / done instead: tools/prof_noslip_trees.py; pass noslip_trees=0 on the command line for pgs_groups' probes)
# noslip probes (pgs_groups): group steps, cycles gathering a sliding contact's block, cycles in sliding-contact branches, sliding steps,
# multiplier iterations, cycles in the pass; then box-box and other narrow-phase cycles (collide)
print("probe slots per substep [nstep, gather cyc, sliding cyc, sliding steps, multiplier its, noslip cyc, boxbox cyc, narrow cyc]:", (out[:, 18:26].mean(0) / 20).round(1))
d = sim.diag()
print("ncon percentiles", np.percentile(d[:, 0], [50, 90, 99, 100]), "newton max iters", np.percentile((d[:, 3] >> 28) & 0xf, [50, 90, 99, 100]))
# the most expensive envs of the launch on their own
top = np.argsort(-tot)[: max(1, N // 100)]
print("slowest 1 % of the envs: total", tot[top].mean().round(0), " phases", (out[top, :8].mean(0) / 20).round(0), " noslip", (out[top, 16].mean() / 20).round(0),
      " narrow", (out[top, 9].mean() / 21).round(0), " ncon", d[top, 0].mean().round(1), " probe slots", (out[top, 18:26].mean(0) / 20).round(1))
print("   their Newton split:", "  ".join(f"{n} {v:.0f}" for n, v in zip(nn, out[top, 10:18].mean(0) / 20)), f"  iterations/substep {((d[top, 3] >> 16) & 0xfff).mean() / 20:.2f}")
