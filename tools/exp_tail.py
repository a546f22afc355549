"""Config 4's tail (HookPackage-2Arms random walk, 4096 envs): how long is a launch against what list scheduling of the per-env costs would give?
Per step: every env's cycles (profile_phases), the launch's wall time (kernel_timing), and three simulated makespans on S = 2048 wave slots:
greedy with the order the kernel uses (cost of the PREVIOUS step, most expensive first), greedy with this step's true costs (oracle order), and the
two lower bounds sum / S and max.  usage: python tools/exp_tail.py [N] [opt=val ...]   (GPU box)"""
import os, sys, heapq, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from av_aloha_amd.sim import BatchedSim
from av_aloha_amd import workloads as W
from test_oracle_physics import model_dict

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
opts = {"profile_phases": 1, "export_contacts": 0, "kernel_timing": 1}
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("="); opts[k] = float(v)
task, arms = os.environ.get("TASK", "hook_package"), int(os.environ.get("ARMS", "2"))
sim = BatchedSim(task, arms, N, options=opts)
md = model_dict(task, arms)
seed = {"hook_package": 3000, "sew_needle": 2000}.get(task, 1000)
sim.reset(W.object_poses(task, np.arange(N), seed))
nj = 21 if arms == 3 else 14
T0, T = 30, 20
acts = W.walk_actions(md["qpos_home"], md["act_ctrlrange"], np.arange(N), T0 + T, nj, seed)


def greedy(costs, order, S):
    h = [0.0] * S
    heapq.heapify(h)
    end = 0.0
    for i in order:
        t = heapq.heappop(h) + costs[i]
        end = max(end, t)
        heapq.heappush(h, t)
    return end


def ktime():
    ms, n = C.c_double(0), C.c_int64(0)
    sim.h.check(sim.h.L.avsim_kernel_time(sim.h.h, 1, C.byref(ms), C.byref(n)))
    return ms.value / max(1, n.value)

prev = None
rows = []
S = 2048
for t in range(T0 + T):
    sim.step(acts[t])
    out = np.zeros((N, 26), dtype=np.int64)
    sim.h.check(sim.h.L.avsim_get_phase_cycles(sim.h.h, out.ctypes.data))
    tot = out[:, :8].sum(1).astype(np.float64)
    k_ms = ktime()
    if t >= T0 and prev is not None:
        cyc = k_ms * 1e-3 * 2.4e9
        rows.append((cyc, greedy(tot, np.argsort(-prev), S), greedy(tot, np.argsort(-tot), S), tot.sum() / S, tot.max(),
                     np.corrcoef(prev, tot)[0, 1], np.mean(np.argsort(-prev)[:N // 100][:, None] == np.argsort(-tot)[:N // 100][None]).item() * (N // 100)))
    prev = tot
r = np.array(rows)
print(f"{task}-{arms}arms {N} envs, {T} steps (Mcycles at 2.4 GHz; profile probes on):")
print("  launch (HIP events)            %.2f" % (r[:, 0].mean() / 1e6))
print("  greedy, previous step's order  %.2f" % (r[:, 1].mean() / 1e6))
print("  greedy, this step's true order %.2f" % (r[:, 2].mean() / 1e6))
print("  sum / 2048 slots               %.2f" % (r[:, 3].mean() / 1e6))
print("  slowest env                    %.2f" % (r[:, 4].mean() / 1e6))
print("  corr(prev cost, cost) %.3f; of this step's slowest 1 %% of envs, fraction that was in the previous step's slowest 1 %%: %.2f" % (r[:, 5].mean(), r[:, 6].mean()))

# ---- what a ROLLOUT launch (every wave steps its env through all K steps before it takes another) would be as long as: list scheduling of the per-env TOTALS over K steps ----
K = int(os.environ.get("ROLLOUT_K", "0"))
if K:
    sim.reset(W.object_poses(task, np.arange(N), seed))
    acts = W.walk_actions(md["qpos_home"], md["act_ctrlrange"], np.arange(N), K, nj, seed)
    per = np.zeros((K, N))
    launch = np.zeros(K)
    for t in range(K):
        sim.step(acts[t])
        out = np.zeros((N, 26), dtype=np.int64)
        sim.h.check(sim.h.L.avsim_get_phase_cycles(sim.h.h, out.ctypes.data))
        per[t] = out[:, :8].sum(1)
        launch[t] = ktime() * 1e-3 * 2.4e9
    tot = per.sum(0)
    print(f"rollout of {K} steps from the reset: sum of the per-step launches {launch.sum() / 1e6:.1f} M cycles; per-env totals: mean {tot.mean() / 1e6:.1f} max {tot.max() / 1e6:.1f} M;")
    print(f"  list scheduling of the totals on {S} slots (most expensive first): {greedy(tot, np.argsort(-tot), S) / 1e6:.1f} M; in index order: {greedy(tot, np.arange(N), S) / 1e6:.1f} M; balanced {tot.sum() / S / 1e6:.1f} M")
    for kk in (10, 25, 50):
        if kk < K:
            c = per[:K // kk * kk].reshape(-1, kk, N).sum(1)
            print(f"  chunks of {kk} steps: sum over chunks of greedy (previous chunk's order) {sum(greedy(c[i], np.argsort(-c[max(i - 1, 0)]), S) for i in range(len(c))) / 1e6:.1f} M")
