"""Per-phase cycles of the physics kernel on bench.py's own workload at several points of the episode (debug aid; run on the GPU box):
what the kernel spends more on at step 100 than at step 20 of config 2.  usage: prof_phases_bench.py [config] [steps,...]"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 2
marks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "10,20,50,100,200,300").split(",")]
args = types.SimpleNamespace(pgs_iters=20, solver="newton", newton_iters=0, option=["profile_phases=1"])
torch.cuda.set_device(0)
job = bench.Workload(args, cfg_id, 4096, 0, 1, 0, False, max(marks), torch)
names = ["kin", "crb", "rne", "smooth", "collide", "rows", "solve", "euler"]
nn = ["init", "grad", "hess", "chol", "search", "final", "noslip", "backsub"]
out = torch.zeros((4096, 26), dtype=torch.int64, device="cuda:0")
for t in range(max(marks)):
    job.step(t)
    if t + 1 in marks:
        job.h.check(job.L.avsim_get_phase_cycles(job.h.h, out.data_ptr()))
        o = out.cpu().numpy().astype(np.float64)
        d = job.diag.cpu().numpy()
        tot = o[:, :8].sum(1) / 20
        print(f"step {t + 1:4d}: total {tot.mean():8.0f}  " + " ".join(f"{n} {v:6.0f}" for n, v in zip(names, o[:, :8].mean(0) / 20)))
        print("           newton " + " ".join(f"{n} {v:6.0f}" for n, v in zip(nn, o[:, 10:18].mean(0) / 20)) +
              f"  iters/substep {((d[:, 3] >> 16) & 0xfff).mean() / 20:.2f}  broad {o[:, 8].mean() / 21:.0f} narrow {o[:, 9].mean() / 21:.0f}")
        print("           per-env total p50/p90/p99/max", np.percentile(tot, [50, 90, 99, 100]).round(0), " ncon", np.percentile(d[:, 0], [50, 90, 99, 100]),
              " nefc", np.percentile(d[:, 1], [50, 90, 99, 100]), " probes", (o[:, 18:26].mean(0) / 20).round(0))
job.close()
