"""bench.py's multi-rank path on the one GPU of the test box: two ranks (gloo rendezvous, both on cuda:0 -- the RCCL run itself
needs the 8-GPU node the driver owns) shard 8192 HookPackage-2Arms envs contiguously (BASELINE configs[3]), all-gather the
per-env (return f32, success i32), and every env's result equals the one a single rank computes for the same global env id."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(extra, dump, nproc, port="29533"):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--config", "4", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--dump", dump] + extra
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--backend", "gloo", "--share-gpu"] + common
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[-1])


def test_two_ranks_gather_what_one_rank_computes(tmp_path):
    d2, d1 = str(tmp_path / "two.npz"), str(tmp_path / "one.npz")
    two = _bench(["--envs-per-gpu", "4096"], d2, 2)
    assert two["n_gpus"] == 2 and two["config"]["gathered_envs"] == 8192 and two["config"]["envs_total"] == 8192
    assert two["scaling"] == "weak" and two["cpu_baseline"] is None and two["config"]["nan_envs"] == 0
    one = _bench(["--envs-per-gpu", "8192"], d1, 1)
    assert one["config"]["gathered_envs"] == 8192
    a, b = np.load(d2), np.load(d1)
    assert a["ret"].dtype == np.float32 and a["succ"].dtype == np.int32
    for k in ("ret", "succ", "agent_sum"):
        assert a[k].shape == (8192,) and np.array_equal(a[k], b[k]), k
    assert np.isfinite(a["agent_sum"]).all() and np.ptp(a["agent_sum"]) > 0      # per-env random walks: the rows differ


def test_eight_ranks_keep_the_rank_order_of_the_gathered_vector(tmp_path):
    """The 8-GPU form of BASELINE configs[3] as a dry run on one GPU: eight ranks x 512 envs (gloo rendezvous, all on cuda:0).  The
    gathered vector is ordered by rank = by global env id: entry i is the env a single rank computes as its env i."""
    d8, d1 = str(tmp_path / "eight.npz"), str(tmp_path / "one.npz")
    eight = _bench(["--envs-per-gpu", "512"], d8, 8, port="29541")
    assert eight["n_gpus"] == 8 and eight["n_ranks_seen"] == 8 and eight["config"]["gathered_envs"] == 4096 and eight["scaling"] == "weak"
    assert eight["is_headline_metric"] is False and "HookPackage" in eight["metric"]
    one = _bench(["--envs-per-gpu", "4096"], d1, 1)
    assert one["n_ranks_seen"] == 1
    a, b = np.load(d8), np.load(d1)
    for k in ("ret", "succ", "agent_sum"):
        assert a[k].shape == (4096,) and np.array_equal(a[k], b[k]), k
    # strong-scaling flag: a fixed total split over the ranks
    st = _bench(["--envs-total", "1024"], str(tmp_path / "s.npz"), 2, port="29547")
    assert st["scaling"] == "strong" and st["config"]["envs_per_gpu"] == 512 and st["config"]["envs_total"] == 1024


def test_rccl_collectives_execute_for_a_world_of_one_rank(tmp_path):
    """The RCCL call path on the one GPU of the test box (two ranks cannot share a GPU under RCCL): `bench.py --dist-always` initialises
    torch.distributed with backend nccl (= RCCL) for a world of one rank and runs its barrier, the max-over-ranks all-reduce of the
    elapsed time and the end-of-rollout all-gather of (return, success) through it; the gathered vector equals the plain single-process
    run's."""
    dn, d1 = str(tmp_path / "rccl.npz"), str(tmp_path / "plain.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--dist-always", "--config", "4", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--envs-per-gpu", "1024", "--dump", dn]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-3000:]
    r = json.loads(lines[-1])
    assert r["n_ranks_seen"] == 1 and r["config"]["gathered_envs"] == 1024
    plain = _bench(["--envs-per-gpu", "1024"], d1, 1)
    assert plain["config"]["gathered_envs"] == 1024
    a, b = np.load(dn), np.load(d1)
    for k in ("ret", "succ", "agent_sum"):
        assert np.array_equal(a[k], b[k]), k


def test_bench_starts_its_own_ranks_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2` with no torch.distributed.run in front (the shape of the driver's N = 1 command): bench.py starts the
    two ranks itself, one process per GPU (here --share-gpu / gloo: both on the one GPU of the test box), and the line says so."""
    d2 = str(tmp_path / "self.npz")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--config", "4", "--steps", "4", "--warmup", "1",
           "--no-cpu-baseline", "--envs-per-gpu", "256", "--dump", d2]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == r["n_ranks_seen"] == 2 and r["config"]["envs_total"] == 512 and r["config"]["gathered_envs"] == 512
    assert abs(r["value_per_gpu"] * 2 - r["value"]) < 1e-6 * r["value"]
    # without --share-gpu the one-GPU box must refuse, loudly, instead of printing n_gpus: 1
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    import torch
    if torch.cuda.device_count() < 2:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert out.returncode != 0 and "needs 2 GPUs" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
