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


def _bench(extra, dump, nproc):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--config", "4", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--dump", dump] + extra
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--backend", "gloo", "--share-gpu"] + common
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
