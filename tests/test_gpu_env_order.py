"""The launch order of the envs (option "order_envs": workgroups take the envs by their cost in the previous step, most
expensive first) is a scheduling matter only: every env is stepped by one wavefront from its own state, so the results must be
bit-identical with the order on and off, on a workload whose per-env costs differ widely (HookPackage random walk)."""
import numpy as np
import pytest

from av_aloha_amd import workloads as W
from test_oracle_physics import model_dict

pytestmark = pytest.mark.gpu


def _run(order):
    from av_aloha_amd.sim import BatchedSim
    cfg = W.CONFIGS[4]
    n, T = 300, 12                       # not a multiple of the envs per workgroup: the last block is partly empty
    ids = np.arange(n)
    md = model_dict(cfg["task"], 2)
    sim = BatchedSim(cfg["task"], 2, n, options={"solver": 1, "order_envs": order})
    sim.reset(W.object_poses(cfg["task"], ids, cfg["seed"]))
    acts = W.walk_actions(md["qpos_home"], md["act_ctrlrange"], ids, T, 14, cfg["seed"])
    out = []
    for t in range(T):
        ap, rw, su = sim.step(acts[t])
        out.append((ap.copy(), rw.copy(), su.copy()))
    q, v, c, _ = sim.get_state()
    d = sim.diag().copy()
    sim.close()
    return out, q.copy(), v.copy(), d


def test_env_order_does_not_change_results():
    a, qa, va, da = _run(1)
    b, qb, vb, db = _run(0)
    assert np.array_equal(qa, qb) and np.array_equal(va, vb)
    for (ap1, r1, s1), (ap2, r2, s2) in zip(a, b):
        assert np.array_equal(ap1, ap2) and np.array_equal(r1, r2) and np.array_equal(s1, s2)
    assert np.array_equal(da[:, :3], db[:, :3])
    assert len(np.unique(da[:, 0])) > 1          # the envs do differ (contact counts), i.e. the order was not the identity
