"""Multi-GPU glue: envs shard contiguously over ranks, one process per GPU; the only collective on the path is the
end-of-rollout all-gather of per-env (return, success).  Backend-agnostic (nccl = RCCL on the GPU
node, gloo in the CPU tests)."""
import numpy as np


def shard_ids(rank: int, world: int, envs_per_rank: int) -> np.ndarray:
    """Global env ids owned by `rank`: env i -> rank i // envs_per_rank.  RNG streams are keyed by these global
    ids, so results do not depend on the number of ranks."""
    assert 0 <= rank < world
    return np.arange(rank * envs_per_rank, (rank + 1) * envs_per_rank)


def _all_gather(t, dist):
    import torch
    dev = t.device
    if dist.get_backend() == "gloo" and t.is_cuda:      # gloo moves host memory only (CPU tests, one-GPU debugging)
        t = t.cpu()
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.cat(out, dim=0).to(dev)


def gather_episode_stats(ret, success, dist=None, always=False):
    """All-gather of the local envs' (return f32, success i32) into rank-ordered [world*N] tensors on every rank: one
    collective per dtype, sendcount N words each.  `ret` / `success` are torch tensors on the rank's
    device; `dist` is torch.distributed (None = single process).  always: run the collective for a world of one rank too (the
    RCCL call path on a one-GPU box)."""
    import torch
    ret = ret.to(torch.float32).contiguous()
    success = success.to(torch.int32).contiguous()
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not always):
        return ret.clone(), success.clone()
    return _all_gather(ret, dist), _all_gather(success, dist)
