"""Multi-GPU glue: envs shard contiguously over ranks, one process per GPU; the only collective on the path is the
end-of-rollout all-gather of per-env (return, success).  Backend-agnostic (nccl = RCCL on the GPU
node, gloo in the CPU tests)."""
import numpy as np


def shard_ids(rank: int, world: int, envs_per_rank: int) -> np.ndarray:
    """Global env ids owned by `rank`: env i -> rank i // envs_per_rank.  RNG streams are keyed by these global
    ids, so results do not depend on the number of ranks."""
    assert 0 <= rank < world
    return np.arange(rank * envs_per_rank, (rank + 1) * envs_per_rank)


def gather_episode_stats(ret, success, dist=None):
    """All-gather (return f32, success i32) of the local envs into rank-ordered [world*N] tensors on every rank.
    `ret` / `success` are torch tensors on the rank's device; `dist` is torch.distributed (None = single process)."""
    import torch
    pack = torch.stack([ret.to(torch.float32), success.to(torch.float32)], dim=1).contiguous()
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return pack[:, 0].clone(), pack[:, 1].to(torch.int32)
    dev = pack.device
    if dist.get_backend() == "gloo" and pack.is_cuda:      # gloo moves host memory only (CPU tests, one-GPU debugging)
        pack = pack.cpu()
    out = [torch.empty_like(pack) for _ in range(dist.get_world_size())]
    dist.all_gather(out, pack)
    allp = torch.cat(out, dim=0).to(dev)
    return allp[:, 0].contiguous(), allp[:, 1].to(torch.int32)
