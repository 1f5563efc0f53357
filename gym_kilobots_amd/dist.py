"""Multi-GPU plumbing: envs are independent (one b2World per env object in the reference,
gym_kilobots/envs/kilobots_env.py:45), so the env axis is sharded in contiguous blocks, one process
per GPU, with NO traffic during stepping.  The only collective is the gather of per-env episode
returns (RCCL all-gather over xGMI on GPUs, gloo in the CPU tests)."""
import torch


def env_shard(total_envs, rank, world_size):
    """[lo, hi) of the contiguous env block owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world_size):
        raise ValueError('rank %d outside world of size %d' % (rank, world_size))
    base, rem = divmod(total_envs, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_returns(local_returns, dist=None):
    """All-gather per-env returns [E_local] -> [E_total] in global env order on every rank.

    `dist` is torch.distributed (initialised) or None for a single process.  Shards may have
    unequal sizes (total_envs % world_size != 0): they are padded to the largest shard."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_returns.clone()
    world = dist.get_world_size()
    n = torch.tensor([local_returns.numel()], device=local_returns.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    padded = torch.zeros(m, device=local_returns.device, dtype=local_returns.dtype)
    padded[:local_returns.numel()] = local_returns
    out = torch.empty(world * m, device=local_returns.device, dtype=local_returns.dtype)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)])
