"""Scene sharding across ranks (one process per GPU).  Scenes are fully independent on the forward
path (no cross-scene op anywhere), so ranks take contiguous blocks of the global
scene list and exchange nothing on the data path; the only collective is the MAX-over-ranks of the
wall time used for reporting (and, for a training step, the gradient all-reduce: see DESIGN.md)."""
import torch
import torch.distributed as dist


def scene_shard(global_batch, rank, world):
    """Contiguous block [lo, hi) of the global scene list owned by `rank`; ragged tails go to the
    lowest ranks so every scene is owned exactly once."""
    if world < 1 or not (0 <= rank < world) or global_batch < 0:
        raise ValueError('bad shard arguments')
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, device=None):
    """Wall-clock of a step = slowest rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_volumes(local_volumes, device=None):
    """Optional: assemble the per-rank TSDF volumes [b_local,1,R,R,R] on every rank (evaluation
    tooling; not on the timed path).  Ragged shards are handled with all_gather_object-free padding."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_volumes
    world = dist.get_world_size()
    n = torch.tensor([local_volumes.shape[0]], dtype=torch.int64, device=local_volumes.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    mx = int(max(int(c.item()) for c in counts))
    pad = torch.zeros((mx,) + tuple(local_volumes.shape[1:]), dtype=local_volumes.dtype, device=local_volumes.device)
    pad[:local_volumes.shape[0]] = local_volumes
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:int(c.item())] for p, c in zip(parts, counts)], 0)
