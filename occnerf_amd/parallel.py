"""Multi-GPU: rays shard, one process per GPU, one gather per frame.

Rays are independent (the only cross-sample dependency is the scan inside one ray), so the
`ray_mask`-compacted ray list is cut into `world_size` contiguous blocks; every rank holds the
full (61 MiB) model, renders its block with no data-path collective, and the `[R/N, 5]`
(rgb, alpha, depth) blocks are gathered on rank 0 over RCCL/xGMI (<= 5.2 MB per 512^2 frame:
latency-bound, one collective).  This replaces the reference's nn.DataParallel over *samples*
with its per-call weight broadcast (network.py:68-72,142-146), it does not mirror it.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rays, world_size):
    """[lo, hi) of every rank's contiguous block; sizes differ by at most one."""
    base, rem = divmod(int(n_rays), int(world_size))
    bounds, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < rem else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def shard_indices(n_rays, rank, world_size, chunk=4096):
    """Ray indices of `rank` when rays are dealt in chunks of `chunk` consecutive rays (chunk c -> rank
    c % world_size).  Contiguous blocks of a pixel-ordered ray list are horizontal image bands with very
    different amounts of body in them; dealing chunks balances the work, and a chunk is still a compact
    set of pixels for the kNN tiles.  Every rank gets the same number of rays +- one chunk."""
    idx = torch.arange(int(n_rays))
    return idx[(idx // int(chunk)) % int(world_size) == int(rank)]


def shard_frame(data, rank, world_size):
    """Slice the per-ray entries of a frame dict (rays[2,R,3], near/far[R,1]) for `rank`."""
    R = data['rays'].shape[1]
    lo, hi = shard_bounds(R, world_size)[rank]
    out = dict(data)
    out['rays'] = data['rays'][:, lo:hi]
    out['near'], out['far'] = data['near'][lo:hi], data['far'][lo:hi]
    return out, (lo, hi)


def gather_rays(block, n_rays, dst=0, group=None):
    """Gather per-rank `[r_i, C]` blocks into `[n_rays, C]` on rank `dst` (None elsewhere).
    Blocks are padded to the largest shard so one fixed-size gather suffices."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return block
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = shard_bounds(n_rays, world)
    width = max(hi - lo for lo, hi in bounds)
    padded = block.new_zeros((width,) + tuple(block.shape[1:]))
    padded[:block.shape[0]] = block
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:hi - lo] for b, (lo, hi) in zip(bufs, bounds)], 0)


def render_frame_sharded(net, data, iter_val=1e7, group=None, chunk=4096):
    """Render this rank's share of `data` (chunks of `chunk` rays dealt round-robin, see shard_indices) and
    gather (rgb, alpha, depth) on rank 0 -- the path's only collective.  Returns the full-frame dict in the
    caller's ray order on rank 0 and None on the other ranks."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    R = data['rays'].shape[1]
    dev = data['rays'].device
    mine = shard_indices(R, rank, world, chunk).to(dev)
    local = dict(data)
    local['rays'] = data['rays'][:, mine]
    local['near'], local['far'] = data['near'][mine], data['far'][mine]
    out = net(**local, iter_val=iter_val)
    packed = torch.cat([out['rgb'], out['alpha'][:, None], out['depth'][:, None]], dim=1)
    if world == 1:
        full = packed
    else:
        sizes = [int(shard_indices(R, r, world, chunk).numel()) for r in range(world)]
        width = max(sizes)
        padded = packed.new_zeros((width, packed.shape[1]))
        padded[:packed.shape[0]] = packed
        bufs = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
        dist.gather(padded, bufs, dst=0, group=group)
        if rank != 0:
            return None
        full = packed.new_empty((R, packed.shape[1]))
        for r in range(world):
            full[shard_indices(R, r, world, chunk).to(dev)] = bufs[r][:sizes[r]]
    return {'rgb': full[:, :3], 'alpha': full[:, 3], 'depth': full[:, 4]}
