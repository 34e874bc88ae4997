"""Order in which a frame's rays are rendered and dealt to GPUs: a 2-D Morton walk of their directions.

Rays are independent, so the order is free; 64 consecutive rays of the walk form a compact ~8x8 pixel patch, which is what
the kNN tiles and the hash-grid gathers want (occnerf_amd/network.py), and 256 consecutive rays the ~16x16 patch that
occnerf_amd/parallel.py deals to a rank.  No host synchronisation.  Rays on the GPU: three small kernels + one radix sort
(csrc/rays.hip `occnerf_ray_order`, deterministic -- a free-view orbit has a new camera, hence a new order, every frame);
rays on the host (frames handed over as pinned host tensors, the CPU tests): the same construction in plain torch ops.
The two agree up to rounding in the projection (a handful of neighbouring rays exchanged); every rank of a sharded render
computes the walk on the same kind of device from the same frame, so the ranks agree exactly.
"""
import torch


def ray_patch_order(rays_d):
    """Permutation that walks the rays along a 2-D Morton curve of their directions (projected on the plane normal
    to the mean direction).  Stable sort: ranks that compute it from the same frame get the same walk."""
    if rays_d.is_cuda:
        from . import ops
        d = rays_d if rays_d.dtype == torch.float32 else rays_d.float()
        return ops.ray_order(d if d.stride(-1) == 1 else d.contiguous())
    return _order_fp32(rays_d.float())


def _order_fp32(rays_d):
    return torch.argsort(_keys_fp32(rays_d), stable=True)


def _keys_fp32(rays_d):
    d = rays_d / rays_d.norm(dim=1, keepdim=True).clamp_min(1e-20)
    m = d.mean(dim=0)
    axis = torch.zeros(3, device=d.device, dtype=d.dtype).scatter_(0, m.abs().argmin().view(1), 1.0)
    e1 = torch.linalg.cross(m, axis)
    e1 = e1 / e1.norm().clamp_min(1e-20)
    e2 = torch.linalg.cross(m, e1)
    e2 = e2 / e2.norm().clamp_min(1e-20)
    uv = torch.stack([d @ e1, d @ e2], dim=1)
    lo = uv.min(dim=0, keepdim=True).values
    span = (uv.max(dim=0, keepdim=True).values - lo).max().clamp_min(1e-20)
    q = ((uv - lo) / span * 65535.0).long().clamp_(0, 65535)

    def spread(x):                       # 16 bits -> every second bit
        x = (x | (x << 8)) & 0x00FF00FF
        x = (x | (x << 4)) & 0x0F0F0F0F
        x = (x | (x << 2)) & 0x33333333
        x = (x | (x << 1)) & 0x55555555
        return x
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1)
