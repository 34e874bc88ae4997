"""Device-side ray generation for a frame.

The reference's datasets build the ray batch with numpy on the CPU for every frame and ship it to the GPU
(tpose.py:155-182, freeview.py:190-218: get_rays_from_KRT + rays_intersect_3d_bbox, 262 144 pixels x 32 B at
512 x 512).  Here the per-pixel work is one HIP kernel (ops.gen_rays) and the compaction by `ray_mask` is a
device nonzero + gather; only the camera (K, E) and the skeleton bbox cross PCIe, and one integer (the
number of kept rays) comes back because the output shapes depend on it.
"""
import torch

from . import ops


def frame_rays(K, E, H, W, bbox_min, bbox_max, device):
    """-> dict with the ray keys of the reference's frame dict, as device tensors:
    rays[2,R,3], near[R,1], far[R,1], ray_mask[H*W] (bool)."""
    rays8, mask8 = ops.gen_rays(K, E, H, W, bbox_min, bbox_max, device)
    mask = mask8.bool()
    kept = rays8[mask]                                         # pixel order, like ray_o[mask_at_box]
    return {'rays': torch.stack([kept[:, 0:3], kept[:, 3:6]], 0),
            'near': kept[:, 6:7].contiguous(), 'far': kept[:, 7:8].contiguous(),
            'ray_mask': mask, 'img_width': W, 'img_height': H}
