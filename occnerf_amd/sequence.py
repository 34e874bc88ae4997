"""Rendering a sequence of frames: the loop of the reference's run.py (`_freeview` run.py:66-119, `run_movement`
:137-186) around the sharded renderer, and the synthetic frame source that stands in for the reference's datasets.

`render_sequence` is what `run.py` executes and what `bench.py`'s `movement` leg times: frame t+1 is generated and
submitted (this rank's share of its rays rendered, the gather of (rgb, alpha, depth) started) before frame t's gather
is waited for and its image handed to the consumer, so the collective and the image assembly of one frame sit under the
kernels of the next (occnerf_amd/parallel.py).
"""
import numpy as np
import torch

from . import synth
from .rays import frame_rays

EXCLUDE_KEYS_TO_GPU = ['frame_name', 'img_width', 'img_height', 'ray_mask',
                       'camera_K', 'camera_E', 'dst_bbox_min', 'dst_bbox_max']
HOST_KEYS = ('bgcolor', 'cnl_bbox_min_xyz', 'cnl_bbox_max_xyz', 'cnl_bbox_scale_xyz')   # float[3]: taken by value


class SyntheticFrames:
    """Frame source with the reference's per-frame dict (leading batch dimension included, as a DataLoader with
    batch_size=1 adds and run.py strips, run.py:85-86).  tpose: 1 frame, zero pose; freeview: `total_frames` orbit frames of
    one seeded pose; movement: `total_frames` frames of a seeded smooth pose walk from one camera; allview: the freeview pose
    from the 23 cameras of a ZJU-MoCap-like ring (allview.py:69); progress: up to 300 frames of the walk with the camera
    moving along (create_dataset.py:40-42 `maxframes = 300` under evaluate)."""

    def __init__(self, data_type, img_size=512, render_frames=100, bgcolor=(255., 255., 255.), device_rays=True,
                 freeview_frame_idx=0, frame_range=None):
        self.data_type, self.img_size = data_type, int(img_size)
        self.bgcolor, self.device_rays, self.freeview_frame_idx = bgcolor, bool(device_rays), int(freeview_frame_idx)
        self.avg_betas = np.zeros(10, dtype='float32')
        self.total_frames = {'tpose': 1, 'allview': 23, 'progress': min(300, int(render_frames))}.get(
            data_type, int(render_frames))
        self.dataset = self                     # run.py reads test_loader.dataset.avg_betas
        # frame_range=(first, count): iterate over a window of the sequence only (bench.py: 8 consecutive orbit frames)
        self.frame_range = None if frame_range is None else (int(frame_range[0]), int(frame_range[1]))

    def __len__(self):
        return self.total_frames if self.frame_range is None else self.frame_range[1]

    def pose(self, idx):
        if self.data_type == 'tpose':
            return None
        if self.data_type in ('movement', 'progress'):
            return synth.movement_pose(idx, self.total_frames)
        return synth.seeded_pose(self.freeview_frame_idx + 1)

    def frame(self, idx):
        """Frame idx as numpy (occnerf_amd.synth.make_frame)."""
        return synth.make_frame(
            img_size=self.img_size, pose72=self.pose(idx),
            orbit_frame=idx if self.data_type in ('freeview', 'allview', 'progress') else 0,
            orbit_period=max(self.total_frames, 1), bgcolor=self.bgcolor, with_rays=not self.device_rays)

    def __iter__(self):
        lo, n = (0, self.total_frames) if self.frame_range is None else self.frame_range
        for idx in range(lo, lo + n):
            batch = {}
            for k, v in self.frame(idx).items():
                batch[k] = torch.as_tensor(np.asarray(v))[None] if not np.isscalar(v) else v
            batch['frame_name'] = [f'frame_{idx:06d}']
            yield batch


def frames_to_device(loader, data_type, device='cuda'):
    """The loader's frames as (renderer inputs, camera key, bookkeeping): tensors on the device (asynchronously), the three
    float[3] constants by value, the ray batch generated on the GPU when the loader hands over a camera instead of rays."""
    for idx, batch in enumerate(loader):
        batch = {k: (v[0] if torch.is_tensor(v) or isinstance(v, list) else v) for k, v in batch.items()}
        data = {k: (v if k in HOST_KEYS else v.to(device, non_blocking=True)) for k, v in batch.items()
                if k not in EXCLUDE_KEYS_TO_GPU and torch.is_tensor(v)}
        if 'rays' not in batch:          # the ray batch is generated on the GPU (occnerf_amd/rays.py)
            fr = frame_rays(batch['camera_K'].numpy(), batch['camera_E'].numpy(), int(batch['img_height']),
                            int(batch['img_width']), batch['dst_bbox_min'].numpy(), batch['dst_bbox_max'].numpy(), device)
            data.update(rays=fr['rays'], near=fr['near'], far=fr['far'])
            ray_index = torch.nonzero(fr['ray_mask']).squeeze(1)
        else:                            # host mask: the index list is formed on the host, no device round trip
            ray_index = torch.nonzero(batch['ray_mask']).squeeze(1).to(device, non_blocking=True)
        # a movement sequence is shot by one camera: the Morton walk of the rays (shard plan, render order) is computed
        # once per ray count
        key = ('movement', int(ray_index.numel())) if data_type == 'movement' else None
        yield data, key, {'idx': idx, 'ray_index': ray_index, 'width': int(batch['img_width']),
                          'height': int(batch['img_height'])}


def render_sequence(renderer, loader, data_type, iter_val, on_frame, device='cuda'):
    """Every frame of `loader` through `renderer` (a ShardedRenderer) with one frame of lag; on rank 0
    `on_frame(out, meta)` receives each frame's gathered {'rgb','alpha','depth'} in frame order.  -> frames rendered."""
    prev, n = None, 0

    def deliver(pending, meta):
        out = renderer.finish(pending)
        if out is not None:                           # ranks > 0: their rays went to rank 0
            on_frame(out, meta)

    with torch.no_grad():
        for data, key, meta in frames_to_device(loader, data_type, device):
            cur = renderer.submit(data, iter_val=iter_val, ray_order_key=key)
            if prev is not None:
                deliver(*prev)
            prev = (cur, meta)
            n += 1
        if prev is not None:
            deliver(*prev)
    return n
