"""The seeded model and frame plumbing shared by bench.py, the tools, smoke() and the tests: a `Network` carrying the
seeded checkpoint (occnerf_amd/checkpoint.py -- there is no network for trained weights or datasets), and a synthetic
frame as device tensors / pinned host tensors with the keys `Network.forward` takes."""
import numpy as np
import torch

from . import checkpoint
from .config import _finish, default_cfg, set_cfg

FRAME_KEYS = ['rays', 'near', 'far', 'bgcolor', 'dst_Rs', 'dst_Ts', 'cnl_gtfms',
              'motion_weights_priors', 'cnl_bbox_min_xyz', 'cnl_bbox_max_xyz', 'cnl_bbox_scale_xyz',
              'dst_posevec']


def build_network(seed=0, amplify=False, S=128, non_rigid=False, device='cuda:0', mlp_precision='fp32', state_dict=None):
    """Network with the seeded checkpoint loaded (strict), on `device`, in eval mode.  amplify: the "trained-like"
    variant (O(1) hash features and visibility counts)."""
    from .network import Network
    cfg = default_cfg()
    _finish(cfg)
    cfg.N_samples = S
    cfg.perturb = 0.
    cfg.ignore_non_rigid_motions = not non_rigid
    cfg.smpl_model = 'synthetic'
    cfg.mlp_precision = mlp_precision
    set_cfg(cfg)
    net = Network()
    net.generate_neural_points(np.zeros(10, 'float32'))
    if state_dict is None:
        state_dict = checkpoint.make_state_dict(net.point_base.detach().numpy(), float(net.bound), seed=seed,
                                                amplify=amplify)
    net.load_state_dict(state_dict, strict=True)
    return net.to(device).deploy_mlps_to_secondary_gpus().eval()


def frame_to_device(frame, device):
    """A frame dict (numpy, occnerf_amd.synth.make_frame) as tensors on `device`."""
    return {k: torch.from_numpy(np.ascontiguousarray(frame[k])).to(device) for k in FRAME_KEYS}


def host_frame(frame):
    """The frame as a dataset would hand it over: pinned host tensors; the three float[3] constants stay on the host
    (the kernels take them by value)."""
    return {k: torch.from_numpy(np.ascontiguousarray(frame[k])).pin_memory() for k in FRAME_KEYS}


def patch_ray_selection(frame, rng, n_patches=6, size=32, full=False):
    """Ray indices (into the frame's bbox-hitting ray list) of `n_patches` random size x size pixel patches -- the training
    batch of the reference (configs/default.yaml:147-150 `patch`, core/data/human_nerf/train.py sample_patch_rays): a patch
    counts when more than half of its pixels hit the bbox (`full`: all of them, which fixes the batch at n_patches x size^2
    rays -- the benchmark's 6 x 32 x 32 = 6 144); patches may overlap, as the reference's do."""
    H = W = int(frame['img_width'])
    index_of = -np.ones(H * W, dtype=np.int64)
    index_of[np.nonzero(np.asarray(frame['ray_mask']).reshape(-1))[0]] = np.arange(frame['rays'].shape[1])
    sel, tries = [], 0
    while len(sel) < n_patches:
        tries += 1
        if tries > 100000:
            raise RuntimeError('patch_ray_selection: no patch of this size fits the rays of the frame')
        y, x = rng.randint(0, H - size), rng.randint(0, W - size)
        pix = (np.arange(y, y + size)[:, None] * W + np.arange(x, x + size)[None, :]).ravel()
        rays = index_of[pix]
        if (rays >= 0).all() if full else (rays >= 0).mean() > 0.5:
            sel.append(rays[rays >= 0])
    return np.concatenate(sel)

