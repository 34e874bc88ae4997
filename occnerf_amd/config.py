"""Configuration node + defaults for the hot path.

The reference builds a YACS singleton at import time from sys.argv
(configs/config.py:36-72).  The keys the renderer reads are kept with the same names and
default values (configs/default.yaml, configs/occnerf/zju_mocap/387/occnerf.yaml); the
node itself is a small attribute dict with the merge order defaults <- yaml file <- KEY VALUE
list.  `configs/__init__.py` at the repo root exposes `cfg`/`args` under the reference's
import path for the run.py / train.py entry points.
"""
import argparse
import ast
import copy
import os

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return dict.get(self, k, default)

    @staticmethod
    def wrap(d):
        if isinstance(d, dict) and not isinstance(d, CfgNode):
            return CfgNode({k: CfgNode.wrap(v) for k, v in d.items()})
        return d

    def merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = CfgNode.wrap(copy.deepcopy(v))
        return self

    def merge_from_file(self, path):
        with open(path) as f:
            return self.merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, 'opts must be KEY VALUE pairs'
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                node = node[p]
            try:
                val = ast.literal_eval(val)
            except (ValueError, SyntaxError):
                pass
            node[parts[-1]] = val
        return self

    def clone(self):
        return CfgNode.wrap(copy.deepcopy(dict(self)))


_DEFAULTS = {
    'category': 'occnerf', 'task': 'zju_mocap', 'subject': 'p387', 'experiment': 'occnerf',
    'resume': False, 'eval_iter': 10000000, 'render_folder_name': '',
    'ignore_non_rigid_motions': False, 'render_skip': 1, 'render_frames': 100, 'num_workers': 4,
    'network_module': 'core.nets.occnerf.network',
    'embedder': {'module': 'core.nets.occnerf.embedders.fourier'},
    'non_rigid_embedder': {'module': 'core.nets.occnerf.embedders.hannw_fourier'},
    'canonical_mlp': {'module': 'core.nets.occnerf.canonical_mlps.occnerf_mlp', 'mlp_depth': 4,
                      'mlp_width': 256, 'multires': 10, 'i_embed': 0},
    'mweight_volume': {'module': 'core.nets.occnerf.mweight_vol_decoders.deconv_vol_decoder',
                       'embedding_size': 256, 'volume_size': 32, 'dst_voxel_size': 0.0625},
    'non_rigid_motion_mlp': {'module': 'core.nets.occnerf.non_rigid_motion_mlps.mlp_offset',
                             'condition_code_size': 69, 'mlp_width': 128, 'mlp_depth': 6,
                             'skips': [4], 'multires': 6, 'i_embed': 0, 'kick_in_iter': 100000,
                             'full_band_iter': 200000},
    'pose_decoder': {'module': 'core.nets.occnerf.pose_decoders.mlp_delta_body_pose',
                     'embedding_size': 69, 'mlp_width': 256, 'mlp_depth': 4,
                     'kick_in_iter': 2000000},
    'sex': 'neutral', 'total_bones': 24, 'bbox_offset': 0.3, 'load_net': 'latest',
    'N_samples': 128, 'perturb': 1.0, 'netchunk_per_gpu': 300000, 'chunk': 32768, 'n_gpus': 1,
    'bgcolor': [0.0, 0.0, 0.0], 'resize_img_scale': 0.5, 'show_alpha': False, 'show_truth': False,
    'patch': {'sample_subject_ratio': 0.8, 'N_patches': 6, 'size': 32},
    'freeview': {'frame_idx': 0}, 'tpose': {}, 'movement': {}, 'train': {},
    # build-specific keys
    'smpl_model': 'auto',            # 'auto' | 'synthetic' | directory holding the SMPL pickles
    # samples resident per pipeline pass (~470 B each): 2^28 keeps a whole 1024^2 x 192 frame (141 M samples, 63 GiB peak of the
    # 288 GB) in ONE pass; larger frames run in several, bit-identically.  The renderer additionally caps a pass at half of
    # the device memory that is free when the frame starts (Network.forward: min(this, free / 2 / 470 B)), so a smaller GPU
    # or ranks sharing one device split a large frame instead of running out of memory.
    'max_samples_per_pass': 1 << 28,
    # 'fp32': exact fp32 MFMA (default, the parity/benchmark path); 'f16x3': split-fp16 MFMA with scaled pieces, 22
    # significand bits per operand -- fp32-grade (csrc/split.h; domain: hidden activations below 4 094); 'bf16x3':
    # split-bf16 MFMA, 16 bits per operand (meets the pixel gate on the random-init checkpoint only)
    'mlp_precision': 'fp32',
    # mlp_precision='f16x3': True / 'sync' reads the kernels' out-of-domain flag after the frame and re-renders in fp32 when set;
    # 'deferred' checks it when a later frame starts / in Network.check_f16x3_domain() and raises; False: no check
    'f16x3_domain_check': True,
    'train_fused_trunks': True,      # bf16 training step: the trunks' forward as one kernel (csrc/trunks.hip); False: ten layer passes
    'skip_empty_samples': True,      # drop samples whose motion-weight sum is exactly 0 (identical pixels)
    'device_rays': True,             # run.py: generate the frame's ray batch on the GPU (occnerf_amd/rays.py)
    'ray_patch_order': True,         # render rays in Morton-ordered pixel patches (any order is exact)
    'knn_culling': True,             # cluster-culled exact kNN (False: brute force)
    'knn_center_cache': True,        # kNN queries within a proven radius of the frame's collapse point take its cached lists (same bits)
    'warp_bone_culling': True,       # warp kernel skips bones whose weight channel cannot reach a wave's samples (same bits)
}

_cfg = None


def default_cfg():
    return CfgNode.wrap(copy.deepcopy(_DEFAULTS))


def get_cfg():
    global _cfg
    if _cfg is None:
        _cfg = default_cfg()
        _finish(_cfg)
    return _cfg


def set_cfg(cfg):
    global _cfg
    _cfg = cfg
    return cfg


def _finish(cfg):
    import torch
    cfg.logdir = os.path.join('experiments', cfg.category, cfg.task, cfg.subject, cfg.experiment)
    n = torch.cuda.device_count()
    cfg.n_gpus = n
    cfg.primary_gpus = [0] if n > 0 else ['cpu']
    cfg.secondary_gpus = ([g for g in range(n) if g != 0] or cfg.primary_gpus) if n > 1 \
        else cfg.primary_gpus


def make_cfg(argv=None):
    """Same CLI as the reference (configs/config.py:65-72): --cfg FILE [--type T] [KEY VALUE ...]."""
    parser = argparse.ArgumentParser()
    parser.add_argument('--cfg', required=True, type=str)
    parser.add_argument('--eval', default='full', type=str)
    parser.add_argument('--type', default='skip', type=str)
    parser.add_argument('opts', default=None, nargs=argparse.REMAINDER)
    args = parser.parse_args(argv)
    cfg = default_cfg()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    default_yaml = os.path.join(here, 'configs', 'default.yaml')
    if os.path.exists(default_yaml):
        cfg.merge_from_file(default_yaml)
    cfg.merge_from_file(args.cfg)
    cfg.merge_from_list(args.opts or [])
    _finish(cfg)
    set_cfg(cfg)
    return cfg, args
