"""`CanonicalMLP`: drop-in for core/nets/occnerf/canonical_mlps/occnerf_mlp.py:31-199.

Same constructor keywords, parameter names (encoder.embeddings/offsets, pts_linears.{0,2,4,6},
geo_linear.0, rgb_linears.{0,2,4,6}, output_linear.0) and the same forward keyword surface
-> raw[N,5] = (rgb logits, sigma, signed distance).  Evaluation goes through the HIP
kernels: per-point table (hash encoding of the projected body points), per-sample
geometry + encoding + visibility aggregation, then the two 4x256 trunks on fp32 MFMA.

`Network` does not call this forward -- it keeps per-point arrays and calls the same ops
directly -- but a caller holding the reference's gathered arguments can.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .gridencoder import GridEncoder


def _trunk(in_dim, width, depth):
    layers = [nn.Linear(in_dim, width), nn.ReLU(inplace=True)]
    for _ in range(depth - 1):
        layers += [nn.Linear(width, width), nn.ReLU(inplace=True)]
    return nn.ModuleList(layers)


class CanonicalMLP(nn.Module):
    def __init__(self, mlp_depth=8, mlp_width=256, input_ch=3, skips=None, bound=1,
                 geo_feat_dim=63, **_):
        super().__init__()
        if skips:
            raise ValueError('CanonicalMLP: skip connections are not used by occnerf.yaml '
                             '(network.py:133 passes skips=[]) and are not built')
        self.mlp_depth, self.mlp_width, self.input_ch, self.bound = mlp_depth, mlp_width, input_ch, bound
        self.encoder = GridEncoder(input_dim=4, num_levels=16, level_dim=2, base_resolution=16,
                                   log2_hashmap_size=19, desired_resolution=2048 * bound,
                                   gridtype='hash', align_corners=False)
        self.neural_point_dim = 64
        self.pts_linears = _trunk(1 + 3 + 32 + 32, mlp_width, mlp_depth)
        self.geo_linear = nn.Sequential(nn.Linear(mlp_width, 64 + 1))
        self.rgb_linears = _trunk(64 + 32 + 32 + 3, mlp_width, mlp_depth)
        self.output_linear = nn.Sequential(nn.Linear(mlp_width, 3))

    def linear_params(self):
        """(weights, biases) of the 10 Linear layers in the order the C ABI packs them."""
        mods = [m for m in self.pts_linears if isinstance(m, nn.Linear)] + [self.geo_linear[0]] + \
               [m for m in self.rgb_linears if isinstance(m, nn.Linear)] + [self.output_linear[0]]
        if self.mlp_depth != 4 or self.mlp_width != 256:
            raise RuntimeError('the HIP canonical MLP is built for mlp_depth=4, mlp_width=256 '
                               '(configs/occnerf/zju_mocap/387/occnerf.yaml)')
        return [m.weight.detach() for m in mods], [m.bias.detach() for m in mods]

    def forward(self, xyz, xyz_embedded=None, knn_points=None, point_norms=None, knn_att=None,
                point_cloud=None, point_sdf=None, knn_idxs=None, learnable_points=None, **_):
        """Reference keyword surface (occnerf_mlp.py:142): gathered neighbours in, raw out."""
        N, k = knn_idxs.shape[0], knn_idxs.shape[2]
        enc = self.encoder
        b32 = float(np.float32(self.bound))
        tb32 = float(np.float32(2 * np.float64(self.bound)))
        with torch.no_grad():
            table = ops.point_table(point_cloud.double().contiguous(),
                                    point_sdf.reshape(-1).float().contiguous(),
                                    learnable_points.float().contiguous(), b32, tb32,
                                    enc.embeddings.detach(), enc.offsets, enc.log2_per_level_scale,
                                    enc.base_resolution)
            vpts = knn_points.reshape(-1, 3).float().contiguous()        # gathered rows as a
            vnrm = point_norms.reshape(-1, 3).double().contiguous()      # virtual point array
            geo = torch.arange(N * k, device=xyz.device, dtype=torch.int32).view(N, k)
            mlp_in, raw, _ = ops.sample_features(
                xyz.float().contiguous(), knn_idxs.int().contiguous(), vpts, vnrm,
                ops.unit_normals(vnrm), None, table, b32, tb32, enc.embeddings.detach(), enc.offsets,
                enc.log2_per_level_scale, enc.base_resolution, geo_idxs=geo,
                att_in=knn_att.reshape(N, -1).float().contiguous())
            ops.canonical_mlp(mlp_in, ops.canonical_mlp_pack(*self.linear_params()), raw)
        return raw
