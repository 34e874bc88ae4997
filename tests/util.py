"""Shared helpers for the parity tests: golden loading and the seeded model context."""
import functools
import os

import numpy as np

from occnerf_amd import checkpoint, geometry, synth
from occnerf_amd.gridencoder import grid_offsets

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
GOLDEN_CASES = ['tpose_ri_s32', 'tpose_ri_s128', 'freeview_amp_s32', 'tpose_amp_s32',
                'movement_amp_s32_f3', 'movement_amp_s32_f9']


def load_golden(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    return {k: g[k] for k in g.files}


@functools.lru_cache(maxsize=4)
def model_context(seed=0, amplify=False):
    """Everything Network.generate_neural_points + the checkpoint define, as numpy
    (no reference import): mesh points, float64 normals, FPS scales, state dict, grid."""
    smpl = synth.SyntheticSMPL()
    verts, joints = smpl(np.zeros(72), np.zeros(10))
    bb = synth.skeleton_to_bbox(joints)
    bound = float(np.max(np.abs(list(bb['min_xyz']) + list(bb['max_xyz']))))
    normals = geometry.vertex_normals(verts, smpl.faces)
    fps, ratio = [], 1.0
    for _ in range(3):
        ratio /= 4
        fps.append(geometry.farthest_point_sampling(verts, ratio))
    sd = checkpoint.make_state_dict(verts, bound, seed=seed, amplify=amplify)
    offsets, pls = grid_offsets(4, 16, 2.0, 16, 19, desired_resolution=2048 * bound)
    return {
        'verts': verts, 'joints': joints, 'bound': bound, 'normals': normals, 'fps': fps,
        'sd': sd, 'offsets': offsets, 'S': float(np.log2(pls)), 'H': 16,
        'point_base': sd['point_base'].numpy(),
        'point_cloud': (sd['point_base'] + sd['point_dist']).numpy(),
        'counter': sd['point_counter'].numpy(),
        'embeddings': sd['cnl_mlp.module.encoder.embeddings'].numpy(),
    }


def mlp_params(sd, prefix, idxs):
    W = [sd[f'{prefix}.{i}.weight'].numpy() for i in idxs]
    B = [sd[f'{prefix}.{i}.bias'].numpy() for i in idxs]
    return W, B


def canonical_mlp_params(sd):
    Wg, Bg = mlp_params(sd, 'cnl_mlp.module.pts_linears', (0, 2, 4, 6))
    w, b = mlp_params(sd, 'cnl_mlp.module.geo_linear', (0,))
    Wc, Bc = mlp_params(sd, 'cnl_mlp.module.rgb_linears', (0, 2, 4, 6))
    w2, b2 = mlp_params(sd, 'cnl_mlp.module.output_linear', (0,))
    return Wg + w, Bg + b, Wc + w2, Bc + b2


def nonrigid_params(sd):
    return mlp_params(sd, 'non_rigid_mlp.module.block_mlps', (0, 2, 4, 6, 8, 10, 12))


def knn_mismatch_is_tie(q, s, got, want, rel=2e-6):
    """True when every differing neighbour slot is a near-tie in distance (the only
    legitimate source of kNN disagreement, SURVEY.md section 7 'kNN exactness')."""
    bad = np.argwhere(got != want)
    for i, j in bad:
        dg = np.linalg.norm(q[i].astype(np.float64) - s[got[i, j]].astype(np.float64))
        dw = np.linalg.norm(q[i].astype(np.float64) - s[want[i, j]].astype(np.float64))
        if abs(dg - dw) > rel * max(dg, dw):
            return False
    return True
