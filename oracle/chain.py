"""TEST INFRASTRUCTURE (like everything under oracle/): the whole path on the CPU, assembled from the oracle's stages.

Used by the parity tests, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg -- never by the product
(occnerf_amd/ must not import this package; tests/test_abi.py checks).  `model_context` rebuilds, without the reference,
everything Network.generate_neural_points + the seeded checkpoint define; `stagewise_oracle_render` chains the C oracle's
stages (oracle/occnerf_oracle.c) into a full render; the per-frame modules (pose refiner, motion bases, motion-weight volume
decoder: a few hundred KFLOP per frame, rows a2-a4) are evaluated with the product's torch modules on the CPU, which are
pinned against the reference through the golden `pose.Rs`, `mb.*`, `mw.vol_slice` arrays."""
import functools

import numpy as np
import torch

from occnerf_amd import checkpoint, geometry, synth
from occnerf_amd.gridencoder import grid_offsets
from occnerf_amd.modules import (BodyPoseRefiner, MotionBasisComputer, MotionWeightVolumeDecoder,
                                 hann_window_weights)

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


def golden_frame(g):
    """Frame dict (numpy) of a golden case, including its ray subset."""
    frame = synth.make_frame(img_size=int(g['meta.img_size']), pose72=g['meta.pose72'],
                             orbit_frame=int(g['meta.orbit_frame']))
    for k in ('rays', 'near', 'far'):
        frame[k] = g['in.' + k]
    return frame


def per_frame_cpu(ctx, frame, iter_val=1e7, kick_pose=2000000, kick_nr=100000, full_nr=200000):
    """Pose decoder, motion bases, motion-weight volume on CPU torch (the product's modules;
    pinned against the reference through the golden `pose.Rs`, `mb.*`, `mw.vol_slice`)."""
    sd = ctx['sd']

    def sub(prefix):
        return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    pose = BodyPoseRefiner()
    pose.load_state_dict(sub('pose_decoder.'))
    dec = MotionWeightVolumeDecoder()
    dec.load_state_dict(sub('mweight_vol_decoder.'))
    with torch.no_grad():
        dst_Rs = torch.from_numpy(frame['dst_Rs'])[None]
        dst_Ts = torch.from_numpy(frame['dst_Ts'])[None]
        posevec = torch.from_numpy(frame['dst_posevec'])[None]
        if iter_val >= kick_pose:
            ref = pose(posevec)['Rs']
            no_root = torch.matmul(dst_Rs[:, 1:].reshape(-1, 3, 3), ref.reshape(-1, 3, 3)).reshape(-1, 23, 3, 3)
            dst_Rs = torch.cat([dst_Rs[:, 0:1], no_root], 1)
        Rs, Ts = MotionBasisComputer()(dst_Rs, dst_Ts, torch.from_numpy(frame['cnl_gtfms'])[None])
        vol = dec(torch.from_numpy(frame['motion_weights_priors'])[None])[0]
        hann = hann_window_weights(6, iter_val, kick_nr, full_nr)
    return Rs[0].numpy(), Ts[0].numpy(), vol.numpy(), hann.numpy(), frame['dst_posevec']


def stagewise_oracle_render(g, ctx, frame=None, S=None, non_rigid=None, preamble=None):
    """Whole path on the CPU from the oracle's stages (the `port` CPU baseline and the
    end-to-end checker).  Returns rgb/alpha/depth + the intermediates.
    preamble=(Rs[24,3,3], Ts[24,3], vol[25,G,G,G]) (numpy): evaluate the per-sample stages on THESE per-frame outputs instead of
    the torch-CPU ones -- a parity test that hands over the HIP preamble's outputs compares the per-sample kernels alone
    (the preamble kernels are pinned separately against the reference's pose.Rs / mb.* / mw.vol_slice goldens)."""
    from oracle import oracle as orc
    frame = golden_frame(g) if frame is None else frame
    S = int(g['meta.S']) if S is None else S
    non_rigid = bool(int(g['meta.non_rigid'])) if non_rigid is None else non_rigid
    Rs, Ts, vol, hann, cond = per_frame_cpu(ctx, frame)
    if preamble is not None:
        Rs, Ts, vol = (np.ascontiguousarray(a, dtype=np.float32) for a in preamble)
    rays8 = np.concatenate([frame['rays'][0], frame['rays'][1], frame['near'], frame['far']], -1).astype(np.float32)
    t_vals = torch.linspace(0., 1., steps=S).numpy()
    z, pts = orc.sample_rays(rays8, t_vals)
    xyz, mask = orc.motion_field(pts, Rs, Ts, vol, frame['cnl_bbox_min_xyz'], frame['cnl_bbox_scale_xyz'])
    x_skel = xyz
    if non_rigid:
        W, B = nonrigid_params(ctx['sd'])
        xyz = orc.nonrigid(xyz, cond, hann, W, B)
    knn = orc.msknn(xyz, ctx['point_base'], ctx['fps'], k=10)
    kb, sdf = orc.point_sdf(ctx['point_cloud'], ctx['point_base'], ctx['normals'])
    table = orc.point_table(kb, sdf, ctx['point_cloud'], ctx['bound'], ctx['embeddings'], ctx['offsets'],
                            ctx['S'], ctx['H'])
    Wg, Bg, Wc, Bc = canonical_mlp_params(ctx['sd'])
    raw, mlp_in = orc.canonical_mlp(xyz, knn, ctx['point_base'], ctx['normals'], ctx['counter'], table,
                                    ctx['bound'], ctx['embeddings'], ctx['offsets'], ctx['S'], ctx['H'],
                                    Wg, Bg, Wc, Bc, want_mlp_in=True)
    n = rays8.shape[0]
    rgb, acc, w, dep, tp = orc.raw2outputs(raw.reshape(n, S, 5), mask.reshape(n, S), z, rays8[:, 3:6],
                                           frame['bgcolor'])
    return {'rgb': rgb, 'alpha': acc, 'depth': dep, 'z': z, 'pts': pts, 'xyz': xyz, 'x_skel': x_skel,
            'mask': mask,
            'knn': knn, 'table': table, 'kb': kb, 'sdf': sdf, 'raw': raw, 'mlp_in': mlp_in,
            'weights': w, 'term': tp, 'Rs': Rs, 'Ts': Ts, 'vol': vol, 'hann': hann, 'cond': cond,
            'rays8': rays8, 't_vals': t_vals}
