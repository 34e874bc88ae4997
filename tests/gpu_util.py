"""Helpers shared by the GPU parity tests, smoke() and bench.py: model construction from the
seeded checkpoint, frame upload, and a full CPU render assembled from the oracle's stages."""
import numpy as np
import torch

from occnerf_amd import checkpoint, synth
from occnerf_amd.config import default_cfg, set_cfg, _finish
from occnerf_amd.modules import (BodyPoseRefiner, MotionBasisComputer, MotionWeightVolumeDecoder,
                                 hann_window_weights)
from tests import util

FRAME_KEYS = ['rays', 'near', 'far', 'bgcolor', 'dst_Rs', 'dst_Ts', 'cnl_gtfms',
              'motion_weights_priors', 'cnl_bbox_min_xyz', 'cnl_bbox_max_xyz', 'cnl_bbox_scale_xyz',
              'dst_posevec']


def build_network(seed=0, amplify=False, S=128, non_rigid=False, device='cuda:0', mlp_precision='fp32'):
    """Network with the seeded checkpoint loaded (strict), on `device`, in eval mode."""
    from occnerf_amd.network import Network
    cfg = default_cfg()
    _finish(cfg)
    cfg.N_samples = S
    cfg.perturb = 0.
    cfg.ignore_non_rigid_motions = not non_rigid
    cfg.smpl_model = 'synthetic'
    cfg.mlp_precision = mlp_precision
    set_cfg(cfg)
    ctx = util.model_context(seed, amplify)
    net = Network()
    net.generate_neural_points(np.zeros(10, 'float32'))
    net.load_state_dict(ctx['sd'], strict=True)
    return net.to(device).deploy_mlps_to_secondary_gpus().eval(), ctx


def golden_frame(g):
    """Frame dict (numpy) of a golden case, including its ray subset."""
    frame = synth.make_frame(img_size=int(g['meta.img_size']), pose72=g['meta.pose72'],
                             orbit_frame=int(g['meta.orbit_frame']))
    for k in ('rays', 'near', 'far'):
        frame[k] = g['in.' + k]
    return frame


def frame_to_device(g_or_frame, device):
    frame = golden_frame(g_or_frame) if 'meta.S' in g_or_frame else g_or_frame
    return {k: torch.from_numpy(np.ascontiguousarray(frame[k])).to(device) for k in FRAME_KEYS}


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


def stagewise_oracle_render(g, ctx, frame=None, S=None, non_rigid=None):
    """Whole path on the CPU from the oracle's stages (the `port` CPU baseline and the
    end-to-end checker).  Returns rgb/alpha/depth + the intermediates."""
    from oracle import oracle as orc
    frame = golden_frame(g) if frame is None else frame
    S = int(g['meta.S']) if S is None else S
    non_rigid = bool(int(g['meta.non_rigid'])) if non_rigid is None else non_rigid
    Rs, Ts, vol, hann, cond = per_frame_cpu(ctx, frame)
    rays8 = np.concatenate([frame['rays'][0], frame['rays'][1], frame['near'], frame['far']], -1).astype(np.float32)
    t_vals = torch.linspace(0., 1., steps=S).numpy()
    z, pts = orc.sample_rays(rays8, t_vals)
    xyz, mask = orc.motion_field(pts, Rs, Ts, vol, frame['cnl_bbox_min_xyz'], frame['cnl_bbox_scale_xyz'])
    x_skel = xyz
    if non_rigid:
        W, B = util.nonrigid_params(ctx['sd'])
        xyz = orc.nonrigid(xyz, cond, hann, W, B)
    knn = orc.msknn(xyz, ctx['point_base'], ctx['fps'], k=10)
    kb, sdf = orc.point_sdf(ctx['point_cloud'], ctx['point_base'], ctx['normals'])
    table = orc.point_table(kb, sdf, ctx['point_cloud'], ctx['bound'], ctx['embeddings'], ctx['offsets'],
                            ctx['S'], ctx['H'])
    Wg, Bg, Wc, Bc = util.canonical_mlp_params(ctx['sd'])
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
