"""Differentiable evaluation of the sample path (config 5).

Rendering (no grad) goes through the fused HIP kernels of Network._render_rays.  When gradients are needed the
same chain is evaluated here, stage by stage, each stage a HIP forward that keeps what its HIP backward needs
(occnerf_amd/train_ops.py, include/occnerf_hip.h section 3):

  sampler + warp      ops.sample_warp forward; backward of `mask` w.r.t. the motion-weight volume and the motion
                      bases (network.py:351-402, 405-432, 456)
  non-rigid MLP       HIP forward only: in the reference's graph `xyz` enters CanonicalMLP through no_grad
                      quantities alone (occnerf_mlp.py:144-167), so this MLP receives no gradient from the
                      rendering loss (the reference's own gradients for it are None; tests/golden/train_*)
  kNN, geometry       HIP (integers / no_grad quantities): ops.msknn_clustered, ops.sample_features
  hash encoding       HIP operator forward/backward behind occnerf_amd/gridencoder.py
  aggregation         HIP forward/backward (ops.aggregate), weights by train_ops.agg_weights
  MLP trunks          train_ops.canonical_trunks: ten MFMA layers forward, wgrad + dgrad backward (bf16 or fp32)
  compositing         train_ops.composite forward/backward (network.py:320-348)
  a18                 comp_loss and the point_counter update: a few elementwise torch ops on [n,S] (network.py:486-519)

The per-point block (network.py:263-284, P = 6 890 rows, gradient to point_dist) stays torch.
`render_rays_autograd_torch` is the first, all-torch-autograd evaluation of the same chain (nn.Linear,
F.grid_sample, cumprod); it is kept as the fp32 reference the HIP stages are tested against.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops, train_ops


def sample_along_rays(rays8, S, perturb, t_rand=None):
    """network.py:405-432,456 -> z_vals[n,S], pts[n,S,3]."""
    near, far = rays8[:, 6:7], rays8[:, 7:8]
    t = torch.linspace(0., 1., steps=S, device=rays8.device)
    z = near * (1. - t) + far * t
    if perturb > 0.:
        mids = .5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        if t_rand is None:
            t_rand = torch.rand(z.shape, device=z.device)
        z = lower + (upper - lower) * t_rand
    pts = rays8[:, None, 0:3] + rays8[:, None, 3:6] * z[..., None]
    return z, pts


def warp_to_canonical(pts, Rs, Ts, vol, bbox_min, bbox_scale):
    """network.py:351-402 for all bones at once -> x_skel[n,S,3], mask[n,S,1]."""
    shape = pts.shape
    p = pts.reshape(-1, 3)
    nb = Rs.shape[0]
    pos = torch.einsum('bij,nj->bni', Rs, p) + Ts[:, None, :]                    # [nb,N,3]
    grid = (pos - bbox_min[None, None, :]) * bbox_scale[None, None, :] - 1.0
    w = F.grid_sample(vol[:nb, None], grid[:, None, None, :, :], padding_mode='zeros',
                      align_corners=True)[:, 0, 0, 0, :]                         # [nb,N]
    wsum = w.sum(0)[:, None]
    x_skel = (w[..., None] * pos).sum(0) / wsum.clamp(min=0.0001)
    return x_skel.reshape(shape), wsum.reshape(shape[0], shape[1], 1)


def motion_bases(net, refine, posevec, dst_Rs, dst_Ts, cnl_gtfms):
    """network.py:557-596 + network_util.py:166-200 with gradients to the pose refiner: (posevec[1,69], dst_Rs[1,24,3,3],
    dst_Ts[1,24,3], cnl_gtfms[1,24,4,4]) -> Rs[24,3,3], Ts[24,3].  Default: the renderer's fused forward kernel and a fused
    backward (train_ops.pose_motion_bases: 2 launches); cfg.train_fused_pose=False: the torch modules under autograd (~560
    launches), kept as the reference the fused pair is tested against."""
    if net.cfg.get('train_fused_pose', True) and posevec.is_cuda:
        if refine:
            return train_ops.pose_motion_bases(net.pose_decoder, posevec[0], dst_Rs[0], dst_Ts[0], cnl_gtfms[0])
        with torch.no_grad():          # before the kick-in iteration the bases are constants of the frame
            return ops.pose_motion_bases(net.pose_decoder, posevec[0].float().contiguous(), False, dst_Rs[0].float().contiguous(),
                                         dst_Ts[0].float().contiguous(), cnl_gtfms[0].float().contiguous())
    if refine:
        refined = net.pose_decoder(posevec)['Rs']
        tb = int(net.cfg.total_bones) - 1
        no_root = torch.matmul(dst_Rs[:, 1:].reshape(-1, 3, 3), refined.reshape(-1, 3, 3)).reshape(-1, tb, 3, 3)
        dst_Rs = torch.cat([dst_Rs[:, 0:1], no_root], dim=1)
    Rs, Ts = net.motion_basis_computer(dst_Rs, dst_Ts, cnl_gtfms)
    return Rs[0], Ts[0]


def point_sdf_block(net):
    """network.py:263-284 with gradients to point_dist -> knn_base[P,3] (f64), dist[P,1].  Default: the renderer's forward
    kernels + one backward kernel (train_ops.point_sdf); cfg.train_fused_points=False: the torch ops under autograd, kept as
    the reference the fused pair is tested against."""
    if net.cfg.get('train_fused_points', True) and net.point_base.is_cuda:
        return train_ops.point_sdf(net)
    pc = net.point_cloud.float()
    base = net.point_base.detach()
    kidx = ops.knn_small(pc.detach().contiguous(), base, 3).long()
    nbr = base[kidx]                                                             # [P,3,3]
    direction = pc[:, None, :] - nbr
    # (the device copy of the float64 normals kept by Network._context: `net.point_norms` is a plain host attribute as in the
    # reference, network.py:122, and moving it every step is a synchronous 165 KB copy -- and not capturable in a hipGraph)
    norms = net._context()['normals'][kidx]                                      # float64
    att = torch.abs(F.cosine_similarity(direction, norms, dim=-1))[..., None]
    knn_base = (att * nbr).sum(1) / att.sum(1)
    # (row-wise dot products; the reference's einsum 'ijk,ijk->ij' runs as 20 670 batched 1x3x1 GEMMs: 0.28 ms)
    inside = ((direction.float() * norms.float()).sum(-1) < 0).sum(1) > 1.5
    dist = torch.norm(direction, dim=-1).mean(1, keepdim=True)
    dist = torch.where(inside[:, None], -dist, dist)
    return knn_base, dist


def canonical_mlp_torch(cm, xyz, knn_idxs, net, knn_base, point_sdf):
    """occnerf_mlp.py:142-199 -> raw[N,5].  cm: CanonicalMLP (parameters), knn_idxs[N,4,10]."""
    N, k = knn_idxs.shape[0], knn_idxs.shape[2]
    base = net.point_base.detach()
    idx0 = knn_idxs[:, 0].long()
    knn_points = base[idx0]                                                      # [N,10,3]
    normals = net._context()['normals'][idx0]                                    # float64
    with torch.no_grad():
        direction = xyz[:, None, :] - knn_points
        # row-wise fp64 dot products (the reference's einsum 'ijk,ijk->ij' runs as a batched fp64 GEMM: 13 ms
        # per step here); only the sign is used
        inside = ((direction.double() * normals.double()).sum(-1) < 0).sum(1) > k * 0.5
        dist = torch.norm(direction, dim=-1).mean(1, keepdim=True)
        dist = torch.where(inside[:, None], -dist, dist)
        normed = torch.clamp((dist + 0.2) / 0.5, 0.0, 1.0)
    bound = cm.bound
    pn = (knn_points + bound) / (2 * bound)
    att = torch.abs(F.cosine_similarity(direction[:, :3], normals[:, :3], dim=-1))[..., None]
    q = (att * pn[:, :3]).sum(1) / att.sum(1)
    h = cm.encoder(torch.cat((q, normed), dim=-1).float(), bound=None)

    pc01 = (knn_base + bound) / (2 * bound)
    sdf01 = torch.clamp((point_sdf + 0.2) / 0.8, 0.0, 1.0)
    feats = cm.encoder(torch.cat((pc01, sdf01), dim=-1).float(), bound=None)
    feats = torch.cat((feats, net.point_cloud.float()), dim=-1)                  # [P,35]
    atts = net.point_counter.detach()[knn_idxs.long()].view(N, -1, 1).clone()
    atts = atts + (1. - atts.min(dim=1, keepdim=True)[0])
    atts = atts / atts.max(dim=1, keepdim=True)[0]
    var = torch.var(atts, dim=1)
    atts = F.softmax(atts, dim=1)
    # sum_j atts[n,j] * feats[knn[n,j]] without materialising feats[knn] ([N,40,35]); HIP forward and
    # atomics backward (ops.aggregate), atts detached as in the reference (occnerf_mlp.py:124)
    agg = ops.aggregate(feats, knn_idxs.reshape(N, -1), atts.detach().reshape(N, -1))

    enc = h
    z = torch.cat([agg, var, enc], dim=-1).float()
    for layer in cm.pts_linears:
        z = layer(z)
    z = cm.geo_linear(z)
    sigma = z[..., [0]]
    z = torch.cat([z[..., 1:], agg, enc], dim=-1)
    for layer in cm.rgb_linears:
        z = layer(z)
    rgb = cm.output_linear(z)
    return torch.cat((rgb, sigma, dist.detach()), dim=-1)


def raw2outputs(raw, mask, z_vals, rays_d, bgcolor):
    """network.py:320-348."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], dim=-1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    alpha = (1.0 - torch.exp(-F.softplus(raw[..., 3]) * dists)) * mask[:, :, 0]
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], dim=-1), dim=-1)[:, :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    term = torch.argmax(alpha, dim=1, keepdim=True)
    depth = torch.sum(weights * z_vals, -1)
    acc = torch.sum(weights, -1)
    rgb_map = rgb_map + (1. - acc[..., None]) * bgcolor[None, :] / 255.
    return rgb_map, acc, depth, term


def render_rays_autograd_torch(net, rays8, Rs, Ts, vol, bbox_min, bbox_scale, bgcolor, cond, hann, t_rand=None):
    """All-torch-autograd evaluation (nn.Linear, F.grid_sample, cumprod) of the chain below: the fp32 reference the
    HIP stages are tested against.  bbox_min, bbox_scale, bgcolor: device tensors."""
    cfg, ctx = net.cfg, net._context()
    S = int(cfg.N_samples)
    n = rays8.shape[0]
    z, pts = sample_along_rays(rays8, S, float(cfg.perturb), t_rand)
    cnl_pts, mask = warp_to_canonical(pts, Rs, Ts, vol, bbox_min, bbox_scale)
    xyz = cnl_pts.reshape(-1, 3)
    if not cfg.ignore_non_rigid_motions:
        nr = net.non_rigid_mlp.module
        emb = torch.cat([hann[j] * fn(xyz * float(2 ** j)) for j in range(int(cfg.non_rigid_motion_mlp.multires))
                         for fn in (torch.sin, torch.cos)], dim=-1)
        xyz = nr(pos_embed=emb, pos_xyz=xyz, condition_code=cond.expand(xyz.shape[0], -1))['xyz']
    with torch.no_grad():
        knn = ops.msknn_clustered(xyz.detach().float().contiguous(), n, S, ctx['clusters'], ctx['seed'])
    knn_base, sdf = point_sdf_block(net)
    raw = canonical_mlp_torch(net.cnl_mlp.module, xyz, knn, net, knn_base, sdf).reshape(n, S, 5)
    rgb, acc, depth, term = raw2outputs(raw, mask, z, rays8[:, 3:6], bgcolor)

    if net.training:                                                             # network.py:486-517
        dist, sigma = raw[..., 4:], raw[..., [3]]
        comp_loss = (dist < 0.).float().detach() * torch.exp(torch.clamp(-F.relu(sigma), min=-10, max=0))
        comp_loss = comp_loss.squeeze(-1) * 10.
        depth_mask = depth.detach() > 0.5
        if int(depth_mask.sum()) > 1:
            tp = term[depth_mask].detach()
            term_pts = torch.gather(cnl_pts[depth_mask].detach(), 1, tp[:, :, None].expand(-1, 1, 3)).squeeze(1)
            kidx = ops.knn_small(term_pts.float().contiguous(), net.point_cloud.detach().float().contiguous(), 10)
            _bump_counter(net, kidx)
    else:
        comp_loss = torch.zeros(1, 1, device=rays8.device)
    return rgb, acc, depth, comp_loss


def _bump_counter(net, kidx):
    """network.py:508-510 `point_counter[idx] += 1` (a non-accumulating index_put: duplicates count once).  A write through
    `.data` does not move the parameter's version counter, and the renderer's cached (geometry, counts) pack is keyed
    on it (Network._point_pack): move it explicitly, or eval renders after training-mode forwards see stale counts."""
    net.point_counter.data[kidx.view(-1).long()] += 1.
    torch.autograd.graph.increment_version(net.point_counter)


def _training_branch(net, raw, depth, term, cnl_pts):
    """network.py:486-517: comp_loss per sample and the visibility-counter update (training mode only).
    The reference selects the rays with depth > 0.5 on the host (`if depth_mask.sum() > 1`, boolean indexing): a device ->
    host round trip in the middle of the step.  Here the search runs for every ray's arg-max-alpha sample (6 144 queries:
    0.3 ms) and the selection is a weight in a device-side scatter -- the same points are incremented (duplicates count once,
    as with the reference's non-accumulating index_put), and the step has no synchronisation point."""
    dist, sigma = raw[..., 4:], raw[..., 3:4]        # (a slice: the reference's list index [3] backpropagates as an index_put)
    comp_loss = (dist < 0.).float().detach() * torch.exp(torch.clamp(-F.relu(sigma), min=-10, max=0))
    comp_loss = comp_loss.squeeze(-1) * 10.
    with torch.no_grad():
        depth_mask = depth.detach() > 0.5
        tp = term.detach().long()
        term_pts = torch.gather(cnl_pts.detach(), 1, tp[:, :, None].expand(-1, 1, 3)).squeeze(1)      # [n,3], every ray
        kidx = ops.knn_small(term_pts.float().contiguous(), net.point_cloud.detach().float().contiguous(), 10)
        w = (depth_mask & (depth_mask.sum() > 1)).to(torch.float32)
        hit = torch.zeros_like(net.point_counter.data, dtype=torch.float32)
        hit.index_add_(0, kidx.view(-1).long(), w[:, None].expand(-1, kidx.shape[1]).reshape(-1))
        net.point_counter.data += (hit > 0).to(net.point_counter.dtype)
        torch.autograd.graph.increment_version(net.point_counter)       # (see _bump_counter)
    return comp_loss


def _use_bf16(cfg):
    """bf16 trunks when the caller runs under torch.autocast(bfloat16) (train.py, `train.bf16`) or asks for it
    with cfg.train_precision = 'bf16'; exact fp32 otherwise."""
    want = str(cfg.get('train_precision', 'auto'))
    if want == 'auto':
        return torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16
    if want not in ('bf16', 'fp32'):
        raise RuntimeError(f"cfg.train_precision must be 'auto', 'bf16' or 'fp32', got {want!r}")
    return want == 'bf16'


def canonical_mlp_hip(cm, xyz, knn_idxs, net, knn_base, point_sdf, ctx, bf16):
    """occnerf_mlp.py:142-199 -> raw[N,5]; every per-sample stage a HIP forward + HIP backward."""
    enc = cm.encoder
    N = knn_idxs.shape[0]
    pc = net.point_cloud.float()
    with torch.no_grad():
        # geometry prelude (:144-167, all no_grad in the reference): encoder input [q, normed] and the signed distance
        table = ops.point_table(knn_base.detach().double().contiguous(), point_sdf.detach().reshape(-1).float().contiguous(),
                                pc.detach().contiguous(), ctx['bound32'], ctx['two_bound32'], enc.embeddings.detach(),
                                enc.offsets, enc.log2_per_level_scale, enc.base_resolution)
        _, raw_d, enc_in = ops.sample_features(
            xyz.detach().contiguous(), knn_idxs, net.point_base.detach(), ctx['normals'], ctx['unit'],
            net.point_counter.detach(), table, ctx['bound32'], ctx['two_bound32'], enc.embeddings.detach(), enc.offsets,
            enc.log2_per_level_scale, enc.base_resolution, want_enc_in=True)
        dist = raw_d[:, 4:5]
        knn40 = knn_idxs.reshape(N, -1)
        atts, var = train_ops.agg_weights(net.point_counter.detach().float().contiguous(), knn40)
    with torch.autocast('cuda', enabled=False):
        h = enc(enc_in, bound=None)                                                  # [N,32], gradient to the table
        bound = cm.bound
        pc01 = (knn_base + bound) / (2 * bound)
        sdf01 = torch.clamp((point_sdf + 0.2) / 0.8, 0.0, 1.0)
        feats = enc(torch.cat((pc01, sdf01), dim=-1).float(), bound=None)
        feats = torch.cat((feats, pc), dim=-1)                                       # [P,35]
        agg = ops.aggregate(feats, knn40, atts)
        raw4 = train_ops.canonical_trunks(cm, agg, var, h, bf16, fused=bool(net.cfg.get('train_fused_trunks', True)),
                                          packed_f32=net._packed_weights()['cnl'] if bf16 else None)
    return torch.cat((raw4, dist), dim=-1)


def render_rays_autograd(net, rays8, Rs, Ts, vol, bbox_min, bbox_scale, bgcolor, cond, hann, t_rand=None, point_block=None):
    """Differentiable counterpart of Network._render_rays (+ the training branch a18).
    bbox_min, bbox_scale, bgcolor: HOST float32[3]; hann: HOST list of the 6 window weights; point_block: the step's
    (knn_base, sdf) of point_sdf_block when the caller has evaluated it already (once per step, inside the hipGraph)."""
    cfg, ctx = net.cfg, net._context()
    bf16 = _use_bf16(cfg)                 # (asked before autocast is switched off for the fp32 stages below)
    S = int(cfg.N_samples)
    n = rays8.shape[0]
    dev = rays8.device
    t_vals = torch.linspace(0., 1., steps=S, device=dev)
    if float(cfg.perturb) > 0.:
        t_rand = (torch.rand(n, S, device=dev) if t_rand is None else t_rand).float().contiguous()
    else:
        t_rand = None
    with torch.autocast('cuda', enabled=False):
        z, cnl, mask = train_ops.sample_warp(rays8, S, t_vals, t_rand, Rs.float(), Ts.float(), vol.float(), bbox_min,
                                             bbox_scale)
    xyz = cnl
    center = None
    with torch.no_grad():
        if not cfg.ignore_non_rigid_motions:
            pk = net._packed_weights()
            condv = cond.reshape(-1).float().contiguous()
            if bf16 and cfg.get('train_nonrigid_f16x3', True):
                # bf16 step: the offsets (no gradient reaches this MLP, occnerf_mlp.py:144-167) on the fp32-grade split-fp16
                # kernels (csrc/split.h; <= 1e-6 m from the fp32 kernel, 2.6x faster)
                # (its out-of-domain flag is copied behind the step and looked at when a later step starts: a step whose
                # activations reached the f16x3 clamp raises there -- Network.check_f16x3_domain)
                flag = net._f16x3_flag_word() if cfg.get('f16x3_domain_check', True) else None
                xyz = ops.nonrigid_bf16x3(cnl, condv, hann, pk['nr_w0'], pk['nr_b0'], pk['nr'], net._nonrigid_f16_pack(),
                                          domain_flag=flag)
                if flag is not None:
                    net._f16x3_enqueue_check(flag, 'training step')
            else:
                xyz = ops.nonrigid(cnl, condv, hann, pk['nr_w0'], pk['nr_b0'], pk['nr'])
        if cfg.get('knn_center_cache', True) and cfg.get('knn_culling', True):
            # the renderer's exact centre cache: queries inside the proven radius around the collapse point take its lists
            center = net._knn_center_lists(cond.reshape(-1).float().contiguous(), hann)
        knn = ops.msknn_clustered(xyz, n, S, ctx['clusters'], ctx['seed'], center=center)
    with torch.autocast('cuda', enabled=False):
        knn_base, sdf = point_sdf_block(net) if point_block is None else point_block
        raw = canonical_mlp_hip(net.cnl_mlp.module, xyz, knn, net, knn_base, sdf, ctx, bf16)
        rgb, acc, depth, term = train_ops.composite(raw, mask, z, rays8, bgcolor)
        if net.training:
            comp_loss = _training_branch(net, raw.reshape(n, S, 5), depth, term.reshape(n, 1), cnl.reshape(n, S, 3))
        else:
            comp_loss = torch.zeros(1, 1, device=dev)
    return rgb, acc, depth, comp_loss
