"""Autograd Functions of the training step over the C ABI's section 3 (include/occnerf_hip.h).

The reference trains through plain torch autograd (trainer.py:239-249 over network.py:444-623 and
occnerf_mlp.py:142-199).  Here every per-sample stage of that graph is a hand-written HIP forward that keeps what
its backward needs, and a hand-written HIP backward:

  canonical_trunks   occnerf_mlp.py:183-199 (pts_linears, geo_linear, rgb_linears, output_linear): ten layers,
                     each one streaming MFMA pass (csrc/linear.hip); backward = per layer a weight-gradient pass
                     (split over row slices, reduced deterministically) and an input-gradient pass with the ReLU
                     mask of the saved activation in its epilogue.  bf16 operands / fp32 accumulation
                     (BASELINE configs[4]) or exact fp32.
  composite          network.py:320-348 (_raw2outputs).
  sample_warp        network.py:405-432,456 and :351-402; only `mask` carries a gradient (to the motion-weight
                     volume and the motion bases): x_skel enters CanonicalMLP through no_grad quantities only.
  agg_weights        occnerf_mlp.py:110-125 (no gradient: the reference detaches the counts).
"""
import ctypes as C

import numpy as np
import torch
from torch.autograd import Function

from . import _lib, ops

PAD = 32                      # every matrix width is padded to a multiple of 32 elements


def _pad(n):
    return (n + PAD - 1) // PAD * PAD


def _esz(bf16):
    return 2 if bf16 else 4


def _dtype(bf16):
    return torch.bfloat16 if bf16 else torch.float32


def _ptr(t, col=0):
    return None if t is None else t.data_ptr() + col * t.element_size()


def linear_pack(W, b, row_map, col_map, bf16, want_t=True):
    """-> Wp[n_pad,k_pad], Wt[k_pad,n_pad] (or None), bias_p[n_pad] fp32."""
    n_pad, k_pad = row_map.numel(), col_map.numel()
    dev, dt = W.device, _dtype(bf16)
    Wp = torch.empty(n_pad, k_pad, device=dev, dtype=dt)
    Wt = torch.empty(k_pad, n_pad, device=dev, dtype=dt) if want_t else None
    bp = torch.empty(n_pad, device=dev, dtype=torch.float32)
    with ops._guard(W):
        rc = _lib.lib().occnerf_linear_pack(
            ops._chk(W, torch.float32, 'W'), ops._opt(b, torch.float32, 'b'), W.shape[0], W.shape[1],
            ops._chk(row_map, torch.int32, 'row_map'), n_pad, ops._chk(col_map, torch.int32, 'col_map'), k_pad,
            int(bf16), Wp.data_ptr(), _ptr(Wt), bp.data_ptr(), ops._stream(W))
    _lib.check(rc, 'linear_pack')
    return Wp, Wt, bp


def linear_forward(x0, k0, W, n_pad, bf16, x1=None, k1=0, bias=None, relu=False, mask=None, out=None,
                   out_f32=False, n_store=None, aux=None, aux_col=0, aux_stride=0):
    """y = epi(x @ W^T).  x0/x1/mask/out: 2-D row-major tensors of the element type (out: fp32 when out_f32);
    W: packed [n_pad, k0 + k1]."""
    M = x0.shape[0]
    dt = _dtype(bf16)
    odt = torch.float32 if (out_f32 or not bf16) else dt
    if out is None:
        out = torch.empty(M, n_pad, device=x0.device, dtype=odt)
    n_store = n_pad if n_store is None else n_store
    for t, name, want in ((x0, 'x0', dt), (x1, 'x1', dt), (mask, 'mask', dt), (W, 'W', dt), (out, 'y', odt)):
        if t is not None:
            if not t.is_cuda or t.dtype != want or t.stride(-1) != 1:
                raise RuntimeError(f'linear_forward: {name} must be a row-major {want} GPU tensor')
    with ops._guard(x0):
        rc = _lib.lib().occnerf_linear_forward(
            x0.data_ptr(), x0.stride(0), int(k0), _ptr(x1), 0 if x1 is None else x1.stride(0), int(k1),
            W.data_ptr(), ops._opt(bias, torch.float32, 'bias'), int(relu), _ptr(mask),
            0 if mask is None else mask.stride(0), out.data_ptr(), out.stride(0), int(out_f32 or not bf16),
            int(n_store), _ptr(aux), int(aux_col), int(aux_stride), M, int(n_pad), int(bf16), ops._stream(x0))
    _lib.check(rc, 'linear_forward')
    return out


class _WgradScratch:
    """Partial-sum buffers of the weight-gradient pass, reused across layers and steps."""
    part = None
    dbpart = None

    @classmethod
    def get(cls, G, dev):
        if cls.part is None or cls.part.device != dev or cls.part.shape[0] < G:
            cls.part = torch.empty(G, 256 * 256, device=dev, dtype=torch.float32)
            cls.dbpart = torch.empty(G, 256, device=dev, dtype=torch.float32)
        return cls.part, cls.dbpart


def linear_wgrad(dz, n_pad, x, k_pad, bf16, row_map, col_map, dW, db=None, accumulate=False):
    """dW[row_map[n], col_map[k]] (+)= sum_m dz[m,n] x[m,k];  db[row_map[n]] (+)= sum_m dz[m,n]."""
    M = dz.shape[0]
    lib = _lib.lib()
    G = int(lib.occnerf_linear_wgrad_slices(M))
    part, dbpart = _WgradScratch.get(G, dz.device)
    with ops._guard(dz):
        st = ops._stream(dz)
        rc = lib.occnerf_linear_wgrad(dz.data_ptr(), dz.stride(0), int(n_pad), x.data_ptr(), x.stride(0), int(k_pad), M,
                                      int(bf16), part.data_ptr(), dbpart.data_ptr(), st)
        _lib.check(rc, 'linear_wgrad')
        rc = lib.occnerf_linear_wgrad_reduce(part.data_ptr(), dbpart.data_ptr(), G, int(n_pad), int(k_pad),
                                             row_map.data_ptr(), col_map.data_ptr(),
                                             ops._chk(dW, torch.float32, 'dW'), dW.shape[1],
                                             ops._opt(db, torch.float32, 'db'), int(accumulate), st)
        _lib.check(rc, 'linear_wgrad_reduce')


# ------------------------------------------------------------------ canonical trunks
_maps_cache = {}


def _trunk_maps(dev):
    """Row/column maps of the ten layers' padded matrices (see the layout notes in _Trunks)."""
    hit = _maps_cache.get(dev)
    if hit is not None:
        return hit

    def m(vals):
        return torch.tensor(vals, dtype=torch.int32, device=dev)
    ident = list(range(256))
    x0_cols = list(range(68)) + [-1] * 28                                   # [agg35, var, enc32 | pad]
    geo_rows = list(range(1, 65)) + [0] + [-1] * 31                         # features 1..64, then sigma
    rgb_seg0 = list(range(64)) + [-1] * 32                                  # geometry features; sigma, pad: no input
    rgb_seg1 = [64 + j for j in range(35)] + [-1] + [63 + j for j in range(36, 68)] + [-1] * 28
    out_rows = [0, 1, 2] + [-1] * 29
    maps = {
        'rows': [m(ident)] * 4 + [m(geo_rows)] + [m(ident)] * 4 + [m(out_rows)],
        'cols': [m(x0_cols)] + [m(ident)] * 4 + [m(rgb_seg0 + rgb_seg1)] + [m(ident)] * 4,
        'rgb_seg0': m(rgb_seg0), 'rgb_seg1': m(rgb_seg1),
    }
    _maps_cache[dev] = maps
    return maps


class _Trunks(Function):
    """raw4[M,4] = (rgb logits, sigma) of occnerf_mlp.py:183-199 from agg[M,35], var[M,1], enc[M,32].

    Buffers (element type T = bf16 or fp32, widths padded to 32):
      X0[M,96]   = [agg 35 | var | enc 32 | 0]            input of pts_linears.0 and second segment of rgb_linears.0
      A1..A4     outputs of pts_linears.{0,2,4,6} (ReLU)
      GEO[M,96]  = geo_linear output with its rows permuted: features 1..64 in columns 0..63, sigma in column
                   64 (also written in fp32 to raw4[:,3]); first segment of rgb_linears.0 (whose weight has no
                   column for sigma)
      B1..B4     outputs of rgb_linears.{0,2,4,6}; output_linear writes raw4[:,0:3] in fp32.
    """

    @staticmethod
    def forward(ctx, agg, var, enc, bf16, fused, packed_f32, *wb):
        W, b = wb[:10], wb[10:]
        dev, M, dt = agg.device, agg.shape[0], _dtype(bf16)
        maps = _trunk_maps(dev)
        packs = [linear_pack(W[l].detach().float().contiguous(), b[l].detach().float().contiguous(),
                             maps['rows'][l], maps['cols'][l], bf16) for l in range(10)]
        if bf16 and fused:
            # one kernel for the ten layers, saved activations written on the way (csrc/trunks.hip): same tensors, same layout
            Wf = [w.detach().float().contiguous() for w in W]
            blob = packed_f32 if packed_f32 is not None else ops.canonical_mlp_pack(Wf, [x.detach().float().contiguous() for x in b])
            X0, A, GEO, B, raw4 = ops.trunks_forward_bf16(agg.contiguous(), var.contiguous(), enc.contiguous(), blob,
                                                          ops.trunks_pack_bf16(Wf))
            ctx.bf16 = bf16
            ctx.acts, ctx.GEO, ctx.B = [X0] + A, GEO, B
            ctx.Wt = [p[1] for p in packs]
            ctx.shapes = [tuple(w.shape) for w in W]
            ctx.needs = [w.requires_grad for w in W]
            return raw4
        X0 = torch.zeros(M, 96, device=dev, dtype=dt)
        X0[:, :35] = agg
        X0[:, 35:36] = var
        X0[:, 36:68] = enc
        acts = [X0]
        for l in range(4):
            acts.append(linear_forward(acts[-1], 96 if l == 0 else 256, packs[l][0], 256, bf16, bias=packs[l][2],
                                       relu=True))
        raw4 = torch.empty(M, 4, device=dev, dtype=torch.float32)
        GEO = linear_forward(acts[4], 256, packs[4][0], 96, bf16, bias=packs[4][2], aux=raw4[:, 3:], aux_col=64,
                             aux_stride=4)
        B = [linear_forward(GEO, 96, packs[5][0], 256, bf16, x1=X0, k1=96, bias=packs[5][2], relu=True)]
        for l in range(6, 9):
            B.append(linear_forward(B[-1], 256, packs[l][0], 256, bf16, bias=packs[l][2], relu=True))
        linear_forward(B[3], 256, packs[9][0], 32, bf16, bias=packs[9][2], out=raw4, out_f32=True, n_store=3)
        ctx.bf16 = bf16
        ctx.acts, ctx.GEO, ctx.B = acts, GEO, B
        ctx.Wt = [p[1] for p in packs]
        ctx.shapes = [tuple(w.shape) for w in W]
        ctx.needs = [w.requires_grad for w in W]
        return raw4

    @staticmethod
    def backward(ctx, draw4):
        bf16, acts, GEO, B, Wt = ctx.bf16, ctx.acts, ctx.GEO, ctx.B, ctx.Wt
        X0 = acts[0]
        dev, M, dt = X0.device, X0.shape[0], _dtype(bf16)
        maps = _trunk_maps(dev)
        draw4 = draw4.contiguous().float()
        dW = [torch.empty(s, device=dev, dtype=torch.float32) for s in ctx.shapes]
        db = [torch.empty(s[0], device=dev, dtype=torch.float32) for s in ctx.shapes]

        def wgrad(l, dz, n_pad, x, k_pad, col_map=None, with_db=True):
            linear_wgrad(dz, n_pad, x, k_pad, bf16, maps['rows'][l], maps['cols'][l] if col_map is None else col_map,
                         dW[l], db[l] if with_db else None)

        dz = torch.zeros(M, 32, device=dev, dtype=dt)
        dz[:, :3] = draw4[:, :3]
        wgrad(9, dz, 32, B[3], 256)
        dz = linear_forward(dz, 32, Wt[9], 256, bf16, mask=B[3])
        for l in (8, 7, 6):                                   # rgb_linears.{6,4,2}: input B[l-6]
            wgrad(l, dz, 256, B[l - 6], 256)
            dz = linear_forward(dz, 256, Wt[l], 256, bf16, mask=B[l - 6])
        wgrad(5, dz, 256, GEO, 96, col_map=maps['rgb_seg0'])
        wgrad(5, dz, 256, X0, 96, col_map=maps['rgb_seg1'], with_db=False)
        dgeo = linear_forward(dz, 256, Wt[5][:96], 96, bf16)
        dz_rgb0 = dz                                          # (X0's gradient takes both of its consumers at once, below)
        dgeo[:, 64] = draw4[:, 3]
        wgrad(4, dgeo, 96, acts[4], 256)
        dz = linear_forward(dgeo, 96, Wt[4], 256, bf16, mask=acts[4])
        for l in (3, 2, 1):                                   # pts_linears.{6,4,2}: input acts[l]
            wgrad(l, dz, 256, acts[l], 256)
            dz = linear_forward(dz, 256, Wt[l], 256, bf16, mask=acts[l])
        wgrad(0, dz, 256, X0, 96)
        # dX0 = dZ(rgb_linears.0) W5[:, X0 part] + dZ(pts_linears.0) W0 as ONE product over K = 256 + 256 (the kernel's two
        # input segments): two launches + a 75 M-element fp32 add (0.9 GB of traffic, rocprofv3 --pmc) before
        dx0 = linear_forward(dz_rgb0, 256, torch.cat([Wt[5][96:], Wt[0]], dim=1).contiguous(), 96, bf16, x1=dz, k1=256,
                             out_f32=True)
        ctx.acts = ctx.GEO = ctx.B = ctx.Wt = None
        return (dx0[:, :35], None, dx0[:, 36:68], None, None, None) + tuple(dW) + tuple(db)


def canonical_trunks(cm, agg, var, enc, bf16, fused=True, packed_f32=None):
    """cm: CanonicalMLP (its ten Linear layers are the parameters); -> raw4[M,4] fp32.
    fused (bf16 only): the forward as ONE kernel that writes the saved activations on the way (csrc/trunks.hip) instead of
    ten layer passes; packed_f32: the caller's current blob of ops.canonical_mlp_pack (biases, head rows), packed here if None."""
    import torch.nn as nn
    mods = [m for m in cm.pts_linears if isinstance(m, nn.Linear)] + [cm.geo_linear[0]] + \
           [m for m in cm.rgb_linears if isinstance(m, nn.Linear)] + [cm.output_linear[0]]
    if cm.mlp_depth != 4 or cm.mlp_width != 256:
        raise RuntimeError('the HIP training trunks are built for mlp_depth=4, mlp_width=256')
    return _Trunks.apply(agg.float(), var.float(), enc.float(), bool(bf16), bool(fused), packed_f32,
                         *[m.weight for m in mods], *[m.bias for m in mods])


# ------------------------------------------------------------------ compositing
class _Composite(Function):
    @staticmethod
    def forward(ctx, raw5, mask, z_vals, rays8, bgcolor):
        raw5, mask = raw5.contiguous(), mask.contiguous()
        rgb, acc, depth, _, term = ops.composite(raw5, mask, z_vals, rays8, bgcolor, want_term=True)
        ctx.save_for_backward(raw5, mask, z_vals, rays8)
        ctx.bg = np.asarray(bgcolor, dtype=np.float32).reshape(3).copy()
        ctx.mark_non_differentiable(term)
        return rgb, acc, depth, term

    @staticmethod
    def backward(ctx, g_rgb, g_acc, g_depth, _g_term):
        raw5, mask, z_vals, rays8 = ctx.saved_tensors
        n, S = z_vals.shape
        d_raw = torch.empty_like(raw5)
        d_mask = torch.empty_like(mask)
        _bg, pbg = ops._host_f32(ctx.bg, 3)

        def g(t, shape):
            return None if t is None else t.contiguous().float().reshape(shape)
        g_rgb, g_acc, g_depth = g(g_rgb, (n, 3)), g(g_acc, (n,)), g(g_depth, (n,))
        with ops._guard(raw5):
            rc = _lib.lib().occnerf_composite_backward(
                raw5.data_ptr(), mask.data_ptr(), z_vals.data_ptr(), rays8.data_ptr(), pbg, n, S, _ptr(g_rgb),
                _ptr(g_acc), _ptr(g_depth), d_raw.data_ptr(), d_mask.data_ptr(), ops._stream(raw5))
        _lib.check(rc, 'composite_backward')
        return d_raw, d_mask, None, None, None


def composite(raw5, mask, z_vals, rays8, bgcolor):
    """raw5[n*S,5] (or [n,S,5]), mask[n*S] -> rgb[n,3], acc[n], depth[n], term[n] (int32, argmax alpha)."""
    return _Composite.apply(raw5.reshape(-1, 5), mask.reshape(-1), z_vals, rays8, bgcolor)


# ------------------------------------------------------------------ sampler + warp
class _SampleWarp(Function):
    @staticmethod
    def forward(ctx, rays8, S, t_vals, t_rand, Rs, Ts, vol, bbox_min, bbox_scale):
        Rs, Ts, vol = Rs.contiguous().float(), Ts.contiguous().float(), vol.contiguous().float()
        z, xs, mk, _ = ops.sample_warp(rays8, S, t_vals, Rs, Ts, vol, bbox_min, bbox_scale, t_rand=t_rand)
        ctx.save_for_backward(rays8, z, Rs, Ts, vol)
        ctx.box = (np.asarray(bbox_min, np.float32).copy(), np.asarray(bbox_scale, np.float32).copy())
        ctx.mark_non_differentiable(z, xs)
        return z, xs, mk

    @staticmethod
    def backward(ctx, _gz, _gx, g_mask):
        rays8, z, Rs, Ts, vol = ctx.saved_tensors
        n, S = z.shape
        nb, G = Rs.shape[0], vol.shape[-1]
        lib = _lib.lib()
        W = int(lib.occnerf_warp_backward_slices(n * S))
        dvp = torch.empty(W, nb, G, G, G, device=z.device, dtype=torch.float32)
        drt = torch.empty(W, nb, 12, device=z.device, dtype=torch.float32)
        _a, pmin = ops._host_f32(ctx.box[0], 3)
        _b, psc = ops._host_f32(ctx.box[1], 3)
        g_mask = g_mask.contiguous().float()
        with ops._guard(z):
            rc = lib.occnerf_warp_backward(rays8.data_ptr(), n, S, z.data_ptr(), g_mask.data_ptr(), Rs.data_ptr(),
                                           Ts.data_ptr(), vol.data_ptr(), nb, G, pmin, psc, dvp.data_ptr(),
                                           drt.data_ptr(), ops._stream(z))
        _lib.check(rc, 'warp_backward')
        d_vol = torch.zeros_like(vol)
        d_vol[:nb] = dvp.sum(0)
        drt = drt.sum(0)
        return None, None, None, None, drt[:, :9].reshape(nb, 3, 3), drt[:, 9:], d_vol, None, None


def sample_warp(rays8, S, t_vals, t_rand, Rs, Ts, vol, bbox_min, bbox_scale):
    """-> z_vals[n,S], x_skel[n*S,3] (no gradient), mask[n*S] (gradient to Rs, Ts, vol)."""
    return _SampleWarp.apply(rays8, int(S), t_vals, t_rand, Rs, Ts, vol, bbox_min, bbox_scale)


# ------------------------------------------------------------------ aggregation weights
def agg_weights(counter, knn):
    """counter[P] fp32, knn[N,K] int32 -> atts[N,K] (softmax), var[N,1]."""
    N, K = knn.shape
    atts = torch.empty(N, K, device=knn.device, dtype=torch.float32)
    var = torch.empty(N, 1, device=knn.device, dtype=torch.float32)
    with ops._guard(knn):
        rc = _lib.lib().occnerf_agg_weights(ops._chk(counter, torch.float32, 'counter'), ops._chk(knn, torch.int32, 'knn'),
                                            N, K, atts.data_ptr(), var.data_ptr(), ops._stream(knn))
    _lib.check(rc, 'agg_weights')
    return atts, var


# ------------------------------------------------------------------ pose refiner -> motion bases
class _PoseMotionBases(Function):
    """(posevec[69], dst_Rs[24,3,3], dst_Ts[24,3], cnl_gtfms[24,4,4], W0..W4, b0..b4) -> Rs[24,3,3], Ts[24,3] with
    refinement on: the fused forward of the renderer (csrc/preamble.hip pose_motion_bases_kernel) and a fused backward
    to the ten parameters (pose_motion_bases_backward_kernel) -- two launches where torch autograd takes ~560."""

    @staticmethod
    def forward(ctx, posevec, dst_Rs, dst_Ts, cnl_gtfms, *wb):
        W, b = [t.detach().float().contiguous() for t in wb[:5]], [t.detach().float().contiguous() for t in wb[5:]]
        dev = dst_Rs.device
        args = [t.detach().float().contiguous() for t in (posevec.reshape(-1), dst_Rs, dst_Ts, cnl_gtfms)]
        Rs = torch.empty(24, 3, 3, device=dev, dtype=torch.float32)
        Ts = torch.empty(24, 3, device=dev, dtype=torch.float32)
        with ops._guard_dev(dev):
            rc = _lib.lib().occnerf_pose_motion_bases(ops._ptr_table(W, 'W'), ops._ptr_table(b, 'b'), args[0].data_ptr(), 1,
                                                      args[1].data_ptr(), args[2].data_ptr(), args[3].data_ptr(), Rs.data_ptr(),
                                                      Ts.data_ptr(), ops._stream(dst_Rs))
        _lib.check(rc, 'pose_motion_bases')
        ctx.save_for_backward(*args, *W, *b)
        return Rs, Ts

    @staticmethod
    def backward(ctx, dRs, dTs):
        saved = ctx.saved_tensors
        args, W, b = saved[:4], list(saved[4:9]), list(saved[9:14])
        dev = args[1].device
        dRs = (torch.zeros(24, 3, 3, device=dev) if dRs is None else dRs).float().contiguous()
        dTs = (torch.zeros(24, 3, device=dev) if dTs is None else dTs).float().contiguous()
        dW, db = [torch.empty_like(w) for w in W], [torch.empty_like(x) for x in b]
        with ops._guard_dev(dev):
            rc = _lib.lib().occnerf_pose_motion_bases_backward(
                ops._ptr_table(W, 'W'), ops._ptr_table(b, 'b'), args[0].data_ptr(), args[1].data_ptr(), args[2].data_ptr(),
                args[3].data_ptr(), dRs.data_ptr(), dTs.data_ptr(), ops._ptr_table(dW, 'dW'), ops._ptr_table(db, 'db'),
                ops._stream(dRs))
        _lib.check(rc, 'pose_motion_bases_backward')
        return (None, None, None, None) + tuple(dW) + tuple(db)


def pose_motion_bases(pose_decoder, posevec, dst_Rs, dst_Ts, cnl_gtfms):
    """Refined motion bases with gradients to the pose refiner's parameters.  pose_decoder: BodyPoseRefiner (69 -> 256 x4 -> 69)."""
    import torch.nn as nn
    lin = [m for m in pose_decoder.block_mlps if isinstance(m, nn.Linear)]
    if len(lin) != 5 or lin[0].in_features != 69 or lin[0].out_features != 256 or lin[4].out_features != 69:
        raise RuntimeError('pose_motion_bases: the fused kernels are built for the 69 -> 256 x4 -> 69 refiner of occnerf.yaml')
    return _PoseMotionBases.apply(posevec, dst_Rs, dst_Ts, cnl_gtfms, *[m.weight for m in lin], *[m.bias for m in lin])


# ------------------------------------------------------------------ per-point SDF block
class _PointSdf(Function):
    """point_dist[P,1] -> knn_base[P,3] (fp64), sdf[P,1] of network.py:263-284: the renderer's forward kernels (3-NN search +
    occnerf_point_sdf) and one backward kernel (occnerf_point_sdf_backward) -- 3 launches where torch autograd takes ~100."""

    @staticmethod
    def forward(ctx, point_dist, point_base, normals, unit):
        pc = (point_base + point_dist).float().contiguous()
        kidx = ops.knn_small(pc, point_base, 3)
        kb, sdf = ops.point_sdf(pc, point_base, normals, unit, kidx)
        ctx.save_for_backward(pc, point_base, normals, unit, kidx)
        ctx.dist_shape = tuple(point_dist.shape)
        return kb, sdf[:, None]

    @staticmethod
    def backward(ctx, d_kb, d_sdf):
        pc, base, normals, unit, kidx = ctx.saved_tensors
        P = pc.shape[0]
        d_kb = (torch.zeros(P, 3, device=pc.device, dtype=torch.float64) if d_kb is None else d_kb.double()).contiguous()
        d_sdf = (torch.zeros(P, device=pc.device) if d_sdf is None else d_sdf.reshape(-1).float()).contiguous()
        out = torch.empty(P, device=pc.device, dtype=torch.float32)
        with ops._guard(pc):
            rc = _lib.lib().occnerf_point_sdf_backward(pc.data_ptr(), base.data_ptr(), normals.data_ptr(), unit.data_ptr(),
                                                       kidx.data_ptr(), P, d_kb.data_ptr(), d_sdf.data_ptr(), out.data_ptr(),
                                                       ops._stream(pc))
        _lib.check(rc, 'point_sdf_backward')
        return out.reshape(ctx.dist_shape), None, None, None


def point_sdf(net):
    """(knn_base[P,3] fp64, sdf[P,1]) of the network's point cloud with the gradient to net.point_dist."""
    ctx = net._context()
    return _PointSdf.apply(net.point_dist, net.point_base.detach().float().contiguous(), ctx['normals'], ctx['unit'])
