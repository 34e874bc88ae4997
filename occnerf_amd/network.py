"""`Network`: the renderer module, drop-in for core/nets/occnerf/network.py:38-623.

Same constructor (no arguments, reads the global cfg), same `generate_neural_points`,
`deploy_mlps_to_secondary_gpus`, `point_cloud`, `forward(**data, iter_val=...)` keyword
surface and output dict, and the same state_dict key names/shapes,
so a reference checkpoint loads with strict=True.  What differs is how a frame is rendered:

  reference                                         here
  ------------------------------------------------  -----------------------------------------
  ray chunks of 32768 x sample chunks of 300000,    one pass over all samples of the frame
  ~150 torch kernels + pykeops per chunk            (HBM holds them), six HIP kernels
  per-point SDF block recomputed in every chunk     hoisted: once per frame (network.py:263-284)
  fourier embedding of xyz computed, never used     skipped (network.py:287; occnerf_mlp.py:180)
  nn.DataParallel over samples                      one process per GPU, rays sharded
                                                    (occnerf_amd/parallel.py)

Per-frame small stuff (pose decoder, motion bases, motion-weight volume) stays torch
(occnerf_amd/modules.py); everything per sample goes through the C ABI (occnerf_amd/ops.py).
There is no CPU path: forward() raises if the parameters are not on a GPU.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import geometry, ops
from .canonical_mlp import CanonicalMLP
from .config import get_cfg
from .rayorder import ray_patch_order
from .modules import (BodyPoseRefiner, MotionBasisComputer, MotionWeightVolumeDecoder,
                      NonRigidMotionMLP, hann_window_weights)


class _Replica(nn.Module):
    """Keeps the `.module.` infix nn.DataParallel puts into the reference's state_dict keys
    (cnl_mlp.module.*, non_rigid_mlp.module.*) without any of its behaviour."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


def _load_smpl(cfg):
    """The SMPL body: the licensed model when present, else the synthetic stand-in."""
    choice = cfg.get('smpl_model', 'auto')
    model_dir = './third_parties/smpl/models' if choice in ('auto', 'synthetic') else choice
    pkl = os.path.join(model_dir, 'basicModel_neutral_lbs_10_207_0_v1.0.0.pkl')
    if choice != 'synthetic' and os.path.exists(pkl):
        from third_parties.smpl.smpl_numpy import SMPL   # the user's reference checkout
        return SMPL(sex='neutral', model_dir=model_dir)
    from .synth import SyntheticSMPL
    return SyntheticSMPL()


class Network(nn.Module):
    def __init__(self, avg_betas=None):
        super().__init__()
        cfg = get_cfg()
        self.cfg = cfg
        self.motion_basis_computer = MotionBasisComputer(total_bones=cfg.total_bones)
        self.mweight_vol_decoder = MotionWeightVolumeDecoder(
            embedding_size=cfg.mweight_volume.embedding_size,
            volume_size=cfg.mweight_volume.volume_size, total_bones=cfg.total_bones)
        nr = cfg.non_rigid_motion_mlp
        self.non_rigid_mlp = _Replica(NonRigidMotionMLP(
            pos_embed_size=nr.multires * 6, condition_code_size=nr.condition_code_size,
            mlp_width=nr.mlp_width, mlp_depth=nr.mlp_depth, skips=nr.skips))
        self.pose_decoder = BodyPoseRefiner(
            embedding_size=cfg.pose_decoder.embedding_size, mlp_width=cfg.pose_decoder.mlp_width,
            mlp_depth=cfg.pose_decoder.mlp_depth, total_bones=cfg.total_bones)
        self._ctx = None           # device-side constants of the sample pipeline
        self._ray_orders = {}      # ray_order_key -> (Morton permutation, its inverse)
        self._wconst = None        # decoded volume logits + per-point table (functions of the weights only)
        self._packed = None        # MFMA-ordered MLP weights (eval: cached)

    # ------------------------------------------------------------------ model set-up
    def generate_neural_points(self, avg_betas):
        """network.py:90-146: body points, normals, bound, 3 FPS scales, canonical MLP."""
        cfg = self.cfg
        self.smpl = _load_smpl(cfg)
        betas = np.zeros(10) if avg_betas is None else np.asarray(avg_betas)
        verts, joints = self.smpl(np.zeros(72,), betas)
        try:
            import trimesh
            normals = np.asarray(trimesh.Trimesh(vertices=verts, faces=self.smpl.faces, process=False,
                                                 maintain_order=True).vertex_normals)
        except ImportError:
            normals = geometry.vertex_normals(verts, self.smpl.faces)
        min_xyz = np.min(joints, axis=0) - cfg.bbox_offset
        max_xyz = np.max(joints, axis=0) + cfg.bbox_offset
        self.bound = np.max(np.abs(list(min_xyz) + list(max_xyz)))
        self.detailed_bound = torch.tensor(np.array([list(min_xyz), list(max_xyz)]))

        self.point_base = nn.Parameter(torch.tensor(verts).float(), requires_grad=False)
        self.point_dist = nn.Parameter(torch.zeros(verts.shape[0], 1).float(), requires_grad=True)
        self.point_dist.data.uniform_(-1e-4, 1e-4)
        self.point_counter = nn.Parameter(torch.ones(verts.shape[0]), requires_grad=False)
        self.point_norms = torch.tensor(normals)            # float64, like trimesh's

        self.fps_index, ratio = [], 1.0
        for _ in range(3):                                   # 1/4, 1/16, 1/64 (network.py:113-118)
            ratio /= 4
            self.fps_index.append(torch.from_numpy(geometry.farthest_point_sampling(verts, ratio)))

        self.cnl_mlp = _Replica(CanonicalMLP(
            mlp_depth=cfg.canonical_mlp.mlp_depth, mlp_width=cfg.canonical_mlp.mlp_width,
            input_ch=63, skips=[], bound=self.bound, detailed_bound=self.detailed_bound))
        self._ctx = self._packed = self._wconst = None

    def deploy_mlps_to_secondary_gpus(self):
        return self          # single device per process; rays are sharded across processes

    @property
    def point_cloud(self):
        return self.point_base + self.point_dist

    def _f16x3_flag_word(self):
        """A zero-initialised device word for f16x3 kernels launched outside the render path (the bf16 training step's offsets)."""
        w = self.__dict__.get('_f16x3_word')
        if w is None or w.device != self.point_base.device:
            w = self.__dict__['_f16x3_word'] = torch.zeros(1, device=self.point_base.device, dtype=torch.int32)
        return w

    def _f16x3_enqueue_check(self, flag, what):
        """Copy the flag word to pinned memory behind the work issued so far, clear it, and queue the copy for check_f16x3_domain."""
        self._frames_rendered = getattr(self, '_frames_rendered', 0) + 1
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(flag, non_blocking=True)
        flag.zero_()
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(flag.device))
        self.__dict__.setdefault('_f16x3_pending', []).append((done, host, f'{what} {self._frames_rendered}'))

    def check_f16x3_domain(self, wait=True):
        """cfg.f16x3_domain_check = 'deferred': look at the out-of-domain flags of the frames rendered so far (wait=False: only
        those whose copy has completed).  Raises RuntimeError naming the first frame whose activations left the f16x3 domain --
        its pixels were computed with saturated activations and must be rendered again with mlp_precision = 'fp32'."""
        pending = self.__dict__.get('_f16x3_pending')
        while pending:
            done, host, frame_no = pending[0]
            if not wait and not done.query():
                return
            done.synchronize()
            pending.pop(0)
            if int(host[0]) != 0:
                pending.clear()
                raise RuntimeError(f"occnerf_amd: {frame_no} of this Network left the domain of the f16x3 kernels (a hidden activation "
                                   'reached 4 094): its results are not fp32-grade -- use mlp_precision=\'fp32\' / '
                                   "train_nonrigid_f16x3=False (a render with cfg.f16x3_domain_check=True falls back by itself, at the "
                                   'price of one wait per frame)')

    def invalidate_cache(self):
        """Drop the device-side constants and packed weights (load_state_dict and .to() do; in-place weight updates
        are noticed by themselves through the parameters' version counters)."""
        self._ctx = self._packed = self._wconst = None
        self._ray_orders = {}
        # the captured per-step graphs read the dropped constants at baked-in addresses (train_graph.py): drop them too
        self.__dict__.pop('_per_step_graph', None)

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.invalidate_cache()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.invalidate_cache()
        return out

    # ------------------------------------------------------------------ device constants
    def _context(self):
        """Padded multi-scale point array, index map, float64 normals: constant per model."""
        dev = self.point_base.device
        if self._ctx is not None and self._ctx['device'] == dev:
            return self._ctx
        base = self.point_base.detach()
        P = base.shape[0]
        sets = [torch.arange(P)] + [f.long() for f in self.fps_index]
        rows, imap, begin = [], [], [0]
        for idx in sets:
            pts = base[idx.to(dev)]
            pad = (-pts.shape[0]) % 4                       # +inf rows: never selected
            rows.append(torch.cat([pts, torch.full((pad, 3), float('inf'), device=dev)]))
            imap.append(torch.cat([idx, torch.zeros(pad, dtype=idx.dtype)]))
            begin.append(begin[-1] + pts.shape[0] + pad)
        pts4 = torch.cat(rows)
        pts4 = torch.cat([pts4, torch.zeros(pts4.shape[0], 1, device=dev)], dim=1).contiguous()
        # scale s may reuse the search radius found at scale s+1 when it contains it
        as_sets = [set(s.tolist()) for s in sets]
        seed = [int(l + 1 < len(sets) and as_sets[l + 1] <= as_sets[l]) for l in range(len(sets))]
        normals = self.point_norms.to(dev).double().contiguous()
        cl = geometry.build_knn_clusters(base.cpu().numpy(), [s.numpy() for s in sets])
        clusters = {k: (torch.from_numpy(np.ascontiguousarray(v)).to(dev) if k in
                        ('points', 'index_map', 'centers', 'ranges', 'radius', 'group_centers', 'group_ranges', 'group_radius') else v)
                    for k, v in cl.items()}
        self._ctx = {
            'clusters': clusters,
            'device': dev, 'points': pts4, 'index_map': torch.cat(imap).int().to(dev),
            'scale_begin': begin, 'seed': seed, 'normals': normals,
            'unit': ops.unit_normals(normals),
            'bound32': float(np.float32(self.bound)),
            'two_bound32': float(np.float32(2 * np.float64(self.bound))),
        }
        return self._ctx

    def _packed_weights(self):
        """MFMA-ordered MLP weights, repacked whenever a packed parameter changed (optimizer steps update
        the tensors in place: their `_version` counters move, whatever mode the module is in)."""
        cm, nr = self.cnl_mlp.module, self.non_rigid_mlp.module
        nr_lin = [m for m in nr.block_mlps if isinstance(m, nn.Linear)]
        cw, cb = cm.linear_params()
        prec = str(self.cfg.get('mlp_precision', 'fp32'))
        if prec not in ('fp32', 'bf16x3', 'f16x3'):
            raise RuntimeError(f"cfg.mlp_precision must be 'fp32', 'f16x3' or 'bf16x3', got {prec!r}")
        # the split-operand kernels take their weights as a second, 2-byte stream: bf16 pieces (bf16x3) or fp16 pieces with a
        # scaled low part (f16x3, fp32-grade: csrc/split.h); the ops dispatch on that stream's dtype
        pack_c = {'bf16x3': ops.canonical_mlp_pack_bf16, 'f16x3': ops.canonical_mlp_pack_f16}.get(prec)
        pack_n = {'bf16x3': ops.nonrigid_pack_bf16, 'f16x3': ops.nonrigid_pack_f16}.get(prec)
        srcs = cw + cb + [m.weight for m in nr_lin] + [m.bias for m in nr_lin]
        key = (prec, str(self.cfg.get('f16x3_domain_check', True))) + tuple((t.data_ptr(), t._version) for t in srcs)
        if self._packed is not None and self._packed['key'] == key:
            return self._packed
        self._packed = {
            'key': key,
            'cnl': ops.canonical_mlp_pack(cw, cb),
            'cnl_bf16': pack_c(cw) if pack_c else None,
            'nr': ops.nonrigid_pack([m.weight.detach() for m in nr_lin],
                                    [m.bias.detach() for m in nr_lin]),
            'nr_bf16': pack_n([m.weight.detach() for m in nr_lin]) if pack_n else None,
            'nr_w0': nr_lin[0].weight.detach(), 'nr_b0': nr_lin[0].bias.detach(),
            # f16x3: the word its kernels set when a hidden activation reaches the mode's clamp (4 094; csrc/split.h)
            'domain_flag': torch.zeros(1, device=cw[0].device, dtype=torch.int32)
            if (prec == 'f16x3' and self.cfg.get('f16x3_domain_check', True)) else None,
        }
        return self._packed

    def _weight_constants(self):
        """Everything of the per-frame preamble that is a function of the weights alone, cached per weight version:
        the decoded motion-weight logits (deconv_vol_decoder.py:25-31: the decoder's input is the constant embedding, only
        `log prior` changes from frame to frame) and the per-point feature table (network.py:263-284 +
        occnerf_mlp.py:171-175, which the reference recomputes in every sample chunk)."""
        dec_mod = self.mweight_vol_decoder
        enc = self.cnl_mlp.module.encoder
        srcs = list(dec_mod.parameters()) + [self.point_dist, self.point_base, enc.embeddings]
        key = tuple((t.data_ptr(), t._version) for t in srcs)
        if self._wconst is not None and self._wconst['key'] == key:
            return self._wconst
        with torch.no_grad():
            dec = dec_mod.decoder.forward_gemm(dec_mod.const_embedding[None])[0].contiguous()
            table = self._point_stage(self._context())
        self._wconst = {'key': key, 'dec': dec, 'table': table}
        return self._wconst

    def _point_pack(self, wc):
        """ops.point_pack of the cached table: per weight version AND per visibility-counter version (the counter moves
        in training steps only, trainer-side validation renders see the new counts)."""
        srcs = [self.point_counter, self.point_base]
        key = (wc['table'].data_ptr(),) + tuple((t.data_ptr(), t._version) for t in srcs)
        if wc.get('pack_key') != key:
            ctx = self._context()
            wc['pack'] = ops.point_pack(self.point_base.detach(), ctx['normals'], ctx['unit'],
                                        self.point_counter.detach(), wc['table'])
            wc['pack_key'] = key
        return wc['pack']

    def _point_stage(self, ctx):
        """network.py:263-284 + occnerf_mlp.py:171-175, once per frame."""
        enc = self.cnl_mlp.module.encoder
        pc = self.point_cloud.detach().float().contiguous()
        base = self.point_base.detach()
        kidx = ops.knn_small(pc, base, 3)
        knn_base, sdf = ops.point_sdf(pc, base, ctx['normals'], ctx['unit'], kidx)
        table = ops.point_table(knn_base, sdf, pc, ctx['bound32'], ctx['two_bound32'],
                                enc.embeddings.detach(), enc.offsets, enc.log2_per_level_scale,
                                enc.base_resolution)
        return table

    # ------------------------------------------------------------------ sample pipeline
    def _stage_features(self, rays8, z, xyz, mask, pk, cond, hann, table, pack, center=None):
        """First half of a pass on the live-sample list (VALU / texture-path kernels + the non-rigid MLP): live list, non-rigid
        offsets, repeated-sample heads, kNN, features.  -> state for `_stage_mlp_composite`.  List and count of the live samples
        stay on the device: no host round trip in the frame."""
        cfg, ctx = self.cfg, self._context()
        S = int(cfg.N_samples)
        enc = self.cnl_mlp.module.encoder
        rows, count = ops.live_rows(mask)
        self.last_live_count = count
        dedup = bool(cfg.get('dedup_repeated_samples', True))
        # Repeated samples (ops.repeat_heads): consecutive live samples with a bitwise identical canonical position
        # share the neighbour search and the features, consecutive feature rows that are bitwise identical share the
        # MLP result.
        # Each distinct input is evaluated once and every sample receives its head's result: bit-identical pixels
        # (cfg.dedup_repeated_samples=False evaluates every live sample; tested).
        scan_a = scan_b = mrows = mcount = None
        frows, fcount, kmask = rows, count, mask
        if not cfg.ignore_non_rigid_motions:
            if pk['nr_bf16'] is not None:      # opt-in split-bf16 MFMA path (cfg.mlp_precision): same list, same count
                ops.nonrigid_bf16x3_rows(xyz, rows, count, cond, hann, pk['nr_w0'], pk['nr_b0'], pk['nr'], pk['nr_bf16'],
                                         domain_flag=pk['domain_flag'])
            else:
                ops.nonrigid_rows(xyz, rows, count, cond, hann, pk['nr_w0'], pk['nr_b0'], pk['nr'])
        if dedup:       # (the positions before the offset differ in their last bits; after it they coincide)
            scan_a, frows, fcount, kmask = ops.repeat_heads(xyz, 3, count, rows=rows,
                                                            want_mask=not cfg.get('knn_query_list', True))
            # (the positions' repeats that are not neighbours in the list are few -- 9.28 M -> 8.39 M on the benchmark
            # frame -- and finding them, 1.7 ms, costs more than the kNN + feature work they save, 0.6 ms)
            if cfg.get('dedup_global_positions', False):
                frows, fcount = ops.unique_heads(xyz, 3, frows, fcount, scan=scan_a, scan_count=count)
        self.last_head_counts = (fcount, None)
        if cfg.get('knn_query_list', True):    # tiles formed over the listed samples only (same indices, tested)
            knn = ops.msknn_clustered(xyz, rays8.shape[0], S, ctx['clusters'], ctx['seed'], rows=frows, count=fcount, center=center)
        else:
            knn = ops.msknn_clustered(xyz, rays8.shape[0], S, ctx['clusters'], ctx['seed'], mask=kmask, center=center)
        if center is not None and len(center) > 3 and center[3] != self._center_stamp(table):
            raise RuntimeError('stale kNN centre: its cached feature row was computed from another table / counter / embedding '
                               'version than this frame\'s (ops.sample_features freshness contract)')
        mlp_in, raw_c, _ = ops.sample_features(
            xyz, knn, self.point_base.detach(), ctx['normals'], ctx['unit'],
            self.point_counter.detach(), table, ctx['bound32'], ctx['two_bound32'],
            enc.embeddings.detach(), enc.offsets, enc.log2_per_level_scale, enc.base_resolution,
            rows=frows, count=fcount, pack=pack, center=None if center is None else center[0],
            center_agg=None if center is None else center[2])
        del knn
        if dedup:
            scan_b, mrows, mcount, _ = ops.repeat_heads(mlp_in, 68, fcount)
            if cfg.get('dedup_global', True):       # ... and the repeats that are not neighbours in the list
                mrows, mcount = ops.unique_heads(mlp_in, 68, mrows, mcount, scan=scan_b, scan_count=fcount)
            self.last_head_counts = (fcount, mcount)
        return {'dedup': dedup, 'rays8': rays8, 'z': z, 'mask': mask, 'rows': rows, 'count': count, 'mlp_in': mlp_in,
                'raw_c': raw_c, 'scan_a': scan_a, 'scan_b': scan_b, 'mrows': mrows, 'mcount': mcount, 'cnl': pk['cnl'],
                'cnl_bf16': pk['cnl_bf16'], 'domain_flag': pk['domain_flag'], 'N': xyz.shape[0]}

    @staticmethod
    def _stage_mlp_composite(st, bgcolor, out, out_rows):
        """Second half of a pass (the matrix-pipe kernel + the per-ray scan): canonical MLP on the feature rows, results back
        to their samples, alpha compositing into the frame's rows."""
        dev, N = st['mlp_in'].device, st['N']

        def mlp(raw_out, count, in_rows=None):
            if st['cnl_bf16'] is not None:          # opt-in split-bf16 MFMA path (cfg.mlp_precision)
                ops.canonical_mlp_bf16x3(st['mlp_in'], st['cnl'], st['cnl_bf16'], raw_out, count=count, in_rows=in_rows,
                                         domain_flag=st['domain_flag'])
            else:
                ops.canonical_mlp(st['mlp_in'], st['cnl'], raw_out, count=count, in_rows=in_rows)
        if st['dedup']:
            raw_h = torch.empty(st['mlp_in'].shape[0], 5, device=dev)
            mlp(raw_h, st['mcount'], st['mrows'])
            raw = ops.scatter_raw_heads(raw_h, st['raw_c'], st['rows'], st['count'], st['scan_a'], st['scan_b'],
                                        torch.zeros(N, 5, device=dev))
        else:
            mlp(st['raw_c'], st['count'])
            raw = ops.scatter_raw(st['raw_c'], st['rows'], st['count'], torch.zeros(N, 5, device=dev))
        st['mlp_in'] = None
        return ops.composite(raw, st['mask'], st['z'], st['rays8'], bgcolor, out=out, out_rows=out_rows)[:3]

    def _nonrigid_f16_pack(self):
        """Split-fp16 operand stream of the non-rigid MLP for the bf16 training step (kept beside the fp32 pack, same key)."""
        pk = self._packed_weights()
        if pk.get('nr_f16_train') is None:
            nr_lin = [m for m in self.non_rigid_mlp.module.block_mlps if isinstance(m, nn.Linear)]
            pk['nr_f16_train'] = pk['nr_bf16'] if (pk['nr_bf16'] is not None and pk['nr_bf16'].dtype == torch.float16) else \
                ops.nonrigid_pack_f16([m.weight.detach() for m in nr_lin])
        return pk['nr_f16_train']

    def _knn_center_lists(self, cond, hann):
        """(center[4], idx[4,10]) of ops.knn_center at this frame's collapse point (see _knn_center): what the kNN kernel's
        centre cache takes -- without the cached feature row the renderer adds."""
        ctx, pk = self._context(), self._packed_weights()
        c = torch.zeros(64, 3, device=self.point_base.device)
        if not self.cfg.ignore_non_rigid_motions:
            c = ops.nonrigid(c, cond, hann, pk['nr_w0'], pk['nr_b0'], pk['nr'], direct=True)
        return ops.knn_center(c[0].contiguous(), ctx['points'], ctx['index_map'], ctx['scale_begin'])

    def _knn_center(self, cond, hann, table, pack):
        """(center, idx) of ops.knn_center for this frame's collapse point: wherever a sample's motion-weight sum is far below
        the 1e-4 clamp of the reference's warp (network.py:388) its warped position lands within a micrometre of the origin,
        hence -- after the non-rigid offset -- of offset(0): two thirds of a frame's live samples.  The kNN kernel hands those
        queries c's neighbour lists when they lie inside the radius in which the lists provably hold (exact: DESIGN.md 3.2)."""
        ctx, pk = self._context(), self._packed_weights()
        c = torch.zeros(64, 3, device=self.point_base.device)
        if not self.cfg.ignore_non_rigid_motions:
            # (any point near the cluster serves as c -- the radius is proven for whatever c is searched -- so the first-version
            # kernel's offset(0), equal to the LDS-staged kernel's to fp32 rounding, does; it keeps the per-kernel profiles of the
            # frame's one big nr16 launch clean)
            c = ops.nonrigid(c, cond, hann, pk['nr_w0'], pk['nr_b0'], pk['nr'], direct=True)
        center, idx = ops.knn_center(c[0].contiguous(), ctx['points'], ctx['index_map'], ctx['scale_begin'])
        # ... and the 36 leading feature columns every sample with c's neighbour lists has (functions of the 40 ids, the
        # visibility counts and the per-point table alone): the feature kernel on samples at c
        enc = self.cnl_mlp.module.encoder
        row, _, enc_in = ops.sample_features(
            c[:8].contiguous(), idx[None].expand(8, -1, -1).contiguous(), self.point_base.detach(), ctx['normals'], ctx['unit'],
            self.point_counter.detach(), table, ctx['bound32'], ctx['two_bound32'], enc.embeddings.detach(), enc.offsets,
            enc.log2_per_level_scale, enc.base_resolution, pack=pack, want_enc_in=True)
        # (the 4th entry pins what the cached row was computed from: ops.sample_features' freshness contract)
        return center, idx, ops.center_row(row[0], enc_in[0]), self._center_stamp(table)

    def _center_stamp(self, table):
        return (table.data_ptr(), table._version, self.point_counter._version, self.cnl_mlp.module.encoder.embeddings._version)

    def _render_rays(self, rays8, Rs, Ts, vol, bbox_min, bbox_scale, bgcolor, cond, hann,
                     table, t_rand=None, out=None, out_rows=None, pack=None, boxes=None, center=None):
        """out: (rgb[R,3], alpha[R], depth[R]) of the whole frame; this pass's rays land in rows out_rows (their index
        in the caller's order) or, without a permutation, in the slice the caller passes."""
        cfg, ctx = self.cfg, self._context()
        S = int(cfg.N_samples)
        enc = self.cnl_mlp.module.encoder
        t_vals = torch.linspace(0., 1., steps=S, device=rays8.device)
        z, xyz, mask, _ = ops.sample_warp(rays8, S, t_vals, Rs, Ts, vol, bbox_min, bbox_scale,
                                          t_rand=t_rand, boxes=boxes)
        pk = self._packed_weights()
        # Samples whose motion-weight sum is exactly 0 (outside every bone's prior support or outside the
        # canonical volume) cannot contribute: their alpha is multiplied by that sum (network.py:330).  They
        # are dropped here -- a quarter of the samples of the benchmark frame -- and the pixel values are
        # bit-identical to evaluating them (cfg.skip_empty_samples=False; tested).
        N = xyz.shape[0]
        if cfg.get('skip_empty_samples', True) and cfg.get('knn_culling', True):
            # (fp32 and the opt-in split-bf16 kernels alike: list and count of the live samples on the device, no host sync)
            st = self._stage_features(rays8, z, xyz, mask, pk, cond, hann, table, pack, center)
            return self._stage_mlp_composite(st, bgcolor, out, out_rows)

        # every sample evaluated (cfg.skip_empty_samples off) and / or the brute-force neighbour search (cfg.knn_culling off)
        if not cfg.ignore_non_rigid_motions:
            if pk['nr_bf16'] is not None:
                ops.nonrigid_bf16x3(xyz, cond, hann, pk['nr_w0'], pk['nr_b0'], pk['nr'], pk['nr_bf16'], out=xyz, domain_flag=pk['domain_flag'])
            else:
                ops.nonrigid(xyz, cond, hann, pk['nr_w0'], pk['nr_b0'], pk['nr'], out=xyz)
        if cfg.get('knn_culling', True):     # same results, ~5x fewer distance evaluations
            knn = ops.msknn_clustered(xyz, rays8.shape[0], S, ctx['clusters'], ctx['seed'], center=center)
        else:
            knn = ops.msknn(xyz, ctx['points'], ctx['index_map'], ctx['scale_begin'], ctx['seed'])
        mlp_in, raw, _ = ops.sample_features(
            xyz, knn, self.point_base.detach(), ctx['normals'], ctx['unit'],
            self.point_counter.detach(), table, ctx['bound32'], ctx['two_bound32'],
            enc.embeddings.detach(), enc.offsets, enc.log2_per_level_scale, enc.base_resolution, pack=pack)
        del knn
        if pk['cnl_bf16'] is not None:          # opt-in split-bf16 MFMA path (cfg.mlp_precision)
            ops.canonical_mlp_bf16x3(mlp_in, pk['cnl'], pk['cnl_bf16'], raw, domain_flag=pk['domain_flag'])
        else:
            ops.canonical_mlp(mlp_in, pk['cnl'], raw)
        del mlp_in
        return ops.composite(raw, mask, z, rays8, bgcolor, out=out, out_rows=out_rows)[:3]

    @staticmethod
    def _ray_patch_order(rays_d):
        """Morton walk of the rays (occnerf_amd/rayorder.py): 64 consecutive rays form a compact ~8x8 pixel patch.
        Outputs are returned in the caller's order."""
        return ray_patch_order(rays_d)

    @staticmethod
    def _host3(v):
        """Per-frame float[3] constants (bbox min / scale, background colour) as host float32: the kernels take
        them by value (ops.host_float3: no copy for host arrays, one per tensor object for device tensors)."""
        return ops.host_float3(v)

    @torch.no_grad()
    def live_samples_per_ray(self, rays, dst_Rs, dst_Ts, cnl_gtfms, motion_weights_priors, dst_posevec=None,
                             near=None, far=None, iter_val=1e7, **kwargs):
        """Number of samples with a non-zero motion-weight sum on each of the given rays (int64 [n], on the device, no
        host round trip): the per-frame preamble and the sampler/warp kernel only -- what a frame costs per ray, since
        every later stage runs on the live samples.  occnerf_amd/parallel.py probes a sixteenth of a frame's rays with it
        to deal rays to GPUs by cost."""
        cfg, dev = self.cfg, self.point_base.device
        f32 = lambda t: t.detach().float().contiguous().to(dev)          # noqa: E731
        wc = self._weight_constants()
        refine = iter_val >= cfg.pose_decoder.get('kick_in_iter', 0)
        Rs, Ts = ops.pose_motion_bases(self.pose_decoder, f32(dst_posevec).reshape(-1), refine, f32(dst_Rs), f32(dst_Ts),
                                       f32(cnl_gtfms))
        vol = ops.prior_softmax(wc['dec'], f32(motion_weights_priors))
        rays8 = ops.pack_rays(f32(rays).reshape(2, -1, 3), f32(near).reshape(-1), f32(far).reshape(-1), None)
        S = int(cfg.N_samples)
        t_vals = torch.linspace(0., 1., steps=S, device=dev)
        boxes = ops.bone_boxes(vol, Rs.shape[0]) if cfg.get('warp_bone_culling', True) else None
        _, _, mask, _ = ops.sample_warp(rays8, S, t_vals, Rs, Ts, vol, self._host3(kwargs['cnl_bbox_min_xyz']),
                                        self._host3(kwargs['cnl_bbox_scale_xyz']), boxes=boxes)
        return (mask.view(-1, S) != 0).sum(dim=1)

    @torch.no_grad()
    def render_preamble(self, data, iter_val=1e7):
        """(Rs[24,3,3], Ts[24,3], vol[25,G,G,G]) of a frame exactly as the render branch of `forward` computes them (the fused
        kernels of csrc/preamble.hip) -- for parity tools that feed a checker the same per-frame outputs."""
        f32 = lambda t: t.detach().float().contiguous().to(self.point_base.device)          # noqa: E731
        wc = self._weight_constants()
        refine = iter_val >= self.cfg.pose_decoder.get('kick_in_iter', 0)
        Rs, Ts = ops.pose_motion_bases(self.pose_decoder, f32(data['dst_posevec']).reshape(-1), refine, f32(data['dst_Rs']),
                                       f32(data['dst_Ts']), f32(data['cnl_gtfms']))
        return Rs, Ts, ops.prior_softmax(wc['dec'], f32(data['motion_weights_priors']))

    def forward(self, rays, dst_Rs, dst_Ts, cnl_gtfms, motion_weights_priors, dst_posevec=None,
                near=None, far=None, iter_val=1e7, **kwargs):
        cfg = self.cfg
        dev = self.point_base.device
        if dev.type != 'cuda':
            raise RuntimeError('occnerf_amd.Network renders on a GPU only; move the module with '
                               '.cuda() first (the reference has no CPU path either)')
        dst_Rs, dst_Ts = dst_Rs[None], dst_Ts[None]
        dst_posevec, cnl_gtfms = dst_posevec[None], cnl_gtfms[None]
        motion_weights_priors = motion_weights_priors[None]
        want_grad = torch.is_grad_enabled()
        # fused HIP render when no graph is wanted and the module is in eval mode; the staged differentiable
        # path otherwise (training mode under no_grad still produces comp_loss and the counter update, as the
        # reference's `if self.training` branch does, network.py:486)
        fused = not want_grad and not self.training
        R = int(near.numel())
        shape = list(rays[1].shape[:-1])
        if R == 0:                                   # a rank's empty shard / a frame that misses the bbox
            z = torch.zeros(0, device=dev)
            return {'rgb': z.reshape(shape + [3]), 'alpha': z.reshape(shape), 'depth': z.reshape(shape),
                    'comp_loss': torch.zeros(0 if self.training else 1, device=dev)}

        nr = cfg.non_rigid_motion_mlp
        hann = hann_window_weights(nr.multires, iter_val, nr.kick_in_iter, nr.full_band_iter)       # host
        refine = iter_val >= cfg.pose_decoder.get('kick_in_iter', 0)
        bbox_min = self._host3(kwargs['cnl_bbox_min_xyz'])
        bbox_scale = self._host3(kwargs['cnl_bbox_scale_xyz'])
        bgcolor = self._host3(kwargs['bgcolor'])
        S = int(cfg.N_samples)
        rays_o, rays_d = rays

        def morton_order(dirs):
            """Morton order of the rays (kNN tiles, gather locality).  It depends on the camera only: a caller that
            renders many frames from one camera (movement sequences, the benchmark) names it with
            ray_order_key=<hashable> and the permutation (an argsort of R keys) is computed once."""
            if not cfg.get('ray_patch_order', True):
                return None
            key = kwargs.get('ray_order_key')
            hit = self._ray_orders.get(key) if key is not None else None
            if hit is not None and hit.numel() == R and hit.device == dirs.device:
                return hit
            order = self._ray_patch_order(dirs)
            if key is not None:
                if len(self._ray_orders) >= 8:
                    self._ray_orders.clear()
                self._ray_orders[key] = order
            return order

        if fused:
            # ---- render: 3 launches of per-frame preamble (csrc/preamble.hip), then the sample pipeline ----
            self.check_f16x3_domain(wait=False)      # (cfg.f16x3_domain_check = 'deferred': earlier frames whose flag has arrived)
            with torch.no_grad():
                wc = self._weight_constants()
                pack = self._point_pack(wc)
                f32 = lambda t: t.detach().float().contiguous()          # noqa: E731
                cond = f32(dst_posevec).reshape(-1) if iter_val >= nr.kick_in_iter else \
                    torch.zeros(dst_posevec.numel(), device=dev)
                # The collapse point's chain (its non-rigid offset, its neighbour lists and radius, its feature row: three
                # single-workgroup kernels) runs on the caller's stream like everything else.  (Round 5 carried an opt-in
                # side-stream placement and a two-stream chunk pipeline; both measured neutral to negative and were removed in
                # round 6 -- the renderer uses ONE stream, HISTORY.md 3.7.)
                center = None
                if cfg.get('knn_center_cache', True) and cfg.get('knn_culling', True):
                    center = self._knn_center(cond, hann.tolist(), wc['table'], pack)
                Rs, Ts = ops.pose_motion_bases(self.pose_decoder, f32(dst_posevec).reshape(-1), refine, f32(dst_Rs[0]),
                                               f32(dst_Ts[0]), f32(cnl_gtfms[0]))
                vol = ops.prior_softmax(wc['dec'], f32(motion_weights_priors[0]))
                # support box of every bone's weight channel (one tiny launch): the warp kernel skips the bones that cannot
                # reach a wave's samples -- same bits (cfg.warp_bone_culling=False: every bone for every sample)
                boxes = ops.bone_boxes(vol, Rs.shape[0]) if cfg.get('warp_bone_culling', True) else None
                rays_f = f32(torch.stack([rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)]) if not torch.is_tensor(rays) else
                             rays.reshape(2, -1, 3))
                order = morton_order(rays_f[1])
                rays8 = ops.pack_rays(rays_f, f32(near).reshape(-1), f32(far).reshape(-1), order)
                out = (torch.empty(R, 3, device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev))
                # all rays of the frame in as few passes as memory allows
                cap = int(cfg.get('max_samples_per_pass', 1 << 28))
                if R * S > (1 << 26):      # (a frame of > 30 GB: ~470 B per resident sample, at most half of what is free)
                    # free = what the driver reports + the allocator's cached, reusable blocks (after the first large frame
                    # the cache holds the frame's buffers: counting them as used would split later frames further)
                    free = (torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev)
                            - torch.cuda.memory_allocated(dev))
                    cap = min(cap, max(1 << 22, int(free // 2 // 470)))
                rays_per_pass = max(1, cap // S)

                def passes():
                    for i in range(0, R, rays_per_pass):
                        n = min(rays_per_pass, R - i)
                        if order is not None:
                            self._render_rays(rays8[i:i + n], Rs, Ts, vol, bbox_min, bbox_scale, bgcolor, cond, hann.tolist(),
                                              wc['table'], out=out, out_rows=order[i:i + n], pack=pack, boxes=boxes, center=center)
                        else:
                            self._render_rays(rays8[i:i + n], Rs, Ts, vol, bbox_min, bbox_scale, bgcolor, cond, hann.tolist(),
                                              wc['table'], out=tuple(t[i:i + n] for t in out), pack=pack, boxes=boxes, center=center)
                passes()
                # f16x3 is exact-grade only inside its domain (hidden activations below 4 094: csrc/split.h); its kernels report
                # leaving it -- never silently wrong pixels.  cfg.f16x3_domain_check: True / 'sync' (default) reads the flag after
                # the frame (the read waits for the frame) and renders the frame AGAIN with the fp32 kernels when it is set;
                # 'deferred' copies the flag to pinned memory behind the frame and looks at it when a later frame starts (or in
                # check_f16x3_domain()): no wait in a pipelined loop, and a violation RAISES there, naming the frame, because
                # that frame's pixels have already been handed out; False skips the check and the guarantee.
                mode = cfg.get('f16x3_domain_check', True)
                flag = self._packed_weights().get('domain_flag') if (R > 0 and mode) else None
                if flag is not None and mode == 'deferred':
                    self._f16x3_enqueue_check(flag, 'frame')
                elif flag is not None and int(flag.item()) != 0:
                    flag.zero_()
                    self.f16x3_fallback_frames = getattr(self, 'f16x3_fallback_frames', 0) + 1
                    if self.f16x3_fallback_frames == 1:
                        import warnings
                        warnings.warn("occnerf_amd: a hidden activation left the domain of cfg.mlp_precision='f16x3' (>= 4 094); "
                                      'this frame -- and every later one that does -- is rendered again with the fp32 kernels')
                    prec0 = cfg.mlp_precision
                    cfg.mlp_precision = 'fp32'
                    try:
                        passes()
                    finally:
                        cfg.mlp_precision = prec0
                rgb, acc, depth = out
                comp_loss = torch.zeros(1, device=dev)
        else:
            # ---- differentiable path: per-frame modules in torch (gradients to the pose refiner and the volume
            # decoder), then HIP forward + HIP backward per stage (train_path.py) ----
            from . import train_path
            self.check_f16x3_domain(wait=False)      # (the bf16 step's f16x3 offsets: flags of earlier steps that have arrived)
            with torch.set_grad_enabled(want_grad):
                # the per-frame modules always run in fp32 (bone transforms in bf16 would move every sample); a caller's
                # torch.autocast(bfloat16) selects the arithmetic of the MLP trunks only (train_path._use_bf16)
                with torch.autocast('cuda', enabled=False):
                    dst_Rs, dst_Ts, cnl_gtfms = dst_Rs.float(), dst_Ts.float(), cnl_gtfms.float()
                    dst_posevec = dst_posevec.float()
                    cond = dst_posevec if iter_val >= nr.kick_in_iter else torch.zeros_like(dst_posevec)
                    static = None
                    if want_grad and cfg.get('train_graph', True):
                        # pose refiner, Rodrigues, motion bases, volume decoder and the per-point SDF block
                        # -- ~250 tiny launches forward, ~450 backward -- replayed as two hipGraphs (train_graph.py)
                        from . import train_graph
                        static = train_graph.get(self)(refine, dst_posevec, dst_Rs, dst_Ts, cnl_gtfms, motion_weights_priors)
                    if static is not None:
                        Rs, Ts, vol, knn_base, point_sdf = static
                    else:
                        Rs, Ts = train_path.motion_bases(self, refine, dst_posevec, dst_Rs, dst_Ts, cnl_gtfms)
                        vol = self.mweight_vol_decoder(motion_weights_priors=motion_weights_priors.float())[0]
                        knn_base, point_sdf = train_path.point_sdf_block(self)      # once per step (the reference: every chunk)
                rays8 = torch.cat([rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float(),
                                   near.reshape(-1, 1).float(), far.reshape(-1, 1).float()], -1)
                order = morton_order(rays8[:, 3:6])
                rays8 = (rays8[order] if order is not None else rays8).contiguous()
                t_rand = kwargs.get('t_rand')            # optional injected jitter [R,S] (tests)
                if t_rand is not None and order is not None:
                    t_rand = t_rand[order]
                outs = []
                for i in range(0, rays8.shape[0], int(cfg.chunk)):
                    outs.append(train_path.render_rays_autograd(
                        self, rays8[i:i + cfg.chunk], Rs, Ts, vol, bbox_min, bbox_scale, bgcolor, cond.float(),
                        hann.tolist(), None if t_rand is None else t_rand[i:i + cfg.chunk], point_block=(knn_base, point_sdf)))
                rgb, acc, depth, comp_loss = (torch.cat(t, 0) if len(outs) > 1 else outs[0][j]
                                              for j, t in enumerate(zip(*outs)))
                if order is not None:            # back to the caller's ray order
                    inv = torch.empty_like(order)
                    inv[order] = torch.arange(order.numel(), device=order.device)
                    rgb, acc, depth = rgb[inv], acc[inv], depth[inv]
                    if comp_loss.shape[0] == order.numel():
                        comp_loss = comp_loss[inv]
        return {'rgb': rgb.reshape(shape + [3]), 'alpha': acc.reshape(shape),
                'depth': depth.reshape(shape),
                'comp_loss': comp_loss.reshape(-1)}
