"""GPU, SURVEY 8(a) rows: the HIP path (through the C ABI) against the CPU oracle and the reference goldens.

Runs FIRST in the driver's `pytest -m gpu` session (files run in name order): a1-a21 stage by stage and end to end.
Bars (prompt rule 3): bit-exact for integer/index work (hash indices via the encoder on identical inputs, kNN indices, argmax);
floating point within the tolerance written next to each assert; the end-to-end gate is BASELINE.json's 1e-4 per-pixel L-infinity.
"""
import os
import numpy as np
import pytest
import torch

from tests import util
from tests.gpu_util import (DEV, T, same, build_network, frame_to_device, per_frame_cpu, stagewise_oracle_render, _dev_model,
                            _clusters, stagewise_table, _torchrun)

pytestmark = pytest.mark.gpu


def test_library_is_the_hip_build(ops):
    from occnerf_amd import _lib
    assert _lib.lib().occnerf_abi_version() == 5
    assert torch.cuda.is_available() and 'gfx950' in torch.cuda.get_device_properties(0).gcnArchName


def test_network_end_to_end(case):
    """Network.forward (module seam) vs the reference's own output on identical rays and the
    seeded checkpoint: BASELINE.json's gate, 1e-4 per-pixel L-infinity for the random-init
    checkpoint.  The amplified ("trained-like") checkpoint is held to 1e-3: its O(1) hash
    features turn a 1-ulp encoder-input difference into ~3e-4 of feature (see
    tests/test_oracle_golden.py::test_canonical_mlp)."""
    g, ctx, o = case
    net, _ = build_network(int(g['meta.seed']), util.level(g), S=int(g['meta.S']),
                           non_rigid=bool(int(g['meta.non_rigid'])))
    with torch.no_grad():
        out = net(**frame_to_device(g, DEV), iter_val=1e7)
    tol = util.pixel_tol(g)
    print()
    for k in ('rgb', 'alpha', 'depth'):
        got = out[k].cpu().numpy()
        assert got.shape == g['out.' + k].shape
        print(f"   {k:5s}: max |hip - reference| {np.abs(got - g['out.' + k]).max():.3e}   max |hip - cpu oracle| {np.abs(got - o[k]).max():.3e}"
              f"   max |cpu oracle - reference| {np.abs(o[k] - g['out.' + k]).max():.3e}   (gate {tol:g})")
        assert np.abs(got - g['out.' + k]).max() <= tol, k
        # The second checker, the CPU oracle chain.  On the trained-like field (a density head with a gain of 640) two fp32
        # evaluations of the same function differ by more than the gate on a few rays in a thousand -- MEASURED against a
        # float64 run of the reference (profiles/r05_parity_truth.md, test_trained_truth_three_way below: the reference's own
        # fp32 output is up to 8.8e-4 of depth from its float64 output, HIP and the oracle as far, each within 2.4e-4 of the
        # other two).  So there the oracle comparison is held to 3x the gate with the oracle's own torch-CPU preamble, and
        # -- separating the per-frame modules from the per-sample kernels -- to the gate itself when the oracle is fed the
        # HIP preamble's outputs (Rs, Ts, volume; they are pinned against the reference on their own).
        assert np.abs(got - o[k]).max() <= (3 * tol if util.level(g) == 2 else tol), k
    if util.level(g) == 2:
        pre = tuple(t.cpu().numpy() for t in net.render_preamble(frame_to_device(g, DEV)))
        o2 = stagewise_oracle_render(g, ctx, preamble=pre)
        for k in ('rgb', 'alpha', 'depth'):
            e = np.abs(out[k].cpu().numpy() - o2[k]).max()
            print(f"   {k:5s}: max |hip - cpu oracle fed the HIP preamble's outputs| {e:.3e}")
            assert e <= tol, k
    assert out['comp_loss'].numel() == 1


@pytest.mark.parametrize('name', util.GOLDEN_CASES)
def test_per_frame_modules_on_gpu_against_reference(name):
    """Rows a2-a4 on the GPU, directly against what the reference's own modules produced (goldens recorded by
    oracle/ref_harness/make_golden.py): pose refiner -> `pose.Rs`, motion bases -> `mb.Rs`, `mb.Ts`, motion-weight
    volume (the GEMM + HIP gather decoder) -> `mw.vol_slice`, `mw.vol_sum`."""
    from tests.gpu_util import golden_frame
    g = util.load_golden(name)
    net, ctx = build_network(int(g['meta.seed']), util.level(g), S=int(g['meta.S']),
                             non_rigid=bool(int(g['meta.non_rigid'])))
    frame = golden_frame(g)
    d = frame_to_device(frame, DEV)
    with torch.no_grad():
        posevec = d['dst_posevec'][None]
        dst_Rs, dst_Ts = d['dst_Rs'][None], d['dst_Ts'][None]
        if 'pose.Rs' in g:
            refined = net.pose_decoder(posevec)['Rs']
            assert np.abs(refined.cpu().numpy() - g['pose.Rs']).max() <= 1e-6
            no_root = torch.matmul(dst_Rs[:, 1:].reshape(-1, 3, 3), refined.reshape(-1, 3, 3)).reshape(-1, 23, 3, 3)
            dst_Rs = torch.cat([dst_Rs[:, 0:1], no_root], dim=1)
        Rs, Ts = net.motion_basis_computer(dst_Rs, dst_Ts, d['cnl_gtfms'][None])
        assert np.abs(Rs.cpu().numpy() - g['mb.Rs']).max() <= 2e-6
        assert np.abs(Ts.cpu().numpy() - g['mb.Ts']).max() <= 2e-6
        vol = net.mweight_vol_decoder(motion_weights_priors=d['motion_weights_priors'][None])[0]
        assert np.abs(vol[:, ::4, ::4, ::4].cpu().numpy() - g['mw.vol_slice']).max() <= 1e-5
        assert abs(float(vol.double().sum()) - float(g['mw.vol_sum'])) <= 1e-3 * abs(float(g['mw.vol_sum']))
        # the render path's fused preamble (csrc/preamble.hip): one launch for a2 + a3, one for the softmax over
        # (cached decoded logits + log prior) -- against the same reference outputs
        from occnerf_amd import ops as o
        Rs2, Ts2 = o.pose_motion_bases(net.pose_decoder, d['dst_posevec'].float().contiguous(), 'pose.Rs' in g,
                                       d['dst_Rs'].float().contiguous(), d['dst_Ts'].float().contiguous(),
                                       d['cnl_gtfms'].float().contiguous())
        assert np.abs(Rs2.cpu().numpy() - g['mb.Rs'][0]).max() <= 2e-6
        assert np.abs(Ts2.cpu().numpy() - g['mb.Ts'][0]).max() <= 2e-6
        wc = net._weight_constants()
        vol2 = o.prior_softmax(wc['dec'], d['motion_weights_priors'].float().contiguous())
        assert np.abs(vol2[:, ::4, ::4, ::4].cpu().numpy() - g['mw.vol_slice']).max() <= 1e-5
        assert float((vol2 - vol).abs().max()) <= 1e-6
        assert net._weight_constants() is wc                      # cached: same weights, same object
        net.point_dist.add_(1e-3)                                 # an in-place update, as an optimiser step does (under no_grad)
        assert net._weight_constants() is not wc


def test_sample_warp(case, ops):
    g, ctx, o = case
    S = int(g['meta.S'])
    z, xs, mk, pts = ops.sample_warp(T(o['rays8']), S, T(o['t_vals']), T(o['Rs']), T(o['Ts']), T(o['vol']),
                                     g['in.cnl_bbox_min_xyz'], g['in.cnl_bbox_scale_xyz'], want_pts=True)
    same(z.cpu().numpy(), o['z'], 'z_vals')                             # bit-exact vs oracle
    same(pts.cpu().numpy().reshape(o['pts'].shape), o['pts'], 'pts')
    same(mk.cpu().numpy(), o['mask'], 'mask')
    same(xs.cpu().numpy(), o['x_skel'], 'x_skel')
    # and within fp32 reordering of the reference's torch ops
    assert np.abs(z.cpu().numpy() - g['comp.z_vals']).max() == 0
    assert np.abs(mk.cpu().numpy() - g['warp.mask'].ravel()).max() <= 5e-6
    # x_skel = sum(w pos) / clamp(sum w, 1e-4): where the weight sum is ~1e-4 a 1e-7 difference of the
    # (CPU-torch, machine-dependent: this box's host is not the one the goldens were written on) motion-weight volume and
    # bone transforms is amplified by 1 / sum w, so the comparison is on the numerator's scale; plain 1e-4 where the
    # weight sum is not tiny
    dx = np.abs(xs.cpu().numpy() - g['warp.x_skel'].reshape(-1, 3)).max(1)
    den = np.maximum(g['warp.mask'].ravel(), 1e-4)
    print('x_skel vs reference: max', dx.max(), 'max scaled by weight sum', (dx * den).max())
    assert (dx * den).max() <= 5e-6
    assert dx[den >= 1e-2].max(initial=0.0) <= 1e-4


def test_sample_warp_stratified(ops, oracle):
    rng = np.random.RandomState(3)
    n, S = 37, 64
    rays = np.concatenate([rng.randn(n, 3), rng.randn(n, 3), rng.uniform(4, 5, (n, 1)), rng.uniform(6, 7, (n, 1))], 1).astype(np.float32)
    t_vals = torch.linspace(0., 1., steps=S).numpy()
    t_rand = rng.rand(n, S).astype(np.float32)
    Rs = np.tile(np.eye(3, dtype=np.float32), (24, 1, 1))
    Ts = rng.randn(24, 3).astype(np.float32) * 0.1
    vol = rng.rand(25, 8, 8, 8).astype(np.float32)
    bmin, bsc = np.array([-3, -3, -3], np.float32), np.array([0.2, 0.3, 0.25], np.float32)
    z, xs, mk, pts = ops.sample_warp(T(rays), S, T(t_vals), T(Rs), T(Ts), T(vol), bmin, bsc, t_rand=T(t_rand), want_pts=True)
    wz, wpts = oracle.sample_rays(rays, t_vals, t_rand)
    wxs, wmk = oracle.motion_field(wpts, Rs, Ts, vol, bmin, bsc)
    assert np.array_equal(z.cpu().numpy(), wz)
    assert np.array_equal(xs.cpu().numpy(), wxs) and np.array_equal(mk.cpu().numpy(), wmk)


def test_warp_bone_culling_is_exact(ops, oracle):
    """Round 4: the warp kernel skips, per wave of 64 samples of one ray, the bones whose motion-weight channel (its non-zero
    support box from occnerf_bone_boxes, widened by the tap reach) cannot reach those samples.  Every skipped (sample, bone)
    pair would have contributed a weight of exactly +0: z, x_skel and the motion-weight sum are bit-identical to the unculled
    kernel -- on posed frames at S = 64 / 128 / 192, rays that miss the body included -- the support boxes equal numpy's, and
    with S not a multiple of 64 the call falls back to every bone."""
    from occnerf_amd import synth
    ctx = util.model_context(0, False)
    for size, S, pose in ((64, 64, 1), (96, 128, 3), (64, 192, 5), (48, 96, 2)):
        frame = synth.make_frame(img_size=size, pose72=synth.seeded_pose(pose), orbit_frame=17 * pose)
        Rs, Ts, vol, hann, cond = per_frame_cpu(ctx, frame)
        rays8 = T(np.concatenate([frame['rays'][0], frame['rays'][1], frame['near'], frame['far']], -1).astype(np.float32))
        t_vals = torch.linspace(0., 1., steps=S, device=DEV)
        vd = T(vol)
        boxes = ops.bone_boxes(vd, 24)
        v = vol[:24]
        for b in range(24):
            nz = np.argwhere(v[b] != 0)                     # (z, y, x)
            want = [G for G in (32, -1) * 3] if nz.size == 0 else [nz[:, 2].min(), nz[:, 2].max(), nz[:, 1].min(), nz[:, 1].max(),
                                                                    nz[:, 0].min(), nz[:, 0].max()]
            assert boxes[b].tolist() == [int(x) for x in want], b
        a = ops.sample_warp(rays8, S, t_vals, T(Rs), T(Ts), vd, frame['cnl_bbox_min_xyz'], frame['cnl_bbox_scale_xyz'])
        c = ops.sample_warp(rays8, S, t_vals, T(Rs), T(Ts), vd, frame['cnl_bbox_min_xyz'], frame['cnl_bbox_scale_xyz'], boxes=boxes)
        for x, y, name in zip(a[:3], c[:3], ('z', 'x_skel', 'mask')):
            assert torch.equal(x.view(torch.int32), y.view(torch.int32)), (size, S, name)
        assert float(a[2].max()) > 0.5 and float((a[2] == 0).float().mean()) > 0.1
    # a channel that is zero everywhere and one that fills the grid
    vz = vd.clone()
    vz[3] = 0
    vz[5] = 1e-3
    bz = ops.bone_boxes(vz, 24)
    assert bz[3, 0] > bz[3, 1] and bz[5].tolist() == [0, 31, 0, 31, 0, 31]
    a = ops.sample_warp(rays8, 64, torch.linspace(0., 1., 64, device=DEV), T(Rs), T(Ts), vz, frame['cnl_bbox_min_xyz'], frame['cnl_bbox_scale_xyz'])
    c = ops.sample_warp(rays8, 64, torch.linspace(0., 1., 64, device=DEV), T(Rs), T(Ts), vz, frame['cnl_bbox_min_xyz'], frame['cnl_bbox_scale_xyz'], boxes=bz)
    assert all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(a[:3], c[:3]))


def test_nonrigid(case, ops):
    g, ctx, o = case
    W, B = util.nonrigid_params(ctx['sd'])
    Wd, Bd = [T(w) for w in W], [T(b) for b in B]
    packed = ops.nonrigid_pack(Wd, Bd)
    rng = np.random.RandomState(0)
    xyz = g['nr.xyz_in'] if 'nr.xyz_in' in g else rng.uniform(-1, 1, (4099, 3)).astype(np.float32)
    cond = (g['nr.cond'] if 'nr.cond' in g else rng.randn(1, 69) * 0.3).astype(np.float32).ravel()
    from oracle import oracle as orc
    for hann in (np.ones(6, np.float32), np.array([1, 1, 0.75, 0.25, 0, 0], np.float32)):
        got = ops.nonrigid(T(xyz), T(cond), hann, Wd[0], Bd[0], packed).cpu().numpy()
        want = orc.nonrigid(xyz, cond, hann, W, B)
        assert np.abs(got - want).max() <= 1e-6          # sinf/cosf + MFMA k-order vs libm/serial
        gotd = ops.nonrigid(T(xyz), T(cond), hann, Wd[0], Bd[0], packed, direct=True).cpu().numpy()
        assert np.abs(gotd - want).max() <= 1e-6         # the 32-sample-wave direct-load kernel
    for n in (1, 15, 16, 17, 33, 127, 128, 129):         # partial tiles / waves / workgroups, in place
        buf = torch.full((n + 2, 3), 5.0, device=DEV)
        buf[:n] = T(xyz[:n])
        ops.nonrigid(buf[:n], T(cond), np.ones(6, np.float32), Wd[0], Bd[0], packed, out=buf[:n])
        got = buf.cpu().numpy()
        assert np.abs(got[:n] - orc.nonrigid(xyz[:n], cond, np.ones(6, np.float32), W, B)).max() <= 1e-6
        assert (got[n:] == 5.0).all()
    # split-bf16 variant: offsets are <= ~0.1 m (amplified checkpoint); 2^-17 relative split error per
    # product through 7 layers -> a few 1e-6 m at most
    ph = ops.nonrigid_pack_bf16(Wd)
    gotb = ops.nonrigid_bf16x3(T(xyz), T(cond), np.ones(6, np.float32), Wd[0], Bd[0], packed, ph).cpu().numpy()
    assert np.abs(gotb - orc.nonrigid(xyz, cond, np.ones(6, np.float32), W, B)).max() <= 5e-6
    # the fp32-grade split (f16x3): the fp32 kernel's own 1e-6, both window settings, in place on a row list too
    pf = ops.nonrigid_pack_f16(Wd)
    for hann in (np.ones(6, np.float32), np.array([1, 1, 0.75, 0.25, 0, 0], np.float32)):
        gotf = ops.nonrigid_bf16x3(T(xyz), T(cond), hann, Wd[0], Bd[0], packed, pf).cpu().numpy()
        assert np.abs(gotf - orc.nonrigid(xyz, cond, hann, W, B)).max() <= 1e-6
    lrows = torch.arange(0, xyz.shape[0], 3, device=DEV, dtype=torch.int32)
    lcount = torch.tensor([lrows.numel() - 2], device=DEV, dtype=torch.int32)
    inpl = ops.nonrigid_bf16x3_rows(T(xyz).clone(), lrows, lcount, T(cond), np.ones(6, np.float32), Wd[0], Bd[0], packed, pf).cpu().numpy()
    sel = lrows.cpu().numpy()[:lrows.numel() - 2]
    keep = np.ones(xyz.shape[0], bool)
    keep[sel] = False
    assert np.abs(inpl[sel] - orc.nonrigid(xyz[sel], cond, np.ones(6, np.float32), W, B)).max() <= 1e-6
    assert np.array_equal(inpl[keep], xyz[keep])
    if 'nr.xyz_out' in g:                                # what the reference's torch MLP returned
        got = ops.nonrigid(T(xyz), T(cond), np.ones(6, np.float32), Wd[0], Bd[0], packed).cpu().numpy()
        assert np.abs(got - g['nr.xyz_out']).max() <= 1e-6
        gotf = ops.nonrigid_bf16x3(T(xyz), T(cond), np.ones(6, np.float32), Wd[0], Bd[0], packed, pf).cpu().numpy()
        assert np.abs(gotf - g['nr.xyz_out']).max() <= 1e-6


def test_msknn_bit_exact(case, ops):
    g, ctx, o = case
    m = _dev_model(ctx, ops)
    got = ops.msknn(T(o['xyz']), m['points'], m['imap'], m['begin'], m['seed']).cpu().numpy()
    same(got, o['knn'], 'knn vs oracle')
    # on the reference's own query points: exactly the indices the reference got
    gotg = ops.msknn(T(g['cnl.xyz']), m['points'], m['imap'], m['begin'], m['seed']).cpu().numpy()
    same(gotg, g['cnl.knn_idxs'].astype(np.int32), 'knn vs reference golden')
    # the radius carry-over is an optimisation only: same result without it
    got2 = ops.msknn(T(o['xyz']), m['points'], m['imap'], m['begin'], [0, 0, 0, 0]).cpu().numpy()
    same(got2, got, 'knn without radius carry-over')


def test_msknn_clustered_bit_exact(case, ops):
    """Cluster culling changes the work, never the result."""
    g, ctx, o = case
    cl = _clusters(ctx)
    S = int(g['meta.S'])
    n = o['xyz'].shape[0] // S
    for seed in ([1, 1, 1, 0], [0, 0, 0, 0]):
        got = ops.msknn_clustered(T(o['xyz']), n, S, cl, seed).cpu().numpy()
        same(got, o['knn'], f'clustered knn vs oracle (seed={seed})')
    gotg = ops.msknn_clustered(T(g['cnl.xyz']), n, S, cl, [1, 1, 1, 0]).cpu().numpy()
    same(gotg, g['cnl.knn_idxs'].astype(np.int32), 'clustered knn vs reference golden')


def test_msknn_clustered_query_list(ops):
    """Query-list mode (tiles formed over the listed samples of each ray) == mask mode, index for index, on every listed
    sample: ragged lists (rays with 0, 1, S listed samples), a count below the list's capacity, a ray count that is not a
    multiple of 64."""
    ctx = util.model_context(0, False)
    cl = _clusters(ctx)
    n_rays, S = 150, 23
    g = torch.Generator(device='cpu').manual_seed(11)
    q = ((torch.rand(n_rays * S, 3, generator=g) - 0.5) * 1.6).to(DEV)
    keep = torch.rand(n_rays, S, generator=g) < 0.4
    keep[3] = False
    keep[4] = True
    keep[5] = False
    keep[5, 7] = True
    keep[n_rays - 1] = True
    mask = keep.reshape(-1).float().to(DEV)
    rows, count = ops.live_rows(mask)
    want = ops.msknn_clustered(q, n_rays, S, cl, [1, 1, 1, 0], mask=mask)
    got = ops.msknn_clustered(q, n_rays, S, cl, [1, 1, 1, 0], rows=rows, count=count)
    sel = rows[:int(count)].long()
    assert sel.numel() > 500 and torch.equal(got[sel], want[sel])
    # a shorter count: only the first entries are queried, and they still agree
    short = torch.tensor([int(count) // 3], device=DEV, dtype=torch.int32)
    got2 = ops.msknn_clustered(q, n_rays, S, cl, [1, 1, 1, 0], rows=rows, count=short)
    sel2 = rows[:int(short)].long()
    assert torch.equal(got2[sel2], want[sel2])


def test_msknn_clustered_edge_cases(ops, oracle):
    ctx = util.model_context(0, False)
    cl = _clusters(ctx)
    rng = np.random.RandomState(7)
    base = ctx['point_base']
    for n_rays, S in ((37, 13), (5, 128), (64, 8), (1, 1)):      # ragged tiles in both directions
        N = n_rays * S
        q = np.concatenate([rng.uniform(-1.5, 1.5, (N - N // 2, 3)),
                            base[rng.randint(0, len(base), N // 2)] + rng.randn(N // 2, 3) * 1e-6]).astype(np.float32)
        q[::7] = base[rng.randint(0, len(base), len(q[::7]))]            # exact hits (distance 0)
        q[1::11] = rng.uniform(-40, 40, (len(q[1::11]), 3))              # far outside
        rng.shuffle(q)
        got = ops.msknn_clustered(T(q), n_rays, S, cl, [1, 1, 1, 0]).cpu().numpy()
        same(got, oracle.msknn(q, base, ctx['fps'], k=10), f'clustered knn {n_rays}x{S}')


def test_msknn_edge_cases(ops, oracle):
    ctx = util.model_context(0, False)
    m = _dev_model(ctx, ops)
    rng = np.random.RandomState(5)
    base = ctx['point_base']
    q = np.concatenate([
        rng.uniform(-1.5, 1.5, (1000, 3)),                 # anywhere in the bbox
        base[rng.randint(0, len(base), 500)],              # exactly on support points (distance 0)
        base[:300] + rng.randn(300, 3) * 1e-6,             # adversarial near-ties
        rng.uniform(-50, 50, (200, 3)),                    # far outside
        np.zeros((3, 3)),                                  # ragged tail (N % 1024 != 0)
    ]).astype(np.float32)
    got = ops.msknn(T(q), m['points'], m['imap'], m['begin'], m['seed']).cpu().numpy()
    want = oracle.msknn(q, base, ctx['fps'], k=10)
    same(got, want, 'knn edge cases')
    assert ops.msknn(torch.empty(0, 3, device=DEV), m['points'], m['imap'], m['begin'], m['seed']).shape == (0, 4, 10)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_msknn_tie_suite(ops, seed):
    """both HIP kNN kernels on the adversarial tie model (tests/util.py::knn_tie_model -- duplicated support
    points, queries exactly equidistant to up to 24 points at all four scales, the k = 10 cut inside a tie group) return,
    index for index, what the documented KeOps rule gives in exact integer arithmetic (knn.py:77-85: ascending distance, the
    lowest row of the scale's block first): brute force, clustered in mask mode, clustered with a query list, with and
    without the radius carry-over; the k = 3 single-scale kernel too."""
    from occnerf_amd import geometry
    n_rays, S = 64, 8
    base, sets, q, want = util.knn_tie_model(n_rays, S, seed)
    rows, imap, begin = [], [], [0]
    for idx in sets:
        pad = (-len(idx)) % 4
        rows.append(np.concatenate([base[idx], np.full((pad, 3), np.inf, np.float32)]))
        imap.append(np.concatenate([idx, np.zeros(pad, idx.dtype)]))
        begin.append(begin[-1] + len(idx) + pad)
    p4 = np.concatenate(rows)
    p4 = np.concatenate([p4, np.zeros((p4.shape[0], 1), np.float32)], 1)
    contains = [int(l + 1 < 4 and set(sets[l + 1].tolist()) <= set(sets[l].tolist())) for l in range(4)]
    assert contains == [1, 1, 1, 0]
    for seedflags in (contains, [0, 0, 0, 0]):
        got = ops.msknn(T(q), T(p4), T(np.concatenate(imap).astype(np.int32)), begin, seedflags).cpu().numpy()
        same(got, want, f'brute-force kNN on the tie model (carry-over {seedflags})')
    cl = geometry.build_knn_clusters(base, sets)
    cl = {k: (T(v) if k in ('points', 'index_map', 'centers', 'ranges', 'radius', 'group_centers', 'group_ranges', 'group_radius')
              else v) for k, v in cl.items()}
    for seedflags in (contains, [0, 0, 0, 0]):
        got = ops.msknn_clustered(T(q), n_rays, S, cl, seedflags).cpu().numpy()
        same(got, want, f'clustered kNN on the tie model (carry-over {seedflags})')
    keep = np.random.RandomState(seed).rand(n_rays * S) < 0.6
    lrows, count = ops.live_rows(T(keep.astype(np.float32)))
    got = ops.msknn_clustered(T(q), n_rays, S, cl, contains, rows=lrows, count=count).cpu().numpy()
    same(got[keep], want[keep], 'clustered kNN, query-list mode, on the tie model')
    same(ops.knn_small(T(q), T(base), 3).cpu().numpy(), want[:, 0, :3], 'k = 3 kernel on the tie model')


def test_knn_center_cache_is_exact(ops, oracle):
    """Round 4: queries inside the radius ops.knn_center derives for a point c take c's cached neighbour lists instead of a
    search.  (a) queries at 0 ... 0.999 r and 1.001 ... 100 r around several c (on the body, inside it, at the frame's collapse
    point): clustered-with-cache == brute force == oracle, index for index, and the inside ones really equal c's lists;
    (b) a c whose 11 nearest points tie (lattice cell centre of the tie model) gets r = 0;
    (c) the benchmark frame rendered with the cache on and off: test_c_configs_full_size.py."""
    from occnerf_amd import geometry, synth
    ctx = util.model_context(0, False)
    m = _dev_model(ctx, ops)
    cl = _clusters(ctx)
    rng = np.random.RandomState(3)
    base = ctx['point_base']
    served = 0
    for c in (np.array([-4.1e-5, -1.5e-5, -5.7e-6], np.float32), base[100] + np.float32(0.013), np.array([0.2, -0.3, 0.05], np.float32),
              base[4000] * np.float32(0.5), np.array([0.0, 0.45, 0.02], np.float32)):
        center, idx = ops.knn_center(T(c), m['points'], m['imap'], m['begin'])
        r = float(center[3].sqrt())
        assert torch.equal(center[:3].cpu(), torch.from_numpy(c))
        want_c = oracle.msknn(c[None], base, ctx['fps'], k=10)[0]
        same(idx.cpu().numpy(), want_c, 'centre lists')
        if r == 0.0:
            continue
        assert 1e-7 < r < 1e-2
        n_rays, S = 64, 8
        dirs = rng.randn(n_rays * S, 3)
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        rad = rng.choice([0.0, 0.3, 0.9, 0.999, 1.001, 1.5, 3.0, 100.0], n_rays * S)
        q = (c[None].astype(np.float64) + dirs * (rad * r)[:, None]).astype(np.float32)
        want = oracle.msknn(q, base, ctx['fps'], k=10)
        brute = ops.msknn(T(q), m['points'], m['imap'], m['begin'], m['seed']).cpu().numpy()
        same(brute, want, 'brute force near a centre')
        for kw in ({}, {'mask': T((rng.rand(n_rays * S) < 0.7).astype(np.float32))}):
            got = ops.msknn_clustered(T(q), n_rays, S, cl, [1, 1, 1, 0], center=(center, idx), **kw).cpu().numpy()
            keep = np.ones(n_rays * S, bool) if not kw else kw['mask'].cpu().numpy() > 0
            same(got[keep], want[keep], 'clustered kNN with the centre cache')
        inside = np.linalg.norm(q.astype(np.float64) - c, axis=1) < 0.99 * r
        assert inside.sum() > 100 and (want[inside] == want_c[None]).all()
        served += int(inside.sum())
    assert served > 300
    tb, tsets, tq, twant = util.knn_tie_model()
    rows, imap, begin = [], [], [0]
    for ix in tsets:
        pad = (-len(ix)) % 4
        rows.append(np.concatenate([tb[ix], np.full((pad, 3), np.inf, np.float32)]))
        imap.append(np.concatenate([ix, np.zeros(pad, ix.dtype)]))
        begin.append(begin[-1] + len(ix) + pad)
    p4 = np.concatenate(rows)
    p4 = np.concatenate([p4, np.zeros((p4.shape[0], 1), np.float32)], 1)
    cen, _ = ops.knn_center(T(np.array([2.5 / 8, 2.5 / 8, 2.5 / 8], np.float32)), T(p4), T(np.concatenate(imap).astype(np.int32)), begin)
    assert float(cen[3]) == 0.0                                         # 8 equidistant corners: no radius
    # (d) the feature kernel's cached aggregate: samples inside the radius -- whole groups of 8 and mixed groups -- with and
    # without the centre: mlp_in and the signed distance bit for bit
    c = np.array([-4.1e-5, -1.5e-5, -5.7e-6], np.float32)
    center, idx = ops.knn_center(T(c), m['points'], m['imap'], m['begin'])
    r = float(center[3].sqrt())
    n_rays, S = 96, 16
    off = rng.randn(n_rays * S, 3) * (0.2 * r)
    off[:160] = rng.randn(160, 3) * 1e-10                                   # 20 whole groups as close to c as the frame's collapsed samples
    far = rng.rand(n_rays * S) < 0.15
    far[:320] = False                                                       # 40 whole groups of 8 inside
    off[far] = rng.randn(int(far.sum()), 3) * 0.05
    q = T((c[None].astype(np.float64) + off).astype(np.float32))
    knn = ops.msknn_clustered(q, n_rays, S, cl, [1, 1, 1, 0], center=(center, idx))
    table = T(np.concatenate([stagewise_table(ctx, oracle), np.zeros((len(base), ops.table_stride() - 35), np.float32)], 1))
    args = (m['base'], m['normals'], m['unit'], T(ctx['counter']), table, m['b32'], m['tb32'], m['emb'], m['off'], ctx['S'], ctx['H'])
    cm, _, ce = ops.sample_features(T(np.tile(c, (8, 1))), idx[None].expand(8, -1, -1).contiguous(), *args, want_enc_in=True)
    row = ops.center_row(cm[0], ce[0])
    assert row.shape == (72,)
    plain = ops.sample_features(q, knn, *args, want_enc_in=True)
    fast = ops.sample_features(q, knn, *args, center=center, center_agg=row, want_enc_in=True)
    assert torch.equal(plain[0].view(torch.int32), fast[0].view(torch.int32)) and torch.equal(plain[1][:, 4], fast[1][:, 4])
    assert torch.equal(plain[2].view(torch.int32), fast[2].view(torch.int32))
    assert torch.equal(plain[0][:320, :36], row[None, :36].expand(320, -1))      # the inside samples do carry the centre's columns
    # ... and most of them the centre's encoder input, hence its encoded columns (the rest differ in the last bit of x and
    # are encoded as usual)
    same_x = (plain[2][:320].view(torch.int32) == row[36:40].view(torch.int32)).all(1)
    assert float(same_x[:160].float().mean()) > 0.5 and torch.equal(plain[0][:320][same_x][:, 36:], row[None, 40:].expand(int(same_x.sum()), -1))


def test_msknn_cluster_groups_change_nothing(ops):
    """The two-level culling (group spheres first, then the clusters of the groups in reach) against the flat scan over every
    cluster sphere, and across group sizes: index-for-index identical on scattered queries near and far from the body."""
    from occnerf_amd import geometry
    ctx = util.model_context(0, False)
    sets = [np.arange(len(ctx['point_base']))] + [np.asarray(f) for f in ctx['fps']]
    rng = np.random.RandomState(5)
    n_rays, S = 96, 16
    q = (ctx['point_base'][rng.randint(0, 6890, n_rays * S)] + rng.randn(n_rays * S, 3).astype(np.float32) *
         rng.choice([0.002, 0.05, 0.6], (n_rays * S, 1)).astype(np.float32))
    dev = ('points', 'index_map', 'centers', 'ranges', 'radius', 'group_centers', 'group_ranges', 'group_radius')
    outs = []
    for per_group in (8, 3, 27, None):
        cl = geometry.build_knn_clusters(ctx['point_base'], sets, clusters_per_group=per_group or 8)
        if per_group is None:                                   # flat: no groups handed over
            for k in ('group_centers', 'group_ranges', 'group_radius'):
                cl.pop(k)
            cl['ngrp'] = 0
        cl = {k: (T(v) if k in dev else v) for k, v in cl.items()}
        outs.append(ops.msknn_clustered(T(q), n_rays, S, cl, [1, 1, 1, 0]).cpu().numpy())
    for o in outs[1:]:
        same(o, outs[0], 'cluster groups')


def test_msknn_small_and_large_launch_forms_agree(ops):
    """Round 6: launches of at most 4 tiles per resident wave run `msknn_clustered_kernel<SPLIT>` (four tickets per tile, tiles wider
    than 0.2 m searched as four one-query jobs), larger ones the one-ticket form.  The SAME queries through both -- one launch of
    12 800 tiles, and the same rays in two launches of 6 400 -- must give the same indices, mask, query-list and centre-cache
    modes included; a sample of them is checked against the brute-force kernel as well."""
    ctx = util.model_context(0, False)
    m = _dev_model(ctx, ops)
    cl = _clusters(ctx)
    rng = np.random.RandomState(17)
    n_rays, S = 64 * 400, 128                                   # 400 ray blocks x 32 chunks = 12 800 tiles > 12 288
    # rays of 128 samples marching through the body: neighbouring samples close, neighbouring rays anywhere (spread tiles)
    start = ctx['point_base'][rng.randint(0, 6890, n_rays)] + rng.randn(n_rays, 3).astype(np.float32) * 0.05
    step = rng.randn(n_rays, 3).astype(np.float32)
    step *= (0.004 / np.linalg.norm(step, axis=1, keepdims=True)).astype(np.float32)
    q = (start[:, None, :] + step[:, None, :] * np.arange(S, dtype=np.float32)[None, :, None]).reshape(-1, 3).astype(np.float32)
    qd = T(q)
    half = n_rays // 2
    keep = T((rng.rand(n_rays * S) < 0.5).astype(np.float32))
    rows_all, count_all = ops.live_rows(keep)
    c = np.array([-4.1e-5, -1.5e-5, -5.7e-6], np.float32)
    center = ops.knn_center(T(c), m['points'], m['imap'], m['begin'])

    def two_launches(**kw):
        outs = []
        for lo, hi in ((0, half), (half, n_rays)):
            sub = {}
            if 'mask' in kw:
                sub['mask'] = kw['mask'][lo * S:hi * S].contiguous()
            if 'rows' in kw:
                r, cnt = ops.live_rows(keep[lo * S:hi * S].contiguous())
                sub['rows'], sub['count'] = r, cnt
            if 'center' in kw:
                sub['center'] = kw['center']
            outs.append(ops.msknn_clustered(qd[lo * S:hi * S].contiguous(), hi - lo, S, cl, [1, 1, 1, 0], **sub))
        return torch.cat(outs)
    live = keep.cpu().numpy() > 0
    for kw in ({}, {'mask': keep}, {'rows': rows_all, 'count': count_all}, {'center': center}):
        whole = ops.msknn_clustered(qd, n_rays, S, cl, [1, 1, 1, 0], **kw)
        parts = two_launches(**kw)
        sel = live if ('mask' in kw or 'rows' in kw) else np.ones(n_rays * S, bool)
        same(parts.cpu().numpy()[sel], whole.cpu().numpy()[sel], f'split form vs one-ticket form ({sorted(kw)})')
    pick = rng.choice(n_rays * S, 20000, replace=False)
    brute = ops.msknn(T(q[pick]), m['points'], m['imap'], m['begin'], m['seed'])
    same(whole.cpu().numpy()[pick], brute.cpu().numpy(), 'clustered (centre cache) vs brute force')


def test_point_stage_bit_exact(case, ops, oracle):
    g, ctx, o = case
    m = _dev_model(ctx, ops)
    pc = T(ctx['point_cloud'])
    kidx = ops.knn_small(pc, m['base'], 3)
    same(kidx.cpu().numpy(), oracle.knn(ctx['point_cloud'], ctx['point_base'], 3), 'kidx')
    kb, sdf = ops.point_sdf(pc, m['base'], m['normals'], m['unit'], kidx)
    same(sdf.cpu().numpy(), o['sdf'], 'sdf')
    same(kb.cpu().numpy(), o['kb'], 'knn_base')
    table = ops.point_table(kb, sdf, pc, m['b32'], m['tb32'], m['emb'], m['off'], ctx['S'], ctx['H'])
    same(table.cpu().numpy()[:, :35], o['table'], 'table')
    assert np.abs(kb.cpu().numpy() - g['cnl.point_cloud']).max() <= 1e-7      # vs the reference
    assert np.abs(sdf.cpu().numpy() - g['cnl.point_sdf'].ravel()).max() <= 1e-7


def test_sample_features_and_mlp(case, ops):
    g, ctx, o = case
    m = _dev_model(ctx, ops)
    mlp_in, raw, enc_in = ops.sample_features(T(o['xyz']), T(o['knn']), m['base'], m['normals'], m['unit'],
                                              T(ctx['counter']), T(np.concatenate([o['table'], np.zeros((o['table'].shape[0], ops.table_stride() - 35), np.float32)], 1)),
                                              m['b32'], m['tb32'], m['emb'], m['off'], ctx['S'], ctx['H'], want_enc_in=True)
    mi = mlp_in.cpu().numpy()
    same(raw.cpu().numpy()[:, 4], o['raw'][:, 4], 'signed distance')         # bit-exact
    same(mi[:, 36:], o['mlp_in'][:, 36:], 'hash encoding')                   # bit-exact
    amp = bool(g['meta.amplify'])
    # aggregation: device expf differs from libm by <= 2 ulp -> 1e-6 relative to O(1) features
    assert np.abs(mi[:, :36] - o['mlp_in'][:, :36]).max() <= (5e-6 if amp else 1e-6)
    Wg, Bg, Wc, Bc = util.canonical_mlp_params(ctx['sd'])
    W = [T(w) for w in Wg + Wc]
    B = [T(b) for b in Bg + Bc]
    packed = ops.canonical_mlp_pack(W, B)
    ops.canonical_mlp(mlp_in, packed, raw)
    got = raw.cpu().numpy()
    # fp32 MFMA sums the same products in a different k order than the oracle's serial chain
    tol = util.pick(g, 2e-6, 2e-4, 5e-4)         # (trained-like: |sigma| up to 30 behind a gain of 640)
    assert np.abs(got[:, :4] - o['raw'][:, :4]).max() <= tol
    # MLP alone on identical inputs (oracle's), against float64
    raw2 = torch.zeros_like(raw)
    ops.canonical_mlp(T(o['mlp_in']), packed, raw2)
    from tests.test_oracle_golden import _mlp_f64
    ref = _mlp_f64(o['mlp_in'], Wg, Bg, Wc, Bc)
    assert np.abs(raw2.cpu().numpy()[:, :4] - ref).max() <= util.pick(g, 1e-6, 5e-5, 5e-4)
    # the 32-sample-wave direct-load kernel: same packed buffer, same bound
    raw3 = torch.zeros_like(raw)
    ops.canonical_mlp(T(o['mlp_in']), packed, raw3, direct=True)
    assert np.abs(raw3.cpu().numpy()[:, :4] - ref).max() <= util.pick(g, 1e-6, 5e-5, 5e-4)


def test_sample_features_generic_level_layout(case, ops, oracle):
    """A level layout the reference's constructor never produces -- hashed levels whose size is not a power of two (the
    reference's loop + modulo, `GENERIC` instantiation of the 8-lanes-per-sample kernel) -- against the oracle's encoder on
    the kernel's own encoder inputs, bit for bit; dense levels and power-of-two levels side by side in the same wave."""
    g, ctx, o = case
    m = _dev_model(ctx, ops)
    off = ctx['offsets'].astype(np.int64)
    sizes = np.diff(off)
    sizes[3], sizes[6], sizes[15] = 300000, 123456, 500008          # multiples of 8, not powers of two, hashed
    off2 = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    rng = np.random.default_rng(5)
    emb2 = rng.uniform(-1, 1, (int(off2[-1]), 2)).astype(np.float32)
    table = T(np.concatenate([o['table'], np.zeros((o['table'].shape[0], ops.table_stride() - 35), np.float32)], 1))
    N = o['xyz'].shape[0]
    rows = torch.arange(N, device=DEV, dtype=torch.int32)            # (a row list routes to the 8-lanes kernel)
    count = torch.tensor([N], device=DEV, dtype=torch.int32)
    mlp_in, raw, enc_in = ops.sample_features(T(o['xyz']), T(o['knn']), m['base'], m['normals'], m['unit'],
                                              T(ctx['counter']), table, m['b32'], m['tb32'], T(emb2), T(off2), ctx['S'],
                                              ctx['H'], want_enc_in=True, rows=rows, count=count)
    x = enc_in.cpu().numpy()
    want, _ = oracle.grid_encode_forward(x, emb2, off2, ctx['S'], ctx['H'])               # [L, B, C]
    same(mlp_in.cpu().numpy()[:, 36:], want.transpose(1, 0, 2).reshape(N, -1), 'hash encoding, generic level layout')
    same(raw.cpu().numpy()[:, 4], o['raw'][:, 4], 'signed distance')


def test_grid_encode_forward_bit_exact(case, ops, oracle):
    g, ctx, _ = case
    m = _dev_model(ctx, ops)
    for tag in ('enc_sample', 'enc_point'):
        x = g[tag + '.in']
        B, L = x.shape[0], 16
        out = torch.empty(L, B, 2, device=DEV)
        dy = torch.empty(B, L * 4 * 2, device=DEV)
        ops.grid_encode_forward(T(x), m['emb'], m['off'], out, B, 4, 2, L, ctx['S'], ctx['H'], dy)
        want, want_dy = oracle.grid_encode_forward(x, ctx['embeddings'], ctx['offsets'], ctx['S'], ctx['H'],
                                                   want_dy_dx=True)
        assert np.array_equal(out.cpu().numpy(), want)                       # bit-exact
        assert np.array_equal(dy.cpu().numpy(), want_dy)
        # ... and equal to what the reference's module returned ([B, L*C] after its permute)
        assert np.array_equal(out.permute(1, 0, 2).reshape(B, -1).cpu().numpy(), g[tag + '.out'])


@pytest.mark.parametrize('D,Cc,gridtype,interp,align', [(2, 1, 0, 0, False), (3, 2, 0, 1, False),
                                                       (3, 4, 1, 0, True), (4, 8, 0, 0, False),
                                                       (5, 2, 0, 0, False), (2, 2, 1, 1, True)])
def test_grid_encode_variants_and_edges(ops, oracle, D, Cc, gridtype, interp, align):
    from occnerf_amd.gridencoder import grid_offsets
    rng = np.random.RandomState(D * 10 + Cc)
    L = 8
    offsets, pls = grid_offsets(D, L, 1.6, 4, 12, align_corners=align)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), Cc)).astype(np.float32)
    x = rng.uniform(0, 1, (777, D)).astype(np.float32)     # ragged size (not a multiple of 256)
    x[0] = 0.0                                              # exact cell corners
    x[1] = 1.0
    x[2] = -1e-6                                            # out of range -> zero row
    x[3, -1] = 1.0 + 1e-6
    x[4] = 0.5
    S = float(np.log2(pls))
    out = torch.empty(L, x.shape[0], Cc, device=DEV)
    dy = torch.empty(x.shape[0], L * D * Cc, device=DEV)
    ops.grid_encode_forward(T(x), T(emb), T(offsets), out, x.shape[0], D, Cc, L, S, 4, dy, gridtype, align, interp)
    want, want_dy = oracle.grid_encode_forward(x, emb, offsets, S, 4, True, gridtype, align, interp)
    assert np.array_equal(out.cpu().numpy(), want)
    assert np.array_equal(dy.cpu().numpy(), want_dy)
    assert not out[:, 2].any() and not out[:, 3].any()
    # empty batch is a no-op
    ops.grid_encode_forward(torch.empty(0, D, device=DEV), T(emb), T(offsets), torch.empty(L, 0, Cc, device=DEV),
                            0, D, Cc, L, S, 4)
    # backward: atomics reorder the fp32 sums -> tolerance 1e-5 relative to the largest entry
    grad = rng.randn(L, x.shape[0], Cc).astype(np.float32)
    ge = torch.zeros_like(T(emb))
    gi = torch.zeros(x.shape[0], D, device=DEV)
    ops.grid_encode_backward(T(grad), T(x), T(emb), T(offsets), ge, x.shape[0], D, Cc, L, S, 4, dy, gi,
                             gridtype, align, interp)
    wge, wgi = oracle.grid_encode_backward(grad, x, offsets, emb.shape[0], Cc, S, 4, want_dy, gridtype, align, interp)
    assert np.abs(ge.cpu().numpy() - wge).max() <= 1e-5 * max(1.0, np.abs(wge).max())
    assert np.abs(gi.cpu().numpy() - wgi).max() <= 1e-5 * max(1.0, np.abs(wgi).max())


@pytest.mark.parametrize('D,Cc,gridtype,interp,align', [(4, 2, 0, 0, False), (3, 2, 0, 1, False), (3, 4, 1, 0, True),
                                                       (2, 8, 0, 0, False), (5, 2, 0, 0, False), (4, 1, 0, 0, False)])
def test_grid_encode_half_dispatch(ops, oracle, D, Cc, gridtype, interp, align):
    """The at::Half dispatch case of the operator (gridencoder.cu:467,500; what grid.py:44-45 feeds under autocast) against
    the oracle's restatement of c10::Half arithmetic: outputs and dy_dx bit for bit (the per-corner order is fixed),
    also within one half-ulp of the fp32 evaluation rounded to half; the input gradient bit for bit; the embedding
    gradient (packed-half atomics in free order) within half rounding of the sequential sum."""
    from occnerf_amd.gridencoder import grid_offsets
    rng = np.random.RandomState(100 + D * 10 + Cc)
    L = 8
    offsets, pls = grid_offsets(D, L, 1.6, 4, 12, align_corners=align)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), Cc)).astype(np.float16)
    x = rng.uniform(0, 1, (1031, D)).astype(np.float32)
    x[0], x[1], x[2], x[4] = 0.0, 1.0, -1e-6, 0.5
    x[3, -1] = 1.0 + 1e-6
    S, B = float(np.log2(pls)), x.shape[0]
    out = torch.empty(L, B, Cc, device=DEV, dtype=torch.float16)
    dy = torch.empty(B, L * D * Cc, device=DEV, dtype=torch.float16)
    ops.grid_encode_forward(T(x), T(emb), T(offsets), out, B, D, Cc, L, S, 4, dy, gridtype, align, interp)
    want, want_dy = oracle.grid_encode_forward_f16(x, emb, offsets, S, 4, True, gridtype, align, interp)
    same(out.cpu().numpy().view(np.uint16), want.view(np.uint16), 'half outputs')
    same(dy.cpu().numpy().view(np.uint16), want_dy.view(np.uint16), 'half dy_dx')
    assert not out[:, 2].any() and not out[:, 3].any()
    # against the fp32 operator on the same (half-valued) table: the 2^D half-rounded accumulation steps stay within
    # a few half-ulps of the largest partial sum (|result| <= 1 here: ulp 2^-11 .. 2^-10)
    f32, _ = oracle.grid_encode_forward(x, emb.astype(np.float32), offsets, S, 4, False, gridtype, align, interp)
    assert np.abs(out.float().cpu().numpy() - f32).max() <= (1 << D) * 2.0 ** -11
    if Cc == 1:
        return                                               # no half backward for odd C (refused by name, tested above)
    grad = (rng.randn(L, B, Cc) * 0.1).astype(np.float16)
    ge = torch.zeros(emb.shape, device=DEV, dtype=torch.float16)
    gi = torch.zeros(B, D, device=DEV, dtype=torch.float16)
    ops.grid_encode_backward(T(grad), T(x), T(emb), T(offsets), ge, B, D, Cc, L, S, 4, dy, gi, gridtype, align, interp)
    wge, wgi = oracle.grid_encode_backward_f16(grad, x, offsets, emb.shape[0], Cc, S, 4, want_dy, gridtype, align, interp)
    same(gi.cpu().numpy().view(np.uint16), wgi.view(np.uint16), 'half grad_inputs')
    got = ge.float().cpu().numpy()
    ref32, _ = oracle.grid_encode_backward(grad.astype(np.float32), x, offsets, emb.shape[0], Cc, S, 4, None, gridtype,
                                           align, interp)
    # per cell n half-rounded additions: error <= n * half-ulp of the running sum; cells of the coarse levels collect
    # hundreds of terms, so the bound is relative to the largest entry -- and the device is as close to the exact sum
    # as the sequential restatement is
    scale = np.abs(ref32).max()
    assert np.abs(got - ref32).max() <= 0.02 * scale
    # (the order of the device's packed-half atomics is free and differs from run to run: one draw of n half-rounded
    # additions lands within a small factor of another -- 2x was exceeded once in five driver / builder runs, D = 2, C = 8)
    assert np.abs(got - ref32).max() <= 4.0 * max(np.abs(wge.astype(np.float32) - ref32).max(), 2.0 ** -11 * scale)


def test_operator_forward_d4c2_equals_reference_shaped_kernel(ops):
    """The 8-lanes-per-sample D = 4, C = 2 operator forward (host-side level modes) against the reference-shaped
    thread-per-(sample, level) kernel reached through the reference's own signature: bit-identical, including rows
    outside [0,1], exact cell corners and a ragged batch; both table layouts (dense + hashed, all-hashed)."""
    from occnerf_amd import _lib
    from occnerf_amd.gridencoder import GridEncoder
    for bound, B in ((1.4, 70001), (0.3, 4097)):
        enc = GridEncoder(input_dim=4, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                          desired_resolution=2048 * bound).to(DEV)
        enc.embeddings.data.uniform_(-1.0, 1.0)
        g = torch.Generator(device='cpu').manual_seed(B)
        x = torch.rand(B, 4, generator=g)
        x[:7] = torch.tensor([[0, 0, 0, 0], [1, 1, 1, 1], [0.5, 0.25, 0.125, 1.0], [-1e-7, 0.5, 0.5, 0.5],
                              [0.5, 1.0000001, 0.5, 0.5], [1.0, 0.0, 1.0, 0.0], [0.999999, 0.999999, 0.999999, 0.999999]])
        x = x.to(DEV)
        L, S, H = 16, enc.log2_per_level_scale, enc.base_resolution
        fast = torch.full((L, B, 2), 7.0, device=DEV)
        ops.grid_encode_forward(x, enc.embeddings.detach(), enc.offsets, fast, B, 4, 2, L, S, H)
        slow = torch.full((L, B, 2), -7.0, device=DEV)
        rc = _lib.lib().occnerf_grid_encode_forward(x.data_ptr(), enc.embeddings.data_ptr(), enc.offsets.data_ptr(),
                                                    slow.data_ptr(), B, 4, 2, L, float(S), int(H), None, 0, 0, 0,
                                                    torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        same(fast.cpu().numpy(), slow.cpu().numpy(), f'operator forward, bound {bound}')
        assert float(fast[:, 3].abs().max()) == 0.0 and float(fast[:, 4].abs().max()) == 0.0      # out-of-range rows


@pytest.mark.parametrize('n', [1, 15, 16, 17, 63, 64, 65, 1000, 4097])
def test_canonical_mlp_ragged(ops, n):
    """Workgroups of 4 waves x 16 samples: partial waves, partial workgroups, column 4 untouched."""
    ctx = util.model_context(0, False)
    Wg, Bg, Wc, Bc = util.canonical_mlp_params(ctx['sd'])
    packed = ops.canonical_mlp_pack([T(w) for w in Wg + Wc], [T(b) for b in Bg + Bc])
    rng = np.random.default_rng(n)
    x = (rng.standard_normal((n, 68)) * 0.3).astype(np.float32)
    raw = torch.full((n + 3, 5), 7.0, device=DEV)             # 3 guard rows behind the batch
    ops.canonical_mlp(T(x), packed, raw[:n])
    from tests.test_oracle_golden import _mlp_f64
    got = raw.cpu().numpy()
    assert np.abs(got[:n, :4] - _mlp_f64(x, Wg, Bg, Wc, Bc)).max() <= 1e-6
    assert (got[:n, 4] == 7.0).all() and (got[n:] == 7.0).all()


def test_canonical_mlp_module_gathered_interface(case, ops):
    """CanonicalMLP.forward with the reference's keyword surface (gathered neighbours)."""
    g, ctx, o = case
    from occnerf_amd.canonical_mlp import CanonicalMLP
    cm = CanonicalMLP(mlp_depth=4, mlp_width=256, input_ch=63, skips=[], bound=ctx['bound'])
    cm.load_state_dict({k[len('cnl_mlp.module.'):]: v for k, v in ctx['sd'].items()
                        if k.startswith('cnl_mlp.module.')})
    cm = cm.to(DEV)
    idx = g['cnl.knn_idxs'].astype(np.int64)
    N = idx.shape[0]
    raw = cm(xyz=T(g['cnl.xyz']), xyz_embedded=None,
             knn_points=T(ctx['point_base'][idx[:, 0]].reshape(N, 10, 3)),
             point_norms=T(ctx['normals'][idx[:, 0]].reshape(N, 10, 3)),
             knn_att=T(ctx['counter'][idx].reshape(N, 40, 1)),
             point_cloud=T(g['cnl.point_cloud']), point_sdf=T(g['cnl.point_sdf']),
             knn_idxs=T(idx), learnable_points=T(g['cnl.learnable_points']))
    want = g['cnl.raw']
    assert np.abs(raw.cpu().numpy()[:, 4] - want[:, 4]).max() <= 1e-6
    assert np.abs(raw.cpu().numpy()[:, :4] - want[:, :4]).max() <= util.pick(g, 2e-5, 5e-4, 2e-2)


def test_canonical_mlp_rows(ops):
    """occnerf_canonical_mlp_rows == occnerf_canonical_mlp_counted on the gathered rows, bit for bit."""
    ctx = util.model_context(0, False)
    Wg, Bg, Wc, Bc = util.canonical_mlp_params(ctx['sd'])
    W = [torch.from_numpy(w).to(DEV) for w in Wg + Wc]
    B = [torch.from_numpy(b).to(DEV) for b in Bg + Bc]
    packed = ops.canonical_mlp_pack(W, B)
    g = torch.Generator(device='cpu').manual_seed(3)
    mlp_in = (torch.randn(1000, 68, generator=g) * 0.3).to(DEV)
    rows = torch.randint(0, 1000, (777,), generator=g).int().to(DEV)
    count = torch.tensor([700], device=DEV, dtype=torch.int32)
    a = ops.canonical_mlp(mlp_in, packed, torch.zeros(777, 5, device=DEV), count=count, in_rows=rows)
    b = ops.canonical_mlp(mlp_in[rows.long()].contiguous(), packed, torch.zeros(777, 5, device=DEV), count=count)
    assert torch.equal(a[:700, :4], b[:700, :4]) and float(a[700:].abs().max()) == 0.0
    assert float(a[:700, :4].abs().max()) > 0


def test_composite(case, ops, oracle):
    g, ctx, o = case
    raw, mask, z = g['comp.raw'], g['comp.mask'][..., 0], g['comp.z_vals']
    n, S = z.shape
    rgb, acc, dep, w, tp = ops.composite(T(raw.reshape(-1, 5)), T(mask.reshape(-1)), T(z), T(o['rays8']),
                                         g['in.bgcolor'], want_weights=True, want_term=True)
    # wave scan re-associates the transmittance product: a few ulp
    assert np.abs(rgb.cpu().numpy() - g['comp.rgb']).max() <= 2e-6
    assert np.abs(acc.cpu().numpy() - g['comp.acc']).max() <= 2e-6
    assert np.abs(dep.cpu().numpy() - g['comp.depth']).max() <= 1e-5
    assert np.abs(w.cpu().numpy() - g['comp.weights']).max() <= 2e-6
    util.assert_term_points(tp.cpu().numpy(), g)      # arg-max alpha: index for index, ties apart


def test_composite_edge_cases(ops, oracle):
    rng = np.random.RandomState(1)
    for S in (1, 63, 64, 65, 192):
        n = 19
        raw = rng.randn(n, S, 5).astype(np.float32) * 3
        raw[0, :, 3] = 50.0                       # saturated alpha, softplus linear branch
        raw[1, :, 3] = -50.0                      # transparent
        mask = rng.rand(n, S).astype(np.float32)
        mask[2] = 0.0                             # fully masked ray -> background colour
        z = np.sort(rng.uniform(4, 7, (n, S)).astype(np.float32), axis=1)
        rays = rng.randn(n, 8).astype(np.float32)
        bg = np.array([255., 128., 0.], np.float32)
        rgb, acc, dep, w, tp = ops.composite(T(raw.reshape(-1, 5)), T(mask.reshape(-1)), T(z), T(rays), bg,
                                             want_weights=True, want_term=True)
        wr, wa, ww, wd, wt = oracle.raw2outputs(raw, mask, z, rays[:, 3:6], bg)
        assert np.abs(rgb.cpu().numpy() - wr).max() <= 3e-6
        assert np.abs(acc.cpu().numpy() - wa).max() <= 3e-6
        assert np.abs(dep.cpu().numpy() - wd).max() <= 3e-5
        assert np.abs(w.cpu().numpy() - ww).max() <= 3e-6
        assert np.array_equal(tp.cpu().numpy(), wt)
        assert np.allclose(rgb.cpu().numpy()[2], bg / 255.0)


@pytest.mark.parametrize('golden', ['train_ri_s32', 'train_amp_s32'])
def test_training_step_against_reference(golden):
    """Rows a18/a19, config 5: training-mode forward (jitter, comp_loss, visibility counter) and the
    gradients of a scalar loss, against the reference's own autograd (tests/golden/train_*_s32)."""
    from occnerf_amd import synth
    g = util.load_golden(golden)
    amp = bool(int(g['meta.amplify']))
    net, ctx = build_network(0, amp, S=32, non_rigid=True)
    net.cfg.perturb = 1.0
    net.train()
    frame = synth.make_frame(img_size=32, pose72=g['meta.pose72'], orbit_frame=7)
    for k in ('rays', 'near', 'far'):
        frame[k] = g['in.' + k]
    data = frame_to_device(frame, DEV)
    out = net(**data, iter_val=1e7, t_rand=T(g['in.t_rand']))
    for k, tol in (('rgb', 2e-4), ('alpha', 2e-4), ('depth', 1e-3), ('comp_loss', 1e-2)):   # comp_loss = 10 exp(-relu(sigma)): 10x the logit tolerance
        assert out[k].shape == g['out.' + k].shape, k
        assert np.abs(out[k].detach().cpu().numpy() - g['out.' + k]).max() <= tol, k
    same(net.point_counter.detach().cpu().numpy(), g['out.point_counter'], 'point_counter after the step')
    loss = (out['rgb'] ** 2).mean() + 0.5 * out['alpha'].mean() + 0.01 * out['depth'].mean() \
        + 0.1 * out['comp_loss'].mean()
    assert abs(float(loss.detach()) - float(g['out.loss'])) <= 1e-4
    loss.backward()
    grads = {n: p.grad for n, p in net.named_parameters()}
    assert sorted(n for n, v in grads.items() if v is None) == sorted(str(x) for x in g['grad.none'])
    report = {}
    for key in g:
        if not key.startswith('grad.') or key in ('grad.none',) or key.startswith('grad.emb'):
            continue
        name = key[len('grad.'):]
        want = g[key].astype(np.float64)
        got = grads[name].detach().cpu().numpy().astype(np.float64)
        report[name] = (np.abs(got - want).max() / max(np.abs(want).max(), 1e-30),
                        np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))
    print({k: (f'{a:.2e}', f'{b:.2e}') for k, (a, b) in report.items()})
    for name, (emax, el2) in report.items():
        # point_dist's gradient passes through d(encoding)/d(input), a piecewise-constant slope of an
        # O(1) random table (amplified checkpoint): a 1-ulp input difference can change the cell at the
        # finest levels, so it is compared in the L2 sense; everything else entry-wise.
        # With the amplified checkpoint (O(1) random hash table) the encoder is ill-conditioned in its
        # input (a few-ulp difference of the projected point moves fine-level features by ~1e-3 and
        # their input-slopes by O(1)), so the two gradients that pass through it -- point_dist and the
        # first geometry layer's weight -- are compared in the L2 sense there; the random-init
        # checkpoint has no such amplification and everything is compared entry-wise.
        if amp and name in ('point_dist', 'cnl_mlp.module.pts_linears.0.weight'):
            assert el2 <= 5e-2, (name, emax, el2)
        else:
            assert emax <= 5e-3, (name, emax, el2)
    ge = grads['cnl_mlp.module.encoder.embeddings'].reshape(-1)
    gv = ge[torch.from_numpy(g['grad.emb.idx']).to(DEV)].cpu().numpy()
    tol = 2e-2 if amp else 2e-3
    assert np.abs(gv - g['grad.emb.val']).max() <= tol * np.abs(g['grad.emb.val']).max()
    assert abs(float(ge.abs().double().sum()) - float(g['grad.emb.abs_sum'])) <= tol * float(g['grad.emb.abs_sum'])


def test_autograd_path_matches_render_path(case):
    """With gradients enabled Network.forward takes the differentiable route (torch autograd over
    the HIP kNN and the HIP grid-encoder Function); in eval mode it must render the same image."""
    g, ctx, o = case
    net, _ = build_network(int(g['meta.seed']), util.level(g), S=int(g['meta.S']),
                           non_rigid=bool(int(g['meta.non_rigid'])))
    data = frame_to_device(g, DEV)
    out = net(**data, iter_val=1e7)                      # grad mode
    assert out['rgb'].requires_grad
    for k in ('rgb', 'alpha', 'depth'):
        # (trained-like checkpoint: the staged training kernels -- layer-by-layer MFMA with intermediates in HBM, another
        # summation order than the fused render kernel -- land at 2.6e-4 of depth, in scene units up to 6; rgb / alpha meet
        # the render gate)
        tol = 5e-4 if (util.level(g) == 2 and k == 'depth') else util.pixel_tol(g)
        assert np.abs(out[k].detach().cpu().numpy() - g['out.' + k]).max() <= tol, k


def test_grid_encoder_module_autograd(ops):
    """GridEncoder module: forward == op, backward runs and matches finite differences."""
    from occnerf_amd.gridencoder import GridEncoder
    torch.manual_seed(0)
    enc = GridEncoder(input_dim=3, num_levels=4, level_dim=2, base_resolution=4, log2_hashmap_size=10,
                      desired_resolution=32).to(DEV)
    enc.embeddings.data.uniform_(-1, 1)
    x = torch.rand(64, 3, device=DEV, requires_grad=True)
    y = enc(x, bound=None)
    assert y.shape == (64, 8)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    assert enc.embeddings.grad is not None and x.grad is not None
    eps = 1e-3
    xd = x.detach().clone()
    xd[:, 0] += eps
    fd = ((enc(xd, bound=None) - y.detach()) * w).sum(1) / eps
    # piecewise-linear field: the finite difference is exact except where the step crosses a cell
    close = (fd - x.grad[:, 0]).abs() <= 1e-2 * (1 + x.grad[:, 0].abs())
    assert close.float().mean() >= 0.8


def test_grid_grad_runs_merge(ops):
    """occnerf_grid_grad_runs (the module backward's transposition [B, L*C] -> [L,B,C] with runs of bitwise identical inputs
    merged) against the plain permutation: (a) with no identical neighbours the output IS the permutation, bit for bit;
    (b) with runs of every length across the 64-sample chunk boundaries -- and -0.0 against +0.0, which are different bit
    patterns and must not merge -- the embedding gradient of the full backward equals the unmerged one (B = 20 000 takes the
    scatter kernel: fp32 global atomics in hardware order, up to 5 000 terms on one cell -- 2e-5 of the largest entry), and every run's rows sit summed in its first sample with zeros behind."""
    from occnerf_amd.gridencoder import grid_offsets
    L, H, D, C = 16, 16, 4, 2
    off, pls = grid_offsets(D, L, 2.0, H, 19, desired_resolution=2048 * 1.4)
    S_ = float(np.log2(pls))
    offsets = torch.tensor(np.asarray(off), dtype=torch.int32, device=DEV)
    total = int(off[-1])
    rng = np.random.default_rng(9)
    B = 20000
    x = rng.random((B, D), dtype=np.float32)
    g = rng.standard_normal((B, L * C)).astype(np.float32)
    plain = ops.grid_grad_runs(T(g), T(x), B, D, L, C)
    assert torch.equal(plain, T(g).view(B, L, C).permute(1, 0, 2).contiguous())                      # (a)
    pos, runs = 5, []
    for ln in (2, 3, 63, 64, 65, 129, 700, 1, 2, 5000):
        x[pos:pos + ln] = x[pos]
        runs.append((pos, ln))
        pos += ln + 3
    x[pos, 0], x[pos + 1] = 0.0, x[pos]
    x[pos + 1, 0] = -0.0                                                                             # not the same bits
    xt, gt = T(x), T(g)
    merged = ops.grid_grad_runs(gt, xt, B, D, L, C)
    perm = gt.view(B, L, C).permute(1, 0, 2).contiguous()
    def check_runs(merged, perm):
        for p0, ln in runs:
            # inside a 64-sample chunk a run collapses onto its first sample; a run crossing chunk boundaries has one head per chunk
            bounds = sorted({p0} | {b for b in range((p0 // 64 + 1) * 64, p0 + ln, 64)}) + [p0 + ln]
            for a, b in zip(bounds[:-1], bounds[1:]):
                want = perm[:, a:b].double().sum(1)
                assert float((merged[:, a].double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
                assert float(merged[:, a + 1:b].abs().max()) == 0.0 if b - a > 1 else True
    check_runs(merged, perm)
    assert torch.equal(merged[:, pos:pos + 2], perm[:, pos:pos + 2])
    emb = torch.zeros(total, C, device=DEV)
    ga, gb = torch.zeros(total, C, device=DEV), torch.zeros(total, C, device=DEV)
    ops.grid_encode_backward(merged, xt, emb, offsets, ga, B, D, C, L, S_, H)
    ops.grid_encode_backward(perm, xt, emb, offsets, gb, B, D, C, L, S_, H)
    assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max())
    # and through the module: encoder(x).backward(g) takes the merged route for B >= 4096
    from occnerf_amd.gridencoder import GridEncoder
    enc = GridEncoder(input_dim=4, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                      desired_resolution=2048 * 1.4).to(DEV)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    enc(xt, bound=None).backward(gt)
    assert float((enc.embeddings.grad - gb).abs().max()) <= 2e-5 * float(gb.abs().max())
    # any other row width takes the general kernel (the 16 x 2 rows above the LDS-tile one): same contract
    L2, C2 = 5, 3
    g2 = T(rng.standard_normal((B, L2 * C2)).astype(np.float32))
    m2, p2 = ops.grid_grad_runs(g2, xt, B, D, L2, C2), g2.view(B, L2, C2).permute(1, 0, 2).contiguous()
    assert torch.equal(m2[:, :5], p2[:, :5]) and torch.equal(m2[:, pos:], p2[:, pos:])
    check_runs(m2, p2)


def test_grid_backward_tiled_vs_scatter(ops):
    """Large batches take the tiled, atomics-free backward (workgroup-owned table tiles in LDS); small ones the
    scatter kernel with global atomics.  Same sums: compare one 40 000-sample call against the same samples fed
    in chunks of 10 000 (scatter path), and both against the CPU oracle's backward on a subset of levels."""
    from occnerf_amd.gridencoder import grid_offsets
    L, H, D, C = 16, 16, 4, 2
    off, pls = grid_offsets(D, L, 2.0, H, 19, desired_resolution=2048 * 1.4)
    S_ = float(np.log2(pls))
    offsets = torch.tensor(np.asarray(off), dtype=torch.int32, device=DEV)
    total = int(off[-1])
    B = 40000
    rng = np.random.default_rng(5)
    x = rng.random((B, D), dtype=np.float32)
    x[::97, 1] = 1.5                                            # out of range rows: no gradient
    x[: B // 2, :3] = x[0, :3] + 0.002 * rng.standard_normal((B // 2, 3)).astype(np.float32)   # contended cells
    g = rng.standard_normal((L, B, C)).astype(np.float32)
    emb = torch.zeros(total, C, device=DEV)
    xt, gt = T(np.clip(x, -1, 2)), T(g)
    tiled = torch.zeros(total, C, device=DEV)
    ops.grid_encode_backward(gt, xt, emb, offsets, tiled, B, D, C, L, S_, H)
    scat = torch.zeros(total, C, device=DEV)
    for i in range(0, B, 10000):
        ops.grid_encode_backward(gt[:, i:i + 10000].contiguous(), xt[i:i + 10000].contiguous(), emb, offsets, scat,
                                 10000, D, C, L, S_, H)
    scale = scat.abs().max().item()
    assert (tiled - scat).abs().max().item() <= 2e-5 * scale    # fp32 sums in different orders
    assert tiled.abs().sum().item() > 0
    # the same tiled kernel without the tile-set pre-pass (no scratch from the caller): every job re-hashes every sample
    from occnerf_amd import _lib
    gt2 = gt.clone()
    gt2[:, 5::11] = 0.0                                         # exact-zero gradient rows are skipped by both
    outs = []
    for use_scratch in (False, True):
        o = torch.zeros(total, C, device=DEV)
        scratch = torch.empty(L * B, dtype=torch.int64, device=DEV) if use_scratch else None
        rc = _lib.lib().occnerf_grid_encode_backward_h(
            gt2.data_ptr(), xt.data_ptr(), emb.data_ptr(), offsets.data_ptr(), ops._host_offsets(offsets), o.data_ptr(), B, D, C,
            L, S_, H, None, None, 0, 0, 0, None if scratch is None else scratch.data_ptr(), 0 if scratch is None else L * B * 8,
            torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        outs.append(o)
    assert (outs[0] - outs[1]).abs().max().item() <= 1e-6 * scale, 'masked and plain scans add the same terms (fp64 tiles)'
    assert outs[0].abs().sum().item() > 0


def test_skip_empty_samples_is_exact(ops):
    """Dropping the samples whose motion-weight sum is exactly 0 changes no output bit (posed free-view frame,
    non-rigid on): rgb, alpha and depth with cfg.skip_empty_samples on and off."""
    from occnerf_amd import synth
    from tests.gpu_util import build_network, frame_to_device
    net, ctx = build_network(seed=0, amplify=True, S=64, non_rigid=True)
    frame = synth.make_frame(img_size=96, pose72=synth.seeded_pose(1), orbit_frame=28)
    data = frame_to_device(frame, DEV)
    outs = []
    for skip in (True, False):
        net.cfg.skip_empty_samples = skip
        with torch.no_grad():
            o = net(**data, iter_val=1e7)
        outs.append({k: o[k].clone() for k in ('rgb', 'alpha', 'depth')})
    net.cfg.skip_empty_samples = True
    for k in ('rgb', 'alpha', 'depth'):
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert float(outs[0]['alpha'].max()) > 0.05                 # the frame is not empty


@pytest.mark.parametrize('n', [1, 255, 70001])
def test_live_rows_and_scatter(ops, n):
    """Device-side live-sample list (no host sync) against torch.nonzero; scatter of compact raw rows."""
    g = torch.Generator(device='cpu').manual_seed(n)
    mask = torch.rand(n, generator=g)
    mask[torch.rand(n, generator=g) < 0.4] = 0.0
    if n == 255:
        mask[:] = 0.0                                               # nothing alive
    md = mask.to(DEV)
    rows, count = ops.live_rows(md)
    want = torch.nonzero(md).squeeze(1).int()
    m = int(count)
    assert m == want.numel() and torch.equal(rows[:m], want)
    raw_c = torch.arange(n * 5, device=DEV, dtype=torch.float32).reshape(n, 5)
    full = ops.scatter_raw(raw_c, rows, count, torch.zeros(n, 5, device=DEV))
    ref = torch.zeros(n, 5, device=DEV)
    ref[want.long()] = raw_c[:m]
    assert torch.equal(full, ref)


def test_empty_ray_batch():
    """A frame (or a rank's shard) without rays returns empty outputs instead of failing inside a kernel wrapper."""
    from occnerf_amd import synth
    net, ctx = build_network(seed=0, amplify=False, S=32, non_rigid=True)
    frame = synth.make_frame(img_size=32, pose72=np.zeros(72, np.float32), orbit_frame=0)
    for k in ('near', 'far'):
        frame[k] = frame[k][:0]
    frame['rays'] = frame['rays'][:, :0]
    with torch.no_grad():
        out = net(**frame_to_device(frame, DEV), iter_val=1e7)
    assert out['rgb'].shape == (0, 3) and out['alpha'].shape == (0,) and out['depth'].shape == (0,)


def test_caches_follow_in_place_weight_updates():
    """A render after optimiser steps must use the updated weights (the packed MFMA weight streams, the decoded volume
    logits and the per-point table are cached per weight version), whatever sequence of train()/eval() and
    no_grad the caller goes through -- the reference trainer's progress renders do exactly this."""
    from occnerf_amd import synth
    from occnerf_amd.optim import FusedAdam
    net, ctx = build_network(seed=0, amplify=True, S=32, non_rigid=True)
    frame = synth.make_frame(img_size=48, pose72=synth.seeded_pose(1), orbit_frame=5)
    data = frame_to_device(frame, DEV)

    def render():
        net.eval()
        with torch.no_grad():
            return net(**data, iter_val=1e7)['rgb'].clone()
    a = render()
    assert torch.equal(render(), a)
    net.train()
    with torch.no_grad():                                   # a render in train mode under no_grad in between
        net(**data, iter_val=1e7)
    opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-2)
    out = net(**data, iter_val=1e7)
    ((out['rgb'] - 0.3) ** 2).mean().backward()
    opt.step(max_grad_norm=1.0)
    b = render()
    assert float((a - b).abs().max()) > 1e-4, 'the eval render after the step still shows the old weights'
    fresh, _ = build_network(seed=0, amplify=True, S=32, non_rigid=True)
    fresh.load_state_dict(net.state_dict(), strict=True)
    fresh.eval()
    with torch.no_grad():
        c = fresh(**data, iter_val=1e7)['rgb']
    assert torch.equal(b, c), 'a freshly built network with the same weights renders the same bits'


@pytest.mark.parametrize('n,kcols,ccols', [(1, 3, 3), (300, 3, 3), (70001, 68, 68), (5000, 4, 8), (257, 3, 3)])
def test_repeat_heads(ops, n, kcols, ccols):
    """Run-length elimination of repeated rows against numpy: scan, head list, head count, head mask -- with and without a
    row list, 16-byte and dword compare paths, a count below the capacity, -0.0 != +0.0 (bit patterns)."""
    rng = np.random.default_rng(n)
    base = rng.standard_normal((max(n // 7, 1), ccols)).astype(np.float32)
    pick = np.sort(rng.integers(0, base.shape[0], n))            # runs of equal rows
    keys = base[pick].copy()
    if n > 10:
        keys[5, 0], keys[6] = 0.0, keys[5]
        keys[6, 0] = -0.0                                          # equal as floats, different as bits
        keys[9, ccols - 1] += 1.0                                   # a column outside the key when kcols < ccols
    kd = torch.from_numpy(keys).to(DEV)
    for use_rows in (False, True):
        if use_rows:
            rows_np = np.sort(rng.choice(n, size=max(n * 2 // 3, 1), replace=False)).astype(np.int32)
            rows = torch.from_numpy(rows_np).to(DEV)
        else:
            rows_np, rows = np.arange(n, dtype=np.int32), None
        cap = rows_np.shape[0]
        cnt = cap if n != 257 else cap // 2                         # a list shorter than its buffer
        count = torch.tensor([cnt], device=DEV, dtype=torch.int32)
        scan, heads, hcount, hmask = ops.repeat_heads(kd, kcols, count, rows=rows, want_mask=True)
        kb = keys.view(np.uint32)[rows_np[:cnt], :kcols]
        flag = np.ones(cnt, bool)
        flag[1:] = (kb[1:] != kb[:-1]).any(1)
        want_scan = np.cumsum(flag)
        assert int(hcount) == int(flag.sum())
        assert np.array_equal(scan[:cnt].cpu().numpy(), want_scan)
        assert np.array_equal(heads[:int(hcount)].cpu().numpy(), rows_np[:cnt][flag])
        want_mask = np.zeros(n, np.float32)
        want_mask[rows_np[:cnt][flag]] = 1.0
        assert np.array_equal(hmask.cpu().numpy(), want_mask)
        # every entry finds its head's result
        raw_h = torch.arange(cap * 5, device=DEV, dtype=torch.float32).reshape(cap, 5)
        raw_c = -torch.arange(cap * 5, device=DEV, dtype=torch.float32).reshape(cap, 5)
        rows_d = rows if rows is not None else torch.arange(n, device=DEV, dtype=torch.int32)
        full = ops.scatter_raw_heads(raw_h, raw_c, rows_d, count, scan, None, torch.zeros(n, 5, device=DEV)).cpu().numpy()
        ref = np.zeros((n, 5), np.float32)
        a = want_scan - 1
        ref[rows_np[:cnt], :4] = raw_h.cpu().numpy()[a, :4]
        ref[rows_np[:cnt], 4] = raw_c.cpu().numpy()[a, 4]
        assert np.array_equal(full, ref)


@pytest.mark.parametrize('n,kcols,ccols', [(1, 3, 3), (4000, 3, 3), (50000, 68, 68), (3000, 4, 8)])
def test_unique_heads(ops, n, kcols, ccols):
    """Distinct rows of a whole list against numpy: one representative per distinct key (bit patterns), ascending order of
    the representatives, every entry mapped to a representative with an equal key; through a head list and a scan map."""
    rng = np.random.default_rng(n + 1)
    base = rng.standard_normal((max(n // 9, 1), ccols)).astype(np.float32)
    keys = base[rng.integers(0, base.shape[0], n)].copy()              # repeats scattered over the list
    if n > 10:
        keys[7, 0], keys[8] = 0.0, keys[7]
        keys[8, 0] = -0.0
    kd = torch.from_numpy(keys).to(DEV)
    kb = np.ascontiguousarray(keys.view(np.uint32)[:, :kcols])
    for use_heads in (False, True):
        if use_heads:
            heads_np = np.sort(rng.choice(n, size=max(n * 3 // 4, 1), replace=False)).astype(np.int32)
            heads = torch.from_numpy(heads_np).to(DEV)
        else:
            heads_np, heads = np.arange(n, dtype=np.int32), None
        cnt = heads_np.shape[0] if n != 4000 else heads_np.shape[0] - 17
        count = torch.tensor([cnt], device=DEV, dtype=torch.int32)
        # a map onto the entries (1-based), as occnerf_repeat_heads' scan is
        scan_np = rng.integers(1, cnt + 1, size=cnt + 5).astype(np.int32)
        scan = torch.from_numpy(scan_np.copy()).to(DEV)
        scan_count = torch.tensor([cnt + 3], device=DEV, dtype=torch.int32)
        out, ocount = ops.unique_heads(kd, kcols, heads, count, scan=scan, scan_count=scan_count)
        m = int(ocount)
        got = out[:m].cpu().numpy()
        ent = kb[heads_np[:cnt]]
        assert m == np.unique(ent, axis=0).shape[0]
        assert np.all(np.diff(np.searchsorted(heads_np[:cnt], got)) > 0)           # ascending entry order
        assert np.unique(kb[got], axis=0).shape[0] == m                                 # all distinct
        new_scan = scan.cpu().numpy()
        assert np.array_equal(new_scan[cnt + 3:], scan_np[cnt + 3:])                  # beyond its length: untouched
        mapped = got[new_scan[:cnt + 3] - 1]                                           # representative rows
        assert np.array_equal(kb[mapped], ent[scan_np[:cnt + 3] - 1])                  # ... with the entry's key


@pytest.mark.parametrize('name', ['freeview_trained_s32', 'freeview_trained_s128'])
def test_rays_dropped_from_the_tie_free_fixtures(name):
    """Nothing is hidden by the tie-free selection of the trained-like fixtures: the rays the generator dropped (a live sample
    within 2e-5 of a neighbour-set change or an inside-vote flip) travel with the fixture, with what the reference rendered for
    them.  They are rendered here too: finite, and within a bound that a flipped neighbour set stays inside (5e-3 of rgb /
    alpha, 5e-2 of depth -- the one flip observed moved a ray by 8.5e-4 / 1.7e-3 / 1.1e-2); how many of them exceed the 1e-4
    gate on this hardware is printed (0 when HIP breaks every tie the way the reference's CPU run did)."""
    from tests.gpu_util import golden_frame
    g = util.load_golden(name)
    net, _ = build_network(int(g['meta.seed']), util.level(g), S=int(g['meta.S']), non_rigid=True)
    frame = golden_frame(g)
    frame['rays'], frame['near'], frame['far'] = g['dropped.rays'], g['dropped.near'], g['dropped.far']
    with torch.no_grad():
        out = net(**frame_to_device(frame, DEV), iter_val=1e7)
    k_rays = g['dropped.rays'].shape[1]
    assert 1 <= k_rays <= 12
    beyond = 0
    for k, bound in (('rgb', 5e-3), ('alpha', 5e-3), ('depth', 5e-2)):
        got = out[k].cpu().numpy()
        assert np.isfinite(got).all()
        err = np.abs(got - g['dropped.out.' + k]).reshape(k_rays, -1).max(1)
        assert err.max() <= bound, (k, err)
        beyond = max(beyond, int((err > 1e-4).sum()))
    print(f'\n   {name}: {k_rays} dropped rays rendered, {beyond} beyond 1e-4 of the reference')
    # (round 4 measured 0 on the MI355X box -- HIP broke every tie the way the reference's CPU run did; a flipped neighbour set is
    # legitimate on other hardware, but more than a couple of them would mean something else moved)
    assert beyond <= 2, (name, beyond)


@pytest.mark.parametrize('name', ['freeview_trained_truth_s32', 'freeview_trained_truth_s128'])
def test_trained_truth_three_way(name, oracle):
    """the trained-like field on 2 048 rays per fixture, rendered by the UNMODIFIED reference twice -- in
    its own float32 (`out.*`: what the 1e-4 gate is defined against) and in float64 (`truth.*`; make_golden.py
    run_truth_case) -- against HIP and against the CPU oracle.  Rays holding a live sample within 2e-5 of a neighbour-set /
    inside-vote discontinuity stay in the file, flagged; the gate is asserted on the others, the flagged ones are bounded.

    What is asserted, and why it is phrased this way: on this field fp32 itself is not a 1e-4 evaluation of the function --
    the reference's float32 output is up to 8.8e-4 (S=32) / 3.8e-4 (S=128) of depth away from its own float64 output (depth
    is in scene units, up to 6.3), 1.5e-4 of alpha.  So (a) rgb and alpha: every non-fragile ray within 1e-4 of the reference;
    (b) depth: within 1e-4 on >= 99.5 % of them and nowhere further from the reference's fp32 output than that output is from
    the truth; (c) HIP and the reference's fp32 run are interchangeable estimators of the truth -- mean / p99 / max distance
    to the truth within 10 % of the reference's, and the per-ray statement `|hip - truth| <= max(|ref - truth|, 5e-5)` holds
    as often (to 0.5 %) as the same statement with the two exchanged; (d) the oracle shares HIP's discrete decisions bit
    for bit, so HIP vs oracle is summation-order noise alone: within 1e-4 on >= 99.5 % of ALL rays."""
    g = util.load_golden(name)
    ctx = util.model_context(int(g['meta.seed']), util.level(g))
    net, _ = build_network(int(g['meta.seed']), util.level(g), S=int(g['meta.S']), non_rigid=True)
    assert g['in.rays'].shape[1] >= 2000
    data = frame_to_device(g, DEV)
    with torch.no_grad():
        out = net(**data, iter_val=1e7)
    o = stagewise_oracle_render(g, ctx, preamble=tuple(t.cpu().numpy() for t in net.render_preamble(data)))
    ok = ~g['fragile']
    assert ok.sum() >= 1800

    def per_ray(a, b):
        e = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
        return e.reshape(e.shape[0], -1).max(1)
    print(f'\n   {name}: {ok.size} rays, {int((~ok).sum())} flagged fragile; distance to the float64 truth on the others '
          '(max / p99 / mean) and HIP against the reference fp32 / the oracle (fed the HIP preamble)')
    for k in ('rgb', 'alpha', 'depth'):
        hip = out[k].cpu().numpy()
        assert hip.shape == g['out.' + k].shape and np.isfinite(hip).all()
        e_h, e_r, e_o = (per_ray(x, g['truth.' + k])[ok] for x in (hip, g['out.' + k], o[k]))
        hr, ho = per_ray(hip, g['out.' + k]), per_ray(hip, o[k])
        print(f'   {k:5s}: to truth: reference {e_r.max():.2e} / {np.percentile(e_r, 99):.2e} / {e_r.mean():.2e}   oracle '
              f'{e_o.max():.2e} / {np.percentile(e_o, 99):.2e} / {e_o.mean():.2e}   HIP {e_h.max():.2e} / {np.percentile(e_h, 99):.2e} / '
              f'{e_h.mean():.2e}  | HIP - reference: max {hr[ok].max():.2e}, {int((hr[ok] > 1e-4).sum())} rays > 1e-4 (fragile rays: '
              f'max {hr[~ok].max():.2e})  | HIP - oracle: max {ho.max():.2e}, {int((ho > 1e-4).sum())} rays > 1e-4')
        if k != 'depth':
            assert hr[ok].max() <= 1e-4, (k, hr[ok].max())                              # (a)
        else:
            assert (hr[ok] <= 1e-4).mean() >= 0.995 and hr[ok].max() <= e_r.max(), (hr[ok].max(), e_r.max())      # (b)
        assert e_h.mean() <= 1.1 * e_r.mean() + 1e-7 and np.percentile(e_h, 99) <= 1.1 * np.percentile(e_r, 99) + 1e-7 \
            and e_h.max() <= 1.1 * e_r.max() + 1e-7, k                                 # (c)
        f_h, f_r = (e_h <= np.maximum(e_r, 5e-5)).mean(), (e_r <= np.maximum(e_h, 5e-5)).mean()
        assert f_h >= f_r - 0.005, (k, f_h, f_r)
        assert (ho <= 1e-4).mean() >= 0.995 and ho.max() <= 5e-4, (k, ho.max())         # (d)
        # the flagged rays: a flipped neighbour set moves a ray by up to ~1e-3 / 1e-2 (depth); bounded, not gated
        assert hr[~ok].max() <= (5e-2 if k == 'depth' else 5e-3), (k, hr[~ok].max())
