"""GPU: the opt-in alternatives to the headline kernels (split-operand MLPs `f16x3` / `bf16x3`), after every section-8 row.
"""
import os
import numpy as np
import pytest
import torch

from tests import util
from tests.gpu_util import (DEV, T, same, build_network, frame_to_device, per_frame_cpu, stagewise_oracle_render, _dev_model,
                            _clusters, stagewise_table, _torchrun)

pytestmark = pytest.mark.gpu


def test_canonical_mlp_bf16x3(case, ops):
    """Split-bf16 MFMA variant: hi/lo operands, three products, fp32 accumulation.  Bound from
    the operand split (2^-17 relative per product): raw logits within 3e-5 (random init) of
    float64; the pixel-level effect is checked end to end (test_network_end_to_end_bf16x3)."""
    g, ctx, o = case
    Wg, Bg, Wc, Bc = util.canonical_mlp_params(ctx['sd'])
    W = [T(w) for w in Wg + Wc]
    B = [T(b) for b in Bg + Bc]
    packed, packed_h = ops.canonical_mlp_pack(W, B), ops.canonical_mlp_pack_bf16(W)
    from tests.test_oracle_golden import _mlp_f64
    ref = _mlp_f64(o['mlp_in'], Wg, Bg, Wc, Bc)
    outs = []
    for variant in (0, 1):                       # LDS-staged and direct-load weight streams
        raw = torch.zeros(o['mlp_in'].shape[0], 5, device=DEV)
        ops.canonical_mlp_bf16x3(T(o['mlp_in']), packed, packed_h, raw, variant=variant)
        err = np.abs(raw.cpu().numpy()[:, :4] - ref).max()
        assert err <= util.pick(g, 3e-5, 2e-4, 2e-2), (variant, err)      # (trained-like: sigma carries a gain of 640)
        outs.append(raw.cpu().numpy())
    same(outs[0], outs[1], 'bf16x3 LDS vs direct')   # same products, same order


def test_canonical_mlp_f16x3(case, ops):
    """The fp32-grade split (cfg.mlp_precision = 'f16x3', csrc/split.h): two fp16 pieces per operand kept in the normal range,
    three MFMA products, fp32 accumulation -- held to the fp32 kernel's OWN tolerances: the tolerances of
    test_sample_features_and_mlp's "MLP alone on identical inputs against float64" for the three checkpoints
    (1e-6 random-init, 5e-5 amplified, 5e-4 trained-like), with the fp32 kernel's error printed beside it."""
    g, ctx, o = case
    Wg, Bg, Wc, Bc = util.canonical_mlp_params(ctx['sd'])
    W = [T(w) for w in Wg + Wc]
    B = [T(b) for b in Bg + Bc]
    packed, packed_h = ops.canonical_mlp_pack(W, B), ops.canonical_mlp_pack_f16(W)
    assert packed_h.dtype == torch.float16
    from tests.test_oracle_golden import _mlp_f64
    ref = _mlp_f64(o['mlp_in'], Wg, Bg, Wc, Bc)
    raw = torch.zeros(o['mlp_in'].shape[0], 5, device=DEV)
    ops.canonical_mlp_bf16x3(T(o['mlp_in']), packed, packed_h, raw)
    raw32 = torch.zeros_like(raw)
    ops.canonical_mlp(T(o['mlp_in']), packed, raw32)
    err, err32 = np.abs(raw.cpu().numpy()[:, :4] - ref).max(), np.abs(raw32.cpu().numpy()[:, :4] - ref).max()
    print(f'\n   f16x3 vs float64 {err:.3e}   (fp32 kernel {err32:.3e}; |outputs| up to {np.abs(ref).max():.3g})')
    assert err <= util.pick(g, 1e-6, 5e-5, 5e-4), err


@pytest.mark.parametrize('n', [1, 31, 32, 33, 127, 128, 129, 1000, 4097])
def test_canonical_mlp_f16x3_ragged(ops, n):
    """As test_canonical_mlp_ragged (the fp32 kernel's test, same inputs, same 1e-6 against float64): workgroups of 4 waves x
    32 samples, partial waves and workgroups, column 4 and the guard rows untouched; and through a row list with the count on
    the device.  Inputs of amplitude 1e-4 (what a random-init hash table produces) and 30 (beyond any trained activation
    seen) exercise the subnormal-piece and the large-value ends of the fp16 range."""
    ctx = util.model_context(0, False)
    Wg, Bg, Wc, Bc = util.canonical_mlp_params(ctx['sd'])
    W = [T(w) for w in Wg + Wc]
    packed, packed_h = ops.canonical_mlp_pack(W, [T(b) for b in Bg + Bc]), ops.canonical_mlp_pack_f16(W)
    from tests.test_oracle_golden import _mlp_f64
    rng = np.random.default_rng(n)
    for amp, tol in ((0.3, 1e-6), (1e-4, 1e-6), (30.0, 1e-4)):        # (amp 30: outputs ~1e2, the fp32 kernel's error there is 3e-5)
        x = (rng.standard_normal((n, 68)) * amp).astype(np.float32)
        raw = torch.full((n + 3, 5), 7.0, device=DEV)             # 3 guard rows behind the batch
        ops.canonical_mlp_bf16x3(T(x), packed, packed_h, raw[:n])
        got = raw.cpu().numpy()
        want = _mlp_f64(x, Wg, Bg, Wc, Bc)
        assert np.abs(got[:n, :4] - want).max() <= tol * max(1.0, np.abs(want).max() if amp > 1 else 1.0), (amp, np.abs(got[:n, :4] - want).max())
        assert (got[:n, 4] == 7.0).all() and (got[n:] == 7.0).all()
    if n >= 127:
        rows = torch.randperm(n, device=DEV).int()
        count = torch.tensor([n - 5], device=DEV, dtype=torch.int32)
        a = ops.canonical_mlp_bf16x3(T(x), packed, packed_h, torch.zeros(n, 5, device=DEV), count=count, in_rows=rows)
        b = ops.canonical_mlp_bf16x3(T(x)[rows.long()][:n - 5].contiguous(), packed, packed_h, torch.zeros(n - 5, 5, device=DEV))
        assert torch.equal(a[:n - 5], b) and float(a[n - 5:].abs().max()) == 0.0


def test_network_end_to_end_bf16x3(case):
    """Opt-in split-bf16 MLP (cfg.mlp_precision='bf16x3'): meets the 1e-4 pixel gate on the random-init checkpoint (and the
    amplified one's 1e-3).  On the TRAINED-LIKE checkpoint it does NOT (round 4, tools/parity_budget.py): operands split into
    hi + lo bf16 carry 2^-17 of relative residue each, and a density head with a gain of 640 behind a 256-term dot product
    turns that into 1.5e-4 of alpha and 9e-4 of depth (depth is in scene units, up to 6) -- measured, stated in DESIGN.md 3.5,
    and held here to 5e-4 / 3e-3 so that it cannot get worse unnoticed.  The fp32 path meets 1e-4 on the same fixture."""
    g, ctx, o = case
    net, _ = build_network(int(g['meta.seed']), util.level(g), S=int(g['meta.S']),
                           non_rigid=bool(int(g['meta.non_rigid'])), mlp_precision='bf16x3')
    with torch.no_grad():
        out = net(**frame_to_device(g, DEV), iter_val=1e7)
    for k in ('rgb', 'alpha', 'depth'):
        tol = util.pixel_tol(g) if util.level(g) != 2 else (3e-3 if k == 'depth' else 5e-4)
        assert np.abs(out[k].cpu().numpy() - g['out.' + k]).max() <= tol, k


def test_network_end_to_end_f16x3(case):
    """cfg.mlp_precision = 'f16x3' (the fp32-grade split of round 5) against the reference's own output: the fp32 path's gate on
    ALL THREE checkpoints -- 1e-4 random-init, 1e-3 amplified, 1e-4 trained-like (where bf16x3 fails it) -- and the kNN indices
    downstream of the split non-rigid offsets unchanged on the fixture (pixels would move by 1e-3 where a neighbour set flips)."""
    g, ctx, o = case
    nr = bool(int(g['meta.non_rigid']))
    net, _ = build_network(int(g['meta.seed']), util.level(g), S=int(g['meta.S']), non_rigid=nr, mlp_precision='f16x3')
    net32, _ = build_network(int(g['meta.seed']), util.level(g), S=int(g['meta.S']), non_rigid=nr)
    with torch.no_grad():
        out = net(**frame_to_device(g, DEV), iter_val=1e7)
        out32 = net32(**frame_to_device(g, DEV), iter_val=1e7)
    print()
    for k in ('rgb', 'alpha', 'depth'):
        e, e32 = np.abs(out[k].cpu().numpy() - g['out.' + k]).max(), np.abs(out32[k].cpu().numpy() - g['out.' + k]).max()
        print(f'   {k:5s}: max |f16x3 - reference| {e:.3e}   (fp32 path {e32:.3e}; gate {util.pixel_tol(g):g})')
        assert e <= util.pixel_tol(g), k


def test_f16x3_keeps_the_neighbour_sets(ops):
    """kNN indices downstream of the f16x3 non-rigid offsets == those downstream of the fp32 offsets on the golden cases that
    run the non-rigid MLP (wherever the fp32 kernel's own neighbour sets are not at a 1e-6 tie)."""
    for name in ('freeview_amp_s32', 'freeview_trained_s32', 'freeview_trained_s128', 'movement_amp_s32_f3'):
        g = util.load_golden(name)
        ctx = util.model_context(int(g['meta.seed']), util.level(g))
        W, B = util.nonrigid_params(ctx['sd'])
        Wd, Bd = [T(w) for w in W], [T(b) for b in B]
        packed, pf = ops.nonrigid_pack(Wd, Bd), ops.nonrigid_pack_f16(Wd)
        xyz, cond, hann = T(g['nr.xyz_in']), T(g['nr.cond'].astype(np.float32).ravel()), np.ones(6, np.float32)
        a = ops.nonrigid(xyz, cond, hann, Wd[0], Bd[0], packed)
        b = ops.nonrigid_bf16x3(xyz, cond, hann, Wd[0], Bd[0], packed, pf)
        m = _dev_model(ctx, ops)
        ka = ops.msknn(a, m['points'], m['imap'], m['begin'], m['seed']).cpu().numpy()
        kb = ops.msknn(b, m['points'], m['imap'], m['begin'], m['seed']).cpu().numpy()
        diff = np.flatnonzero((ka != kb).reshape(ka.shape[0], -1).any(1))
        print(f'\n   {name}: max |offset f16x3 - fp32| {float((a - b).abs().max()):.2e}; samples whose neighbour lists differ: {diff.size} of {ka.shape[0]}')
        for i in diff:      # only genuine ties may differ
            sets = [np.arange(ctx['point_base'].shape[0])] + [np.asarray(f) for f in ctx['fps']]
            for lvl in range(4):
                if not np.array_equal(ka[i, lvl], kb[i, lvl]):
                    assert util.knn_mismatch_is_tie(a[i:i + 1].cpu().numpy(), ctx['point_base'], ka[i:i + 1, lvl], kb[i:i + 1, lvl], rel=2e-6)


def test_bf16x3_row_list_entry_points(ops):
    """the split-bf16 kernels take the device-side live list like the fp32 ones.
    occnerf_canonical_mlp_bf16x3_rows (count on the device, input row through an index, compact output; both weight-stream
    variants) == the plain call on the gathered rows, bit for bit; occnerf_nonrigid_bf16x3_rows (in place on the listed
    samples) == the plain call on the gathered samples, untouched elsewhere."""
    ctx = util.model_context(0, True)
    Wg, Bg, Wc, Bc = util.canonical_mlp_params(ctx['sd'])
    W = [torch.from_numpy(w).to(DEV) for w in Wg + Wc]
    B = [torch.from_numpy(b).to(DEV) for b in Bg + Bc]
    packed, packed_h = ops.canonical_mlp_pack(W, B), ops.canonical_mlp_pack_bf16(W)
    g = torch.Generator(device='cpu').manual_seed(4)
    mlp_in = (torch.randn(1000, 68, generator=g) * 0.3).to(DEV)
    rows = torch.randint(0, 1000, (777,), generator=g).int().to(DEV)
    count = torch.tensor([700], device=DEV, dtype=torch.int32)
    for variant in (0, 1):
        a = ops.canonical_mlp_bf16x3(mlp_in, packed, packed_h, torch.zeros(777, 5, device=DEV), variant=variant, count=count, in_rows=rows)
        b = ops.canonical_mlp_bf16x3(mlp_in[rows.long()][:700].contiguous(), packed, packed_h, torch.zeros(700, 5, device=DEV), variant=variant)
        assert torch.equal(a[:700, :4], b[:, :4]) and float(a[700:].abs().max()) == 0.0 and float(b[:, :4].abs().max()) > 0
        c = ops.canonical_mlp_bf16x3(mlp_in, packed, packed_h, torch.zeros(1000, 5, device=DEV), variant=variant, count=count)
        d = ops.canonical_mlp_bf16x3(mlp_in[:700].contiguous(), packed, packed_h, torch.zeros(700, 5, device=DEV), variant=variant)
        assert torch.equal(c[:700, :4], d[:, :4]) and float(c[700:].abs().max()) == 0.0
    Wn, Bn = util.nonrigid_params(ctx['sd'])
    Wd, Bd = [torch.from_numpy(w).to(DEV) for w in Wn], [torch.from_numpy(b).to(DEV) for b in Bn]
    pk, ph = ops.nonrigid_pack(Wd, Bd), ops.nonrigid_pack_bf16(Wd)
    xyz = ((torch.rand(5000, 3, generator=g) - 0.5) * 1.5).to(DEV)
    cond = (torch.randn(69, generator=g) * 0.2).to(DEV)
    lrows = torch.sort(torch.randperm(5000, generator=g)[:1900]).values.int().to(DEV)
    lcount = torch.tensor([1777], device=DEV, dtype=torch.int32)
    hann = np.ones(6, np.float32)
    want = ops.nonrigid_bf16x3(xyz[lrows.long()][:1777].contiguous(), cond, hann, Wd[0], Bd[0], pk, ph)
    got = ops.nonrigid_bf16x3_rows(xyz.clone(), lrows, lcount, cond, hann, Wd[0], Bd[0], pk, ph)
    assert torch.equal(got[lrows.long()[:1777]], want) and float((want - xyz[lrows.long()][:1777]).abs().max()) > 1e-4
    untouched = torch.ones(5000, dtype=torch.bool, device=DEV)
    untouched[lrows.long()[:1777]] = False
    assert torch.equal(got[untouched], xyz[untouched])


def test_bf16x3_render_uses_the_device_list(ops):
    """The opt-in bf16x3 render takes the same path as fp32 -- live list and count on the device (no torch.nonzero), repeated
    samples evaluated once -- and skipping / eliminating changes no output bit of it either."""
    from occnerf_amd import synth
    net, ctx = build_network(seed=0, amplify=True, S=64, non_rigid=True, mlp_precision='bf16x3')
    data = frame_to_device(synth.make_frame(img_size=96, pose72=synth.seeded_pose(1), orbit_frame=28), DEV)
    real_nonzero, calls = torch.nonzero, []
    torch.nonzero = lambda *a, **k: (calls.append(1), real_nonzero(*a, **k))[1]
    outs = []
    try:
        for skip, dedup in ((True, True), (True, False), (False, False)):
            net.cfg.skip_empty_samples, net.cfg.dedup_repeated_samples = skip, dedup
            with torch.no_grad():
                o = net(**data, iter_val=1e7)
            outs.append(torch.cat([o['rgb'], o['alpha'][:, None], o['depth'][:, None]], 1))
    finally:
        torch.nonzero = real_nonzero
        net.cfg.skip_empty_samples, net.cfg.dedup_repeated_samples = True, True
    assert not calls
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


def test_f16x3_reports_leaving_its_domain():
    """f16x3 clamps hidden activations at 4 094 (csrc/split.h) -- a checkpoint outside that domain must not
    render wrong pixels silently.  (a) kernel level: inputs that push a first-layer activation past the clamp set the flag word,
    inputs inside the domain leave it zero; (b) renderer: a checkpoint whose first canonical layer is scaled by 3e4 is detected,
    the frame is rendered again by the fp32 kernels (bit-identical to an fp32 network's frame), a warning is issued once and the
    counter moves; the unscaled checkpoint never falls back."""
    import warnings
    from occnerf_amd import ops
    g = util.load_golden('freeview_amp_s32')
    nr = bool(int(g['meta.non_rigid']))
    data = frame_to_device(g, DEV)
    # (a)
    net, _ = build_network(int(g['meta.seed']), util.level(g), S=int(g['meta.S']), non_rigid=nr, mlp_precision='f16x3')
    pk = net._packed_weights()
    assert pk['domain_flag'] is not None and int(pk['domain_flag'].item()) == 0
    x = torch.randn(1000, 68, device=DEV) * 0.3
    raw = torch.zeros(1000, 5, device=DEV)
    ops.canonical_mlp_bf16x3(x, pk['cnl'], pk['cnl_bf16'], raw, domain_flag=pk['domain_flag'])
    assert int(pk['domain_flag'].item()) == 0
    big = x.clone()
    big[777, :34] = 5000.0                                  # 16 x 5000 > 65504: the input clamp itself
    ops.canonical_mlp_bf16x3(big, pk['cnl'], pk['cnl_bf16'], raw, domain_flag=pk['domain_flag'])
    assert int(pk['domain_flag'].item()) == 1
    pk['domain_flag'].zero_()
    # (b)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        with torch.no_grad():
            net(**data, iter_val=1e7)                       # inside the domain: no warning, no fallback
    assert getattr(net, 'f16x3_fallback_frames', 0) == 0
    net32, _ = build_network(int(g['meta.seed']), util.level(g), S=int(g['meta.S']), non_rigid=nr)
    for n in (net, net32):
        with torch.no_grad():
            n.cnl_mlp.module.pts_linears[0].weight.mul_(3e4)
            n.cnl_mlp.module.pts_linears[0].bias.mul_(3e4)
    with torch.no_grad():
        want = net32(**data, iter_val=1e7)
        with pytest.warns(UserWarning, match='f16x3'):
            got = net(**data, iter_val=1e7)
        got2 = net(**data, iter_val=1e7)                    # the second frame falls back too, without a second warning
    assert net.f16x3_fallback_frames == 2
    for k in ('rgb', 'alpha', 'depth'):
        assert torch.equal(got[k], want[k]) and torch.equal(got2[k], want[k]), k
    assert net.cfg.mlp_precision == 'f16x3'


def test_f16x3_deferred_domain_check_raises_for_the_right_frame():
    """cfg.f16x3_domain_check = 'deferred': no wait per frame; a frame that left the domain is named when its flag has arrived
    (explicit check_f16x3_domain(), or the start of a later frame), frames inside the domain pass silently."""
    g = util.load_golden('freeview_amp_s32')
    nr = bool(int(g['meta.non_rigid']))
    data = frame_to_device(g, DEV)
    net, _ = build_network(int(g['meta.seed']), util.level(g), S=int(g['meta.S']), non_rigid=nr, mlp_precision='f16x3')
    net.cfg.f16x3_domain_check = 'deferred'
    with torch.no_grad():
        for _ in range(3):
            net(**data, iter_val=1e7)
    net.check_f16x3_domain()                                   # three frames inside the domain
    with torch.no_grad():
        net.cnl_mlp.module.pts_linears[0].weight.mul_(3e4)
        net.cnl_mlp.module.pts_linears[0].bias.mul_(3e4)
        net(**data, iter_val=1e7)                              # frame 4: outside
    with pytest.raises(RuntimeError, match='frame 4'):
        net.check_f16x3_domain()
    net.check_f16x3_domain()                                   # reported once, then clean
