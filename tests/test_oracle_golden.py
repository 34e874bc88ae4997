"""CPU: the C oracle against golden vectors recorded from the reference's own Python
(oracle/ref_harness/make_golden.py).  This is what pins the oracle (prompt rule 3)."""
import os
import numpy as np
import pytest
import torch

from tests import util

CASES = util.GOLDEN_CASES


@pytest.fixture(scope='module', params=CASES)
def case(request):
    g = util.load_golden(request.param)
    ctx = util.model_context(int(g['meta.seed']), util.level(g))
    return g, ctx


def test_checkpoint_recipe_reproduces(case):
    g, ctx = case
    from occnerf_amd.checkpoint import tensor_digest
    assert list(g['sd.keys']) == list(ctx['sd'].keys())
    for k, d in zip(g['sd.keys'], g['sd.digests']):
        assert tensor_digest(ctx['sd'][str(k)]) == str(d), k
    for i in range(3):
        assert np.array_equal(g[f'model.fps{i}'], ctx['fps'][i])
    assert tensor_digest(torch.from_numpy(ctx['normals'])) == str(g['model.point_norms_digest'])
    assert ctx['bound'] == float(g['meta.bound'])


def test_sampler_and_warp(case, oracle):
    g, ctx = case
    S = int(g['meta.S'])
    rays = np.concatenate([g['in.rays'][0], g['in.rays'][1], g['in.near'], g['in.far']], -1)
    t_vals = torch.linspace(0., 1., steps=S).numpy()
    z, pts = oracle.sample_rays(rays, t_vals)
    assert np.array_equal(z, g['comp.z_vals'])
    assert np.abs(pts - g['warp.pts']).max() <= 5e-7
    vol = _volume(g, ctx)
    xs, mk = oracle.motion_field(g['warp.pts'], g['warp.Rs'], g['warp.Ts'], vol,
                                 g['in.cnl_bbox_min_xyz'], g['in.cnl_bbox_scale_xyz'])
    assert np.abs(mk - g['warp.mask'].reshape(-1)).max() <= 2e-6
    assert np.abs(xs - g['warp.x_skel'].reshape(-1, 3)).max() <= 2e-5


def _volume(g, ctx):
    """Motion-weight volume: decoded by the product's torch module from the seeded
    checkpoint (CPU here); pinned against the golden slices."""
    from occnerf_amd.modules import MotionWeightVolumeDecoder
    dec = MotionWeightVolumeDecoder()
    dec.load_state_dict({k[len('mweight_vol_decoder.'):]: v for k, v in ctx['sd'].items()
                         if k.startswith('mweight_vol_decoder.')})
    frame = _frame(g)
    with torch.no_grad():
        vol = dec(torch.from_numpy(frame['motion_weights_priors'])[None])[0].numpy()
    assert np.abs(vol[:, ::4, ::4, ::4] - g['mw.vol_slice']).max() <= 1e-5
    return vol


def _frame(g):
    from occnerf_amd import synth
    return synth.make_frame(img_size=int(g['meta.img_size']), pose72=g['meta.pose72'],
                            orbit_frame=int(g['meta.orbit_frame']))


def test_msknn_bit_exact(case, oracle):
    g, ctx = case
    xyz = g['cnl.xyz']
    got = oracle.msknn(xyz, ctx['point_base'], ctx['fps'], k=10)
    want = g['cnl.knn_idxs'].astype(np.int32)
    assert np.array_equal(got, want)
    # independent float64 brute force on the finest scale (ties excluded)
    d = np.linalg.norm(xyz[:256, None].astype(np.float64) - ctx['point_base'][None].astype(np.float64), axis=-1)
    ref = np.argsort(d, axis=1, kind='stable')[:, :10]
    assert util.knn_mismatch_is_tie(xyz[:256], ctx['point_base'], got[:256, 0], ref)


def test_msknn_tie_suite_oracle(oracle):
    """the kNN restatement on an adversarial tie model (tests/util.py::knn_tie_model: exact duplicates,
    queries equidistant to 2 ... 24 support points at every scale, the k = 10 cut inside a tie group) against the documented
    KeOps rule -- lowest row of the scale's block first -- evaluated in exact integer arithmetic."""
    for seed in (0, 1, 2):
        base, sets, q, want = util.knn_tie_model(seed=seed)
        got = oracle.msknn(q, base, sets[1:], k=10)
        assert np.array_equal(got, want), seed
    # k = 3 single-scale search (network.py:265, the per-point block) on the same model: ties to the lower row as well
    base, sets, q, want = util.knn_tie_model(seed=3)
    assert np.array_equal(oracle.knn(q, base, 3), want[:, 0, :3])


def test_point_sdf(case, oracle):
    g, ctx = case
    kb, dist = oracle.point_sdf(ctx['point_cloud'], ctx['point_base'], ctx['normals'])
    assert np.abs(kb - g['cnl.point_cloud']).max() <= 1e-7
    assert np.abs(dist - g['cnl.point_sdf'].ravel()).max() <= 1e-7
    assert np.array_equal(g['cnl.learnable_points'], ctx['point_cloud'])


def test_grid_encode(case, oracle):
    g, ctx = case
    for tag in ('enc_sample', 'enc_point'):
        out, _ = oracle.grid_encode_forward(g[tag + '.in'], ctx['embeddings'], ctx['offsets'],
                                            ctx['S'], ctx['H'])
        got = out.transpose(1, 0, 2).reshape(out.shape[1], -1)
        assert np.array_equal(got, g[tag + '.out'])


def test_canonical_mlp(case, oracle):
    g, ctx = case
    kb, dist = oracle.point_sdf(ctx['point_cloud'], ctx['point_base'], ctx['normals'])
    table = oracle.point_table(kb, dist, ctx['point_cloud'], ctx['bound'], ctx['embeddings'],
                               ctx['offsets'], ctx['S'], ctx['H'])
    tol = util.pick(g, 1e-7, 2e-3, 3e-4)     # 1-ulp input change x O(1) (amplified) / 0.05 (trained-like) fine-level features
    assert np.abs(table[:, :32] - g['enc_point.out']).max() <= tol
    Wg, Bg, Wc, Bc = util.canonical_mlp_params(ctx['sd'])
    raw, mlp_in = oracle.canonical_mlp(g['cnl.xyz'], g['cnl.knn_idxs'].astype(np.int32),
                                       ctx['point_base'], ctx['normals'], ctx['counter'], table,
                                       ctx['bound'], ctx['embeddings'], ctx['offsets'], ctx['S'],
                                       ctx['H'], Wg, Bg, Wc, Bc, want_mlp_in=True)
    # encoder input / output as the reference fed / got them
    assert np.abs(mlp_in[:, 36:] - g['enc_sample.out']).max() <= tol
    want = g['cnl.raw']
    assert np.abs(raw[:, 4] - want[:, 4]).max() <= 1e-6            # signed distance
    # rgb logits, sigma.  With O(1) hash features (amplify) a 1-ulp difference in the encoder
    # input moves finest-level features by ~3e-4 (scale 4.4e3 cells x 6e-8); that conditioning
    # is the reference's own, so the tight check is done on reference-fed inputs below.
    assert np.abs(raw[:, :4] - want[:, :4]).max() <= util.pick(g, 2e-5, 5e-4, 2e-2)
    x = mlp_in.copy()
    x[:, 36:] = g['enc_sample.out']
    assert np.abs(_mlp_f64(x, Wg, Bg, Wc, Bc) - want[:, :4]).max() <= util.pick(g, 2e-5, 2e-4, 5e-3)
    # (trained-like: sigma = 640 x a 256-term dot product - 28, |sigma| up to 30: fp32 summation noise scales with it)
    assert np.abs(_mlp_f64(mlp_in, Wg, Bg, Wc, Bc) - raw[:, :4]).max() <= util.pick(g, 1e-5, 1e-5, 5e-4)


def _mlp_f64(x, Wg, Bg, Wc, Bc):
    h = x.astype(np.float64)
    for W, b in zip(Wg[:-1], Bg[:-1]):
        h = np.maximum(h @ W.T.astype(np.float64) + b, 0)
    geo = h @ Wg[-1].T.astype(np.float64) + Bg[-1]
    h = np.concatenate([geo[:, 1:], x[:, :35], x[:, 36:]], -1)
    for W, b in zip(Wc[:-1], Bc[:-1]):
        h = np.maximum(h @ W.T.astype(np.float64) + b, 0)
    rgb = h @ Wc[-1].T.astype(np.float64) + Bc[-1]
    return np.concatenate([rgb, geo[:, :1]], -1)


def test_nonrigid(case, oracle):
    g, ctx = case
    if not int(g['meta.non_rigid']):
        pytest.skip('T-pose case: non-rigid MLP is skipped (run.py:130)')
    W, B = util.nonrigid_params(ctx['sd'])
    out = oracle.nonrigid(g['nr.xyz_in'], g['nr.cond'], np.ones(6, np.float32), W, B)
    assert np.abs(out - g['nr.xyz_out']).max() <= 1e-6


def test_composite(case, oracle):
    g, ctx = case
    rgb, acc, w, dep, tp = oracle.raw2outputs(g['comp.raw'], g['comp.mask'][..., 0],
                                              g['comp.z_vals'], g['comp.rays_d'], g['in.bgcolor'])
    assert np.abs(rgb - g['comp.rgb']).max() <= 2e-6
    assert np.abs(acc - g['comp.acc']).max() <= 2e-6
    assert np.abs(dep - g['comp.depth']).max() <= 1e-5
    assert np.abs(w - g['comp.weights']).max() <= 2e-6
    util.assert_term_points(tp, g)
    assert np.array_equal(g['comp.rgb'], g['out.rgb'])


# ----------------------------------------------------------------------------- ray generation
@pytest.mark.parametrize('tag', ['t32', 'f32', 'c64'])
def test_gen_rays_oracle(tag):
    """oracle.gen_rays against the reference's own camera_util functions (golden rays_cameras.npz:
    float32 T-pose camera, float32 orbit camera, float64 off-centre camera)."""
    from oracle import oracle as orc
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'rays_cameras.npz'))
    img = int(g[f'{tag}.img'])
    ro, rd, near, far, mask = orc.gen_rays(g[f'{tag}.K'], g[f'{tag}.E'], img, img, g[f'{tag}.bbox_min'],
                                           g[f'{tag}.bbox_max'])
    assert np.array_equal(mask, g[f'{tag}.mask'])
    assert rd.dtype == g[f'{tag}.rays_d'].dtype
    assert np.array_equal(rd, g[f'{tag}.rays_d']) and np.array_equal(ro, g[f'{tag}.rays_o'])
    assert np.array_equal(near, g[f'{tag}.near']) and np.array_equal(far, g[f'{tag}.far'])


# ----------------------------------------------------------------------------- fixture hygiene
@pytest.mark.parametrize('name', util.GOLDEN_CASES)
def test_golden_rays_are_what_synth_generates_at_head(name):
    """The golden ray subsets are the rays the frame generator hands over TODAY (in-place direction clamp of
    camera_util.py:163-212 included): a change to occnerf_amd/synth.py that moves them must regenerate the fixtures.
    Runs everywhere (no reference needed)."""
    from occnerf_amd import synth
    g = util.load_golden(name)
    frame = synth.make_frame(img_size=int(g['meta.img_size']), pose72=g['meta.pose72'],
                             orbit_frame=int(g['meta.orbit_frame']))
    sel = g['in.ray_select']
    assert np.array_equal(frame['rays'][:, sel], g['in.rays'])
    assert np.array_equal(frame['near'][sel], g['in.near']) and np.array_equal(frame['far'][sel], g['in.far'])
    assert np.array_equal(frame['ray_mask'], g['in.ray_mask'])


@pytest.mark.skipif(not os.path.isdir('/root/reference/core/nets/occnerf'),
                    reason='build container only: needs the reference checkout')
def test_make_golden_reproduces_committed_fixtures(tmp_path):
    """`python oracle/ref_harness/make_golden.py` at HEAD regenerates the committed fixtures: inference cases and the
    ray cameras array for array, the training case to thread-order noise in the reference's own autograd."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (the two 2 048-ray float32 + float64 "truth" fixtures take five minutes in full: every 16th of their rays is regenerated
    # -- rays are independent -- and compared with the committed arrays at the same stride, bit for bit)
    env = dict(os.environ, OCCNERF_GOLDEN_DIR=str(tmp_path), OCCNERF_TRUTH_STRIDE='16')
    r = subprocess.run([sys.executable, os.path.join(root, 'oracle', 'ref_harness', 'make_golden.py'),
                        'rays', 'tpose', 'movement', 'train', 'trained', 'truth'], env=env, cwd=str(tmp_path), capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    made = sorted(os.listdir(tmp_path))
    assert made == ['freeview_trained_s128.npz', 'freeview_trained_s32.npz', 'freeview_trained_truth_s128.npz',
                    'freeview_trained_truth_s32.npz', 'movement_amp_s32_f3.npz',
                    'movement_amp_s32_f9.npz', 'rays_cameras.npz', 'tpose_ri_s32.npz', 'train_amp_s32.npz', 'train_ri_s32.npz']
    for f in made:
        a, b = np.load(tmp_path / f), np.load(os.path.join(util.GOLDEN_DIR, f))
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            if 'truth' in f:
                want = b[k]
                if k == 'in.rays':
                    want = want[:, ::16]
                elif want.ndim >= 1 and want.shape[0] == b['fragile'].shape[0] and not k.startswith(('sd.', 'meta.')):
                    want = want[::16]
                assert a[k].dtype == want.dtype and np.array_equal(a[k], want), (f, k)
            elif f.startswith('train') and a[k].dtype.kind == 'f' and (k.startswith('grad.') or k.startswith('out.')):
                scale = max(float(np.abs(b[k]).max()), 1e-30)
                assert np.abs(a[k].astype(np.float64) - b[k]).max() <= 1e-5 * scale + 1e-9, (f, k)
            else:
                assert np.array_equal(a[k], b[k]), (f, k)


def test_oracle_against_the_float64_truth(oracle):
    """The trained-like truth fixture (2 048 rays, reference in float32 and in float64): the CPU oracle chain on every 8th
    ray is as close to the float64 truth as the reference's own float32 run is (three-way table printed); the full set runs
    on the GPU box against HIP (tests/test_a_rows.py::test_trained_truth_three_way)."""
    from oracle.chain import golden_frame, model_context, stagewise_oracle_render
    g = util.load_golden('freeview_trained_truth_s32')
    assert g['truth.depth'].dtype == np.float64 and g['out.depth'].dtype == np.float32 and g['in.rays'].shape[1] == 2048
    sel = np.arange(0, 2048, 8)
    frame = golden_frame(g)
    frame['rays'], frame['near'], frame['far'] = g['in.rays'][:, sel], g['in.near'][sel], g['in.far'][sel]
    o = stagewise_oracle_render(g, model_context(int(g['meta.seed']), util.level(g)), frame=frame)
    ok = ~g['fragile'][sel]
    print()
    for k in ('rgb', 'alpha', 'depth'):
        t, r = g['truth.' + k][sel], g['out.' + k][sel]
        e_o = np.abs(o[k] - t).reshape(len(sel), -1).max(1)[ok]
        e_r = np.abs(r - t).reshape(len(sel), -1).max(1)[ok]
        e_or = np.abs(o[k] - r).reshape(len(sel), -1).max(1)[ok]
        print(f'   {k:5s}: |reference fp32 - truth| max {e_r.max():.2e} mean {e_r.mean():.2e}   |oracle - truth| max {e_o.max():.2e} '
              f'mean {e_o.mean():.2e}   |oracle - reference fp32| max {e_or.max():.2e}')
        assert e_o.mean() <= 1.15 * e_r.mean() + 1e-7 and e_o.max() <= 1.25 * e_r.max() + 1e-6, k
        assert e_or.max() <= (5e-4 if k == 'depth' else 1e-4), k


def test_oracle_half_conversions_match_ieee(oracle):
    """The integer-arithmetic half <-> float conversions behind the oracle's at::Half restatement (gridencoder.cu:467's
    half dispatch case): every half pattern to float, and float -> half round-to-nearest-even incl. ties, subnormals,
    overflow, against numpy's IEEE binary16."""
    import ctypes as C
    allh = np.arange(65536, dtype=np.uint16)
    f = np.empty(65536, np.float32)
    oracle.lib().oc_half_to_float(allh.ctypes.data_as(C.POINTER(C.c_uint16)), f.ctypes.data_as(C.POINTER(C.c_float)),
                                  C.c_int64(65536))
    ref = allh.view(np.float16).astype(np.float32)
    assert bool(((f.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(f) & np.isnan(ref))).all())
    rng = np.random.RandomState(0)
    with np.errstate(over='ignore', invalid='ignore'):
        mids = ((ref[:-1].astype(np.float64) + ref[1:].astype(np.float64)) / 2)
        mids = mids[np.isfinite(mids)].astype(np.float32)
        v = np.concatenate([(rng.randn(200000) * 10 ** rng.uniform(-9, 5, 200000)).astype(np.float32), mids,
                            np.array([0, -0.0, 65504, 65519.9, 65520, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, np.inf,
                                      -np.inf], np.float32)])
        want = v.astype(np.float16)
    h, _ = oracle.half_roundtrip(v)
    assert np.array_equal(h.view(np.uint16), want.view(np.uint16))
