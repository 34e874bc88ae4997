"""GPU, SURVEY 8(b): the drop-in boundary -- the operator FFI (`_gridencoder`), its dtype dispatch, the module seam.
"""
import os
import numpy as np
import pytest
import torch

from tests import util
from tests.gpu_util import (DEV, T, same, build_network, frame_to_device, per_frame_cpu, stagewise_oracle_render, _dev_model,
                            _clusters, stagewise_table, _torchrun)

pytestmark = pytest.mark.gpu


def test_ops_refuse_cpu_tensors(ops):
    with pytest.raises(RuntimeError):
        ops.knn_small(torch.zeros(4, 3), torch.zeros(8, 3), 3)
    with pytest.raises(RuntimeError):          # unsupported template dims raise like the reference
        x = torch.zeros(4, 7, device=DEV)
        ops.grid_encode_forward(x, torch.zeros(64, 2, device=DEV), torch.tensor([0, 64], dtype=torch.int32, device=DEV),
                                torch.zeros(1, 4, 2, device=DEV), 4, 7, 2, 1, 1.0, 16)
    with pytest.raises(RuntimeError):
        ops.grad_total_variation(None, None, None, None, 1e-7, 1, 4, 2, 16, 0.5, 16)


def test_operator_seam_dtype_errors(ops):
    """gridencoder.cu:467 dispatches float / double / half and all three are built; tensors of a call whose dtypes do not match
    are refused, as data_ptr<scalar_t>() would."""
    x = torch.rand(8, 4, device=DEV)
    off = torch.tensor([0, 64], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):                       # double embeddings need double outputs
        ops.grid_encode_forward(x, torch.zeros(64, 2, device=DEV, dtype=torch.float64), off, torch.zeros(1, 8, 2, device=DEV),
                                8, 4, 2, 1, 1.0, 16)
    with pytest.raises(RuntimeError):
        ops.grid_encode_backward(torch.zeros(1, 8, 2, device=DEV, dtype=torch.float64), x, torch.zeros(64, 2, device=DEV), off,
                                 torch.zeros(64, 2, device=DEV), 8, 4, 2, 1, 1.0, 16)
    with pytest.raises(RuntimeError):                       # ... and integer tensors have no dispatch case
        ops.grid_encode_forward(x, torch.zeros(64, 2, device=DEV, dtype=torch.int32), off, torch.zeros(1, 8, 2, device=DEV),
                                8, 4, 2, 1, 1.0, 16)
    with pytest.raises(RuntimeError):                       # half embeddings need half outputs, as data_ptr<scalar_t>() insists
        ops.grid_encode_forward(x, torch.zeros(64, 2, device=DEV, dtype=torch.float16), off, torch.zeros(1, 8, 2, device=DEV),
                                8, 4, 2, 1, 1.0, 16)
    with pytest.raises(RuntimeError, match='C = 1'):        # the reference's half atomicAdd for odd C is an empty stub
        h = torch.float16
        ops.grid_encode_backward(torch.zeros(1, 8, 1, device=DEV, dtype=h), x, torch.zeros(64, 1, device=DEV, dtype=h), off,
                                 torch.zeros(64, 1, device=DEV, dtype=h), 8, 4, 1, 1, 1.0, 16)


@pytest.mark.parametrize('D,Cc,gridtype,interp,align', [(4, 2, 0, 0, False), (3, 4, 1, 1, True), (2, 1, 0, 0, False), (5, 8, 0, 0, False)])
def test_grid_encode_double_dispatch(ops, oracle, D, Cc, gridtype, interp, align):
    """gridencoder.cu:467,500 with scalar_t = double, through `_gridencoder` (the dtype of the embeddings / grad selects the case,
    as AT_DISPATCH does): outputs and dy_dx BIT-EXACT against the oracle (same float cell arithmetic, one double fma per corner in
    corner order); the backward's atomics reorder double sums: 1e-13 relative to the largest entry."""
    import _gridencoder as _backend
    from occnerf_amd.gridencoder import grid_offsets
    rng = np.random.RandomState(D * 10 + Cc)
    L, B = 6, 777
    offs, pls = grid_offsets(D, L, 1.6, 4, 12, align_corners=align)
    emb = rng.uniform(-1, 1, (int(offs[-1]), Cc))
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[0], x[1], x[2], x[4] = 0.0, 1.0, -1e-6, 0.5
    x[3, -1] = 1.0 + 1e-6
    S = np.log2(pls)
    f64 = torch.float64
    out = torch.empty(L, B, Cc, device=DEV, dtype=f64)
    dy = torch.empty(B, L * D * Cc, device=DEV, dtype=f64)
    _backend.grid_encode_forward(T(x), T(emb), T(offs), out, B, D, Cc, L, S, 4, dy, gridtype, align, interp)
    want, want_dy = oracle.grid_encode_forward_f64(x, emb, offs, float(S), 4, True, gridtype, align, interp)
    same(out.cpu().numpy(), want, 'float64 outputs')
    same(dy.cpu().numpy(), want_dy, 'float64 dy_dx')
    assert not out[:, 2].any() and not out[:, 3].any()
    out2 = torch.empty(L, B, Cc, device=DEV, dtype=f64)
    _backend.grid_encode_forward(T(x), T(emb), T(offs), out2, B, D, Cc, L, S, 4, None, gridtype, align, interp)
    assert torch.equal(out, out2)
    grad = rng.randn(L, B, Cc)
    ge = torch.zeros(emb.shape, device=DEV, dtype=f64)
    gi = torch.zeros(B, D, device=DEV, dtype=f64)
    _backend.grid_encode_backward(T(grad), T(x), T(emb), T(offs), ge, B, D, Cc, L, S, 4, dy, gi, gridtype, align, interp)
    wge, wgi = oracle.grid_encode_backward_f64(grad, x, offs, emb.shape[0], Cc, float(S), 4, want_dy, gridtype, align, interp)
    assert np.abs(ge.cpu().numpy() - wge).max() <= 1e-13 * max(1.0, np.abs(wge).max())
    same(gi.cpu().numpy(), wgi, 'float64 grad_inputs')            # (a fixed-order fma chain per (sample, dimension): exact)


def test_grid_encoder_module_under_autocast(ops):
    """grid.py:42-45: under autocast the module casts the embeddings to half (even C), the output is half, the input
    stays float, and the gradient arrives at the fp32 parameter; without autocast everything stays fp32."""
    from occnerf_amd.gridencoder import GridEncoder
    torch.manual_seed(0)
    enc = GridEncoder(input_dim=3, num_levels=4, level_dim=2, base_resolution=4, log2_hashmap_size=10).to(DEV)
    enc.embeddings.data.uniform_(-1, 1)
    x = torch.rand(257, 3, device=DEV)
    with torch.autocast('cuda', dtype=torch.float16):
        y = enc(x, bound=None)
        assert y.dtype == torch.float16
        y.float().square().sum().backward()
    g16 = enc.embeddings.grad.clone()
    assert g16.dtype == torch.float32 and bool(torch.isfinite(g16).all()) and float(g16.abs().max()) > 0
    enc.embeddings.grad = None
    y32 = enc(x, bound=None)
    assert y32.dtype == torch.float32
    y32.square().sum().backward()
    assert float((y.float() - y32).detach().abs().max()) <= 8 * 2.0 ** -11 * max(1.0, float(y32.detach().abs().max()))
    assert float((g16 - enc.embeddings.grad).abs().max()) <= 0.05 * float(enc.embeddings.grad.abs().max())


def test_reference_state_dict_surface():
    net, ctx = build_network(0, False, S=32)
    keys = list(net.state_dict().keys())
    assert keys == list(ctx['sd'].keys())
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(ctx['sd'][k].shape), k


def test_gridencoder_module_is_importable_under_the_reference_name(ops, oracle):
    """`import _gridencoder as _backend` (grid.py:9) and the EXACT positional calls of grid.py:55 (forward: outputs
    [L,B,C] and the optional dy_dx written in place) and grid.py:83 (backward: into a zero-initialised grad_embeddings, grad_inputs
    only when dy_dx was kept), on the renderer's encoder shape (D = 4, C = 2, L = 16) and a small one, against the oracle."""
    import _gridencoder as _backend
    from occnerf_amd.gridencoder import grid_offsets
    assert sorted(n for n in dir(_backend) if not n.startswith('_')) == ['grad_total_variation', 'grid_encode_backward',
                                                                          'grid_encode_forward']
    for D, C, L, H, log2T, B, pls_in in ((4, 2, 16, 16, 19, 5000, 1.3819), (3, 2, 4, 4, 10, 333, 2.0)):
        rng = np.random.RandomState(B)
        offs, pls = grid_offsets(D, L, pls_in, H, log2T)
        emb_np = rng.uniform(-1, 1, (int(offs[-1]), C)).astype(np.float32)
        x_np = rng.uniform(0, 1, (B, D)).astype(np.float32)
        x_np[1] = 1.0
        x_np[2, 0] = -1e-6                                                  # outside [0, 1]: a zero row
        inputs, embeddings, offsets = T(x_np), T(emb_np), T(offs)
        S = np.log2(pls)                                                    # grid.py:39 passes numpy's float64 scalar
        gridtype, align_corners, interpolation = 0, False, 0
        for calc_grad_inputs in (False, True):
            # ---- grid.py:46-55
            outputs = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)
            dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype) if calc_grad_inputs else None
            _backend.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners,
                                         interpolation)
            want, want_dy = oracle.grid_encode_forward(x_np, emb_np, offs, float(S), H, calc_grad_inputs, gridtype, align_corners,
                                                       interpolation)
            same(outputs.cpu().numpy(), want, 'outputs[L,B,C]')
            assert not outputs[:, 2].any()
            if calc_grad_inputs:
                same(dy_dx.cpu().numpy(), want_dy, 'dy_dx')
            # ---- grid.py:73-83
            g_np = rng.randn(B, L * C).astype(np.float32)
            grad = T(g_np).view(B, L, C).permute(1, 0, 2).contiguous()
            grad_embeddings = torch.zeros_like(embeddings)
            grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None
            _backend.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                                          gridtype, align_corners, interpolation)
            wge, wgi = oracle.grid_encode_backward(grad.cpu().numpy(), x_np, offs, emb_np.shape[0], C, float(S), H, want_dy,
                                                   gridtype, align_corners, interpolation)
            # (atomics / tile sums reorder the fp32 additions: 1e-5 relative to the largest entry, as in test_a_rows.py)
            assert np.abs(grad_embeddings.cpu().numpy() - wge).max() <= 1e-5 * max(1.0, np.abs(wge).max())
            if grad_inputs is not None:
                assert np.abs(grad_inputs.cpu().numpy() - wgi).max() <= 1e-5 * max(1.0, np.abs(wgi).max())
    with pytest.raises(RuntimeError, match='grad_total_variation'):
        _backend.grad_total_variation(inputs, embeddings, torch.zeros_like(embeddings), offsets, 1e-7, B, D, C, L, S, H, 0, False)


def test_shencoder_module_imports_and_refuses_by_name():
    """occnerf_mlp.py:6 -> shencoder/sphere_harmonics.py:9: `import _shencoder` must succeed; the encoder is never evaluated."""
    import importlib.util
    import _shencoder as _backend
    # (importing the `core.nets` package parses sys.argv, as the reference's does: load the package file itself)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('core_nets_occnerf_shencoder',
                                                  os.path.join(root, 'core', 'nets', 'occnerf', 'shencoder', '__init__.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    SHEncoder = mod.SHEncoder
    enc = SHEncoder(input_dim=3, degree=4)
    assert enc.output_dim == 16
    with pytest.raises(NotImplementedError, match='sh_encode_forward'):
        _backend.sh_encode_forward(None, None, 0, 3, 4, None)
    with pytest.raises(NotImplementedError, match='sh_encode_backward'):
        _backend.sh_encode_backward(None, None, 0, 3, 4, None, None)
    with pytest.raises(NotImplementedError):
        enc(torch.zeros(2, 3))


@pytest.mark.parametrize('L,B', [(1, 300), (5, 1000), (16, 4097), (16, 70001)])
def test_operator_forward_xcd_form_is_bit_identical(ops, L, B):
    """the D = 4, C = 2 operator forward with the level pairs dealt to the XCDs (the default from 32 768 samples up;
    knob grid_xcd = 1 forces it, 2 forbids it) against the sample-major kernel: odd L, L < 16, rows outside [0, 1], B not a
    multiple of 256 -- outputs and dy_dx bit for bit."""
    from occnerf_amd import _lib
    from occnerf_amd.gridencoder import grid_offsets
    rng = np.random.RandomState(L * 7 + B)
    offs, pls = grid_offsets(4, L, 1.3819, 16, 19)
    emb = T(rng.uniform(-1, 1, (int(offs[-1]), 2)).astype(np.float32))
    x = rng.uniform(0, 1, (B, 4)).astype(np.float32)
    x[0] = 0.0
    x[1] = 1.0
    x[5, 2] = 1.0 + 1e-6
    x[B - 1, 0] = -1e-6
    x[B // 2] = -0.25
    xs, off = T(x), T(offs)
    S = float(np.log2(pls))
    res = {}
    try:
        for mode in (1, 2):
            assert _lib.lib().occnerf_experiment_knob(b'grid_xcd', mode) >= 0
            for want_dy in (False, True):
                out = torch.full((L, B, 2), 7.0, device=DEV)
                dy = torch.full((B, L * 4 * 2), 7.0, device=DEV) if want_dy else None
                ops.grid_encode_forward(xs, emb, off, out, B, 4, 2, L, S, 16, dy, 0, False, 0)
                res[mode, want_dy] = (out, dy)
    finally:
        _lib.lib().occnerf_experiment_knob(b'grid_xcd', 0)
    for want_dy in (False, True):
        a, b = res[1, want_dy], res[2, want_dy]
        assert torch.equal(a[0].view(torch.int32), b[0].view(torch.int32))
        if want_dy:
            assert torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))
    assert not res[1, False][0][:, B - 1].any() and not res[1, False][0][:, 5].any()
