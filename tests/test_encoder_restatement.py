"""CPU: a SECOND, independent restatement of the torch-ngp grid encoder (gridencoder/src/gridencoder.cu:50-245), written in
numpy straight from the CUDA source -- vectorised over the batch, uint32 wrap-around by numpy's own modular uint32 arithmetic,
float32 steps as numpy float32 operations, a fused multiply-add as the exactly rounded float64 evaluation (the product of two
floats is exact in float64; the one case in 2^29 where the double rounding could matter is detected and redone in exact
rational arithmetic) -- and asserted BIT-EQUAL to the C oracle (oracle/occnerf_oracle.c) that pins the HIP kernels.

The encoder is one of the two third-party kernels whose binary cannot run here (parity unpinned): what
remains unverifiable is which of the source's `a * b + c` expressions nvcc's default --fmad=true contracts and the last bits
of CUDA's exp2f (documented: 2 ulp).  `test_size_of_the_unverifiable_assumptions` flips each of them and puts the number of
golden outputs that move, and by how much, on record."""
import fractions

import numpy as np
import pytest

from tests import util

PRIMES = np.array([1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737], dtype=np.uint32)
F32 = np.float32


def fma32(a, b, c):
    """Correctly rounded fp32 fma of arrays, via float64 (exact product, one float64 rounding of the sum) with the
    double-rounding corner redone exactly."""
    a, b, c = np.broadcast_arrays(np.asarray(a, F32), np.asarray(b, F32), np.asarray(c, F32))
    s = a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)
    out = s.astype(F32)
    # a float64 whose 29 dropped bits are exactly 1000...0 sits on an fp32 midpoint: only then can the first rounding have
    # decided the second
    risky = np.flatnonzero(((s.view(np.uint64) & np.uint64((1 << 29) - 1)) == np.uint64(1 << 28)) & np.isfinite(s))
    for i in risky:
        exact = fractions.Fraction(float(a.flat[i])) * fractions.Fraction(float(b.flat[i])) + fractions.Fraction(float(c.flat[i]))
        lo, hi = np.nextafter(out.flat[i], F32(-np.inf)), np.nextafter(out.flat[i], F32(np.inf))
        best = min((lo, out.flat[i], hi), key=lambda v: (abs(fractions.Fraction(float(v)) - exact), int(np.asarray(v, F32).view(np.uint32)) & 1))
        out.flat[i] = best
    return out


def level_scale(level, S, H, contract=True, ulps=0):
    """gridencoder.cu:138 `exp2f(level * S) * H - 1.0f`.  contract: one fma (nvcc --fmad=true) or multiply, round, subtract."""
    t = F32(level) * F32(S)
    e = F32(np.exp2(np.float64(t)))                                    # correctly rounded exp2 of the fp32 product
    for _ in range(abs(ulps) if float(t) != np.floor(float(t)) else 0):      # (an integer power of two is exact in any exp2f)
        e = np.nextafter(e, F32(np.inf if ulps > 0 else -np.inf))
    if contract:
        return fma32(e, F32(H), F32(-1.0)).reshape(())[()]
    return F32(F32(e * F32(H)) - F32(1.0))


def grid_index(pos_grid, hashmap_size, resolution, gridtype, align_corners):
    """gridencoder.cu:50-84: pos_grid [B,D] uint32 -> [B] uint32 row (before `* C`)."""
    D = pos_grid.shape[1]
    index = np.zeros(pos_grid.shape[0], np.uint32)
    stride = 1
    for d in range(D):
        if stride > hashmap_size:
            break
        index = index + pos_grid[:, d] * np.uint32(stride & 0xFFFFFFFF)          # uint32: wraps
        stride = (stride * (resolution if align_corners else resolution + 1)) & 0xFFFFFFFF
    if gridtype == 0 and stride > hashmap_size:
        h = np.zeros(pos_grid.shape[0], np.uint32)
        for d in range(D):
            h ^= pos_grid[:, d] * PRIMES[d]
        index = h
    return index % np.uint32(hashmap_size)


def encode_numpy(x, emb, offsets, S, H, gridtype=0, align_corners=False, interp=0, contract_scale=True, contract_pos=True,
                 contract_acc=True, exp2_ulps=0, acc_dtype=None):
    """-> outputs[L,B,C] float32 (gridencoder.cu:87-198).  acc_dtype (e.g. np.longdouble): the scalar_t = double dispatch case
    evaluated above its own precision -- cell position and corner weights in float32 as the template keeps them, the weighted
    sum of the (float64) embeddings in acc_dtype."""
    x = np.asarray(x, F32)
    emb = np.asarray(emb, F32 if acc_dtype is None else np.float64)
    B, D = x.shape
    C, L = emb.shape[1], len(offsets) - 1
    out = np.zeros((L, B, C), F32 if acc_dtype is None else acc_dtype)
    oob = ((x < 0) | (x > 1)).any(1)
    with np.errstate(over='ignore', invalid='ignore'):
        for level in range(L):
            grid = emb[int(np.uint32(offsets[level])):]
            hashmap_size = int(offsets[level + 1] - offsets[level])
            scale = level_scale(level, S, H, contract_scale, exp2_ulps)
            resolution = int(np.ceil(scale)) + 1
            half = F32(0.0 if align_corners else 0.5)
            pos = fma32(x, scale, half) if contract_pos else (x * scale).astype(F32) + half
            pos_grid = np.floor(pos).astype(np.int64).astype(np.uint32)           # floorf, then float -> uint32
            pos = (pos - pos_grid.astype(F32)).astype(F32)
            if interp == 1:
                pos = (pos * pos * fma32(F32(-2.0), pos, F32(3.0))).astype(F32)      # (2 * val is exact: contraction or not)
            res = np.zeros((B, C), F32 if acc_dtype is None else acc_dtype)
            for idx in range(1 << D):
                w = np.ones(B, F32)
                pl = pos_grid.copy()
                for d in range(D):
                    if idx & (1 << d):
                        w = (w * pos[:, d]).astype(F32)
                        pl[:, d] = pos_grid[:, d] + np.uint32(1)
                    else:
                        w = (w * (F32(1) - pos[:, d]).astype(F32)).astype(F32)
                row = grid_index(pl, hashmap_size, resolution, gridtype, align_corners).astype(np.int64)
                for ch in range(C):
                    v = grid[row, ch]
                    if acc_dtype is not None:
                        res[:, ch] = res[:, ch] + w.astype(acc_dtype) * v.astype(acc_dtype)
                        continue
                    res[:, ch] = fma32(w, v, res[:, ch]) if contract_acc else (res[:, ch] + (w * v).astype(F32)).astype(F32)
            res[oob] = 0
            out[level] = res
    return out


@pytest.fixture(scope='module', params=['tpose_ri_s32', 'freeview_amp_s32', 'freeview_trained_s128'])
def enc_case(request):
    g = util.load_golden(request.param)
    return g, util.model_context(int(g['meta.seed']), util.level(g))


def test_numpy_restatement_equals_the_c_oracle_on_golden_inputs(enc_case, oracle):
    """The renderer's encoder (D = 4, C = 2, 16 levels: 2 dense, 14 hashed 2^19) on what the reference fed it -- per-sample
    inputs and per-point inputs of three checkpoints: numpy restatement == C oracle == the recorded outputs, bit for bit."""
    g, ctx = enc_case
    for tag in ('enc_sample', 'enc_point'):
        x = g[tag + '.in']
        x = x[::max(1, len(x) // 1500)]                                     # a spread of <= ~1500 rows keeps the CPU suite short
        mine = encode_numpy(x, ctx['embeddings'], ctx['offsets'], ctx['S'], ctx['H'])
        orc, _ = oracle.grid_encode_forward(x, ctx['embeddings'], ctx['offsets'], ctx['S'], ctx['H'])
        assert np.array_equal(mine.view(np.uint32), orc.view(np.uint32)), tag
        rec = g[tag + '.out'][::max(1, len(g[tag + '.in']) // 1500)]
        assert np.array_equal(mine.transpose(1, 0, 2).reshape(len(x), -1), rec), tag


@pytest.mark.parametrize('D,C,gridtype,interp,align', [(3, 2, 0, 1, False), (3, 4, 1, 0, True), (2, 8, 0, 0, False), (5, 2, 0, 0, False),
                                                       (4, 1, 1, 1, True)])
def test_numpy_restatement_equals_the_c_oracle_generic(oracle, D, C, gridtype, interp, align):
    """Other template instantiations of the operator seam (tiled grids, smoothstep, align_corners, D up to 5), including
    out-of-range inputs and inputs on cell corners."""
    from occnerf_amd.gridencoder import grid_offsets
    rng = np.random.RandomState(D * 10 + C)
    L = 8
    offsets, pls = grid_offsets(D, L, 1.6, 4, 12, align_corners=align)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(F32)
    x = rng.uniform(0, 1, (700, D)).astype(F32)
    x[0], x[1], x[2], x[4] = 0.0, 1.0, -1e-6, 0.5
    x[3, -1] = 1.0 + 1e-6
    x[5:40] = np.round(x[5:40] * 8) / 8                                       # on cell corners of the coarse levels
    S = float(np.log2(pls))
    mine = encode_numpy(x, emb, offsets, S, 4, gridtype, align, interp)
    orc, _ = oracle.grid_encode_forward(x, emb, offsets, S, 4, False, gridtype, align, interp)
    assert np.array_equal(mine.view(np.uint32), orc.view(np.uint32))


@pytest.mark.parametrize('D,C,gridtype,interp,align', [(4, 2, 0, 0, False), (3, 4, 1, 1, True), (2, 1, 0, 0, False)])
def test_double_dispatch_case_of_the_c_oracle(oracle, D, C, gridtype, interp, align):
    """gridencoder.cu:467 with scalar_t = double (round 6: the last dispatch case of the operator seam).  The C oracle's
    restatement (float cell position and weights, double embeddings, one fma per corner) against this file's numpy evaluation
    in extended precision: equal to a few ulps of float64; and with float32-representable embeddings it rounds to the float32
    case's outputs within one float32 ulp.  Backward: the scatter is linear in grad -- checked against the forward's own
    weights through <grad, forward(emb)> == <backward(grad), emb>."""
    from occnerf_amd.gridencoder import grid_offsets
    rng = np.random.RandomState(D * 100 + C)
    L = 6
    offsets, pls = grid_offsets(D, L, 1.7, 4, 11, align_corners=align)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), C))
    x = rng.uniform(0, 1, (500, D)).astype(F32)
    x[0], x[1], x[2] = 0.0, 1.0, -1e-6
    S = float(np.log2(pls))
    got, dy = oracle.grid_encode_forward_f64(x, emb, offsets, S, 4, True, gridtype, align, interp)
    assert got.dtype == np.float64 and dy.dtype == np.float64 and not got[:, 2].any() and not dy[2].any()
    want = encode_numpy(x, emb, offsets, S, 4, gridtype, align, interp, acc_dtype=np.longdouble)
    assert np.abs(got - want.astype(np.float64)).max() <= 8 * np.finfo(np.float64).eps
    emb32 = emb.astype(F32)
    got32, _ = oracle.grid_encode_forward(x, emb32, offsets, S, 4, False, gridtype, align, interp)
    got64, _ = oracle.grid_encode_forward_f64(x, emb32.astype(np.float64), offsets, S, 4, False, gridtype, align, interp)
    assert np.abs(got64 - got32).max() <= 2 ** -22                                   # |values| < 2: a few float32 ulps of the sum
    grad = rng.randn(L, len(x), C)
    ge, gi = oracle.grid_encode_backward_f64(grad, x, offsets, emb.shape[0], C, S, 4, dy, gridtype, align, interp)
    assert ge.dtype == np.float64 and gi.shape == x.shape
    lhs, rhs = float((grad * got).sum()), float((ge * emb).sum())
    assert abs(lhs - rhs) <= 1e-12 * max(1.0, abs(lhs))
    # input gradient: grad_inputs[b, d] = sum_{l, c} grad[l, b, c] dy_dx[b, l, d, c]
    want_gi = np.einsum('lbc,bldc->bd', grad, dy.reshape(len(x), L, D, C))
    assert np.abs(gi - want_gi).max() <= 1e-12 * max(1.0, np.abs(want_gi).max())


def test_uint32_wraparound_is_exercised(enc_case):
    """The hash really overflows 32 bits on the renderer's levels (a restatement in wider integers would differ)."""
    g, ctx = enc_case
    x = g['enc_sample.in'][:64]
    scale = level_scale(15, ctx['S'], ctx['H'])
    pg = np.floor(fma32(x, scale, F32(0.5))).astype(np.int64)
    wide = (pg[:, 1] * 2654435761) ^ (pg[:, 2] * 805459861)
    assert (wide >= 1 << 32).any()


def test_size_of_the_unverifiable_assumptions(enc_case, capsys):
    """Which `a * b + c` nvcc fuses cannot be checked without its binary; CUDA's exp2f is documented to 2 ulp.  Each
    assumption flipped on the golden inputs: how many of the [N, 32] outputs move, and by how much -- relative to the
    largest feature.  The effect is bounded here so that the record stays true.  The position inside a cell is the
    fractional part of a number as large as 2 900 (finest level), whose fp32 ulp is 2.4e-4 of a cell: a differently rounded
    `x * scale + 0.5` (fused or not: <= 6 % of the outputs move, by <= 5.5e-5 of the feature scale) or a level scale that
    is one ulp off (CUDA's exp2f: nearly every output of the 14 levels whose scale is not a power of two moves, by <= 7e-4
    of the feature scale) shift the interpolation weights by that much.  Measured round 4: the `exp2f * H - 1` contraction
    changes nothing (same 16 scales either way), the accumulation's contraction 1.3e-7.  In absolute terms <= 7e-8 on the
    random-init table, 2.6e-5 on the trained-like one, 6e-4 on the amplified one: the golden chain is bit-exact relative to
    THIS set of contractions and a correctly rounded exp2, and the unpinned remainder is worth < 1e-3 of the features."""
    g, ctx = enc_case
    x = g['enc_sample.in']
    x = x[::max(1, len(x) // 1200)]
    base = encode_numpy(x, ctx['embeddings'], ctx['offsets'], ctx['S'], ctx['H'])
    amp = float(np.abs(base).max())
    rows = []
    for name, kw in (('scale = exp2f*H, then - 1 (no fma)', dict(contract_scale=False)),
                     ('pos = x*scale, then + 0.5 (no fma)', dict(contract_pos=False)),
                     ('results += w*grid as multiply + add', dict(contract_acc=False)),
                     ('exp2f one ulp high', dict(exp2_ulps=1)), ('exp2f one ulp low', dict(exp2_ulps=-1))):
        alt = encode_numpy(x, ctx['embeddings'], ctx['offsets'], ctx['S'], ctx['H'], **kw)
        moved = alt != base
        worst = float(np.abs(alt - base).max())
        rows.append((name, int(moved.sum()), moved.size, worst, worst / max(amp, 1e-30)))
        assert worst <= 1e-3 * amp + 1e-12, (name, worst, amp)
    with capsys.disabled():
        print(f"\n   encoder assumptions on {str(g['sd.digests'][0])[:8]}.. (largest feature {amp:.3g}):")
        for name, n, tot, worst, rel in rows:
            print(f'      {name:40s}: {n:6d} of {tot} outputs move, max |change| {worst:.3e} = {rel:.1e} of the feature scale')
