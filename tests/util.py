"""Shared helpers for the parity tests: golden loading and the seeded model context."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
GOLDEN_CASES = ['tpose_ri_s32', 'tpose_ri_s128', 'freeview_amp_s32', 'tpose_amp_s32',
                'movement_amp_s32_f3', 'movement_amp_s32_f9', 'freeview_trained_s32', 'freeview_trained_s128']


def level(g):
    """Checkpoint recipe of a golden case (occnerf_amd/checkpoint.py): 0 random-init, 1 amplified, 2 trained-like."""
    return int(g['meta.amplify'])


def pick(g, random_init, amplified, trained):
    return (random_init, amplified, trained)[level(g)]


def pixel_tol(g):
    """End-to-end gate on rgb / alpha / depth against the reference's output: BASELINE.json's 1e-4 per-pixel L-infinity for
    the random-init AND the trained-like checkpoint; the amplified one (hash table U(+-1): a 1-ulp encoder-input difference
    moves a fine-level feature by 3e-4, in the reference as much as here) is held to 1e-3."""
    return pick(g, 1e-4, 1e-3, 1e-4)


def load_golden(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
    return {k: g[k] for k in g.files}


from oracle.chain import (canonical_mlp_params, mlp_params, model_context,  # noqa: E402,F401  (re-exported)
                          nonrigid_params)


def knn_mismatch_is_tie(q, s, got, want, rel=2e-6):
    """True when every differing neighbour slot is a near-tie in distance (the only
    legitimate source of kNN disagreement, SURVEY.md section 7 'kNN exactness')."""
    bad = np.argwhere(got != want)
    for i, j in bad:
        dg = np.linalg.norm(q[i].astype(np.float64) - s[got[i, j]].astype(np.float64))
        dw = np.linalg.norm(q[i].astype(np.float64) - s[want[i, j]].astype(np.float64))
        if abs(dg - dw) > rel * max(dg, dw):
            return False
    return True


def assert_term_points(tp_got, g):
    """arg-max alpha per ray (network.py:340) against the reference's `comp.term`: index for index, except where the
    candidates tie -- their alphas, recomputed in float64, are equal to 1e-6 of the ray's largest alpha, or EVERY alpha of
    the ray is below what fp32 resolves of 1 - exp(-x) (then each comes out as zero or one ulp by the expf in use)."""
    raw, mask, z = g['comp.raw'], g['comp.mask'][..., 0], g['comp.z_vals']
    n = z.shape[0]
    tp_h, tp_r = np.asarray(tp_got).ravel().astype(np.int64), g['comp.term'].ravel().astype(np.int64)
    bad = np.flatnonzero(tp_h != tp_r)
    if bad.size:
        d64 = np.concatenate([np.diff(z.astype(np.float64), axis=1), np.full((n, 1), 1e10)], 1) * \
            np.linalg.norm(g['comp.rays_d'].astype(np.float64), axis=-1, keepdims=True)
        sig = raw[..., 3].astype(np.float64)
        a64 = (1.0 - np.exp(-np.where(sig > 20, sig, np.log1p(np.exp(np.minimum(sig, 20)))) * d64)) * mask
        for r in bad:
            assert abs(a64[r, tp_h[r]] - a64[r, tp_r[r]]) <= 1e-6 * a64[r].max() or a64[r].max() < 6e-8, (r, tp_h[r], tp_r[r])
        assert bad.size <= max(1, n // 50)
