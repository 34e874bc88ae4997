"""Shared helpers for the parity tests: golden loading and the seeded model context."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
GOLDEN_CASES = ['tpose_ri_s32', 'tpose_ri_s128', 'freeview_amp_s32', 'tpose_amp_s32',
                'movement_amp_s32_f3', 'movement_amp_s32_f9']


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
