"""Where the 1e-4 pixel budget goes: a golden case rendered by Network.forward on the GPU and, stage by stage, the HIP
kernels fed with the REFERENCE's recorded inputs -- every difference against what the unmodified reference produced
(tests/golden/*.npz, oracle/ref_harness/make_golden.py).
    python3 tools/parity_budget.py freeview_trained_s32 freeview_trained_s128 ..."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
from occnerf_amd import seeded  # noqa: E402
from oracle.chain import golden_frame, model_context, stagewise_oracle_render  # noqa: E402  (the checker)

DEV = 'cuda:0'
GOLDEN = os.path.join(os.getcwd(), 'tests', 'golden')


def budget(name, precision='fp32'):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    g = {k: z[k] for k in z.files}
    ctx = model_context(int(g['meta.seed']), int(g['meta.amplify']))
    net = seeded.build_network(int(g['meta.seed']), int(g['meta.amplify']), S=int(g['meta.S']),
                               non_rigid=bool(int(g['meta.non_rigid'])), mlp_precision=precision, state_dict=ctx['sd'])
    S = int(g['meta.S'])
    from occnerf_amd import ops
    rec = {}
    real = {n: getattr(ops, n) for n in ('sample_warp', 'nonrigid_rows', 'nonrigid', 'nonrigid_bf16x3', 'msknn_clustered', 'sample_features',
                                         'canonical_mlp', 'canonical_mlp_bf16x3', 'composite')}

    def wrap(n):
        def f(*a, **k):
            out = real[n](*a, **k)
            rec[n] = (a, k, out)
            return out
        return f
    for n in real:
        setattr(ops, n, wrap(n))
    net.cfg.dedup_repeated_samples = False
    net.cfg.skip_empty_samples = False                 # every sample in the reference's order: row i here = row i there
    try:
        with torch.no_grad():
            out = net(**seeded.frame_to_device(golden_frame(g), DEV), iter_val=1e7, ray_order_key=None)
    finally:
        for n in real:
            setattr(ops, n, real[n])
    print(f'== {name} ({precision}): S={S}, {g["out.alpha"].size} rays, alpha in [{g["out.alpha"].min():.3f}, {g["out.alpha"].max():.3f}]')
    for k in ('rgb', 'alpha', 'depth'):
        d = np.abs(out[k].cpu().numpy() - g['out.' + k])
        print(f'   pixel {k:5s}: max |hip - reference| = {d.max():.3e}   (gate 1e-4)   p99 {np.percentile(d, 99):.2e}')
    o = stagewise_oracle_render(g, ctx)
    for k in ('rgb', 'alpha', 'depth'):
        print(f'   pixel {k:5s}: max |cpu oracle - reference| = {np.abs(o[k] - g["out." + k]).max():.3e}')
    # ---- the frame's own intermediates against the reference's (rays are rendered in Morton order: undo it)
    n = g['out.alpha'].size
    a, k, _ = rec['composite']
    order = k.get('out_rows')
    order = np.arange(n) if order is None else order.cpu().numpy()
    inv = np.empty(n, np.int64)
    inv[order] = np.arange(n)
    raw = a[0].cpu().numpy().reshape(n, S, 5)[inv]
    mask = a[1].cpu().numpy().reshape(n, S)[inv]
    xyz = rec['msknn_clustered'][0][0].cpu().numpy().reshape(n, S, 3)[inv]          # positions the kNN saw (after the offset)
    knn = rec['msknn_clustered'][2].cpu().numpy().reshape(n, S, 40)[inv]
    gx = g['cnl.xyz'].reshape(n, S, 3)
    print(f'   stage warp+nonrigid: max |xyz - reference| = {np.abs(xyz - gx).max():.3e}; mask {np.abs(mask - g["comp.mask"].reshape(n, S)).max():.3e}')
    gk = g['cnl.knn_idxs'].reshape(n, S, 40).astype(np.int64)
    live = mask > 0
    mism = (knn != gk).any(-1) & live
    print(f'   stage kNN: samples (live) with any differing neighbour index: {int(mism.sum())} of {int(live.sum())}')
    graw = g['comp.raw'].reshape(n, S, 5)
    dsig, ddist, drgb = np.abs(raw[..., 3] - graw[..., 3]), np.abs(raw[..., 4] - graw[..., 4]), np.abs(raw[..., :3] - graw[..., :3]).max(-1)
    print(f'   stage features+MLP (live samples): max |sigma diff| {dsig[live].max():.3e}  |rgb logit diff| {drgb[live].max():.3e}  '
          f'|signed dist diff| {ddist[live].max():.3e}; sign(dist) flips: {int(((np.sign(raw[..., 4]) != np.sign(graw[..., 4])) & live).sum())}')
    w = g['comp.weights'].reshape(n, S)
    worst = int(np.argmax(np.abs(out['alpha'].cpu().numpy() - g['out.alpha'])))
    print(f'   worst ray {worst}: alpha {float(out["alpha"][worst]):.6f} vs {float(g["out.alpha"][worst]):.6f}; its samples with weight > 1e-3 or |dsigma| > 1e-2:')
    for sidx in range(S):
        if w[worst, sidx] > 1e-3 or (dsig[worst, sidx] > 1e-2 and live[worst, sidx]):
            print(f'      s={sidx:3d} weight {w[worst, sidx]:.4f} mask {mask[worst, sidx]:.4f} sigma {raw[worst, sidx, 3]:+.5f} vs {graw[worst, sidx, 3]:+.5f} '
                  f'dist {raw[worst, sidx, 4]:+.6f} vs {graw[worst, sidx, 4]:+.6f} knn differs {bool((knn[worst, sidx] != gk[worst, sidx]).any())} '
                  f'|dxyz| {np.abs(xyz[worst, sidx] - gx[worst, sidx]).max():.2e}')
    return out


if __name__ == '__main__':
    for name in sys.argv[1:] or ['freeview_trained_s32', 'freeview_trained_s128']:
        budget(name)
        budget(name, 'bf16x3')
