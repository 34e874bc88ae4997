"""Parity soak: many frames (poses x cameras) of a checkpoint rendered by the HIP path and by the CPU oracle chain on the same
rays; per-ray errors, and every ray above the 1e-4 gate classified -- does it hold a live sample that sits on a discontinuity
of the function (a neighbour-set tie or an inside-vote flip within 2e-5, oracle/ref_harness/make_golden.py::fragile_rays'
criterion), where the reference on other hardware flips as well?
    python3 tools/parity_soak.py --level 2 --frames 40 --rays 128 --spp 64 > profiles/rNN_parity_soak.md
The oracle is the checker here (test infrastructure): it is pinned to the unmodified reference by tests/golden/."""
import argparse
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
from occnerf_amd import seeded, synth  # noqa: E402
from oracle.chain import model_context, stagewise_oracle_render  # noqa: E402


def fragile(xyz, mask, ctx, rel=2e-5):
    """[n] bool per ray, as make_golden.py::fragile_rays (float64 distances to the four point sets)."""
    n, S = mask.shape
    base = ctx['point_base'].astype(np.float64)
    normals = ctx['normals'].astype(np.float64)
    sets = [np.arange(base.shape[0])] + [np.asarray(f).astype(np.int64) for f in ctx['fps']]
    live = np.flatnonzero(mask.reshape(-1) > 0)
    q = xyz.astype(np.float64)[live]
    frag = np.zeros(live.size, bool)
    for lvl, idx in enumerate(sets):
        pts = base[idx]
        for lo in range(0, live.size, 2048):
            d = np.linalg.norm(q[lo:lo + 2048, None, :] - pts[None], axis=-1)
            part = np.argpartition(d, 11, axis=1)[:, :12]
            ds = np.take_along_axis(d, part, 1)
            o = np.argsort(ds, axis=1)
            ds, part = np.take_along_axis(ds, o, 1), np.take_along_axis(part, o, 1)
            gap = lambda j: (ds[:, j + 1] - ds[:, j]) / np.maximum(ds[:, j + 1], 1e-30)          # noqa: E731
            f = gap(9) < rel
            if lvl == 0:
                f |= gap(2) < rel
                nb = idx[part[:, :10]]
                dirs = q[lo:lo + 2048, None, :] - base[nb]
                dots = (dirs * normals[nb]).sum(-1)
                cnt = (dots < 0).sum(1)
                near0 = (np.abs(dots) < 1e-6 * np.linalg.norm(dirs, axis=-1) * np.linalg.norm(normals[nb], axis=-1)).any(1)
                f |= near0 & ((cnt == 5) | (cnt == 6))
            frag[lo:lo + 2048] |= f
    bad = np.zeros(n * S, bool)
    bad[live[frag]] = True
    return bad.reshape(n, S).any(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--level', type=int, default=2, help='checkpoint recipe: 0 random-init, 1 amplified, 2 trained-like')
    ap.add_argument('--frames', type=int, default=40)
    ap.add_argument('--rays', type=int, default=128)
    ap.add_argument('--spp', type=int, default=64)
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--precision', default='fp32')
    ap.add_argument('--same-preamble', action='store_true',
                    help="feed the oracle chain the HIP preamble's outputs (Rs, Ts, volume): per-sample kernels alone")
    args = ap.parse_args()
    from oracle import oracle as orc
    orc.build()
    ctx = model_context(0, args.level)
    net = seeded.build_network(0, args.level, S=args.spp, non_rigid=True, mlp_precision=args.precision, state_dict=ctx['sd'])
    rng = np.random.RandomState(0)
    errs = {k: [] for k in ('rgb', 'alpha', 'depth')}
    frag_all, alpha_all = [], []
    for t in range(args.frames):
        frame = synth.make_frame(img_size=args.size, pose72=synth.seeded_pose(100 + t), orbit_frame=int(rng.randint(0, 100)))
        R = frame['rays'].shape[1]
        sel = np.sort(rng.choice(R, min(args.rays, R), replace=False))
        frame['rays'], frame['near'], frame['far'] = frame['rays'][:, sel], frame['near'][sel], frame['far'][sel]
        data = seeded.frame_to_device(frame, 'cuda:0')
        with torch.no_grad():
            out = net(**data, iter_val=1e7)
        pre = tuple(t.cpu().numpy() for t in net.render_preamble(data)) if args.same_preamble else None
        o = stagewise_oracle_render(None, ctx, frame=frame, S=args.spp, non_rigid=True, preamble=pre)
        for k in errs:
            e = np.abs(out[k].cpu().numpy() - o[k])
            errs[k].append(e.reshape(len(sel), -1).max(1))
        frag_all.append(fragile(o['xyz'], o['mask'].reshape(len(sel), args.spp), ctx))
        alpha_all.append(o['alpha'])
    frag_all, alpha_all = np.concatenate(frag_all), np.concatenate(alpha_all)
    n = frag_all.size
    print(f'# Parity soak: HIP ({args.precision}) vs CPU oracle chain' + (' fed the HIP preamble outputs' if args.same_preamble else '') + f', checkpoint recipe {args.level} '
          f'({["random-init", "amplified", "trained-like"][args.level]}), {args.frames} frames (seeded poses 100.., random orbit '
          f'cameras, {args.size}x{args.size} image), {n} rays x {args.spp} samples\n')
    print(f'rays with alpha in (0.05, 0.95): {int(((alpha_all > 0.05) & (alpha_all < 0.95)).sum())}; alpha max {alpha_all.max():.3f}; '
          f'rays holding a live sample on a discontinuity (tie criterion 2e-5): {int(frag_all.sum())}\n')
    print('| output | p50 | p99 | max over all rays | max over the rays WITHOUT a fragile sample | rays > 1e-4 | of them fragile |')
    print('|---|---|---|---|---|---|---|')
    for k in errs:
        e = np.concatenate(errs[k])
        over = e > 1e-4
        print(f'| {k} | {np.percentile(e, 50):.2e} | {np.percentile(e, 99):.2e} | {e.max():.2e} | {e[~frag_all].max():.2e} | '
              f'{int(over.sum())} | {int((over & frag_all).sum())} |')


if __name__ == '__main__':
    main()
