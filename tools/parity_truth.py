"""Three-way parity on the trained-like truth fixtures (tests/golden/freeview_trained_truth_s{32,128}.npz: 2 048 rays each,
rendered by the unmodified reference in its own float32 AND in float64): HIP and the CPU oracle chain on the same rays;
which of reference-fp32 / oracle / HIP is closest to the float64 truth, and where HIP exceeds the 1e-4 gate against the
reference, how far the reference itself is from the truth on that ray.
    python3 tools/parity_truth.py [--precision fp32] [--same-preamble] [--dump out.npz]  > profiles/rNN_parity_truth.md
--same-preamble: the oracle chain is fed HIP's per-frame preamble outputs (Rs, Ts, volume) instead of its own torch-CPU
evaluation, which separates preamble noise from the per-sample kernels.
The oracle and the fixtures are the checkers here (test infrastructure)."""
import argparse
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
from occnerf_amd import ops, seeded  # noqa: E402
from oracle import chain  # noqa: E402

GOLD = os.path.join(os.getcwd(), 'tests', 'golden')


def hip_preamble(net, data, iter_val=1e7):
    """(Rs, Ts, vol) exactly as Network.forward's render branch computes them (csrc/preamble.hip)."""
    return tuple(t.cpu().numpy() for t in net.render_preamble(data, iter_val))


def three_way(name, precision='fp32', same_preamble=False, with_oracle=True):
    gz = np.load(os.path.join(GOLD, name + '.npz'))
    g = {k: gz[k] for k in gz.files}
    ctx = chain.model_context(int(g['meta.seed']), int(g['meta.amplify']))
    net = seeded.build_network(int(g['meta.seed']), int(g['meta.amplify']), S=int(g['meta.S']), non_rigid=True,
                               mlp_precision=precision, state_dict=ctx['sd'])
    frame = chain.golden_frame(g)
    data = seeded.frame_to_device(frame, 'cuda:0')
    with torch.no_grad():
        out = net(**data, iter_val=1e7)
    res = {'hip.' + k: out[k].cpu().numpy() for k in ('rgb', 'alpha', 'depth')}
    if with_oracle:
        o = chain.stagewise_oracle_render(g, ctx, frame=frame, preamble=hip_preamble(net, data) if same_preamble else None)
        res.update({'oracle.' + k: o[k] for k in ('rgb', 'alpha', 'depth')})
    for k in ('rgb', 'alpha', 'depth'):
        res['ref.' + k], res['truth.' + k] = g['out.' + k], g['truth.' + k]
    res['fragile'] = g['fragile']
    return res


def per_ray(a, b):
    e = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    return e.reshape(e.shape[0], -1).max(1)


def report(name, res, precision, same_preamble):
    ok = ~res['fragile']
    n = ok.size
    print(f'## {name}: {n} rays ({int((~ok).sum())} flagged fragile: a live sample within 2e-5 of a neighbour-set / inside-vote '
          f'discontinuity), HIP {precision}' + (', oracle fed HIP\'s preamble outputs' if same_preamble else '') + '\n')
    who = [('reference fp32', 'ref'), ('CPU oracle chain', 'oracle'), ('HIP', 'hip')]
    who = [w for w in who if (w[1] + '.rgb') in res]
    print('Distance to the float64 truth, non-fragile rays (max | p99 | rays > 1e-4):\n')
    print('| output | ' + ' | '.join(w[0] for w in who) + ' |')
    print('|---|' + '---|' * len(who))
    for k in ('rgb', 'alpha', 'depth'):
        cells = []
        for _, tag in who:
            e = per_ray(res[f'{tag}.{k}'], res[f'truth.{k}'])[ok]
            cells.append(f'{e.max():.2e} / {np.percentile(e, 99):.2e} / {int((e > 1e-4).sum())}')
        print(f'| {k} | ' + ' | '.join(cells) + ' |')
    print('\nHIP against the reference\'s float32 run (the 1e-4 gate), non-fragile rays:\n')
    print('| output | max | p99 | rays > 1e-4 | of them: reference itself > 1e-4 from the truth | HIP no further from the truth than '
          'max(reference, 5e-5) |')
    print('|---|---|---|---|---|---|')
    for k in ('rgb', 'alpha', 'depth'):
        e = per_ray(res[f'hip.{k}'], res[f'ref.{k}'])[ok]
        et_h = per_ray(res[f'hip.{k}'], res[f'truth.{k}'])[ok]
        et_r = per_ray(res[f'ref.{k}'], res[f'truth.{k}'])[ok]
        over = e > 1e-4
        print(f'| {k} | {e.max():.2e} | {np.percentile(e, 99):.2e} | {int(over.sum())} | {int((over & (et_r > 1e-4)).sum())} | '
              f'{100.0 * float((et_h <= np.maximum(et_r, 5e-5)).mean()):.2f} % |')
    if 'oracle.rgb' in res:
        print('\nHIP against the CPU oracle chain, all rays (the two share every discrete decision bit for bit):\n')
        print('| output | max | p99 | rays > 1e-4 |')
        print('|---|---|---|---|')
        for k in ('rgb', 'alpha', 'depth'):
            e = per_ray(res[f'hip.{k}'], res[f'oracle.{k}'])
            print(f'| {k} | {e.max():.2e} | {np.percentile(e, 99):.2e} | {int((e > 1e-4).sum())} |')
    fr = res['fragile']
    if fr.any():
        print(f'\nThe {int(fr.sum())} fragile rays (kept in the file, flagged): HIP vs reference max '
              + ', '.join(f"{k} {per_ray(res['hip.' + k], res['ref.' + k])[fr].max():.2e}" for k in ('rgb', 'alpha', 'depth')))
    print()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--precision', default='fp32')
    ap.add_argument('--same-preamble', action='store_true')
    ap.add_argument('--no-oracle', action='store_true')
    ap.add_argument('--dump')
    args = ap.parse_args()
    from oracle import oracle as orc
    orc.build()
    print(f'# Trained-like checkpoint, float64 truth: reference fp32 / CPU oracle / HIP ({args.precision})\n')
    dump = {}
    for name in ('freeview_trained_truth_s32', 'freeview_trained_truth_s128'):
        res = three_way(name, args.precision, args.same_preamble, not args.no_oracle)
        report(name, res, args.precision, args.same_preamble)
        dump.update({f'{name}.{k}': v for k, v in res.items()})
    if args.dump:
        np.savez_compressed(args.dump, **dump)


if __name__ == '__main__':
    main()
