"""Per-kernel timing at BASELINE configs[1] size on one MI355X (developer tool, not the headline
bench): python tools/kernel_bench.py [--samples N]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occnerf_amd import ops  # noqa: E402
from oracle import chain as util  # noqa: E402  (weights of the seeded checkpoint as numpy)


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


torch.set_grad_enabled(False)


def bench_nonrigid():
    ctx = util.model_context(0, False)
    W, B = util.nonrigid_params(ctx['sd'])
    Wd = [torch.from_numpy(w).cuda() for w in W]
    Bd = [torch.from_numpy(b).cuda() for b in B]
    pk, ph = ops.nonrigid_pack(Wd, Bd), ops.nonrigid_pack_bf16(Wd)
    N = 183784 * 128
    xyz = (torch.rand(N, 3, device='cuda') - 0.5) * 2
    cond = torch.randn(69, device='cuda') * 0.3
    hann = np.ones(6, np.float32)
    out = torch.empty_like(xyz)
    t32 = timeit(lambda: ops.nonrigid(xyz, cond, hann, Wd[0], Bd[0], pk, out=out))
    r32 = out.clone()
    tb = timeit(lambda: ops.nonrigid_bf16x3(xyz, cond, hann, Wd[0], Bd[0], pk, ph, out=out))
    print(f'nonrigid fp32   : {t32:8.2f} ms')
    td = timeit(lambda: ops.nonrigid(xyz, cond, hann, Wd[0], Bd[0], pk, out=out, direct=True))
    print(f'nonrigid direct : {td:8.2f} ms   max|diff| = {float((out - r32).abs().max()):.3e}')
    print(f'nonrigid bf16x3 : {tb:8.2f} ms   max|diff| = {float((out - r32).abs().max()):.3e}')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=183784 * 128)
    args = ap.parse_args()
    N = args.samples
    dev = 'cuda:0'
    ctx = util.model_context(0, False)
    Wg, Bg, Wc, Bc = util.canonical_mlp_params(ctx['sd'])
    W = [torch.from_numpy(w).to(dev) for w in Wg + Wc]
    B = [torch.from_numpy(b).to(dev) for b in Bg + Bc]
    packed, packed_h = ops.canonical_mlp_pack(W, B), ops.canonical_mlp_pack_bf16(W)
    mlp_in = torch.randn(N, 68, device=dev) * 0.3
    raw = torch.zeros(N, 5, device=dev)
    t32 = timeit(lambda: ops.canonical_mlp(mlp_in, packed, raw))
    r32 = raw.clone()
    flop = 923136.0 * N
    print(f'canonical_mlp fp32          : {t32:8.2f} ms  {flop / t32 / 1e9:8.1f} TFLOP/s')
    td = timeit(lambda: ops.canonical_mlp(mlp_in, packed, raw, direct=True))
    print(f'canonical_mlp fp32 direct   : {td:8.2f} ms  {flop / td / 1e9:8.1f} TFLOP/s  '
          f'max|diff vs default| = {float((raw[:, :4] - r32[:, :4]).abs().max()):.3e}')
    for variant, name in ((1, 'direct'), (0, 'lds   ')):
        tb = timeit(lambda: ops.canonical_mlp_bf16x3(mlp_in, packed, packed_h, raw, variant=variant))
        print(f'canonical_mlp bf16x3 {name} : {tb:8.2f} ms  {flop / tb / 1e9:8.1f} TFLOP/s (algorithmic)  '
              f'max|diff vs fp32| = {float((raw[:, :4] - r32[:, :4]).abs().max()):.3e}')


def bench_frame_stages():
    """knn variants on the real benchmark frame (needs the whole pipeline up to xyz)."""
    from occnerf_amd import synth
    from occnerf_amd.seeded import build_network, frame_to_device
    net = build_network(0, False, S=128, non_rigid=True)
    frame = synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28)
    data = frame_to_device(frame, 'cuda:0')
    c = net._context()
    import time
    for culling in (False, True):
        net.cfg.knn_culling = culling
        net(**data, iter_val=1e7)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = net(**data, iter_val=1e7)
        torch.cuda.synchronize()
        print(f'frame, knn_culling={culling}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms')
        if culling:
            assert torch.equal(out['rgb'], ref['rgb'])
        ref = out


def bench_knn():
    """kNN variants alone on the benchmark frame's canonical sample positions."""
    from occnerf_amd import synth
    from occnerf_amd.seeded import build_network, frame_to_device
    net = build_network(0, False, S=128, non_rigid=True)
    frame = synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28)
    data = frame_to_device(frame, 'cuda:0')
    grabbed = {}
    real = ops.msknn_clustered

    def grab(xyz, n_rays, S, cl, seed, mask=None, **kw):
        grabbed.update(xyz=xyz.clone(), n=n_rays, S=S, cl=cl, seed=seed, mask=mask, kw=kw)
        return real(xyz, n_rays, S, cl, seed, mask=mask, **kw)
    ops.msknn_clustered = grab
    with torch.no_grad():
        net(**data, iter_val=1e7)
    ops.msknn_clustered = real
    c = net._context()
    x, n, S, cl, seed = grabbed['xyz'], grabbed['n'], grabbed['S'], grabbed['cl'], grabbed['seed']
    tb = timeit(lambda: ops.msknn(x, c['points'], c['index_map'], c['scale_begin'], c['seed']))
    tc = timeit(lambda: ops.msknn_clustered(x, n, S, cl, seed))
    print(f'msknn brute     : {tb:8.2f} ms')
    print(f'msknn clustered : {tc:8.2f} ms')
    if grabbed['mask'] is not None:
        tm = timeit(lambda: ops.msknn_clustered(x, n, S, cl, seed, mask=grabbed['mask']))
        print(f'msknn clustered, dead samples skipped : {tm:8.2f} ms')
    if grabbed['kw'].get('rows') is not None:
        tm = timeit(lambda: ops.msknn_clustered(x, n, S, cl, seed, **grabbed['kw']))
        print(f'msknn clustered, query list           : {tm:8.2f} ms')
        from occnerf_amd import ops as _o
        m = torch.zeros(x.shape[0], device=x.device)
        m[grabbed['kw']['rows'][:int(grabbed['kw']['count'])].long()] = 1.0
        tm = timeit(lambda: ops.msknn_clustered(x, n, S, cl, seed, mask=m))
        print(f'msknn clustered, same samples by mask : {tm:8.2f} ms')


def bench_stage(name):
    """Time one pipeline op in isolation on the benchmark frame (inputs grabbed from a real forward)."""
    from occnerf_amd import synth
    from occnerf_amd.seeded import build_network, frame_to_device
    net = build_network(0, False, S=128, non_rigid=True)
    net.cfg.dedup_repeated_samples = '--dedup' in sys.argv      # default: every live sample (17.6 M rows)
    frame = synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28)
    data = frame_to_device(frame, 'cuda:0')
    real = getattr(ops, name)
    grabbed = {}

    def grab(*a, **k):
        grabbed['a'] = tuple(x.clone() if torch.is_tensor(x) else x for x in a)
        grabbed['k'] = {kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in k.items()}
        return real(*a, **k)
    setattr(ops, name, grab)
    net(**data, iter_val=1e7)
    setattr(ops, name, real)
    t = timeit(lambda: real(*grabbed['a'], **grabbed['k']), n=5)
    print(f'{name:24s}: {t:8.3f} ms')


if __name__ == '__main__':
    if '--nonrigid' in sys.argv:
        bench_nonrigid()
        sys.exit(0)
    if '--stage' in sys.argv:
        bench_stage(sys.argv[sys.argv.index('--stage') + 1])
        sys.exit(0)
    if '--knn' in sys.argv:
        bench_knn()
        sys.exit(0)
    if '--frame' in sys.argv:
        sys.argv.remove('--frame')
        bench_frame_stages()
        sys.exit(0)
    main()
