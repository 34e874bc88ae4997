"""Per-tile cost of the cluster-culled kNN on the benchmark frame and a rank's eighth of it.

    tools/knn_tile_stats.py --build                  (here or on the GPU box: compiles the variants below into tools/bin)
    OCCNERF_HIP_LIB=tools/bin/<variant>.so python3 tools/knn_tile_stats.py [--world 8] [--stats]

Variants (csrc/knn.hip compiled with -D flags and linked with the product's other objects; the shipped library has no
diagnostic code): `knn_q0` (OCC_KNN_QUEUE=0: round 5's immediate insertion), `knn_q4` (the shipped per-lane candidate queues),
`knn_q4_lazy` (queues drained only when full), `knn_q8`, and `*_stats` = the same with OCC_KNN_TILE_STATS (s_memtime per tile,
squared extent, searched queries, points scanned, insertion-chain executions -> a device buffer through
occnerf_debug_knn_tile_stats)."""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {'knn_q0': ['-DOCC_KNN_QUEUE=0'], 'knn_q0_stats': ['-DOCC_KNN_QUEUE=0', '-DOCC_KNN_TILE_STATS'],
            'knn_q4': [], 'knn_q4_stats': ['-DOCC_KNN_TILE_STATS'], 'knn_q4_lazy': ['-DOCC_KNN_LAZY_DRAIN=1'],
            'knn_q4_lazy_stats': ['-DOCC_KNN_LAZY_DRAIN=1', '-DOCC_KNN_TILE_STATS'],
            'knn_q2': ['-DOCC_KNN_QUEUE=2'], 'knn_q8': ['-DOCC_KNN_QUEUE=8']}


def build():
    src = os.path.join(ROOT, 'occnerf_amd', 'csrc')
    subprocess.check_call(['make', '-s', '-j8', '-C', src])
    objs = [os.path.join(src, 'build', f) for f in sorted(os.listdir(os.path.join(src, 'build'))) if f.endswith('.o') and f != 'knn.o']
    flags = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fvisibility=hidden', '-ffp-contract=off', '-Wno-unused-function']
    out = os.path.join(ROOT, 'tools', 'bin')
    os.makedirs(out, exist_ok=True)
    for name, defs in VARIANTS.items():
        o = os.path.join(out, name + '.o')
        subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + defs + ['-c', os.path.join(src, 'knn.hip'), '-o', o])
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(out, name + '.so'), o] + objs)
        os.remove(o)
        print('built', name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--build', action='store_true')
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--stats', action='store_true')
    args = ap.parse_args()
    if args.build:
        return build()
    import numpy as np
    import torch
    from occnerf_amd import _lib, ops, synth
    from occnerf_amd.parallel import ShardedRenderer
    from occnerf_amd.seeded import build_network, frame_to_device
    dev = torch.device('cuda:0')
    net = build_network(seed=0, amplify=False, S=128, non_rigid=True, device=dev)
    data = frame_to_device(synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28), dev)
    grabbed = []
    real = ops.msknn_clustered

    def grab(xyz, n_rays, S, cl, seed, **kw):
        grabbed.append(dict(xyz=xyz.clone(), n=n_rays, S=S, cl=cl, seed=seed, kw={k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()}))
        return real(xyz, n_rays, S, cl, seed, **kw)
    ops.msknn_clustered = grab
    with torch.no_grad():
        net(**data, iter_val=1e7)
        host = {k: (v.cpu() if k in ('cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor') else v) for k, v in data.items()}
        r = ShardedRenderer(net, dev, single=True)
        r.world, r.rank, r.collective, r.verify_plan = args.world, 0, True, False
        mine = r._build_plan(host)['mine']['cuda']
        sub = dict(host)
        sub['rays'], sub['near'], sub['far'] = data['rays'][:, mine].contiguous(), data['near'][mine], data['far'][mine]
        net(**sub, iter_val=1e7)
    ops.msknn_clustered = real
    print('library', _lib.LIB_PATH)
    for name, g in zip(('full frame', f'rank 0 of {args.world}'), grabbed):
        call = lambda: real(g['xyz'], g['n'], g['S'], g['cl'], g['seed'], **g['kw'])      # noqa: E731
        ref = call()
        ms = []
        for _ in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            call()
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        tiles = ((g['n'] + 63) // 64) * ((g['S'] + 3) // 4)
        print(f'{name}: {g["n"]} rays, {tiles} tiles, kernel + list pre-pass {np.median(ms):.3f} ms (min {min(ms):.3f})')
        # identical indices on the listed queries, whatever the variant: checksum for cross-variant comparison
        rows = g['kw'].get('rows')
        sel = ref if rows is None else ref[rows[:int(g['kw']['count'])].long()]
        print(f'  index checksum {int(sel.long().sum())} over {sel.shape[0]} queries')
        if args.stats:
            lib = ctypes.CDLL(_lib.LIB_PATH)
            buf = torch.zeros(tiles, 6, device=dev)
            assert lib.occnerf_debug_knn_tile_stats(ctypes.c_void_p(buf.data_ptr())) == 0
            call()
            torch.cuda.synchronize()
            lib.occnerf_debug_knn_tile_stats(None)
            st = buf.cpu().numpy()
            st = st[st[:, 0] > 0]
            cyc, ext, nq, sc, dr = st[:, 0], np.sqrt(st[:, 1]), st[:, 2], st[:, 3], st[:, 4]
            print(f'  searched jobs {len(st)}, cycles: mean {cyc.mean():.0f} p50 {np.median(cyc):.0f} p99 {np.percentile(cyc, 99):.0f} max {cyc.max():.0f}; '
                  f'sum {cyc.sum() / 1e6:.0f} M')
            print(f'  per scanned point: {cyc.sum() / sc.sum():.0f} cycles, {dr.sum() / sc.sum():.2f} insertion-chain executions')
            print('  extent bin (m)   jobs   share of cycles   mean cycles   max cycles   mean points scanned   mean queries   mean chain executions')
            edges = [0, 0.05, 0.1, 0.2, 0.3, 0.5, 0.8, 1.2, 10]
            for lo, hi in zip(edges[:-1], edges[1:]):
                m = (ext >= lo) & (ext < hi)
                if m.any():
                    print(f'  {lo:4.2f}-{hi:5.2f}   {int(m.sum()):7d}   {cyc[m].sum() / cyc.sum():8.3f}   {cyc[m].mean():12.0f}   {cyc[m].max():10.0f}   '
                          f'{sc[m].mean():10.0f}   {nq[m].mean():8.1f}   {dr[m].mean():10.0f}')
            top = np.argsort(-cyc)[:8]
            print('  longest jobs (cycles, extent m, queries, points):', [(int(cyc[i]), round(float(ext[i]), 3), int(nq[i]), int(sc[i])) for i in top])


if __name__ == '__main__':
    main()
