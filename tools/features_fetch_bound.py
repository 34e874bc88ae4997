"""Upper bound of any fetch-side change to sample_features8_kernel: the benchmark frame's feature launch with
the shipped kernel and with diagnostic builds whose hash-corner gathers (OCC_FEAT_EXP_RESIDENT_HASH) or table-row gathers
(OCC_FEAT_EXP_RESIDENT_ROWS) all hit a few cache-resident lines -- same instruction stream, no misses.
    tools/features_fetch_bound.py --build ;  OCCNERF_HIP_LIB=tools/bin/<variant>.so python3 tools/features_fetch_bound.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {'feat_shipped': [], 'feat_resident_hash': ['-DOCC_FEAT_EXP_RESIDENT_HASH'], 'feat_resident_rows': ['-DOCC_FEAT_EXP_RESIDENT_ROWS'],
            'feat_resident_both': ['-DOCC_FEAT_EXP_RESIDENT_HASH', '-DOCC_FEAT_EXP_RESIDENT_ROWS']}


def build():
    src = os.path.join(ROOT, 'occnerf_amd', 'csrc')
    subprocess.check_call(['make', '-s', '-j8', '-C', src])
    objs = [os.path.join(src, 'build', f) for f in sorted(os.listdir(os.path.join(src, 'build'))) if f.endswith('.o') and f != 'features.o']
    flags = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fvisibility=hidden', '-ffp-contract=off', '-Wno-unused-function']
    out = os.path.join(ROOT, 'tools', 'bin')
    os.makedirs(out, exist_ok=True)
    for name, defs in VARIANTS.items():
        o = os.path.join(out, name + '.o')
        subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + defs + ['-c', os.path.join(src, 'features.hip'), '-o', o])
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(out, name + '.so'), o] + objs)
        os.remove(o)
        print('built', name)


if __name__ == '__main__':
    if '--build' in sys.argv:
        build()
        sys.exit(0)
    import numpy as np
    import torch
    from occnerf_amd import _lib, ops, synth
    from occnerf_amd.seeded import build_network, frame_to_device
    net = build_network(seed=0, amplify=False, S=128, non_rigid=True)
    data = frame_to_device(synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28), 'cuda:0')
    calls = []
    real = ops.sample_features

    def grab(*a, **k):
        if a[0].shape[0] > 100000:
            calls.append((a, k))
        return real(*a, **k)
    ops.sample_features = grab
    with torch.no_grad():
        net(**data, iter_val=1e7)
    ops.sample_features = real
    a, k = calls[0]
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    n = int(k['count']) if k.get('count') is not None else a[0].shape[0]
    print(f'{os.path.basename(_lib.LIB_PATH)}: {n} listed samples, feature launch {np.median(ts):.3f} ms (min {min(ts):.3f})')
