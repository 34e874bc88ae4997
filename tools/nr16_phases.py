"""nr16::nonrigid_lds_kernel (the fp32 non-rigid MLP, 0.86 of the fp32-MFMA peak executed against the canonical kernel's 0.915): how
much of the gap is the embedding's sincosf?  The shipped kernel against a diagnostic build that replaces them by two VALU ops
(OCC_NR16_EXP_NO_SINCOS: wrong offsets, a timing build only).
    tools/nr16_phases.py --build ;  OCCNERF_HIP_LIB=tools/bin/<variant>.so python3 tools/nr16_phases.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {'nr16_shipped': [], 'nr16_no_sincos': ['-DOCC_NR16_EXP_NO_SINCOS']}


def build():
    src = os.path.join(ROOT, 'occnerf_amd', 'csrc')
    subprocess.check_call(['make', '-s', '-j8', '-C', src])
    objs = [os.path.join(src, 'build', f) for f in sorted(os.listdir(os.path.join(src, 'build'))) if f.endswith('.o') and f != 'nonrigid16.o']
    flags = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fvisibility=hidden', '-ffp-contract=off', '-Wno-unused-function']
    out = os.path.join(ROOT, 'tools', 'bin')
    os.makedirs(out, exist_ok=True)
    for name, defs in VARIANTS.items():
        o = os.path.join(out, name + '.o')
        subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + defs + ['-c', os.path.join(src, 'nonrigid16.hip'), '-o', o])
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(out, name + '.so'), o] + objs)
        os.remove(o)
        print('built', name)


if __name__ == '__main__':
    if '--build' in sys.argv:
        build()
        sys.exit(0)
    import numpy as np
    import torch
    from occnerf_amd import _lib, ops
    from occnerf_amd.seeded import build_network
    net = build_network(seed=0, amplify=False, S=128, non_rigid=True)
    pk = net._packed_weights()
    N = 17598062
    xyz = (torch.rand(N, 3, device='cuda') - 0.5) * 1.6
    cond = torch.randn(69, device='cuda') * 0.1
    hann = [1.0, 1.0, 1.0, 0.7, 0.3, 0.0]
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.nonrigid(xyz, cond, hann, pk['nr_w0'], pk['nr_b0'], pk['nr'], out=xyz)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f'{os.path.basename(_lib.LIB_PATH)}: {N} samples, non-rigid launch {np.median(ts):.3f} ms (min {min(ts):.3f})')
