"""Where the fused trunk forward (csrc/trunks.hip) spends its time: the shipped kernel against diagnostic builds that skip the
global stores of the saved activations (OCC_TRUNKS_EXP_NO_STORE) or the whole saving (OCC_TRUNKS_EXP_NO_SAVE).
    tools/trunks_phases.py --build ;  OCCNERF_HIP_LIB=tools/bin/<variant>.so python3 tools/trunks_phases.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {'trunks_shipped': [], 'trunks_no_store': ['-DOCC_TRUNKS_EXP_NO_STORE'], 'trunks_no_save': ['-DOCC_TRUNKS_EXP_NO_SAVE']}


def build():
    src = os.path.join(ROOT, 'occnerf_amd', 'csrc')
    subprocess.check_call(['make', '-s', '-j8', '-C', src])
    objs = [os.path.join(src, 'build', f) for f in sorted(os.listdir(os.path.join(src, 'build'))) if f.endswith('.o') and f != 'trunks.o']
    flags = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fvisibility=hidden', '-ffp-contract=off', '-Wno-unused-function']
    out = os.path.join(ROOT, 'tools', 'bin')
    os.makedirs(out, exist_ok=True)
    for name, defs in VARIANTS.items():
        o = os.path.join(out, name + '.o')
        subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + defs + ['-c', os.path.join(src, 'trunks.hip'), '-o', o])
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(out, name + '.so'), o] + objs)
        os.remove(o)
        print('built', name)


if __name__ == '__main__':
    if '--build' in sys.argv:
        build()
        sys.exit(0)
    import torch
    from occnerf_amd import _lib, ops
    from occnerf_amd.seeded import build_network
    net = build_network(0, False, S=128, non_rigid=True)
    cm = net.cnl_mlp.module
    W, b = cm.linear_params()
    W = [w.detach().float().contiguous() for w in W]
    blob = ops.canonical_mlp_pack(W, [x.detach().float().contiguous() for x in b])
    pk = ops.trunks_pack_bf16(W)
    M = 786432
    agg, var, enc = torch.randn(M, 35, device='cuda'), torch.rand(M, 1, device='cuda'), torch.randn(M, 32, device='cuda')
    ts = []
    for i in range(8):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.trunks_forward_bf16(agg, var, enc, blob, pk)
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e))
    print(f'{os.path.basename(_lib.LIB_PATH)}: {M} rows, {min(ts):.3f} ms (median {sorted(ts)[4]:.3f}) incl. the output allocations')
