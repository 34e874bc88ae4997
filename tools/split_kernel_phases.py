"""The split-operand MLP kernels (cfg.mlp_precision 'f16x3' / 'bf16x3') under the experiment knob split_refill:
  0 = shipped (refill pieces spread over the k-step), 1 = refill pieces at the barrier, 2 / 3 = the two with s_memtime phase stamps
(workgroup 1000 writes its phase lengths over its own output rows).  Prints the HIP-event time of a frame-sized launch of each
form, checks that their outputs are bit-identical, then the phase table; finally the non-rigid split kernel's launch time.
    python3 tools/split_kernel_phases.py [rows, default 17600000]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
from occnerf_amd import _lib, ops  # noqa: E402
from occnerf_amd.seeded import build_network  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 17_600_000
dev = torch.device('cuda:0')
net = build_network(seed=0, amplify=False, S=128, non_rigid=True, device=dev)
W, B = net.cnl_mlp.module.linear_params()
W, B = [w.detach() for w in W], [b.detach() for b in B]
packed = ops.canonical_mlp_pack(W, B)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, 68, device=dev, generator=g) * 0.3


def knob(v):
    assert _lib.lib().occnerf_experiment_knob(b'split_refill', v) >= 0


def timed(fn, reps=4):
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return ms


for name, ph in (('f16x3', ops.canonical_mlp_pack_f16(W)), ('bf16x3', ops.canonical_mlp_pack_bf16(W))):
    outs = []
    for k in (0, 1, 0, 1):
        knob(k)
        raw = torch.zeros(N, 5, device=dev)
        ms = timed(lambda: ops.canonical_mlp_bf16x3(x, packed, ph, raw))
        print(f'canonical {name} split_refill={k}: {min(ms[1:]):.2f} ms (runs {", ".join(f"{m:.2f}" for m in ms)})', flush=True)
        outs.append(raw)
    print(f'canonical {name}: outputs bit-identical between the forms: {all(torch.equal(outs[0], o) for o in outs[1:])}', flush=True)
    del outs
# the same instruction stream on all-zero operands (weights, biases and inputs 0): what the clock gives back when the matrix
# pipe's operands do not toggle -- if the launch gets faster, the sustained clock under load is what bounds the kernel
knob(0)
Wz, Bz = [torch.zeros_like(w) for w in W], [torch.zeros_like(b) for b in B]
pz, phz = ops.canonical_mlp_pack(Wz, Bz), ops.canonical_mlp_pack_f16(Wz)
raw = torch.zeros(N, 5, device=dev)
phr = ops.canonical_mlp_pack_f16(W)
ms_r = timed(lambda: ops.canonical_mlp_bf16x3(x, packed, phr, raw))
xz = torch.zeros_like(x)
ms_z = timed(lambda: ops.canonical_mlp_bf16x3(xz, pz, phz, raw))
print(f'canonical f16x3, random operands {min(ms_r[1:]):.2f} ms; all-zero operands {min(ms_z[1:]):.2f} ms (same kernel, same row count)', flush=True)
del xz
ph = ops.canonical_mlp_pack_f16(W)
names = ['prologue (aux copy, inputs, split)', 'sync + geometry L0 + split', '3 hidden layers (2 splits)', 'sigma dot + split',
         'head + bgeo split', 'colour L0 + split', '3 hidden layers (2 splits)', 'rgb dots + drain + store', 'TOTAL',
         'of which: relu_split phases']
for k in (2, 3):
    knob(k)
    raw = torch.zeros(N, 5, device=dev)
    for _ in range(2):
        ops.canonical_mlp_bf16x3(x, packed, ph, raw)
    torch.cuda.synchronize()
    d = raw[1000 * 128:1000 * 128 + 10, 0].cpu().numpy()
    print(f'phases of one workgroup (128 rows; s_memtime ticks = shader cycles), f16x3, refill pieces '
          f'{"spread over the k-step" if k == 2 else "at the barrier"}:')
    for nm, v in zip(names, d):
        print(f'  {nm:40s} {v:10.0f}')
knob(0)
del x, raw
# the non-rigid split kernel (two workgroups per CU since round 5)
nr_lin = [m for m in net.non_rigid_mlp.module.block_mlps if isinstance(m, torch.nn.Linear)]
Wd, Bd = [m.weight.detach() for m in nr_lin], [m.bias.detach() for m in nr_lin]
pk = ops.nonrigid_pack(Wd, Bd)
xyz = torch.rand(N, 3, device=dev, generator=g) * 2 - 1
cond = torch.randn(69, device=dev, generator=g) * 0.3
for name, pf in (('f16x3', ops.nonrigid_pack_f16(Wd)), ('bf16x3', ops.nonrigid_pack_bf16(Wd))):
    ms = timed(lambda: ops.nonrigid_bf16x3(xyz, cond, np.ones(6, np.float32), Wd[0], Bd[0], pk, pf))
    print(f'non-rigid {name}: {min(ms[1:]):.2f} ms for {N} rows (runs {", ".join(f"{m:.2f}" for m in ms)})')
ms = timed(lambda: ops.nonrigid(xyz, cond, np.ones(6, np.float32), Wd[0], Bd[0], pk))
print(f'non-rigid fp32: {min(ms[1:]):.2f} ms')
