"""A/B of the two split-operand canonical MLP kernels (experiment knob split_tail: 0 = chunk-major layers, 1 = layer
boundaries pipelined): bit-equality of the outputs on random rows and HIP-event time of a frame-sized launch.
    python3 tools/split_tail_ab.py [rows, default 17600000]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
from occnerf_amd import _lib, ops  # noqa: E402
from tests import util  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 17_600_000
dev = torch.device('cuda:0')
ctx = util.model_context(0, False)
Wg, Bg, Wc, Bc = util.canonical_mlp_params(ctx['sd'])
W = [torch.tensor(w, device=dev) for w in Wg + Wc]
B = [torch.tensor(b, device=dev) for b in Bg + Bc]
packed = ops.canonical_mlp_pack(W, B)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, 68, device=dev, generator=g) * 0.3
for name, ph in (('f16x3', ops.canonical_mlp_pack_f16(W)), ('bf16x3', ops.canonical_mlp_pack_bf16(W))):
    outs = []
    for knob in (0, 1, 0, 1):
        assert _lib.lib().occnerf_experiment_knob(b'split_tail', knob) >= 0
        raw = torch.zeros(N, 5, device=dev)
        ms = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.canonical_mlp_bf16x3(x, packed, ph, raw)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        print(f'{name} split_tail={knob}: {min(ms[1:]):.2f} ms (runs {", ".join(f"{m:.2f}" for m in ms)})', flush=True)
        outs.append(raw)
    print(f'{name}: outputs bit-identical between the two kernels: {torch.equal(outs[0], outs[1])}; '
          f'max |diff| {float((outs[0] - outs[1]).abs().max()):.3e}', flush=True)
    del outs
_lib.lib().occnerf_experiment_knob(b'split_tail', 0)
