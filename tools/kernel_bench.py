"""Per-kernel timing at BASELINE configs[1] size on one MI355X (developer tool, not the headline
bench): python tools/kernel_bench.py [--samples N]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occnerf_amd import ops  # noqa: E402
from tests import util  # noqa: E402


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
    for variant, name in ((1, 'direct'), (0, 'lds   ')):
        tb = timeit(lambda: ops.canonical_mlp_bf16x3(mlp_in, packed, packed_h, raw, variant=variant))
        print(f'canonical_mlp bf16x3 {name} : {tb:8.2f} ms  {flop / tb / 1e9:8.1f} TFLOP/s (algorithmic)  '
              f'max|diff vs fp32| = {float((raw[:, :4] - r32[:, :4]).abs().max()):.3e}')


if __name__ == '__main__':
    main()
