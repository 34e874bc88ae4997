"""do the hashed levels want to be dealt to the XCDs?  The benchmark frame's 17.6 M encoder inputs (what
sample_features8_kernel encodes) through the operator-level D4C2 forward in the shipped sample-major form and with the level
pairs dealt to the XCDs (knob grid_xcd): same bits, time by HIP events; FETCH_SIZE comes from separate rocprofv3 --pmc passes
over this script (tools/xcd_levels.sh).
    python3 tools/xcd_levels.py [--reps 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
from occnerf_amd import _lib, ops, synth  # noqa: E402
from occnerf_amd.seeded import build_network, frame_to_device  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--mode', default='both', choices=['both', 'shipped', 'xcd'])
args = ap.parse_args()
net = build_network(0, False, S=128, non_rigid=True)
net.cfg.dedup_repeated_samples = False
data = frame_to_device(synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28), 'cuda:0')
grab = {}
real = ops.sample_features


def hook(*a, **k):
    if k.get('rows') is not None and k['rows'].shape[0] > 1000000:
        k2 = dict(k, want_enc_in=True)
        out = real(*a, **k2)
        grab['enc_in'], grab['count'] = out[2], k['count']
        return out[0], out[1], None
    return real(*a, **k)
ops.sample_features = hook
with torch.no_grad():
    net(**data, iter_val=1e7)
ops.sample_features = real
n = int(grab['count'])
x = grab['enc_in'][:n].contiguous()
enc = net.cnl_mlp.module.encoder
emb, off = enc.embeddings.detach(), enc.offsets
L = int(off.shape[0] - 1)
S, H = enc.log2_per_level_scale, enc.base_resolution
print(f'{n} encoder inputs of the benchmark frame; distinct rows: {int(torch.unique(x, dim=0).shape[0])}')
outs = {}
for mode, knobv in (('shipped', 2), ('xcd', 1)):          # knob grid_xcd: 2 = sample-major kernel, 1 = levels dealt to the XCDs
    if args.mode not in ('both', mode):
        continue
    assert _lib.lib().occnerf_experiment_knob(b'grid_xcd', knobv) >= 0
    out = torch.empty(L, n, 2, device='cuda:0')
    for _ in range(2):
        ops.grid_encode_forward(x, emb, off, out, n, 4, 2, L, S, H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        ops.grid_encode_forward(x, emb, off, out, n, 4, 2, L, S, H)
    e1.record()
    torch.cuda.synchronize()
    print(f'{mode:8s}: {e0.elapsed_time(e1) / args.reps:.3f} ms per pass over {n} samples x {L} levels')
    outs[mode] = out
_lib.lib().occnerf_experiment_knob(b'grid_xcd', 0)
if len(outs) == 2:
    print('bit-identical:', bool(torch.equal(outs['shipped'], outs['xcd'])))
