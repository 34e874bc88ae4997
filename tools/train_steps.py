"""A few optimiser steps at config-5 scale without the torch profiler (for rocprofv3 passes):
    rocprofv3 --pmc ... -- python3 tools/train_steps.py [bf16|fp32] [steps]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from occnerf_amd.seeded import build_network, frame_to_device
from occnerf_amd import synth
from occnerf_amd.optim import FusedAdam
net = build_network(0, False, S=128, non_rigid=True)
net.cfg.perturb = 1.0
net.cfg.train_precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
net.train()
frame = synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28)
R = frame['rays'].shape[1]
sel = np.sort(np.random.RandomState(0).choice(R, 6144, replace=False))
for k in ('near', 'far'): frame[k] = frame[k][sel]
frame['rays'] = frame['rays'][:, sel]
data = frame_to_device(frame, 'cuda:0')
opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = net(**data, iter_val=1e7)
    loss = ((out['rgb'] - 0.5) ** 2).mean() + 0.1 * out['comp_loss'].mean()
    loss.backward()
    opt.step(max_grad_norm=1.0)
    torch.cuda.synchronize(); print(f'step {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms')
