"""One optimiser step of the differentiable path at config-5 scale (6144 rays x 128 samples) on one GPU: step time and a
torch-profiler table.  python tools/train_step_profile.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from occnerf_amd.seeded import build_network, frame_to_device
from occnerf_amd import synth
net = build_network(0, False, S=128, non_rigid=True)
net.cfg.perturb = 1.0
net.cfg.train_precision = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
print('train_precision', net.cfg.train_precision)
net.train()
frame = synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28)
R = frame['rays'].shape[1]
rng = np.random.RandomState(0)
sel = np.sort(rng.choice(R, 6144, replace=False))
for k in ('near', 'far'): frame[k] = frame[k][sel]
frame['rays'] = frame['rays'][:, sel]
data = frame_to_device(frame, 'cuda:0')
params = [p for p in net.parameters() if p.requires_grad]
from occnerf_amd.optim import FusedAdam
opt = FusedAdam(params, lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    out = net(**data, iter_val=1e7)
    loss = ((out['rgb'] - 0.5) ** 2).mean() + 0.1 * out['comp_loss'].mean()
    loss.backward()
    opt.step(max_grad_norm=1.0)
    return float(loss)
for i in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(3): step()
torch.cuda.synchronize(); print('train step ms', (time.perf_counter() - t0) / 3 * 1e3, 'rays', 6144, 'samples', 6144 * 128)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=70, max_name_column_width=60))
