"""N optimiser steps of the differentiable path at config-5 scale (6 144 rays x 128 samples, bf16 trunks) for rocprofv3:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -o t -- python3 tools/train_step_trace.py 8
prints ms/step measured with HIP events over the timed steps (the profiler's per-kernel table is divided by the same count)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
from occnerf_amd import synth  # noqa: E402
from occnerf_amd.optim import FusedAdam  # noqa: E402
from occnerf_amd.seeded import build_network, frame_to_device  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
precision = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
net = build_network(0, False, S=128, non_rigid=True)
net.cfg.perturb = 1.0
net.cfg.train_precision = precision
net.train()
frame = synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28)
if os.environ.get('OCC_TRAIN_RAYS', 'patches') == 'patches':      # the reference's batch: 6 patches of 32 x 32 pixels
    from occnerf_amd.seeded import patch_ray_selection
    sel = patch_ray_selection(frame, np.random.RandomState(0), 6, 32, full=True)
else:                                                             # 6 144 rays scattered over the frame (rounds 2-4's batch)
    sel = np.sort(np.random.RandomState(0).choice(frame['rays'].shape[1], 6144, replace=False))
for k in ('near', 'far'):
    frame[k] = frame[k][sel]
frame['rays'] = frame['rays'][:, sel]
data = frame_to_device(frame, 'cuda:0')
for k in ('cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor'):
    data[k] = data[k].cpu()
opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4)


def step():
    opt.zero_grad(set_to_none=True)
    out = net(**data, iter_val=1e7)
    loss = ((out['rgb'] - 0.5) ** 2).mean() + 0.1 * out['comp_loss'].mean()
    loss.backward()
    opt.step(max_grad_norm=1.0)
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    step()
e1.record()
torch.cuda.synchronize()
print(f'train step {e0.elapsed_time(e1) / steps:.3f} ms over {steps} steps (+3 warm-up), precision {precision}')
