"""Kernel times of the opt-in bf16x3 path on the bench frame:  rocprofv3 --kernel-trace --stats -- python3 tools/alt_frame.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from occnerf_amd import synth
from occnerf_amd.seeded import build_network, frame_to_device
net = build_network(seed=0, amplify=False, S=128, non_rigid=True, mlp_precision='bf16x3')
frame = synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28)
data = frame_to_device(frame, 'cuda:0')
for k in ('cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor'):
    data[k] = data[k].cpu()
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        out = net(**data, iter_val=1e7, ray_order_key='alt')
    torch.cuda.synchronize(); print(f'frame {it}: {(time.perf_counter() - t0) * 1e3:.2f} ms')
