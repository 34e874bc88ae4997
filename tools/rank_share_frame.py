"""One rank's share of the benchmark frame (the plan occnerf_amd/parallel.py builds for --world ranks) rendered repeatedly on
this GPU -- for rocprofv3 --kernel-trace: what a rank's frame looks like at N = 8 (kernel tails, launch gaps, fixed work).
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o s -- python3 tools/rank_share_frame.py --world 8 --rank 0"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
from occnerf_amd import synth  # noqa: E402
from occnerf_amd.parallel import ShardedRenderer  # noqa: E402
from occnerf_amd.seeded import build_network, frame_to_device  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--world', type=int, default=8)
ap.add_argument('--rank', type=int, default=0)
ap.add_argument('--frames', type=int, default=40)
ap.add_argument('--dedup', action='store_true')
args = ap.parse_args()
dev = torch.device('cuda:0')
net = build_network(seed=0, amplify=False, S=128, non_rigid=True, device=dev)
net.cfg.dedup_repeated_samples = args.dedup
data = frame_to_device(synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28), dev)
for k in ('cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor'):
    data[k] = data[k].cpu()
with torch.no_grad():
    r = ShardedRenderer(net, dev, single=True)
    r.world, r.rank, r.collective, r.verify_plan = args.world, args.rank, True, False
    mine = r._build_plan(data)['mine']['cuda']
    sub = dict(data)
    sub['rays'], sub['near'], sub['far'] = data['rays'][:, mine].contiguous(), data['near'][mine], data['far'][mine]
    for _ in range(2):
        net(**sub, iter_val=1e7, ray_order_key='s')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.frames):
        net(**sub, iter_val=1e7, ray_order_key='s')
    torch.cuda.synchronize()
print(f'rank {args.rank} of {args.world}: {int(sub["rays"].shape[1])} rays, {(time.perf_counter() - t0) / args.frames * 1e3:.2f} ms per frame')
