"""The benchmark frame (512x512 x 128, non-rigid on) rendered serially and with cfg.overlap_chunks = 2, 4, 8, ...: frame time
and bit-identity of rgb/alpha/depth against the serial render.
    python3 tools/overlap_frame.py [--dedup] [--chunks 0,2,4,8] [--frames 8]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
from occnerf_amd import synth  # noqa: E402
from occnerf_amd.seeded import build_network, frame_to_device  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--dedup', action='store_true')
ap.add_argument('--chunks', default='0,2,4,8')
ap.add_argument('--frames', type=int, default=8)
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--spp', type=int, default=128)
args = ap.parse_args()
net = build_network(seed=0, amplify=False, S=args.spp, non_rigid=True)
net.cfg.dedup_repeated_samples = args.dedup
frame = synth.make_frame(img_size=args.size, pose72=synth.seeded_pose(1), orbit_frame=28)
data = frame_to_device(frame, 'cuda:0')
for k in ('cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor'):
    data[k] = data[k].cpu()
ref = None
for n in [int(c) for c in args.chunks.split(',')]:
    net.cfg.overlap_chunks = n
    times = []
    for it in range(args.frames + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            out = net(**data, iter_val=1e7, ray_order_key='ov')
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    got = torch.cat([out['rgb'], out['alpha'][:, None], out['depth'][:, None]], 1)
    if ref is None:
        ref = got
    t = sorted(times[2:])
    print(f'overlap_chunks={n}: median {t[len(t) // 2]:.2f} ms  min {t[0]:.2f} ms  bit-identical to first: {torch.equal(got, ref)}'
          f'  FEATURES_SMALL={os.environ.get("OCCNERF_FEATURES_SMALL", "0")} dedup={args.dedup}', flush=True)
