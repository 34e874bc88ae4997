"""A/B of cfg.center_side_stream through the emulated-rank path bench.py's predicted_scaling leg uses (ShardedRenderer(emulate=
(N, k)) on a one-rank RCCL group, frames pipelined): median per-frame HIP-event time of rank k of N, alternating the setting.
    python3 tools/center_stream_ab.py [--world 8] [--ranks 0,5] [--rounds 3]"""
import argparse
import os
import socket
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
from occnerf_amd import synth  # noqa: E402
from occnerf_amd.parallel import ShardedRenderer  # noqa: E402
from occnerf_amd.seeded import build_network, host_frame  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--world', type=int, default=8)
ap.add_argument('--ranks', default='0,5')
ap.add_argument('--rounds', type=int, default=3)
ap.add_argument('--frames', type=int, default=12)
args = ap.parse_args()
s = socket.socket()
s.bind(('127.0.0.1', 0))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', str(s.getsockname()[1]))
s.close()
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
net = build_network(seed=0, amplify=False, S=bench.SPP, non_rigid=True, device=dev)
net.cfg.dedup_repeated_samples = False
frame_h = host_frame(synth.make_frame(img_size=bench.IMG, pose72=synth.seeded_pose(1), orbit_frame=28))
host_out = torch.empty(frame_h['rays'].shape[1], 5).pin_memory()
for k in [int(x) for x in args.ranks.split(',')]:
    for rnd in range(args.rounds):
        for side in (True, False):
            net.cfg.center_side_stream = side
            r = ShardedRenderer(net, dev, emulate=(args.world, k))
            _, ms = bench.timed_steps(r, frame_h, args.frames, 3, 0, 1, dev, ('ab', args.world), host_out)
            print(f'rank {k} of {args.world} side_stream={side}: median {np.median(ms):.3f} ms (min {min(ms):.3f}, max {max(ms):.3f})',
                  flush=True)
            del r
dist.destroy_process_group()
