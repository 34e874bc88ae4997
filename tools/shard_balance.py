"""Balance of the multi-GPU shard plan, measured on one GPU: for N in {2, 4, 8} every rank's share of a frame (the plan
occnerf_amd/parallel.py would build) is rendered by itself and its ray count, live-sample count and render time are
printed, with max/mean over the ranks -- the ceiling the plan puts on strong-scaling efficiency before any collective.
    python tools/shard_balance.py [--size 512 --spp 128] [--block 256]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from occnerf_amd import synth  # noqa: E402
from occnerf_amd.parallel import BLOCK, ShardedRenderer  # noqa: E402
from occnerf_amd.seeded import build_network, frame_to_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--spp', type=int, default=128)
    ap.add_argument('--block', type=int, default=BLOCK)
    ap.add_argument('--no-morton', action='store_true')
    ap.add_argument('--static', action='store_true', help='block b -> rank b % N, no cost estimate')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    net = build_network(seed=0, amplify=False, S=args.spp, non_rigid=True, device=dev)
    net.cfg.dedup_repeated_samples = False
    frame = synth.make_frame(img_size=args.size, pose72=synth.seeded_pose(1), orbit_frame=28)
    data = frame_to_device(frame, dev)
    R = data['rays'].shape[1]
    report = {'rays': R, 'spp': args.spp, 'block': args.block, 'morton': not args.no_morton, 'cost_aware': not args.static, 'worlds': {}}
    with torch.no_grad():
        net(**data, iter_val=1e7)
        for W in (1, 2, 4, 8):
            rows = []
            for rank in range(W):
                r = ShardedRenderer(net, dev, block=args.block, single=True, morton=not args.no_morton, balance=not args.static)
                r.world, r.rank, r.collective, r.verify_plan = W, rank, True, False                                # plan arithmetic only
                plan = r._build_plan(data)
                sub = dict(data)
                if W > 1:
                    mine = plan['mine']['cuda']
                    sub['rays'], sub['near'], sub['far'] = data['rays'][:, mine].contiguous(), data['near'][mine], data['far'][mine]
                net(**sub, iter_val=1e7)                                 # warm (ray order of this shard)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    net(**sub, iter_val=1e7, ray_order_key=('b', W, rank))
                torch.cuda.synchronize()
                rows.append((int(sub['rays'].shape[1]), int(net.last_live_count), (time.perf_counter() - t0) / 3 * 1e3))
            rays, live, ms = (np.array(c, dtype=np.float64) for c in zip(*rows))
            report['worlds'][W] = {'rays_max_over_mean': rays.max() / rays.mean(), 'live_max_over_mean': live.max() / live.mean(),
                                   'ms_max': ms.max(), 'ms_mean': ms.mean(), 'ms_max_over_mean': ms.max() / ms.mean(),
                                   'per_rank': rows}
            if W > 1:
                report['worlds'][W]['speedup_bound_vs_1'] = report['worlds'][1]['ms_max'] / ms.max()
    print(json.dumps(report))


if __name__ == '__main__':
    main()
