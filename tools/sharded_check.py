"""Run under torchrun on >= 2 GPUs: frames rendered with their rays sharded over the ranks (RCCL gather, pipelined) must
equal, bit for bit, the same frames rendered by rank 0 alone.  Prints one JSON line.  Default: three 256x256x64 free-view
frames of different poses (host frames); `--kind movement --size 512 --spp 128 --frames 3` = BASELINE configs[2]: frames of
the movement pose walk from one camera, rays generated on the device, camera named (cached shard plans).
With OCC_DIST_BACKEND=gloo OCC_FORCE_DEVICE=0 the ranks share one GPU and exchange through the host (RCCL refuses two
ranks on one device): everything but the collective itself is then the production code path.
`--force-collective` with ONE rank (`--nproc-per-node 1`, nccl): the one-rank RCCL group takes the N > 1 branch -- plan
checksum all-gather, dist.gather on device buffers, work.wait(), un-permutation -- which a single-GPU box can execute.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/sharded_check.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--kind', default='freeview', choices=['freeview', 'movement'])
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--spp', type=int, default=64)
    ap.add_argument('--frames', type=int, default=3)
    ap.add_argument('--force-collective', action='store_true')
    args = ap.parse_args()
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    local = int(os.environ.get('OCC_FORCE_DEVICE', local))        # experiment: several ranks on one GPU
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    backend = os.environ.get('OCC_DIST_BACKEND', 'nccl')             # 'gloo': several processes on one GPU, host exchange
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from occnerf_amd import synth
    from occnerf_amd.parallel import ShardedRenderer
    from occnerf_amd.seeded import build_network, FRAME_KEYS
    net = build_network(seed=0, amplify=True, S=args.spp, non_rigid=True, device=dev)
    frames = []
    for t in range(args.frames):
        if args.kind == 'freeview':
            f = synth.make_frame(img_size=args.size, pose72=synth.seeded_pose(1 + t), orbit_frame=10 * t)
            frames.append({k: torch.from_numpy(np.ascontiguousarray(f[k])) for k in FRAME_KEYS})
        else:
            from occnerf_amd.rays import frame_rays
            f = synth.make_frame(img_size=args.size, pose72=synth.movement_pose(t, args.frames), with_rays=False)
            fr = frame_rays(f['camera_K'], f['camera_E'], args.size, args.size, f['dst_bbox_min'], f['dst_bbox_max'], dev)
            d = {k: torch.from_numpy(np.ascontiguousarray(f[k])).to(dev) for k in FRAME_KEYS if k in f}
            d.update(rays=fr['rays'], near=fr['near'], far=fr['far'])
            frames.append((d, ('movement', int(fr['rays'].shape[1]))))
    with torch.no_grad():
        sr = ShardedRenderer(net, dev, force_collective=args.force_collective or None)
        sharded = list(sr.render_frames(frames))
        ok, worst = True, 0.0
        if rank == 0:
            alone = list(ShardedRenderer(net, dev, single=True).render_frames(frames))
            for a, b in zip(sharded, alone):
                for k in ('rgb', 'alpha', 'depth'):
                    ok = ok and torch.equal(a[k], b[k])
                    worst = max(worst, float((a[k] - b[k]).abs().max()))
            print(json.dumps({'world_size_formed': dist.get_world_size(), 'backend': backend, 'collective': sr.collective,
                              'gathers_issued': sr.gathers_issued, 'plans_verified': sr.plans_verified, 'frames': len(frames), 'bit_identical': bool(ok),
                              'max_abs_diff': worst, 'kind': args.kind, 'size': args.size, 'spp': args.spp,
                              'rays': [int((f[0] if isinstance(f, tuple) else f)['rays'].shape[1]) for f in frames]}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
