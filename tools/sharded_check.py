"""Run under torchrun on >= 2 GPUs: three 256x256x64 frames rendered with their rays sharded over the ranks (RCCL gather,
pipelined) must equal, bit for bit, the same frames rendered by rank 0 alone.  Prints one JSON line.
With OCC_DIST_BACKEND=gloo OCC_FORCE_DEVICE=0 the ranks share one GPU and exchange through the host (RCCL refuses two
ranks on one device): everything but the collective itself is then the production code path.
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
    net = build_network(seed=0, amplify=True, S=64, non_rigid=True, device=dev)
    frames = []
    for t in range(3):
        f = synth.make_frame(img_size=256, pose72=synth.seeded_pose(1 + t), orbit_frame=10 * t)
        frames.append({k: torch.from_numpy(np.ascontiguousarray(f[k])) for k in FRAME_KEYS})
    with torch.no_grad():
        sharded = list(ShardedRenderer(net, dev, chunk=1024).render_frames(frames))
        ok, worst = True, 0.0
        if rank == 0:
            alone = list(ShardedRenderer(net, dev, single=True).render_frames(frames))
            for a, b in zip(sharded, alone):
                for k in ('rgb', 'alpha', 'depth'):
                    ok = ok and torch.equal(a[k], b[k])
                    worst = max(worst, float((a[k] - b[k]).abs().max()))
            print(json.dumps({'world_size_formed': dist.get_world_size(), 'backend': backend, 'frames': len(frames), 'bit_identical': bool(ok),
                              'max_abs_diff': worst, 'rays': [int(f['rays'].shape[1]) for f in frames]}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
