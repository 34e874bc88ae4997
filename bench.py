"""Headline benchmark: rays/s of the volumetric ray-rendering hot path at 512x512 rays x 128
samples/ray (BASELINE.json configs[1]: free-view frame, non-rigid motion on, random-init
checkpoint), on N MI355X of one node.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" renders one synthetic free-view frame per rank through Network.forward (the
reference's module seam) with the frame's inputs already resident in HBM, then gathers the
[R,5] (rgb, alpha, depth) block on rank 0 over RCCL (the path's only exchange step).  Rays
shard by frame across ranks: per-GPU work is fixed as N grows (weak scaling); `value` is
rays of all ranks / wall time (max over ranks).  Rank 0 prints ONE JSON line.

`roofline`: the dominant kernel is the fp32-MFMA canonical MLP (occnerf_amd/csrc/mlp16.hip).
achieved = 923 136 FLOP/sample x samples per launch / average launch duration, measured
with HIP events recorded on the launch stream around every launch inside the timed region.
`cpu_baseline`: the CPU oracle (a port of the reference path, OpenMP on all host cores of
this box) on a bounded ray sample of the same frame -- only the checker being timed, never
part of the product path.
`all_samples`: the same frame with cfg.skip_empty_samples off.  By default the renderer does not
evaluate the samples whose motion-weight sum is exactly 0 (their alpha is multiplied by it, so the
pixels are bit-identical; a quarter of this frame's samples); `roofline` counts FLOPs only for the
samples a launch processes.  `alt`: the opt-in split-bf16 MLP path.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

IMG, SPP = 512, 128
FLOP_PER_SAMPLE_CNL = 923136          # canonical MLP, 461 568 MAC
FLOP_PER_SAMPLE_NR = 200704           # non-rigid MLP (free-view / movement)
PEAK_FP32_MFMA = 157.3e12             # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32


def cpu_baseline(ctx, frame, n_rays):
    """rays/s of the CPU oracle on `n_rays` rays spread over the frame (rank 0, N=1 only)."""
    from oracle import oracle as orc
    from tests.gpu_util import stagewise_oracle_render
    orc.build()
    R = frame['rays'].shape[1]
    sel = np.linspace(0, R - 1, n_rays).astype(np.int64)
    sub = dict(frame)
    sub['rays'], sub['near'], sub['far'] = frame['rays'][:, sel], frame['near'][sel], frame['far'][sel]
    stagewise_oracle_render(None, ctx, frame={**sub, 'rays': sub['rays'][:, :8], 'near': sub['near'][:8],
                                              'far': sub['far'][:8]}, S=SPP, non_rigid=True)   # warm-up
    t0 = time.perf_counter()
    stagewise_oracle_render(None, ctx, frame=sub, S=SPP, non_rigid=True)
    dt = time.perf_counter() - t0
    return {'value': n_rays / dt, 'unit': 'rays/s', 'cores': os.cpu_count(), 'kind': 'port',
            'sample': f'{n_rays} rays x {SPP} samples of the same 512x512 free-view frame, '
                      f'oracle/occnerf_oracle.c (OpenMP), {dt:.1f} s'}


def pmc_traffic(n_samples):
    """HBM bytes per launch of the roofline kernel from the committed PMC passes (rocprofv3 cannot run
    inside this process); only quoted when it was collected at the same launch size."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_hbm.json')
    try:
        d = json.load(open(path))
        if int(d['samples_per_launch']) == int(n_samples):
            k = d['kernels']
            return float((k.get('occ::m16::canonical_mlp_lds_kernel') or k['occ::canonical_mlp_kernel'])['hbm_bytes_corrected'])
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--cpu-rays', type=int, default=4096, help='rays in the CPU baseline sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the opt-in bf16x3 measurement')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from occnerf_amd import ops, synth
    from occnerf_amd.parallel import gather_rays
    from tests.gpu_util import build_network, frame_to_device

    net, ctx = build_network(seed=0, amplify=False, S=SPP, non_rigid=True, device=dev)
    # every rank renders one full frame per step (the same free-view frame: identical work per GPU)
    frame = synth.make_frame(img_size=IMG, pose72=synth.seeded_pose(1), orbit_frame=28)
    data = frame_to_device(frame, dev)
    R = frame['rays'].shape[1]

    # HIP events around every launch of the dominant kernel, on the stream it is launched on
    mlp_events, real_mlp = [], ops.canonical_mlp

    def timed_mlp(mlp_in, packed, raw, count=None, **kw):
        s = torch.cuda.current_stream(mlp_in.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        out = real_mlp(mlp_in, packed, raw, count=count, **kw)
        e1.record(s)
        # rows the launch really processes: the device-side live count when the renderer passes one
        mlp_events.append((e0, e1, mlp_in.shape[0] if count is None else count.clone()))
        return out
    ops.canonical_mlp = timed_mlp

    def step():
        with torch.no_grad():
            out = net(**data, iter_val=1e7)
        packed = torch.cat([out['rgb'], out['alpha'][:, None], out['depth'][:, None]], dim=1)
        if world > 1:                                  # the path's one exchange step: [R,5] to rank 0
            bufs = [torch.empty_like(packed) for _ in range(world)] if rank == 0 else None
            dist.gather(packed, bufs, dst=0)
        return packed

    for _ in range(args.warmup):
        step()
    mlp_events.clear()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(R)], device=dev, dtype=torch.float64)
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, rays_all = float(tmax[0]), float(tsum[1])
    else:
        rays_all = float(R)

    # opt-in split-bf16 MLP path (cfg.mlp_precision='bf16x3'): measured beside the headline, same frame,
    # same steps; never part of `value`
    alt = None
    if world == 1 and not args.no_alt:
        ops.canonical_mlp = real_mlp
        net.cfg.mlp_precision = 'bf16x3'
        net.invalidate_cache()
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dta = time.perf_counter() - ta
        alt = {'mlp_precision': 'bf16x3 (hi/lo bf16 operands, 3 MFMA products, fp32 accumulate; parity-tested '
                                'to the same 1e-4 pixel gate)', 'value': R * args.steps / dta, 'unit': 'rays/s',
               'ms_per_step': dta / args.steps * 1e3}
        net.cfg.mlp_precision = 'fp32'
        net.invalidate_cache()

    # every sample evaluated (cfg.skip_empty_samples off): same pixels bit for bit, reported beside the headline
    full = None
    if world == 1 and not args.no_alt:
        ops.canonical_mlp = real_mlp
        net.cfg.skip_empty_samples = False
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        tf = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dtf = time.perf_counter() - tf
        full = {'skip_empty_samples': False, 'value': R * args.steps / dtf, 'unit': 'rays/s',
                'ms_per_step': dtf / args.steps * 1e3,
                'note': 'all R x 128 samples through every stage; the headline drops the samples whose motion-weight '
                        'sum is exactly 0 (alpha is multiplied by it), with bit-identical rgb/alpha/depth'}
        net.cfg.skip_empty_samples = True

    if rank == 0:
        ms = [e0.elapsed_time(e1) for e0, e1, _ in mlp_events]
        nsmp = [int(n) for _, _, n in mlp_events]          # (read back after the timed region)
        avg_ms = float(np.mean(ms))
        achieved = FLOP_PER_SAMPLE_CNL * float(np.mean(nsmp)) / (avg_ms * 1e-3)
        line = {
            'metric': 'rays/sec at 512x512x128spp, random-init ckpt', 'value': rays_all * args.steps / dt,
            'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: free-view frame, 512x512 image, 128 samples/ray, '
                                   'non-rigid motion on, seeded random-init checkpoint; synthetic SMPL-like body '
                                   f'and camera; {R} rays hit the body bbox (ray_mask), one frame per GPU per step; samples whose '
                                   'motion-weight sum is exactly 0 are dropped after the warp (bit-identical pixels, see all_samples)',
                       'rays_per_frame': R, 'samples_per_ray': SPP, 'image': [IMG, IMG],
                       'samples_evaluated_per_frame': float(np.mean(nsmp)),
                       'skip_empty_samples': bool(net.cfg.get('skip_empty_samples', True)),
                       'parallelism': f'frames x{world} (rays sharded by frame), RCCL gather to rank 0'},
            'roofline': {'bound': 'mfma', 'kernel': 'occ::m16::canonical_mlp_lds_kernel (fp32 MFMA 16x16x4, LDS-staged weights)',
                         'achieved': achieved / 1e12, 'peak': PEAK_FP32_MFMA / 1e12, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_FP32_MFMA, 'traffic': pmc_traffic(float(np.mean(nsmp))),
                         'traffic_note': 'HBM bytes/launch, FETCH_SIZE x2 + WRITE_SIZE from profiles/r01_pmc_hbm.json; '
                                         'algorithmic 288 B/sample',
                         'launch_ms': avg_ms, 'launches_timed': len(ms),
                         'flop_per_launch': FLOP_PER_SAMPLE_CNL * float(np.mean(nsmp))},
        }
        if alt is not None:
            line['alt'] = alt
        if full is not None:
            line['all_samples'] = full
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(ctx, frame, args.cpu_rays)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
