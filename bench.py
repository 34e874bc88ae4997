"""Headline benchmark: rays/s of the volumetric ray-rendering hot path at 512x512 rays x 128
samples/ray (BASELINE.json configs[1]: free-view frame, non-rigid motion on, random-init
checkpoint), on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one synthetic free-view frame through Network.forward (the reference's module seam), by SURVEY.md
section 8(d)'s definition of the metric: the frame's ray batch [R,8] and motion-weight prior start in (pinned) host
memory and are copied to the device inside the step, the [R,5] (rgb, alpha, depth) block is copied back to pinned
host memory inside the step; PNG encoding is not part of it.  With N > 1 the ONE frame's rays are sharded over the
ranks (4 096-ray chunks dealt round-robin, occnerf_amd/parallel.py) and the blocks gathered on rank 0 over RCCL --
the path's only exchange step, issued asynchronously so that frame t's gather runs under frame t+1's kernels:
total work is fixed as N grows (`scaling: strong`); `value` = rays of the frame x steps / wall time (max over
ranks).  Rank 0 prints ONE JSON line.  `median_ms_per_step` comes from HIP events recorded at every step boundary.

`roofline`: the dominant kernel is the fp32-MFMA canonical MLP (occnerf_amd/csrc/mlp16.hip).
achieved = 923 136 FLOP/sample x samples per launch / average launch duration, measured
with HIP events recorded on the launch stream around every launch inside the timed region.
`cpu_baseline`: the CPU oracle (a port of the reference path, OpenMP on all host cores of
this box) on a bounded ray sample of the same frame -- only the checker being timed, never
part of the product path.
`all_samples`: the same frame with cfg.skip_empty_samples off.  By default the renderer does not
evaluate the samples whose motion-weight sum is exactly 0 (their alpha is multiplied by it, so the
pixels are bit-identical; a quarter of this frame's samples); `roofline` counts FLOPs only for the
samples a launch processes.  `dedup`: the same frame with the renderer's default cfg.dedup_repeated_samples on (runs of
bitwise identical samples evaluated once, bit-identical pixels); the headline keeps it OFF so that `value` stays the
every-live-sample rate of rounds 1-2.  `alt`: the opt-in split-bf16 MLP path.  `train`: BASELINE configs[4], one optimisation
step (forward + backward + clip + Adam, all HIP kernels) on 6 144 rays x 128 samples in bf16.  `weak_frames`
(N > 1 only): the round-1 mode, one whole frame per rank per step.
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
TRAIN_RAYS = 6144                     # default.yaml patch config: 6 patches x 32 x 32


def cpu_baseline(ctx, frame, n_rays):
    """rays/s of the CPU oracle on `n_rays` rays spread over the frame (rank 0, N=1 only)."""
    from oracle import oracle as orc
    from oracle.chain import stagewise_oracle_render
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
    for name in ('r02_pmc_hbm.json', 'r01_pmc_hbm.json'):
        try:
            d = json.load(open(os.path.join(ROOT, 'profiles', name)))
            if abs(int(d['samples_per_launch']) - int(n_samples)) <= 1:
                k = d['kernels']
                return float(k['occ::m16::canonical_mlp_lds_kernel']['hbm_bytes_corrected']), name
        except Exception:
            pass
    return None, None


def timed_steps(renderer, frame_h, steps, warmup, rank, world, dev, order_key, host_out):
    """`steps` pipelined frames after `warmup`; -> (wall seconds, per-step ms list from HIP events)."""
    def run(n, events=None):
        prev = None
        stream = torch.cuda.current_stream(dev)
        for _ in range(n):
            if events is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream)
                events.append(e)
            with torch.no_grad():
                cur = renderer.submit(frame_h, ray_order_key=order_key)
            if prev is not None:
                out = renderer.finish(prev)
                if out is not None:
                    host_out.copy_(out['packed'], non_blocking=True)      # one contiguous [R,5] D2H into pinned memory
            prev = cur
        out = renderer.finish(prev)
        if out is not None:
            host_out.copy_(out['packed'], non_blocking=True)
        if events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            events.append(e)
    if warmup:
        run(warmup)
    events = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, events)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    return dt, [events[i].elapsed_time(events[i + 1]) for i in range(len(events) - 1)]


def train_leg(dev, steps, warmup):
    """BASELINE configs[4]: one optimisation step at the reference's patch configuration (6 x 32 x 32 rays, 128
    samples/ray, jitter on): forward + backward through the HIP sampler / kNN / encoder / MLP / compositor kernels
    with bf16 MLP trunks, gradient clipping and Adam on the device."""
    from occnerf_amd import synth
    from occnerf_amd.optim import FusedAdam
    from occnerf_amd.seeded import build_network, frame_to_device
    net = build_network(seed=0, amplify=False, S=SPP, non_rigid=True, device=dev)
    net.cfg.perturb = 1.0
    net.cfg.train_precision = 'bf16'
    net.train()
    frame = synth.make_frame(img_size=IMG, pose72=synth.seeded_pose(1), orbit_frame=28)
    R = frame['rays'].shape[1]
    sel = np.sort(np.random.RandomState(0).choice(R, TRAIN_RAYS, replace=False))
    for k in ('near', 'far'):
        frame[k] = frame[k][sel]
    frame['rays'] = frame['rays'][:, sel]
    data = frame_to_device(frame, dev)
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
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    net.cfg.train_precision = 'auto'
    return {'ms_per_step': dt * 1e3, 'rays_per_step': TRAIN_RAYS, 'samples_per_step': TRAIN_RAYS * SPP,
            'rays_per_s': TRAIN_RAYS / dt, 'dtype': 'bf16 MLP trunks (fp32 accumulate, fp32 master weights); '
            'fp32 sampler, encoder, aggregation, compositor', 'final_loss': float(loss),
            'what': 'forward + backward + clip_grad_norm + Adam, every per-sample stage a HIP kernel '
                    '(occnerf_amd/train_path.py); synthetic target'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--cpu-rays', type=int, default=4096, help='rays in the CPU baseline sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the side measurements (bf16x3, all samples, train)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('OCC_FORCE_DEVICE', os.environ.get('LOCAL_RANK', 0)))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    backend = os.environ.get('OCC_DIST_BACKEND', 'nccl')     # 'gloo' + OCC_FORCE_DEVICE=0: dry run of the N > 1 path on one GPU
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from occnerf_amd import ops, synth
    from occnerf_amd.parallel import ShardedRenderer
    from occnerf_amd.seeded import build_network, host_frame

    net = build_network(seed=0, amplify=False, S=SPP, non_rigid=True, device=dev)
    frame = synth.make_frame(img_size=IMG, pose72=synth.seeded_pose(1), orbit_frame=28)
    frame_h = host_frame(frame)
    R = frame['rays'].shape[1]
    host_out = torch.empty(R, 5).pin_memory()
    renderer = ShardedRenderer(net, dev)

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
    # The headline evaluates every live sample (the definition of rounds 1-2); the renderer's default also evaluates runs of
    # bitwise identical samples once (cfg.dedup_repeated_samples, bit-identical pixels): reported beside it as `dedup`.
    net.cfg.dedup_repeated_samples = False
    ops.canonical_mlp = timed_mlp
    timed_steps(renderer, frame_h, 1, args.warmup, rank, world, dev, ('bench', rank), host_out)   # warm-up (+1 step)
    mlp_events.clear()
    dt, step_ms = timed_steps(renderer, frame_h, args.steps, 0, rank, world, dev, ('bench', rank), host_out)
    main_events = list(mlp_events)
    ops.canonical_mlp = real_mlp
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])

    side = {}
    if not args.no_alt:
        net.cfg.dedup_repeated_samples = True
        if world > 1:
            dist.barrier()
        dtd, _ = timed_steps(renderer, frame_h, args.steps, args.warmup, rank, world, dev, ('bench', rank), host_out)
        if world > 1:
            td = torch.tensor([dtd], device=dev, dtype=torch.float64)
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
            dtd = float(td[0])
        heads = getattr(net, 'last_head_counts', (None, None))
        side['dedup'] = {
            'dedup_repeated_samples': True, 'value': R * args.steps / dtd, 'unit': 'rays/s',
            'ms_per_step': dtd / args.steps * 1e3,
            'distinct_positions_last_pass': None if heads[0] is None else int(heads[0]),
            'distinct_feature_rows_last_pass': None if heads[1] is None else int(heads[1]),
            'note': 'the renderer\'s default: consecutive live samples with bitwise identical canonical positions / feature '
                    'rows are evaluated once (run-length, exact: rgb/alpha/depth bit-identical, tested on this frame); how '
                    'much it removes is a property of the frame -- here most live samples sit where the motion-weight sum '
                    'is far below the reference\'s 1e-4 clamp and collapse onto the origin'}
        net.cfg.dedup_repeated_samples = False
        if world == 1:
            # opt-in split-bf16 MLP path (cfg.mlp_precision='bf16x3'): same frame, same steps; never part of `value`
            net.cfg.mlp_precision = 'bf16x3'
            net.invalidate_cache()
            dta, _ = timed_steps(renderer, frame_h, args.steps, args.warmup, rank, world, dev, ('bench', rank), host_out)
            side['alt'] = {'mlp_precision': 'bf16x3 (hi/lo bf16 operands, 3 MFMA products, fp32 accumulate; parity-tested '
                                            'to the same 1e-4 pixel gate)', 'value': R * args.steps / dta, 'unit': 'rays/s',
                           'ms_per_step': dta / args.steps * 1e3}
            net.cfg.mlp_precision = 'fp32'
            net.invalidate_cache()
            # every sample evaluated (cfg.skip_empty_samples off): same pixels bit for bit
            net.cfg.skip_empty_samples = False
            dtf, _ = timed_steps(renderer, frame_h, args.steps, args.warmup, rank, world, dev, ('bench', rank), host_out)
            side['all_samples'] = {
                'skip_empty_samples': False, 'value': R * args.steps / dtf, 'unit': 'rays/s',
                'ms_per_step': dtf / args.steps * 1e3,
                'note': 'all R x 128 samples through every stage; the headline drops the samples whose motion-weight '
                        'sum is exactly 0 (alpha is multiplied by it), with bit-identical rgb/alpha/depth'}
            net.cfg.skip_empty_samples = True
            side['train'] = train_leg(dev, max(5, args.steps // 2), 3)
        else:
            # round-1 mode: one whole frame per rank per step, same-size gather (weak scaling)
            whole = ShardedRenderer(net, dev, single=True)         # every rank renders the full frame by itself
            dist.barrier()
            dtw, _ = timed_steps(whole, frame_h, max(3, args.steps // 4), 1, rank, 1, dev, ('bench-whole', rank), host_out)
            tw = torch.tensor([dtw], device=dev, dtype=torch.float64)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            n = max(3, args.steps // 4)
            side['weak_frames'] = {'value': world * R * n / float(tw[0]), 'unit': 'rays/s', 'scaling': 'weak',
                                   'ms_per_step': float(tw[0]) / n * 1e3,
                                   'note': 'one whole frame per rank per step, no gather (every rank keeps its frame)'}

    if rank == 0:
        ms = [e0.elapsed_time(e1) for e0, e1, _ in main_events]
        nsmp = [int(n) for _, _, n in main_events]          # (read back after the timed region)
        avg_ms = float(np.mean(ms))
        achieved = FLOP_PER_SAMPLE_CNL * float(np.mean(nsmp)) / (avg_ms * 1e-3)
        traffic, traffic_src = pmc_traffic(float(np.mean(nsmp)))
        line = {
            'metric': 'rays/sec at 512x512x128spp, random-init ckpt', 'value': R * args.steps / dt,
            'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'median_ms_per_step': float(np.median(step_ms)),
            'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: free-view frame, 512x512 image, 128 samples/ray, '
                                   'non-rigid motion on, seeded random-init checkpoint; synthetic SMPL-like body '
                                   f'and camera; {R} rays hit the body bbox (ray_mask); H2D of the ray batch and the prior and '
                                   'D2H of [R,5] inside the step; the frame\'s rays sharded over the ranks; samples whose '
                                   'motion-weight sum is exactly 0 are dropped after the warp (bit-identical pixels, see all_samples); every other '
                                   'sample is evaluated (the renderer\'s run-length elimination of identical samples is OFF here, see dedup)',
                       'rays_per_frame': R, 'samples_per_ray': SPP, 'image': [IMG, IMG],
                       'samples_evaluated_per_launch': float(np.mean(nsmp)),
                       'skip_empty_samples': bool(net.cfg.get('skip_empty_samples', True)),
                       'dedup_repeated_samples': False,
                       'world_size_formed': renderer.formed_world_size(), 'backend': backend if world > 1 else None,
                       'parallelism': f'one frame, rays sharded x{world} in 4096-ray chunks, async RCCL gather to rank 0 '
                                      'overlapped with the next frame'},
            'roofline': {'bound': 'mfma', 'kernel': 'occ::m16::canonical_mlp_lds_kernel (fp32 MFMA 16x16x4, LDS-staged weights)',
                         'achieved': achieved / 1e12, 'peak': PEAK_FP32_MFMA / 1e12, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_FP32_MFMA, 'traffic': traffic,
                         'traffic_note': f'HBM bytes/launch, FETCH_SIZE x2 + WRITE_SIZE from profiles/{traffic_src}; '
                                         'algorithmic 288 B/sample',
                         'launch_ms': avg_ms, 'launches_timed': len(ms),
                         'flop_per_launch': FLOP_PER_SAMPLE_CNL * float(np.mean(nsmp))},
        }
        line.update(side)
        if world == 1 and not args.no_cpu_baseline:
            from oracle.chain import model_context                  # the checker, timed as the stated CPU baseline
            line['cpu_baseline'] = cpu_baseline(model_context(0, False), frame, args.cpu_rays)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
