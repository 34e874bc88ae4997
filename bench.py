"""Headline benchmark: rays/s of the volumetric ray-rendering hot path at 512x512 rays x 128
samples/ray (BASELINE.json configs[1]: free-view frame, non-rigid motion on, random-init
checkpoint), on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      (no launcher: bench.py starts the N ranks itself -- a child
                                       `python -m torch.distributed.run ... bench.py` created before this process
                                       has touched the GPU -- relays the child's JSON line and exits with its code)

A "step" is one synthetic free-view frame through Network.forward (the reference's module seam), by SURVEY.md
section 8(d)'s definition of the metric: the frame's ray batch [R,8] and motion-weight prior start in (pinned) host
memory and are copied to the device inside the step, the [R,5] (rgb, alpha, depth) block is copied back to pinned
host memory inside the step; PNG encoding is not part of it.  With N > 1 the ONE frame's rays are sharded over the
ranks (its Morton walk cut into 256-ray blocks, dealt by estimated cost, occnerf_amd/parallel.py) and the blocks
gathered on rank 0 over RCCL -- the path's only exchange step, issued asynchronously so that frame t's gather runs
under frame t+1's kernels:
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
(N > 1 only): the round-1 mode, one whole frame per rank per step.  `config4` (N = 1): BASELINE configs[3], one 1024x1024 x
192-sample frame with seeded visibility counts, renderer default and every-live-sample.  `movement` (N = 1): BASELINE
configs[2]'s sequence on one GPU through the loop run.py executes (occnerf_amd/sequence.py: device ray generation, named
camera, one frame of lag, device image assembly, uint8 D2H), rays/s over a whole pass.  With N > 1 `config.per_rank_rays` /
`per_rank_live_samples` show the balance of the shard plan.  `no_shortcuts` (N = 1): the frame with round 4's two exact
data-dependent shortcuts off (cfg.knn_center_cache, cfg.warp_bone_culling) -- identical pixels, the way round 3 rendered it.
`rccl_world1` (N = 1): the same frame through the N > 1
branch of the sharded renderer with a ONE-rank `nccl` process group (plan + checksum all-gather, padded send buffer,
asynchronous dist.gather on device buffers, work.wait(), un-permutation): what a single-GPU box can execute of the
multi-GPU path; pixels bit-identical to the headline's, never part of `value`.  It runs in a CHILD process under a time limit
(this script again with a few headline frames of its own), so that a communicator that hangs cannot take the headline line down.
Round 5: `alt2` -- the fp32-GRADE split-fp16 MLP path (cfg.mlp_precision='f16x3'; its kernel's launches timed with HIP events);
`predicted_scaling` -- PREDICTED, SINGLE GPU, not a scaling curve: inside the rccl_world1 child every rank k of the N-rank plans
(N = 2, 4, 8) of the headline frame is emulated through the collective renderer (ShardedRenderer(emulate=(N, k)), RCCL self-gather
into slot k) and {N, slowest_rank_ms, T1_over_slowest, per_rank_fixed_ms} reported; `freeview_orbit` -- 8 consecutive frames of
the orbit, a NEW camera every frame (nothing named or cached: device ray generation, Morton order, render, image assembly, uint8
D2H), with the per-frame cost of the order shown separately; `train.roofline` -- the training step against the compulsory HBM
traffic of its staged design (per-row byte table in this file) and the bf16 MFMA floor.
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
PEAK_HBM_ACHIEVABLE = 6.3e12      # B/s a streaming kernel reaches on MI355X (guide); datasheet 8 TB/s
PEAK_BF16_MFMA = 2.5e15           # dense bf16 FLOP/s (guide)
PEAK_FP32_MFMA = 157.3e12             # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
TRAIN_RAYS = 6144                     # default.yaml patch config: 6 patches x 32 x 32


def cpu_baseline(ctx, frame, n_rays):
    """rays/s of the CPU oracle on `n_rays` rays spread over the frame (rank 0, N=1 only)."""
    from oracle import oracle as orc
    from oracle.chain import stagewise_oracle_render
    orc.build()
    cores = orc.set_threads(orc.effective_cpus())      # the CPUs this process may use (affinity mask and cgroup quota)
    R = frame['rays'].shape[1]
    sel = np.linspace(0, R - 1, n_rays).astype(np.int64)
    sub = dict(frame)
    sub['rays'], sub['near'], sub['far'] = frame['rays'][:, sel], frame['near'][sel], frame['far'][sel]
    stagewise_oracle_render(None, ctx, frame={**sub, 'rays': sub['rays'][:, :8], 'near': sub['near'][:8],
                                              'far': sub['far'][:8]}, S=SPP, non_rigid=True)   # warm-up
    t0 = time.perf_counter()
    stagewise_oracle_render(None, ctx, frame=sub, S=SPP, non_rigid=True)
    dt = time.perf_counter() - t0
    return {'value': n_rays / dt, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n_rays} rays x {SPP} samples of the same 512x512 free-view frame, '
                      f'oracle/occnerf_oracle.c (OpenMP), {dt:.1f} s; a literal serial-fmaf port of the reference\'s '
                      'arithmetic (the checker: summation order kept so that indices match bit for bit), NOT a tuned CPU '
                      'implementation -- the GPU/CPU ratio says nothing about kernel quality'}


def pmc_traffic(n_samples):
    """HBM bytes per launch of the roofline kernel from the committed PMC passes (rocprofv3 cannot run
    inside this process); only quoted when it was collected at the same launch size."""
    import glob
    for name in sorted((os.path.basename(f) for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm.json'))), reverse=True):
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


# Compulsory HBM traffic of one training step of the STAGED design (one kernel per stage, intermediates in HBM), bytes per
# sample row (M = rays x samples rows; bf16 activations = 2 B, everything else fp32).  Reads + writes of every stage's
# operands once; gathers from cache-resident tables (hash table, per-point table, motion volume) not counted.
TRAIN_BYTES_PER_ROW = {
    'sampler + warp forward (xyz 12, mask 4, z 4)': 20,
    'non-rigid MLP forward (xyz in place)': 24,
    'multi-scale kNN (xyz 12 -> 160 B of indices)': 172,
    'geometry / encoder inputs (xyz, idx -> mlp_in 272, raw 20, enc_in 16)': 480,
    'aggregation weights (idx 160 -> atts 160, var 4)': 324,
    'hash encoding forward (enc_in 16 -> 128)': 144,
    'aggregation forward (idx 160, atts 160 -> 140)': 460,
    # round 6: one kernel (csrc/trunks.hip) -- the 68 fp32 inputs read, X0 192 + 8 x 512 + GEO 192 + raw4 16 written, no
    # activation read back (rounds 2-5: X0 assembly 464 + ten layer passes 8 976 = 9 440 B per row)
    'trunks forward, fused (agg 140, var 4, enc 128 -> X0, A1..A4, GEO, B1..B4, raw4)': 4768,
    'compositing forward + backward (raw 20, mask 4, z 4; d_raw 20, d_mask 4)': 80,
    'linear layers backward: 10 weight-gradient passes (dZ + X) and 10 input-gradient passes (dZ + ReLU mask -> dX)': 24768,
    'aggregation backward (grad rows 140, idx 160, atts 160, run sums 140)': 600,
    'hash encoding backward (grad 128, enc_in 16, tile sets 16 x 8)': 272,
    'warp backward (d_mask 4, z 4, rays)': 40,
}
TRAIN_PARAM_BYTES = 60.3e6 * 28 + 3 * 62e6 + 134e6      # Adam: p, m, v read + written, g read; dense embedding-gradient buffers; decoder dW
TRAIN_FLOP_PER_ROW = 3 * FLOP_PER_SAMPLE_CNL + FLOP_PER_SAMPLE_NR      # trunks forward + dgrad + wgrad (bf16 MFMA), non-rigid forward (fp32)


def train_pmc_traffic():
    """HBM bytes of one training step from the committed PMC passes (tools/pmc_train.sh; rocprofv3 cannot run inside this
    process): -> (bytes per step summed over every kernel, file name) or (None, None)."""
    import glob
    for name in sorted((os.path.basename(f) for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_train_hbm_pmc.json'))), reverse=True):
        try:
            return float(json.load(open(os.path.join(ROOT, 'profiles', name)))['hbm_bytes_per_step_total']), name
        except Exception:
            pass
    return None, None


def train_leg(dev, steps, warmup):
    """BASELINE configs[4]: one optimisation step on the reference's patch batch (6 patches of 32 x 32 pixels, 128
    samples/ray, jitter on): forward + backward through the HIP sampler / kNN / encoder / MLP / compositor kernels
    with bf16 MLP trunks, gradient clipping and Adam on the device."""
    from occnerf_amd import synth
    from occnerf_amd.optim import FusedAdam
    from occnerf_amd.seeded import build_network, frame_to_device
    full = synth.make_frame(img_size=IMG, pose72=synth.seeded_pose(1), orbit_frame=28)
    R = full['rays'].shape[1]
    from occnerf_amd.seeded import patch_ray_selection
    # the reference's batch (default.yaml `patch`: N_patches 6, size 32 -- core/data/human_nerf/train.py): 6 random 32 x 32
    # pixel patches, here required to lie wholly on bbox-hitting pixels so that the batch is 6 144 rays; and, for continuity
    # with rounds 2-4, the same number of rays scattered over the frame (`scattered_ms_per_step`).  Each batch trains a fresh
    # copy of the seeded checkpoint (the steps move the field, and with it how many samples collapse).
    batches = {'patches': patch_ray_selection(full, np.random.RandomState(0), 6, 32, full=True),
               'scattered': np.sort(np.random.RandomState(0).choice(R, TRAIN_RAYS, replace=False))}
    timing = {}
    for name in ('scattered', 'patches'):
        net = build_network(seed=0, amplify=False, S=SPP, non_rigid=True, device=dev)
        net.cfg.perturb = 1.0
        net.cfg.train_precision = 'bf16'
        net.train()
        opt = FusedAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4)
        sel = batches[name]
        assert sel.size == TRAIN_RAYS
        frame = dict(full)
        for k in ('near', 'far'):
            frame[k] = full[k][sel]
        frame['rays'] = full['rays'][:, sel]
        data = frame_to_device(frame, dev)
        for k in ('cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor'):
            data[k] = data[k].cpu()

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
        timing[name] = (time.perf_counter() - t0) / steps
    dt = timing['scattered']
    net.cfg.train_precision = 'auto'
    rows = TRAIN_RAYS * SPP
    nbytes = rows * sum(TRAIN_BYTES_PER_ROW.values()) + TRAIN_PARAM_BYTES
    flop = rows * TRAIN_FLOP_PER_ROW
    t_hbm, t_mfma = nbytes / PEAK_HBM_ACHIEVABLE, flop / PEAK_BF16_MFMA
    from occnerf_amd import train_graph
    pg = train_graph.get(net)
    traffic, traffic_src = train_pmc_traffic()
    return {'ms_per_step': dt * 1e3,
            'batch': '6 144 rays scattered over the 512 x 512 frame: the batch rounds 2-5 measured (round 5 reported it as '
                     'scattered_ms_per_step), kept as the headline of this leg for continuity',
            'patches_ms_per_step': timing['patches'] * 1e3,
            'patches_batch': '6 patches of 32 x 32 pixels lying WHOLLY on bbox-hitting pixels (patch_ray_selection(full=True)): the '
                             'shape of the reference\'s patch batch (default.yaml patch.N_patches 6, size 32) but not its sampling '
                             'rule -- the reference accepts patches by mask coverage, which need not fill them; coherent patches '
                             'collapse into more runs and are slightly faster',
            'rays_per_step': TRAIN_RAYS, 'samples_per_step': rows,
            'rays_per_s': TRAIN_RAYS / dt, 'dtype': 'bf16 MLP trunks (fp32 accumulate, fp32 master weights); '
            'fp32 sampler, encoder, aggregation, compositor', 'final_loss': float(loss.detach()),
            'what': 'forward + backward + clip_grad_norm + Adam, every per-sample stage a HIP kernel '
                    '(occnerf_amd/train_path.py); synthetic target',
            'per_step_hipgraph': {'captures': pg.captures, 'replays': pg.replays, 'failed': pg.failed},
            # whichever of the two rooflines takes longer bounds the step: here HBM (the layers are 256 wide: 0.26 FLOP/B)
            'roofline': {'bound': 'hbm' if t_hbm >= t_mfma else 'mfma',
                         'algorithmic_bytes_per_step': nbytes, 'bytes_per_row': sum(TRAIN_BYTES_PER_ROW.values()),
                         'achieved': nbytes / dt / 1e9, 'peak': PEAK_HBM_ACHIEVABLE / 1e9, 'unit': 'GB/s',
                         'frac': t_hbm / dt, 'hbm_floor_ms': t_hbm * 1e3,
                         'traffic': traffic,
                         'traffic_note': None if traffic is None else
                         f'NOT measured in this run: HBM bytes of one step summed over every kernel, from the committed file '
                         f'profiles/{traffic_src} (tools/pmc_train.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over '
                         f'tools/train_step_trace.py, the guide\'s gfx950 corrections); {traffic / nbytes:.2f}x the algorithmic bytes',
                         'peak_note': '6.3 TB/s = what a streaming kernel achieves on this part (MI355X_MICROARCH.md; datasheet 8 TB/s: '
                                      'frac_of_datasheet below)', 'frac_of_datasheet': nbytes / 8.0e12 / dt,
                         'mfma': {'flop_per_step': flop, 'peak': PEAK_BF16_MFMA / 1e12, 'unit': 'TFLOP/s',
                                  'achieved': flop / dt / 1e12, 'frac': t_mfma / dt, 'mfma_floor_ms': t_mfma * 1e3},
                         'bytes_per_row_breakdown': TRAIN_BYTES_PER_ROW,
                         'note': 'compulsory traffic of the staged design (every stage one kernel, intermediates in HBM), not a '
                                 'counter measurement; table gathers served by L2 are not counted'}}


def config4_leg(dev, frames=3):
    """BASELINE configs[3] on one GPU: 1024 x 1024 image, 192 samples/ray, non-rigid on, visibility-weighted aggregation
    with the seeded visibility pattern of SURVEY 8(d) C4; the frame resident on the device."""
    from occnerf_amd import synth
    from occnerf_amd.seeded import build_network, frame_to_device
    net = build_network(seed=0, amplify=False, S=192, non_rigid=True, device=dev)
    rng = np.random.RandomState(4)
    pc = net.point_base.detach().cpu().numpy()
    cnt = np.where(pc[:, 2] > 0, 1.0, 1.0 + rng.poisson(50, pc.shape[0])).astype(np.float32)
    with torch.no_grad():
        net.point_counter.copy_(torch.from_numpy(cnt).to(dev))
    frame = synth.make_frame(img_size=1024, pose72=synth.seeded_pose(1), orbit_frame=28)
    data = frame_to_device(frame, dev)
    R = frame['rays'].shape[1]
    out = {'workload': 'BASELINE configs[3]: 1024x1024 image, 192 samples/ray, non-rigid on, seeded visibility counts',
           'rays_per_frame': R, 'samples_per_frame': R * 192}
    for name, dedup in (('default', True), ('every_live_sample', False)):
        net.cfg.dedup_repeated_samples = dedup
        with torch.no_grad():
            net(**data, iter_val=1e7, ray_order_key='c4')
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(frames):
                net(**data, iter_val=1e7, ray_order_key='c4')
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / frames
        out[name] = {'ms_per_frame': dt * 1e3, 'value': R / dt, 'unit': 'rays/s', 'dedup_repeated_samples': dedup}
    out['peak_mem_GiB'] = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    del net, data
    torch.cuda.empty_cache()
    return out


def movement_leg(net, dev, n_frames=12):
    """BASELINE configs[2]'s sequence on ONE GPU, through the loop `run.py --type movement` executes
    (occnerf_amd/sequence.render_sequence): per frame the synthetic source's host work, ray generation on the device,
    the render with the camera named, device image assembly and the uint8 image copied to pinned host memory (PNG
    encoding excluded, as in the metric).  A warm pass first (weights packed, kNN layout built); the timed pass starts
    with empty ray-order / shard-plan caches like a fresh sequence would."""
    from occnerf_amd.image import assemble_uint8_device
    from occnerf_amd.parallel import ShardedRenderer
    from occnerf_amd.sequence import SyntheticFrames, render_sequence
    loader = SyntheticFrames('movement', img_size=IMG, render_frames=n_frames, device_rays=True)
    stage = torch.empty(IMG, IMG, 3, dtype=torch.uint8).pin_memory()
    bg = np.array([1., 1., 1.])
    out = {'workload': f'BASELINE configs[2] on one GPU: {n_frames} frames of the movement pose walk, 512x512 x 128, '
                       'device ray generation + render + device image assembly + uint8 D2H per frame',
           'frames': n_frames}
    for name, dedup in (('every_live_sample', False), ('default', True)):
        net.cfg.dedup_repeated_samples = dedup
        rays = []

        def on_frame(o, meta):
            img, _ = assemble_uint8_device(meta['width'], meta['height'], meta['ray_index'], bg, o['rgb'], o['alpha'],
                                           want_alpha=False)
            stage.copy_(img, non_blocking=True)
            rays.append(int(meta['ray_index'].numel()))
        renderer = ShardedRenderer(net, dev, single=True)
        render_sequence(renderer, loader, 'movement', 1e7, on_frame, dev)          # warm pass
        torch.cuda.synchronize()
        rays.clear()
        net._ray_orders.clear()
        renderer = ShardedRenderer(net, dev, single=True)
        t0 = time.perf_counter()
        render_sequence(renderer, loader, 'movement', 1e7, on_frame, dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {'value': sum(rays) / dt, 'unit': 'rays/s', 'ms_per_frame': dt / n_frames * 1e3,
                     'rays_per_frame_mean': float(np.mean(rays)), 'dedup_repeated_samples': dedup}
    net.cfg.dedup_repeated_samples = False
    return out


def freeview_orbit_leg(net, dev, n_frames=8, first=28):
    """A free-view orbit has a NEW camera every frame (freeview.py:133-142, camera_util.py:85-110): nothing can be named or
    cached across frames -- rays generated on the device, the Morton walk of THEIR directions computed (csrc/rays.hip
    occnerf_ray_order), render, device image assembly, uint8 D2H -- through the loop `run.py --type freeview` executes.
    `n_frames` consecutive frames of the 100-step orbit starting at the headline's camera; every live sample evaluated.  The
    per-frame cost of the order itself is shown separately (HIP events), beside the torch construction it replaced."""
    from occnerf_amd.image import assemble_uint8_device
    from occnerf_amd.parallel import ShardedRenderer
    from occnerf_amd.rayorder import _order_fp32, ray_patch_order
    from occnerf_amd.sequence import SyntheticFrames, frames_to_device, render_sequence
    loader = SyntheticFrames('freeview', img_size=IMG, render_frames=100, device_rays=True, frame_range=(first, n_frames))
    stage = torch.empty(IMG, IMG, 3, dtype=torch.uint8).pin_memory()
    bg = np.array([1., 1., 1.])
    rays = []

    def on_frame(o, meta):
        img, _ = assemble_uint8_device(meta['width'], meta['height'], meta['ray_index'], bg, o['rgb'], o['alpha'], want_alpha=False)
        stage.copy_(img, non_blocking=True)
        rays.append(int(meta['ray_index'].numel()))
    net.cfg.dedup_repeated_samples = False
    renderer = ShardedRenderer(net, dev, single=True)
    render_sequence(renderer, loader, 'freeview', 1e7, on_frame, dev)          # warm pass
    torch.cuda.synchronize()
    rays.clear()
    net._ray_orders.clear()
    t0 = time.perf_counter()
    render_sequence(renderer, loader, 'freeview', 1e7, on_frame, dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the order alone, on the same frames' ray directions
    ms_hip, ms_torch = [], []
    with torch.no_grad():
        for data, key, meta in frames_to_device(loader, 'freeview', dev):
            assert key is None
            d = data['rays'][1].contiguous()
            for fn, acc in ((ray_patch_order, ms_hip), (_order_fp32, ms_torch)):
                fn(d)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t1 = time.perf_counter()
                e0.record()
                fn(d)
                e1.record()
                torch.cuda.synchronize()
                acc.append((e0.elapsed_time(e1), (time.perf_counter() - t1) * 1e3))
    return {'workload': f'{n_frames} consecutive frames (orbit steps {first}..{first + n_frames - 1} of 100) of the free-view orbit of the '
                        'headline pose, 512x512 x 128: a new camera every frame, ray_order_key=None, device ray generation + Morton '
                        'order + render + device image assembly + uint8 D2H per frame; every live sample evaluated',
            'frames': n_frames, 'value': sum(rays) / dt, 'unit': 'rays/s', 'ms_per_frame': dt / n_frames * 1e3,
            'rays_per_frame_mean': float(np.mean(rays)),
            'ray_order_ms_per_frame': {'hip_kernels_gpu': float(np.mean([a for a, _ in ms_hip])),
                                       'hip_kernels_wall': float(np.mean([b for _, b in ms_hip])),
                                       'torch_ops_gpu': float(np.mean([a for a, _ in ms_torch])),
                                       'torch_ops_wall': float(np.mean([b for _, b in ms_torch])),
                                       'note': 'occnerf_ray_order (3 kernels + hipcub radix sort) is what the frames above used; the '
                                               'torch construction (~35 launches + a stable argsort) is what it replaced'}}


def predicted_scaling_leg(net, frame_h, dev, host_out, t1_ms, frames=6):
    """PREDICTED, SINGLE GPU -- not a scaling curve.  No multi-GPU node has been available: for N in {2, 4, 8} EVERY rank k of
    the N-rank plan of the headline frame is emulated in this process (ShardedRenderer(emulate=(N, k)): the rank's share
    gathered from the pinned host frame, H2D, rendered, copied to the padded send buffer, one gather on the one-rank RCCL
    group into slot k of the N-slot receive buffer; rank 0 also un-permutes and copies the frame back to the host), pipelined
    like the headline.  A real node's frame takes at least the slowest rank's time (+ the gather's wire time: <= 5.2 MB over
    xGMI, ~0.05 ms).  per_rank_fixed_ms = mean over ranks of (rank time - T1 x the rank's share of the live samples): what a
    rank spends beyond its proportional share (preamble, selects, kernel tails at 1/N of the size, launch gaps)."""
    from occnerf_amd.parallel import ShardedRenderer
    out = {'label': 'predicted, single GPU: every rank of the N-rank plan emulated one after the other on ONE MI355X through the '
                    'collective renderer (one-rank RCCL group); NOT a measured scaling curve',
           'T1_ms': t1_ms, 'worlds': {}}
    for N in (2, 4, 8):
        rows = []
        for k in range(N):
            r = ShardedRenderer(net, dev, emulate=(N, k))
            _, step_ms = timed_steps(r, frame_h, frames, 3, 0, 1, dev, ('pred', N), host_out)
            rays_k, live_k = r.shard_stats()
            # median of the per-frame HIP-event times (a rank's first frames at a new shard size pay allocator misses)
            rows.append((float(np.median(step_ms)), int(rays_k), int(live_k if live_k is not None else -1)))
            del r
        ms = np.array([a for a, _, _ in rows])
        live = np.array([c for _, _, c in rows], dtype=np.float64)
        fixed = float(np.mean(ms - t1_ms * live / live.sum())) if (live > 0).all() else None
        out['worlds'][str(N)] = {'N': N, 'slowest_rank_ms': float(ms.max()), 'mean_rank_ms': float(ms.mean()),
                                 'rank0_ms': float(ms[0]), 'T1_over_slowest': t1_ms / float(ms.max()),
                                 'per_rank_fixed_ms': fixed, 'per_rank': [[round(a, 3), b, c] for a, b, c in rows]}
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `torch.distributed.run` (this
    process has made no GPU call: a fresh child, never a re-exec), relay rank 0's JSON line, -> the child's exit code."""
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr',
           '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:                      # (stderr is inherited)
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def rccl_world1_leg(net, frame_h, dev, steps, want, host_out, t1_ms=None):
    """The N > 1 branch of the sharded renderer on the one GPU of this box: a ONE-rank `nccl` (= RCCL) process group,
    `force_collective`: shard plan from the Morton walk + its checksum all-gather, padded send buffer, asynchronous
    dist.gather into the list-of-views receive buffer, work.wait(), un-permutation.  Pixels must equal `want` (the
    headline renderer's [R,5] block) bit for bit."""
    from occnerf_amd.parallel import ShardedRenderer
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(_free_port()))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        r = ShardedRenderer(net, dev, force_collective=True)
        dtc, _ = timed_steps(r, frame_h, steps, 2, 0, 1, dev, 'bench', host_out)
        with torch.no_grad():
            out = r.finish(r.submit(frame_h, ray_order_key='bench'))['packed']
        torch.cuda.synchronize()
        R = out.shape[0]
        res = {'backend': dist.get_backend(), 'world_size_formed': dist.get_world_size(), 'collective': r.collective,
               'gathers_issued': r.gathers_issued, 'plans_verified': r.plans_verified,
               'bit_identical_to_headline': bool(torch.equal(out, want)),
               'value': R * steps / dtc, 'unit': 'rays/s', 'ms_per_step': dtc / steps * 1e3,
               'note': 'one-rank RCCL group through the N > 1 code path (dist.gather on device buffers); not the headline'}
        if t1_ms is not None:          # while the group exists: every rank of the 2 / 4 / 8-rank plans emulated through it
            try:
                res['predicted_scaling'] = predicted_scaling_leg(net, frame_h, dev, host_out, t1_ms)
            except Exception as e:
                res['predicted_scaling'] = {'error': f'{type(e).__name__}: {e}'[:400]}
        return res
    finally:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--cpu-rays', type=int, default=4096, help='rays in the CPU baseline sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the side measurements (bf16x3, all samples, train)')
    ap.add_argument('--rccl-inline', action='store_true', help=argparse.SUPPRESS)      # (the child process of the rccl_world1 leg)
    ap.add_argument('--only', default=None, help='comma-separated side legs to run beside the headline (default: all): '
                                                 'dedup,rccl_world1,alt,alt2,no_shortcuts,all_samples,train,movement,freeview_orbit,config4')
    args = ap.parse_args()
    only = None if args.only is None else set(args.only.split(','))

    def leg(name):
        return not args.no_alt and (only is None or name in only)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:      # no launcher around us: start the ranks ourselves
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('OCC_FORCE_DEVICE', os.environ.get('LOCAL_RANK', 0)))
    # Contract: rank 0 prints ONE JSON line on stdout.  Libraries print there too (RCCL's version banner at communicator
    # creation, through C stdio): keep a private handle on the real stdout for the line and send everything else --
    # Python- and C-level -- to stderr.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; start it with '
                         f'--nproc-per-node {args.gpus}, or without a launcher (bench.py then starts its ranks itself)')
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
    # experiments only (A/B runs of a renderer option through the whole bench, child processes included):
    # OCC_BENCH_CFG="key=value,key=value" is applied to the network's cfg and reported in config.cfg_overrides
    cfg_overrides = {}
    for kv in filter(None, os.environ.get('OCC_BENCH_CFG', '').split(',')):
        import yaml
        k, v = kv.split('=', 1)
        cfg_overrides[k] = net.cfg[k] = yaml.safe_load(v)
    frame = synth.make_frame(img_size=IMG, pose72=synth.seeded_pose(1), orbit_frame=28)
    frame_h = host_frame(frame)
    R = frame['rays'].shape[1]
    host_out = torch.empty(R, 5).pin_memory()
    renderer = ShardedRenderer(net, dev)
    formed_world = renderer.formed_world_size()

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
    nr_events, real_nr = [], ops.nonrigid_rows

    def timed_nr(xyz, rows, count, *a, **kw):
        st = torch.cuda.current_stream(xyz.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        out = real_nr(xyz, rows, count, *a, **kw)
        e1.record(st)
        nr_events.append((e0, e1))
        return out
    # The headline evaluates every live sample (the definition of rounds 1-2); the renderer's default also evaluates runs of
    # bitwise identical samples once (cfg.dedup_repeated_samples, bit-identical pixels): reported beside it as `dedup`.
    net.cfg.dedup_repeated_samples = False
    ops.canonical_mlp, ops.nonrigid_rows = timed_mlp, timed_nr
    timed_steps(renderer, frame_h, 1, args.warmup, rank, world, dev, 'bench', host_out)   # warm-up (+1 step)
    mlp_events.clear()
    nr_events.clear()
    dt, step_ms = timed_steps(renderer, frame_h, args.steps, 0, rank, world, dev, 'bench', host_out)
    main_events, main_nr = list(mlp_events), list(nr_events)
    ops.canonical_mlp, ops.nonrigid_rows = real_mlp, real_nr
    torch.cuda.synchronize()
    headline_out = host_out.clone() if world == 1 else None      # rank 0's [R,5] of the last timed frame
    # balance of the shard plan: rays and live samples every rank rendered in the last frame (read after the timed region)
    my_rays, my_live = renderer.shard_stats()
    per_rank = [[my_rays, my_live if my_live is not None else -1]]
    if world > 1:
        mine = torch.tensor(per_rank[0], device=dev, dtype=torch.int64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [t.tolist() for t in allr]
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])

    side = {}
    if leg('dedup'):
        net.cfg.dedup_repeated_samples = True
        if world > 1:
            dist.barrier()
        dtd, _ = timed_steps(renderer, frame_h, args.steps, args.warmup, rank, world, dev, 'bench', host_out)
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
        if leg('rccl_world1'):
            # A side leg never takes the headline down -- not even by hanging inside a communicator: unless this IS the
            # child, the leg runs in a child process (this script with --rccl-inline: its own few headline frames, then the
            # one-rank nccl group) under a time limit, and only its `rccl_world1` object is taken over.
            try:
                if args.rccl_inline:
                    side['rccl_world1'] = rccl_world1_leg(net, frame_h, dev, max(3, args.steps // 2), headline_out.to(dev), host_out,
                                                          t1_ms=dt / args.steps * 1e3)
                else:
                    import subprocess
                    child = subprocess.run([sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps',
                                            str(max(3, args.steps // 2)), '--warmup', '2', '--no-cpu-baseline', '--only', 'rccl_world1',
                                            '--rccl-inline'], capture_output=True, text=True, timeout=400,
                                           env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')})
                    lines = [l for l in child.stdout.splitlines() if l.startswith('{')]
                    if child.returncode != 0 or not lines:
                        raise RuntimeError(f'child rc {child.returncode}: {child.stderr[-300:]}')
                    cl = json.loads(lines[-1])
                    side['rccl_world1'] = dict(cl['rccl_world1'], headline_ms_per_step_in_child=cl['ms_per_step'])
                    # (the 2 / 4 / 8-rank plans, every rank emulated through the child's one-rank RCCL group: its own key)
                    ps = side['rccl_world1'].pop('predicted_scaling', None)
                    if ps is not None:
                        side['predicted_scaling'] = ps
            except Exception as e:
                side['rccl_world1'] = {'error': f'{type(e).__name__}: {e}'[:400]}
        if leg('alt'):
            # opt-in split-bf16 MLP path (cfg.mlp_precision='bf16x3'): same frame, same steps; never part of `value`
            net.cfg.mlp_precision = 'bf16x3'
            net.invalidate_cache()
            dta, _ = timed_steps(renderer, frame_h, args.steps, args.warmup, rank, world, dev, 'bench', host_out)
            net.cfg.dedup_repeated_samples = True          # ... and in the renderer's default configuration, like `dedup`
            dtb, _ = timed_steps(renderer, frame_h, args.steps, args.warmup, rank, world, dev, 'bench', host_out)
            net.cfg.dedup_repeated_samples = False
            side['alt'] = {'mlp_precision': 'bf16x3 (hi/lo bf16 operands, 3 MFMA products, fp32 accumulate): meets the 1e-4 pixel '
                                            'gate on the random-init checkpoint, NOT on the trained-like one (alpha 1.5e-4, depth '
                                            '9e-4: DESIGN.md 3.5); same device-side live list / repeated-sample elimination as fp32',
                           'value': R * args.steps / dta, 'unit': 'rays/s', 'ms_per_step': dta / args.steps * 1e3,
                           'default': {'dedup_repeated_samples': True, 'value': R * args.steps / dtb, 'unit': 'rays/s',
                                       'ms_per_step': dtb / args.steps * 1e3}}
            net.cfg.mlp_precision = 'fp32'
            net.invalidate_cache()
        if leg('alt2'):
            # opt-in fp32-GRADE split-fp16 MLP path (cfg.mlp_precision='f16x3', csrc/split.h): same frame, same steps; never part
            # of `value` -- the headline stays the exact fp32 kernel.  The kernel's own launches are timed with HIP events.
            net.cfg.mlp_precision = 'f16x3'
            net.invalidate_cache()
            # (the out-of-domain flag: checked after every frame -- one wait per frame, the renderer's default -- for
            # `checked_per_frame_ms_per_step`; checked behind the pipelined loop, Network.check_f16x3_domain(), for the leg's figure)
            dts2, _ = timed_steps(renderer, frame_h, args.steps, args.warmup, rank, world, dev, 'bench', host_out)
            net.cfg.f16x3_domain_check = False            # the unwatched kernels (round 5's): no flag word, nothing reported
            net.invalidate_cache()
            dtu2, _ = timed_steps(renderer, frame_h, args.steps, args.warmup, rank, world, dev, 'bench', host_out)
            net.cfg.f16x3_domain_check = 'deferred'
            net.invalidate_cache()
            ev2, real_mlp2 = [], ops.canonical_mlp_bf16x3

            def timed_mlp2(*a, **k):
                st = torch.cuda.current_stream(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                r = real_mlp2(*a, **k)
                e1.record(st)
                ev2.append((e0, e1))
                return r
            ops.canonical_mlp_bf16x3 = timed_mlp2
            try:
                dta2, _ = timed_steps(renderer, frame_h, args.steps, args.warmup, rank, world, dev, 'bench', host_out)
            finally:
                ops.canonical_mlp_bf16x3 = real_mlp2
            torch.cuda.synchronize()
            mlp2_ms = float(np.mean([a.elapsed_time(b) for a, b in ev2[-args.steps:]])) if ev2 else None
            net.cfg.dedup_repeated_samples = True
            dtb2, _ = timed_steps(renderer, frame_h, args.steps, args.warmup, rank, world, dev, 'bench', host_out)
            net.cfg.dedup_repeated_samples = False
            net.check_f16x3_domain()                      # raises if any of the frames above left the mode's domain
            net.cfg.f16x3_domain_check = True
            net.invalidate_cache()
            live2 = float(net.last_live_count) if getattr(net, 'last_live_count', None) is not None else None
            side['alt2'] = {'mlp_precision': 'f16x3 (two fp16 pieces per operand kept in the normal range: 22 significand bits, 3 MFMA '
                                             'products on v_mfma_f32_32x32x16_f16, fp32 accumulate): meets the fp32 kernel\'s own tolerances '
                                             'and the 1e-4 pixel gate on all three checkpoints (tests/test_y_alternatives.py *_f16x3); domain: '
                                             'hidden activations below 4 094',
                            'value': R * args.steps / dta2, 'unit': 'rays/s', 'ms_per_step': dta2 / args.steps * 1e3,
                            'domain_check': "cfg.f16x3_domain_check='deferred': every frame's out-of-domain flag copied behind the frame "
                                            'and verified after the loop (none set); checked_per_frame_ms_per_step = the default, one '
                                            'wait per frame with an fp32 re-render on violation',
                            'checked_per_frame_ms_per_step': dts2 / args.steps * 1e3,
                            'unchecked_ms_per_step': dtu2 / args.steps * 1e3,
                            'unchecked_note': 'cfg.f16x3_domain_check=False: the kernels without the running maximum (round 5\'s '
                                              'form); a checkpoint outside the domain would then saturate silently',
                            'canonical_mlp_launch_ms': mlp2_ms,
                            'canonical_mlp_algorithmic_tflops': None if not (mlp2_ms and live2) else
                            FLOP_PER_SAMPLE_CNL * live2 / (mlp2_ms * 1e-3) / 1e12,
                            'note': 'algorithmic TFLOP/s = the network\'s 923 136 FLOP/sample over the launch time (the kernel executes 3x '
                                    'that on the fp16 pipe, peak 2 500 dense); not comparable with the fp32 roofline fraction',
                            'default': {'dedup_repeated_samples': True, 'value': R * args.steps / dtb2, 'unit': 'rays/s',
                                        'ms_per_step': dtb2 / args.steps * 1e3}}
            net.cfg.mlp_precision = 'fp32'
            net.invalidate_cache()
        if leg('no_shortcuts'):
            # the two exact data-dependent shortcuts of round 4 off (kNN / feature centre cache, warp bone culling): the same frame
            # the way round 3 rendered it -- identical pixels
            net.cfg.knn_center_cache, net.cfg.warp_bone_culling = False, False
            dtn, _ = timed_steps(renderer, frame_h, args.steps, args.warmup, rank, world, dev, 'bench', host_out)
            net.cfg.knn_center_cache, net.cfg.warp_bone_culling = True, True
            side['no_shortcuts'] = {
                'knn_center_cache': False, 'warp_bone_culling': False, 'value': R * args.steps / dtn, 'unit': 'rays/s',
                'ms_per_step': dtn / args.steps * 1e3,
                'note': 'the headline serves the kNN queries (and their feature aggregate) that lie within a proven radius of the '
                        'frame\'s collapse point from ONE search, and skips bones whose weight channel cannot reach a wave\'s samples; '
                        'both exact (bit-identical pixels, tested); every live sample still goes through both MLPs'}
        if leg('all_samples'):
            # every sample evaluated (cfg.skip_empty_samples off): same pixels bit for bit
            net.cfg.skip_empty_samples = False
            dtf, _ = timed_steps(renderer, frame_h, args.steps, args.warmup, rank, world, dev, 'bench', host_out)
            side['all_samples'] = {
                'skip_empty_samples': False, 'value': R * args.steps / dtf, 'unit': 'rays/s',
                'ms_per_step': dtf / args.steps * 1e3,
                'note': 'all R x 128 samples through every stage; the headline drops the samples whose motion-weight '
                        'sum is exactly 0 (alpha is multiplied by it), with bit-identical rgb/alpha/depth'}
            net.cfg.skip_empty_samples = True
        if leg('train'):
            side['train'] = train_leg(dev, max(5, args.steps // 2), 3)
        if leg('movement'):
            side['movement'] = movement_leg(net, dev)
        if leg('freeview_orbit'):
            side['freeview_orbit'] = freeview_orbit_leg(net, dev)
        if leg('config4'):
            del renderer
            torch.cuda.empty_cache()
            side['config4'] = config4_leg(dev)
    elif leg('weak_frames'):
        # round-1 mode: one whole frame per rank per step, same-size gather (weak scaling)
        whole = ShardedRenderer(net, dev, single=True)         # every rank renders the full frame by itself
        dist.barrier()
        dtw, _ = timed_steps(whole, frame_h, max(3, args.steps // 4), 1, rank, 1, dev, 'bench-whole', host_out)
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
        nr_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in main_nr])) if main_nr else None
        live = float(np.mean(nsmp))
        step_s = dt / args.steps
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
                       'knn_center_cache': bool(net.cfg.get('knn_center_cache', True)),
                       'warp_bone_culling': bool(net.cfg.get('warp_bone_culling', True)),
                       'world_size_formed': formed_world, 'backend': backend if world > 1 else None,
                       **({'cfg_overrides': cfg_overrides} if cfg_overrides else {}),
                       'per_rank_rays': [p[0] for p in per_rank],
                       'per_rank_live_samples': [p[1] for p in per_rank],
                       'parallelism': f'one frame, its Morton walk dealt to {world} rank(s) in 256-ray blocks, async RCCL gather '
                                      'to rank 0 overlapped with the next frame'},
            'roofline': {'bound': 'mfma', 'kernel': 'occ::m16::canonical_mlp_lds_kernel (fp32 MFMA 16x16x4, LDS-staged weights)',
                         'achieved': achieved / 1e12, 'peak': PEAK_FP32_MFMA / 1e12, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_FP32_MFMA, 'traffic': traffic,
                         'traffic_note': f'NOT measured in this run: read from the committed file profiles/{traffic_src} (separate '
                                         'rocprofv3 --pmc passes over this very launch size, tools/pmc_hbm.sh: HBM bytes per '
                                         'launch = FETCH_SIZE x2 + WRITE_SIZE with the guide\'s gfx950 corrections); null when no '
                                         'committed pass matches the launch size; algorithmic 288 B/sample',
                         'launch_ms': avg_ms, 'launches_timed': len(ms),
                         'flop_per_launch': FLOP_PER_SAMPLE_CNL * float(np.mean(nsmp)),
                         # the second MFMA kernel of the frame (rank 0's launches): algorithmic 200 704 FLOP per live sample
                         # (91 520 MAC = 183 040 FLOP executed after folding the 69 per-frame condition inputs into a bias)
                         'nonrigid': None if nr_ms is None else {
                             'kernel': 'occ::nr16::nonrigid_lds_kernel', 'launch_ms': nr_ms,
                             'achieved_algorithmic': FLOP_PER_SAMPLE_NR * live / (nr_ms * 1e-3) / 1e12,
                             'frac_algorithmic': FLOP_PER_SAMPLE_NR * live / (nr_ms * 1e-3) / PEAK_FP32_MFMA,
                             'frac_executed': 183040 * live / (nr_ms * 1e-3) / PEAK_FP32_MFMA},
                         # whole step: canonical + non-rigid algorithmic FLOP of the samples rank 0 evaluated / step time
                         'end_to_end_frac': (FLOP_PER_SAMPLE_CNL + FLOP_PER_SAMPLE_NR) * live / step_s / PEAK_FP32_MFMA},
        }
        line.update(side)
        if world == 1 and not args.no_cpu_baseline:
            from oracle.chain import model_context                  # the checker, timed as the stated CPU baseline
            line['cpu_baseline'] = cpu_baseline(model_context(0, False), frame, args.cpu_rays)
        json_out.write(json.dumps(line) + '\n')
        json_out.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
