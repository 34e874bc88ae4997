"""Render entry point with the reference's command line (run.py:246-247, configs/config.py:65-72):

    python run.py --cfg configs/occnerf/synthetic/occnerf.yaml --type {tpose,freeview,movement,allview,evaluate} [KEY VALUE ...]

Frames go to experiments/<category>/<task>/<subject>/<experiment>/<load_net>/<folder>/NNNNNN.png
exactly like the reference (run.py:79-81, image_util.py:53-75).  `load_net: seeded[:N]` renders the
seeded random-init checkpoint (occnerf_amd/checkpoint.py) when no .tar is on disk.

Several GPUs: start it under torchrun, one process per GPU --
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 run.py --cfg ... --type movement
every rank renders its share of each frame's rays (occnerf_amd/parallel.py) and rank 0 gathers them over
RCCL, assembles and writes the images."""
import os
import time

import numpy as np
import torch

from configs import cfg, args
cfg.bgcolor = [255., 255., 255.]

from core.data import create_dataloader  # noqa: E402
from core.nets import create_network  # noqa: E402
from occnerf_amd.image import ImageWriter, assemble_uint8_device  # noqa: E402
from occnerf_amd.parallel import ShardedRenderer  # noqa: E402
from occnerf_amd.sequence import frames_to_device, render_sequence  # noqa: E402


def load_network(model):
    ckpt_path = os.path.join(cfg.logdir, f'{cfg.load_net}.tar')
    if os.path.exists(ckpt_path):
        ckpt = torch.load(ckpt_path, map_location='cpu')
        model.load_state_dict(ckpt['network'], strict=True)
        print('load network from ', ckpt_path)
    elif str(cfg.load_net).startswith('seeded'):
        from occnerf_amd.checkpoint import make_state_dict
        seed = int(str(cfg.load_net).split(':')[1]) if ':' in str(cfg.load_net) else 0
        model.load_state_dict(make_state_dict(model.point_base.detach().numpy(), float(model.bound), seed=seed),
                              strict=True)
        print(f'seeded random-init checkpoint (seed {seed})')
    else:
        raise FileNotFoundError(ckpt_path)
    return model.cuda().deploy_mlps_to_secondary_gpus()


def _init_ranks():
    """One process per GPU under torchrun (RANK / LOCAL_RANK / WORLD_SIZE); a plain launch is world 1.
    OCC_DIST_BACKEND=gloo OCC_FORCE_DEVICE=0 (tests): several ranks share one GPU and exchange through the host -- RCCL
    refuses two ranks per device; everything but the collective itself is then the production path.
    OCC_FORCE_COLLECTIVE=1 under a launcher with ONE rank: the one-rank `nccl` group is formed too and ShardedRenderer takes
    its N > 1 branch (plan, checksum all-gather, device-buffer gather, un-permutation) -- the RCCL path on a single-GPU box."""
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    torch.cuda.set_device(int(os.environ.get('OCC_FORCE_DEVICE', os.environ.get('LOCAL_RANK', 0))))
    forced = 'WORLD_SIZE' in os.environ and os.environ.get('OCC_FORCE_COLLECTIVE', '0') == '1'
    if (world > 1 or forced) and not torch.distributed.is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = os.environ.get('OCC_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', rank=rank, world_size=world,
                                                 device_id=torch.device('cuda', torch.cuda.current_device()))
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def _setup(data_type, **loader_kw):
    cfg.perturb = 0.
    rank, world = _init_ranks()
    model = create_network()
    loader = create_dataloader(data_type, **loader_kw)
    model.generate_neural_points(loader.dataset.avg_betas)
    model = load_network(model).eval()
    dev = torch.device('cuda', torch.cuda.current_device())
    return rank, world, model, loader, ShardedRenderer(model, dev), dev


def _finish_ranks(rank, world):
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _render(data_type, folder_name):
    """run.py:66-119 (_freeview) and :137-186 (run_movement): every frame of the loader through the network, images to
    <logdir>/<load_net>/<folder>/NNNNNN.png.  One process per GPU: each renders its share of the frame's rays, rank 0
    receives the gathered (rgb, alpha, depth) while the ranks already render the NEXT frame (ShardedRenderer: one frame of
    lag), assembles the image on the device and hands the uint8 pixels to the PNG writer thread."""
    rank, world, model, loader, renderer, dev = _setup(data_type)
    writer = ImageWriter(output_dir=os.path.join(cfg.logdir, str(cfg.load_net).replace(':', '_')),
                         exp_name=folder_name) if rank == 0 else None
    stats = {'rays': 0, 'first_s': 0.0, 'per_frame': [], 'frames': 0}
    torch.cuda.synchronize()
    t_wall0 = time.perf_counter()

    def on_frame(out, meta):                           # rank 0 only
        rgb_img, alpha_img = assemble_uint8_device(meta['width'], meta['height'], meta['ray_index'],
                                                   np.array(cfg.bgcolor) / 255., out['rgb'], out['alpha'],
                                                   want_alpha=bool(cfg.show_alpha))
        img_dev = torch.cat([rgb_img, alpha_img], dim=1) if cfg.show_alpha else rgb_img
        # uint8 over PCIe into a pinned staging buffer; the writer thread waits for the copy and encodes the PNG
        # while the next frame renders (nothing here blocks on the GPU)
        writer.append_device(img_dev, img_name=f"{meta['idx']:06d}" if data_type == 'movement' else None)
        stats['rays'] += int(meta['ray_index'].numel())
        stats['per_frame'].append(int(meta['ray_index'].numel()))
        stats['frames'] += 1
        if meta['idx'] == 0:
            # With one frame of lag this runs after frame 1 has been submitted, so the synchronisation covers frames 0 AND 1
            # (weight packing and the per-model kNN layout included): the steady-state figure below counts neither their
            # time nor their rays.
            torch.cuda.synchronize()
            stats['first_s'] = time.perf_counter() - t_wall0

    render_sequence(renderer, loader, data_type, cfg.eval_iter, on_frame, dev)
    if rank != 0:
        _finish_ranks(rank, world)
        return
    torch.cuda.synchronize()
    t_render = time.perf_counter() - t_wall0
    writer.finalize()
    n_rays = stats['rays']
    print(f'{n_rays} rays in {t_render:.3f} s -> {n_rays / max(t_render, 1e-9):.0f} rays/s (frame generation and image '
          f'assembly included; PNG encoding runs beside it)')
    if stats['frames'] > 2:
        print(f"frames 1-2 (warm-up, pipelined) {stats['first_s'] * 1e3:.0f} ms; frames 3..{stats['frames']}: "
              f"{sum(stats['per_frame'][2:]) / max(t_render - stats['first_s'], 1e-9):.0f} rays/s; wall clock with PNG "
              f'writing {time.perf_counter() - t_wall0:.2f} s')
    _finish_ranks(rank, world)


def PSNR(img1, img2, scale=255.):
    """run.py:21-23."""
    mse = torch.mean((img1 - img2) ** 2)
    return 20 * torch.log10(scale / torch.sqrt(mse))


def run_tpose():
    cfg.ignore_non_rigid_motions = True
    _render('tpose', 'tpose' if not cfg.render_folder_name else cfg.render_folder_name)


def run_freeview():
    _render('freeview', f'freeview_{cfg.freeview.frame_idx}' if not cfg.render_folder_name else cfg.render_folder_name)


def run_movement():
    _render('movement', 'movement' if not cfg.render_folder_name else cfg.render_folder_name)


def run_allview():
    """run.py:188-192: the frame cfg.freeview.frame_idx from every camera of the rig."""
    _render('allview', f'allview_{cfg.freeview.frame_idx}' if not cfg.render_folder_name else cfg.render_folder_name)


def run_evaluate():
    """run.py:194-244: PSNR of the rendered rays against the frames' target colours over the `progress` frames (frames
    4 and 15 skipped, the network called with iter_val = 1 exactly as the reference does, run.py:224-230: pose refinement
    and the non-rigid condition are then below their kick-in iterations).  The
    synthetic source has no photographs: its targets are a teacher's render of the same rays (the seeded, amplified
    checkpoint `cfg.evaluate_teacher`, like train.py's supervision).  Metrics beyond PSNR are out of scope."""
    from occnerf_amd.checkpoint import make_state_dict
    rank, world, model, loader, renderer, dev = _setup('progress', evaluate=True)
    teacher = create_network()
    teacher.generate_neural_points(loader.dataset.avg_betas)
    seed = int(cfg.get('evaluate_teacher_seed', 1))
    teacher.load_state_dict(make_state_dict(teacher.point_base.detach().numpy(), float(teacher.bound), seed=seed,
                                            amplify=True), strict=True)
    teacher = teacher.to(dev).eval()
    teach = ShardedRenderer(teacher, dev)
    psnrs, skips = [], [4, 15]
    with torch.no_grad():
        for data, key, meta in frames_to_device(loader, 'progress', dev):
            if meta['idx'] in skips:
                continue
            target = teach.finish(teach.submit(data, iter_val=cfg.eval_iter))
            out = renderer.finish(renderer.submit(data, iter_val=1))       # run.py:224 `batch['iter_val'] = torch.full((1,), 1)`
            if out is not None:
                psnrs.append(PSNR(out['rgb'], target['rgb'], 1.))
    if rank == 0:
        print('AVG PSNR %.4f' % torch.mean(torch.stack(psnrs)))
    _finish_ranks(rank, world)


if __name__ == '__main__':
    fn = globals().get(f'run_{args.type}')
    if fn is None:
        raise SystemExit(f"--type {args.type}: supported are tpose, freeview, movement, allview, evaluate")
    fn()
