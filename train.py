"""Training entry point with the reference's command line (train.py:48-49):

    python train.py --cfg configs/occnerf/synthetic/occnerf.yaml [train.maxiter 200 ...]

Scope: one optimisation step = Network.forward in
training mode through the differentiable path (occnerf_amd/train_path.py: torch autograd over
the HIP kNN and the HIP grid-encoder forward/backward) + MSE and completeness losses +
clip_grad_norm + Adam with the reference's per-group learning rates (optimizer.py:12-43) +
exponential decay (exp_decay.py:7-19).  LPIPS, real datasets and progress dumps are out of scope;
the supervision here is a synthetic teacher (the same network with a second seeded checkpoint)
rendered through the HIP path.  Checkpoints use the reference's layout
({'iter','network','optimizer'} -> experiments/.../latest.tar, trainer.py:398-406)."""
import os
import time

import numpy as np
import torch

from configs import cfg, args  # noqa: F401
from core.nets import create_network
from occnerf_amd import synth
from occnerf_amd.checkpoint import make_state_dict

LR_GROUPS = (('mweight_vol_decoder', 'lr_mweight_vol_decoder'), ('pose_decoder', 'lr_pose_decoder'),
             ('non_rigid_mlp', 'lr_non_rigid_mlp'), ('point_dist', 'lr_point_dist'))
TRAIN_DEFAULTS = {'maxiter': 100, 'lr': 5e-4, 'lr_point_dist': 1e-4, 'lr_mweight_vol_decoder': 5e-5,
                  'lr_pose_decoder': 5e-5, 'lr_non_rigid_mlp': 5e-5, 'lrate_decay': 500, 'log_interval': 10,
                  'bf16': False, 'lossweights': {'mse': 0.2, 'comp': 1.0}}


def make_optimizer(net, tc):
    groups = []
    for name, p in net.named_parameters():
        if not p.requires_grad:
            continue
        lr = tc['lr']
        for key, lr_name in LR_GROUPS:
            if key in name:
                lr = tc[lr_name]
        groups.append({'params': [p], 'lr': lr, 'name': name, 'base_lr': lr})
    from occnerf_amd.optim import FusedAdam
    return FusedAdam(groups, lr=tc['lr'], betas=(0.9, 0.999))


def patch_rays(frame, rng, n_patches=6, size=32):
    """6 random 32x32 pixel patches (default.yaml:147-150) restricted to rays that hit the bbox."""
    from occnerf_amd.seeded import patch_ray_selection
    return patch_ray_selection(frame, rng, n_patches, size)


def main():
    tc = dict(TRAIN_DEFAULTS)
    tc.update({k: v for k, v in dict(cfg.get('train', {})).items() if k in TRAIN_DEFAULTS})
    dev = torch.device('cuda:0')
    net = create_network()
    net.generate_neural_points(np.zeros(10, 'float32'))
    net.load_state_dict(make_state_dict(net.point_base.detach().numpy(), float(net.bound), seed=0), strict=True)
    teacher = create_network()
    teacher.generate_neural_points(np.zeros(10, 'float32'))
    teacher.load_state_dict(make_state_dict(teacher.point_base.detach().numpy(), float(teacher.bound), seed=1,
                                            amplify=True), strict=True)
    net, teacher = net.to(dev).train(), teacher.to(dev).eval()
    opt = make_optimizer(net, tc)
    cfg.perturb = 1.0
    rng = np.random.RandomState(0)
    size = int(cfg.get('render_size', 256))
    os.makedirs(cfg.logdir, exist_ok=True)
    t0 = time.time()
    for it in range(1, int(tc['maxiter']) + 1):
        frame = synth.make_frame(img_size=size, pose72=synth.seeded_pose(100 + it % 16), orbit_frame=it % 50,
                                 orbit_period=50, bgcolor=cfg.bgcolor)
        sel = patch_rays(frame, rng, int(cfg.patch.N_patches), int(cfg.patch.size))
        frame['rays'], frame['near'], frame['far'] = frame['rays'][:, sel], frame['near'][sel], frame['far'][sel]
        keys = ['rays', 'near', 'far', 'bgcolor', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors',
                'cnl_bbox_min_xyz', 'cnl_bbox_max_xyz', 'cnl_bbox_scale_xyz', 'dst_posevec']
        data = {k: torch.from_numpy(np.ascontiguousarray(frame[k])).to(dev) for k in keys}
        with torch.no_grad():
            target = teacher(**data, iter_val=cfg.eval_iter)['rgb']
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bool(tc['bf16'])):
            out = net(**data, iter_val=it)
            loss = tc['lossweights']['mse'] * torch.mean((out['rgb'].float() - target) ** 2) \
                + tc['lossweights']['comp'] * out['comp_loss'].float().mean()
        loss.backward()
        opt.step(max_grad_norm=1.0)                                     # trainer.py:248-249: clip + Adam, one device pass
        decay = 0.1 ** (it / (tc['lrate_decay'] * 1000))                # exp_decay.py:7-19
        for grp in opt.param_groups:
            grp['lr'] = grp['base_lr'] * decay
        if it % int(tc['log_interval']) == 0 or it == 1:
            print(f'iter {it:5d}  loss {float(loss):.6f}  rays {len(sel)}  {time.time() - t0:.1f} s')
    torch.save({'iter': it, 'network': net.state_dict(), 'optimizer': opt.state_dict()},
               os.path.join(cfg.logdir, 'latest.tar'))
    print('saved', os.path.join(cfg.logdir, 'latest.tar'))


if __name__ == '__main__':
    main()
