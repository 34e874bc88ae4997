"""Which piece of the per-step static part can be captured in a hipGraph?  python3 tools/debug_graph.py"""
import os, sys, traceback
sys.path.insert(0, os.getcwd())
import torch, torch.nn as nn
from occnerf_amd import synth, train_path
from occnerf_amd.seeded import build_network, frame_to_device

net = build_network(0, False, S=32, non_rigid=True)
net.train()
frame = synth.make_frame(img_size=32, pose72=synth.seeded_pose(2), orbit_frame=7)
d = frame_to_device(frame, 'cuda:0')
posevec, dst_Rs, dst_Ts, gt, prior = (d[k][None].float().contiguous() for k in ('dst_posevec', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors'))


class Pose(nn.Module):
    def __init__(s): super().__init__(); s.m = net.pose_decoder
    def forward(s, p): return s.m(p)['Rs']
class FK(nn.Module):
    def __init__(s): super().__init__(); s.m = net.motion_basis_computer; s.w = nn.Parameter(torch.zeros(1, device='cuda'))
    def forward(s, R, T, g): a, b = s.m(R + s.w, T, g); return a, b
class Dec(nn.Module):
    def __init__(s): super().__init__(); s.m = net.mweight_vol_decoder
    def forward(s, pr): return s.m(motion_weights_priors=pr)[0]
class Pts(nn.Module):
    def __init__(s): super().__init__(); s.point_dist = net.point_dist
    def forward(s, dummy): kb, sdf = train_path.point_sdf_block(net); return kb, sdf + dummy.sum() * 0
class Inv(nn.Module):
    def __init__(s): super().__init__(); s.w = nn.Parameter(torch.zeros(1, device='cuda'))
    def forward(s, g): return torch.linalg.inv_ex(g.view(-1, 4, 4) + s.w * 0)[0]
class KnnOnly(nn.Module):
    def __init__(s): super().__init__(); s.w = nn.Parameter(torch.zeros(1, device='cuda'))
    def forward(s, x):
        from occnerf_amd import ops
        k = ops.knn_small(net.point_cloud.detach().float().contiguous(), net.point_base.detach(), 3)
        return x * s.w + k.float().sum()
class IdxPut(nn.Module):
    def __init__(s): super().__init__(); s.w = nn.Parameter(torch.ones(1, device='cuda')); s.j = torch.tensor([1, 2, 3], device='cuda'); s.p = torch.tensor([0, 0, 0], device='cuda')
    def forward(s, x):
        g = (x * s.w).clone()
        g[:, s.j] = torch.matmul(g[:, s.p], x[:, s.j])
        return g
for name, mod, args in (('pose', Pose(), (posevec,)), ('inv_ex', Inv(), (gt,)), ('index_put', IdxPut(), (gt.clone(),)), ('fk', FK(), (dst_Rs, dst_Ts, gt)),
                        ('decoder', Dec(), (prior,)), ('knn_small', KnnOnly(), (posevec,)), ('points', Pts(), (posevec,))):
    try:
        fn = torch.cuda.make_graphed_callables(mod, tuple(a.clone() for a in args), allow_unused_input=True)
        out = fn(*args)
        out = out if isinstance(out, tuple) else (out,)
        sum(o.float().sum() for o in out).backward()
        torch.cuda.synchronize()
        print(name, 'CAPTURED')
    except Exception as e:
        print(name, 'FAILED', type(e).__name__, str(e).splitlines()[0][:200])
        tb = traceback.format_exc().splitlines()
        print('    ', '\n     '.join(t for t in tb if 'occnerf_amd' in t or 'torch/' in t)[-1500:])
        try:
            torch.cuda.synchronize()
        except Exception as e2:
            print('   sync after failure:', e2)
