"""Where do the aggregation backward's scatters go?  One training step at config-5 size; the (grad rows, ids) handed to
ops.agg_backward are analysed in torch: run heads, pairs per point tile, hottest points."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from occnerf_amd import ops, synth
from occnerf_amd.seeded import build_network, frame_to_device
net = build_network(0, False, S=128, non_rigid=True)
net.cfg.perturb = 1.0; net.cfg.train_precision = 'bf16'; net.train()
frame = synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28)
sel = np.sort(np.random.RandomState(0).choice(frame['rays'].shape[1], 6144, replace=False))
for k in ('near', 'far'): frame[k] = frame[k][sel]
frame['rays'] = frame['rays'][:, sel]
data = frame_to_device(frame, 'cuda:0')
grab = {}
real = ops.agg_backward
def hook(g, knn, atts, P):
    grab.update(g=g.clone(), knn=knn.clone(), atts=atts.clone(), P=P)
    return real(g, knn, atts, P)
ops.agg_backward = hook
out = net(**data, iter_val=1e7)
(((out['rgb'] - 0.5) ** 2).mean() + 0.1 * out['comp_loss'].mean()).backward()
torch.cuda.synchronize()
g, knn = grab['g'], grab['knn']
N = g.shape[0]
nz = (g != 0).any(1)
same = torch.zeros(N, dtype=torch.bool, device=g.device)
same[1:] = (knn[1:] == knn[:-1]).all(1)
same[torch.arange(0, N, 64, device=g.device)] = False
print('samples', N, 'nonzero-gradient rows', int(nz.sum()), 'rows identical to predecessor (within 64-chunks)', int(same.sum()))
heads = nz & ~same          # (approximation: a zero row inside a run does not break it)
print('run heads (approx)', int(heads.sum()))
tp = 18432 // g.shape[1]
tiles = (grab['P'] + tp - 1) // tp
pairs = torch.bincount((knn[heads].reshape(-1) // tp).long(), minlength=tiles)
print('tile_points', tp, 'tiles', tiles, '(head, neighbour) pairs per tile:', pairs.tolist(), 'total', int(pairs.sum()))
allpairs = torch.bincount((knn[nz].reshape(-1) // tp).long(), minlength=tiles)
print('without run merging:', allpairs.tolist(), 'total', int(allpairs.sum()))
pts = torch.bincount(knn[heads].reshape(-1).long(), minlength=grab['P'])
top = torch.topk(pts, 12)
print('hottest points (pairs):', list(zip(top.indices.tolist(), top.values.tolist())))
m_per_head = torch.zeros(int(heads.sum()), device=g.device)
t_of = (knn[heads] // tp)
ntile = torch.zeros(int(heads.sum()), tiles, device=g.device).scatter_(1, t_of.long(), 1.0).sum(1)
print('tiles touched per head: mean', float(ntile.mean()), 'max', float(ntile.max()))
# time the two kernels
for _ in range(3): real(g, knn, grab['atts'], grab['P'])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): real(g, knn, grab['atts'], grab['P'])
e1.record(); torch.cuda.synchronize()
print('agg_backward (runs + tiles + partial.sum) ms', e0.elapsed_time(e1) / 10)


def t(gg, kk, aa, n=10):
    for _ in range(2): real(gg, kk, aa, grab['P'])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): real(gg, kk, aa, grab['P'])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
atts = grab['atts']
print('zero gradients (masks all zero: fixed cost)      ', t(torch.zeros_like(g), knn, atts))
print('real                                              ', t(g, knn, atts))
g8 = g.clone(); g8[torch.arange(N, device=g.device) % 8 != 0] = 0
print('every 8th sample only                             ', t(g8, knn, atts))
kr = torch.randint(0, grab['P'], knn.shape, device=knn.device, dtype=torch.int32)
print('random ids (no runs, pairs spread over all tiles)  ', t(g, kr, atts))
k1 = (knn % 526).contiguous()
print('all ids folded into tile 0                         ', t(g, k1, atts))
