import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util   # (golden loader: this is a debugging aid for the tests)
from occnerf_amd.seeded import build_network, frame_to_device
from occnerf_amd import synth
g = util.load_golden('train_amp_s32')
net = build_network(0, True, S=32, non_rigid=True)
net.cfg.perturb = 1.0
net.train()
frame = synth.make_frame(img_size=32, pose72=g['meta.pose72'], orbit_frame=7)
for k in ('rays', 'near', 'far'):
    frame[k] = g['in.' + k]
data = frame_to_device(frame, 'cuda:0')
out = net(**data, iter_val=1e7, t_rand=torch.from_numpy(g['in.t_rand']).cuda())
loss = (out['rgb'] ** 2).mean() + 0.5 * out['alpha'].mean() + 0.01 * out['depth'].mean() + 0.1 * out['comp_loss'].mean()
loss.backward()
got = net.cnl_mlp.module.pts_linears[0].weight.grad.cpu().numpy()
want = g['grad.cnl_mlp.module.pts_linears.0.weight']
err = np.abs(got - want)
print('col max err', np.round(err.max(0) / np.abs(want).max(), 4))
print('col scale  ', np.round(np.abs(want).max(0) / np.abs(want).max(), 4))
r, c = np.unravel_index(err.argmax(), err.shape)
print('worst', r, c, got[r, c], want[r, c])
