"""Row a21: image assembly after the renderer, against the reference's own unpack_to_image."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import util
from tests.gpu_util import DEV, T, same  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unpack_to_image_byte_identical():
    from occnerf_amd.image import unpack_to_image
    g = util.load_golden('tpose_ri_image32')
    rgb_img, alpha_img, _ = unpack_to_image(32, 32, g['in.ray_mask'], g['img.bgcolor'], g['out.rgb'], g['out.alpha'])
    assert np.array_equal(rgb_img, g['img.rgb']) and np.array_equal(alpha_img, g['img.alpha'])


@pytest.mark.gpu
def test_assemble_image_kernel_byte_identical(tmp_path):
    """Row f3: the HIP image kernel (scatter by ray_mask + background fill + uint8 quantisation) on device tensors
    against the reference's own unpack_to_image output; edge values of the quantiser; the asynchronous PNG writer."""
    from PIL import Image
    from occnerf_amd.image import ImageWriter, assemble_uint8_device, unpack_to_image
    g = util.load_golden('tpose_ri_image32')
    dev = 'cuda:0'
    idx = torch.nonzero(torch.from_numpy(g['in.ray_mask'])).squeeze(1).to(dev)
    q, qa = assemble_uint8_device(32, 32, idx, g['img.bgcolor'], torch.from_numpy(g['out.rgb']).to(dev),
                                  torch.from_numpy(g['out.alpha']).to(dev))
    assert q.is_cuda and q.dtype == torch.uint8
    assert np.array_equal(q.cpu().numpy(), g['img.rgb']) and np.array_equal(qa.cpu().numpy(), g['img.alpha'])
    # quantiser edges and a sparse / empty mask at a non-square size, against the host function
    rng = np.random.RandomState(0)
    H, W = 37, 53
    for frac in (0.0, 0.02, 1.0):
        mask = rng.rand(H * W) < frac
        R = int(mask.sum())
        rgb = rng.rand(R, 3).astype(np.float32) * 1.4 - 0.2
        alpha = rng.rand(R).astype(np.float32) * 1.2 - 0.1
        if R > 8:
            rgb[:8, 0] = [0.0, 1.0, 1.0 / 255, 0.99999994, 254.5 / 255, 0.5, -0.0, 2.0]
        bg = np.array([255, 128, 3]) / 255.
        want_rgb, want_a, _ = unpack_to_image(W, H, mask, bg, rgb, alpha)
        ridx = torch.nonzero(torch.from_numpy(mask)).squeeze(1).to(dev)
        q, qa = assemble_uint8_device(W, H, ridx, bg, torch.from_numpy(rgb).to(dev), torch.from_numpy(alpha).to(dev))
        assert np.array_equal(q.cpu().numpy(), want_rgb) and np.array_equal(qa.cpu().numpy(), want_a), frac
    w = ImageWriter(str(tmp_path), 'seq', stages=2)
    imgs = []
    for t in range(5):                                      # more frames than staging buffers
        img = torch.full((16, 24, 3), 10 * t, dtype=torch.uint8, device=dev)
        img[t, t] = 255
        imgs.append(img.cpu().numpy())
        w.append_device(img)
    w.finalize()
    for t, want in enumerate(imgs):
        assert np.array_equal(np.asarray(Image.open(tmp_path / 'seq' / f'{t:06d}.png')), want)
    # frames of mixed sizes, more of each than staging buffers: every shape has its own free queue (a shared one
    # dropped the other shape's buffers for good and the writer blocked forever)
    w = ImageWriter(str(tmp_path), 'mixed', stages=2)
    imgs = []
    for t in range(9):
        shape = (16, 24, 3) if t % 3 else (20, 12, 3)
        img = torch.full(shape, 7 * t, dtype=torch.uint8, device=dev)
        imgs.append(img.cpu().numpy())
        w.append_device(img)
    w.finalize()
    for t, want in enumerate(imgs):
        assert np.array_equal(np.asarray(Image.open(tmp_path / 'mixed' / f'{t:06d}.png')), want)


@pytest.mark.gpu
def test_run_py_tpose_entry_point(tmp_path):
    """python run.py --cfg ... --type tpose renders the golden frame: same CLI, same folder layout,
    pixels within 1 grey level of the reference's PNG (1e-4 float parity -> at most one 8-bit step)."""
    from PIL import Image
    g = util.load_golden('tpose_ri_image32')
    cmd = [sys.executable, os.path.join(ROOT, 'run.py'), '--cfg',
           os.path.join(ROOT, 'configs/occnerf/synthetic/occnerf.yaml'), '--type', 'tpose',
           'render_size', '32', 'N_samples', '32']
    subprocess.check_call(cmd, cwd=str(tmp_path), env={**os.environ, 'PYTHONPATH': ROOT})
    png = tmp_path / 'experiments' / 'occnerf' / 'synthetic' / 'capsule_body' / 'occnerf' / 'seeded' / 'tpose' / '000000.png'
    assert png.exists()
    img = np.asarray(Image.open(png))
    assert img.shape == (32, 32, 3)
    assert np.abs(img.astype(int) - g['img.rgb'].astype(int)).max() <= 1
    assert (img != g['img.rgb']).mean() < 0.01


@pytest.mark.gpu
def test_run_py_allview_and_evaluate_entry_points(tmp_path):
    """run.py:188-244 on the synthetic source: `--type allview` writes the 23 views of the rig under allview_<frame_idx>/,
    `--type evaluate` renders the progress frames at iter_val = 1 (frames 4 and 15 skipped) and prints AVG PSNR against the
    teacher's targets."""
    from PIL import Image
    base = [sys.executable, os.path.join(ROOT, 'run.py'), '--cfg', os.path.join(ROOT, 'configs/occnerf/synthetic/occnerf.yaml')]
    env = {**os.environ, 'PYTHONPATH': ROOT}
    subprocess.check_call(base + ['--type', 'allview', 'render_size', '32', 'N_samples', '16'], cwd=str(tmp_path), env=env)
    folder = tmp_path / 'experiments' / 'occnerf' / 'synthetic' / 'capsule_body' / 'occnerf' / 'seeded' / 'allview_0'
    pngs = sorted(os.listdir(folder))
    assert pngs == [f'{i:06d}.png' for i in range(23)]
    views = [np.asarray(Image.open(folder / f)) for f in pngs]
    assert views[0].shape == (32, 32, 3) and len({v.tobytes() for v in views}) >= 20          # the camera really moves
    out = subprocess.run(base + ['--type', 'evaluate', 'render_size', '32', 'N_samples', '16', 'render_frames', '6'],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('AVG PSNR')]
    assert line, out.stdout[-2000:]
    psnr = float(line[-1].split()[-1])
    assert 5.0 < psnr < 80.0                                   # finite: the loaded and the teacher checkpoints differ


@pytest.mark.gpu
@pytest.mark.parametrize('bf16', ['False', 'True'])
def test_train_py_entry_point(tmp_path, bf16):
    """python train.py --cfg ... runs optimisation steps through the differentiable path (fp32 trunks, and bf16 trunks
    under torch.autocast: BASELINE configs[4]), the loss falls, and the checkpoint has the reference's layout and loads
    back with strict=True."""
    cmd = [sys.executable, os.path.join(ROOT, 'train.py'), '--cfg',
           os.path.join(ROOT, 'configs/occnerf/synthetic/occnerf.yaml'), 'render_size', '128', 'N_samples', '32',
           'train.maxiter', '12', 'train.log_interval', '1', 'patch.size', '16', 'patch.N_patches', '4',
           'train.bf16', bf16]
    out = subprocess.check_output(cmd, cwd=str(tmp_path), env={**os.environ, 'PYTHONPATH': ROOT}, text=True)
    losses = [float(line.split('loss')[1].split()[0]) for line in out.splitlines() if line.startswith('iter')]
    assert len(losses) >= 12 and all(np.isfinite(losses))
    assert np.mean(losses[-3:]) < np.mean(losses[:3])
    ckpt = torch.load(tmp_path / 'experiments' / 'occnerf' / 'synthetic' / 'capsule_body' / 'occnerf' / 'latest.tar',
                      map_location='cpu')
    assert set(ckpt) == {'iter', 'network', 'optimizer'} and ckpt['iter'] == 12
    from tests.gpu_util import build_network
    net, ctx = build_network(0, False, S=32, device='cpu')
    net.load_state_dict(ckpt['network'], strict=True)


@pytest.mark.gpu
def test_ray_order_kernel_walks_the_morton_curve():
    """occnerf_ray_order (three kernels + a radix sort) against the torch construction of occnerf_amd/rayorder.py on the rays
    of three cameras: a permutation; the torch Morton keys read along the HIP walk are sorted up to the few rays whose
    16-bit quantisation differs by rounding in the projection (< 1 % adjacent inversions, none by more than a cell);
    deterministic; accepts the strided direction columns of a rays8 array; R = 1 and R = 0."""
    from occnerf_amd import ops, synth
    from occnerf_amd.rayorder import _keys_fp32, ray_patch_order
    DEV = 'cuda:0'
    for img, orbit in ((64, 0), (200, 17), (512, 28)):
        frame = synth.make_frame(img_size=img, pose72=synth.seeded_pose(1), orbit_frame=orbit)
        d = torch.from_numpy(np.ascontiguousarray(frame['rays'][1])).to(DEV)
        R = d.shape[0]
        o = ops.ray_order(d)
        assert o.dtype == torch.int64 and o.shape == (R,)
        assert torch.equal(torch.sort(o).values, torch.arange(R, device=DEV))
        assert torch.equal(o, ops.ray_order(d)) and torch.equal(o, ray_patch_order(d))
        k = _keys_fp32(d)[o]
        inv = (k[1:] < k[:-1])
        assert float(inv.float().mean()) < 0.01, float(inv.float().mean())
        # the walk is as compact as torch's: mean distance between consecutive directions within 2 %
        ot = torch.argsort(_keys_fp32(d), stable=True)
        step = lambda order: float((d[order][1:] - d[order][:-1]).norm(dim=1).mean())      # noqa: E731
        assert step(o) <= 1.02 * step(ot)
        rays8 = torch.zeros(R, 8, device=DEV)
        rays8[:, 3:6] = d
        assert torch.equal(ops.ray_order(rays8[:, 3:6]), o)
    assert ops.ray_order(d[:1]).tolist() == [0] and ops.ray_order(d[:0]).numel() == 0


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['t32', 'f32', 'c64'])
def test_gen_rays(ops, tag):
    """Device ray generation against what the reference's camera_util produced (rays_cameras.npz)."""
    from occnerf_amd import rays as rays_mod
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'rays_cameras.npz'))
    img = int(g[f'{tag}.img'])
    K, E, lo, hi = g[f'{tag}.K'], g[f'{tag}.E'], g[f'{tag}.bbox_min'], g[f'{tag}.bbox_max']
    rays8, mask = ops.gen_rays(K, E, img, img, lo, hi, DEV)
    rays8, mask = rays8.cpu().numpy(), mask.cpu().numpy().astype(bool)
    want_mask = g[f'{tag}.mask']
    assert (mask != want_mask).sum() == 0
    # float32 cameras: numpy's sgemm may round the two 3-term dot products differently (<= 1 ulp of O(1))
    assert np.abs(rays8[:, 0:3] - g[f'{tag}.rays_o']).max() <= 1e-6
    assert np.abs(rays8[:, 3:6] - g[f'{tag}.rays_d']).max() <= 1e-6
    assert np.abs(rays8[want_mask, 6] - g[f'{tag}.near']).max() <= 2e-5
    assert np.abs(rays8[want_mask, 7] - g[f'{tag}.far']).max() <= 2e-5
    fr = rays_mod.frame_rays(K, E, img, img, lo, hi, DEV)
    R = int(want_mask.sum())
    assert fr['rays'].shape == (2, R, 3) and fr['near'].shape == (R, 1) and fr['far'].shape == (R, 1)
    assert np.abs(fr['rays'][1].cpu().numpy() - g[f'{tag}.rays_d'][want_mask]).max() <= 1e-6
    assert np.array_equal(fr['ray_mask'].cpu().numpy(), want_mask)
