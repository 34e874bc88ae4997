"""GPU: BASELINE.json's configs at their real sizes (C1 t-pose, C2 the benchmark frame, C3 sharded movement sequence,
C4 1024 x 1024 x 192 occlusion-aware), through size-independent properties and shortcut-on / shortcut-off bit identity.
"""
import os
import numpy as np
import pytest
import torch

from tests import util
from tests.gpu_util import (DEV, T, same, build_network, frame_to_device, per_frame_cpu, stagewise_oracle_render, _dev_model,
                            _clusters, stagewise_table, _torchrun)

pytestmark = pytest.mark.gpu


def test_config1_real_size(oracle):
    """BASELINE configs[0] at its real size: T-pose render, 128x128 image, 32 samples/ray, random-init weights --
    every ray against the full CPU oracle (1e-4, the BASELINE gate)."""
    from occnerf_amd import synth
    net, ctx = build_network(seed=0, amplify=False, S=32, non_rigid=False)
    frame = synth.make_frame(img_size=128, pose72=np.zeros(72, np.float32), orbit_frame=0)
    R = frame['rays'].shape[1]
    assert R > 4000
    with torch.no_grad():
        out = net(**frame_to_device(frame, DEV), iter_val=1e7)
    want = stagewise_oracle_render(None, ctx, frame=frame, S=32, non_rigid=False)
    for k in ('rgb', 'alpha', 'depth'):
        assert out[k].shape[0] == R
        assert np.abs(out[k].cpu().numpy() - want[k]).max() <= 1e-4, k


def test_full_size_properties(ops, oracle):
    """BASELINE.json configs[1] sizes (512x512 rays, 128 samples): size-independent checks."""
    from occnerf_amd import synth
    net, ctx = build_network(0, False, S=128, non_rigid=True)
    frame = synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28)
    data = frame_to_device(frame, DEV)
    with torch.no_grad():      # (scoped: a failing assert must not leave later autograd tests in no-grad mode)
        out = net(**data, iter_val=1e7)
        R = frame['rays'].shape[1]
        assert out['rgb'].shape == (R, 3) and out['alpha'].shape == (R,)
        rgb, acc = out['rgb'], out['alpha']
        assert torch.isfinite(rgb).all() and torch.isfinite(out['depth']).all()
        assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
        assert float(rgb.min()) >= -1e-6 and float(rgb.max()) <= 1.0 + 1e-5
        # determinism: same frame twice -> identical bits
        out2 = net(**data, iter_val=1e7)
        assert torch.equal(out2['rgb'], rgb) and torch.equal(out2['depth'], out['depth'])
        # ray sharding: rendering a slice of the rays gives the same pixels (no cross-ray coupling)
        lo, hi = R // 3, R // 3 + 4097
        part = dict(data)
        part['rays'], part['near'], part['far'] = data['rays'][:, lo:hi].contiguous(), data['near'][lo:hi], data['far'][lo:hi]
        outp = net(**part, iter_val=1e7)
        assert torch.equal(outp['rgb'], rgb[lo:hi]) and torch.equal(outp['alpha'], acc[lo:hi])
        # a random subset of rays against the full CPU oracle
        sel = np.sort(np.random.RandomState(0).choice(R, 96, replace=False))
        sub = dict(frame)
        sub['rays'], sub['near'], sub['far'] = frame['rays'][:, sel], frame['near'][sel], frame['far'][sel]
        want = stagewise_oracle_render(None, ctx, frame=sub, S=128, non_rigid=True)
    for k in ('rgb', 'alpha', 'depth'):
        assert np.abs(out[k].cpu().numpy()[sel] - want[k]).max() <= 1e-4, k


def test_skip_empty_samples_is_exact_at_bench_size(ops):
    """The same property on the frame the headline is quoted on (BASELINE configs[1]: 512x512 rays, 128 samples,
    random-init checkpoint, non-rigid on): skipping the dead quarter of the samples changes no output bit."""
    from occnerf_amd import synth
    net, ctx = build_network(seed=0, amplify=False, S=128, non_rigid=True)
    frame = synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28)
    data = frame_to_device(frame, DEV)
    outs = []
    for skip in (True, False):
        net.cfg.skip_empty_samples = skip
        with torch.no_grad():
            o = net(**data, iter_val=1e7)
        outs.append({k: o[k].clone() for k in ('rgb', 'alpha', 'depth')})
        if skip:
            live = int(net.last_live_count)
    net.cfg.skip_empty_samples = True
    R = frame['rays'].shape[1]
    assert 0.5 * R * 128 < live < 0.9 * R * 128, live          # a real fraction of the frame is dead
    for k in ('rgb', 'alpha', 'depth'):
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_knn_center_cache_is_exact_at_bench_size(ops):
    """The benchmark frame (BASELINE configs[1]) rendered with cfg.knn_center_cache on (the default, in the headline) and off:
    identical pixels."""
    from occnerf_amd import synth
    net, _ = build_network(seed=0, amplify=False, S=128, non_rigid=True)
    data = frame_to_device(synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28), DEV)
    outs = []
    for on in (True, False):
        net.cfg.knn_center_cache = on
        with torch.no_grad():
            o = net(**data, iter_val=1e7)
        outs.append(torch.cat([o['rgb'], o['alpha'][:, None], o['depth'][:, None]], 1))
    net.cfg.knn_center_cache = True
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('size,S,amplify', [(96, 64, True), (512, 128, False)])
def test_dedup_repeated_samples_is_exact(ops, size, S, amplify):
    """Evaluating each run of bitwise identical samples once (cfg.dedup_repeated_samples) changes no output bit -- on a
    small amplified-checkpoint frame and on the frame the headline is quoted on -- and does remove work there."""
    from occnerf_amd import synth
    net, ctx = build_network(seed=0, amplify=amplify, S=S, non_rigid=True)
    frame = synth.make_frame(img_size=size, pose72=synth.seeded_pose(1), orbit_frame=28)
    data = frame_to_device(frame, DEV)
    outs = []
    for dedup in (True, False):
        net.cfg.dedup_repeated_samples = dedup
        with torch.no_grad():
            o = net(**data, iter_val=1e7)
        outs.append({k: o[k].clone() for k in ('rgb', 'alpha', 'depth')})
        if dedup:
            live, heads_a, heads_b = int(net.last_live_count), int(net.last_head_counts[0]), int(net.last_head_counts[1])
    net.cfg.dedup_repeated_samples = True
    assert 0 < heads_b <= heads_a <= live
    if size == 512:
        assert heads_a < 0.6 * live and heads_b < 0.4 * live, (live, heads_a, heads_b)
        for glob, gpos in ((False, False), (True, True)):   # run-length only / global on both stages: the same pixels
            net.cfg.dedup_global, net.cfg.dedup_global_positions = glob, gpos
            with torch.no_grad():
                o = net(**data, iter_val=1e7)
            ha, hb = int(net.last_head_counts[0]), int(net.last_head_counts[1])
            assert (ha == heads_a and hb > heads_b) if not glob else (ha < heads_a and hb == heads_b), (ha, hb)
            for k in ('rgb', 'alpha', 'depth'):
                assert torch.equal(o[k], outs[1][k]), k
        net.cfg.dedup_global, net.cfg.dedup_global_positions = True, False
    for k in ('rgb', 'alpha', 'depth'):
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert float(outs[0]['alpha'].max()) > 0.05


def test_config4_shape_occlusion_aware_path(oracle):
    """BASELINE configs[3] shape: 1024x1024 camera, 192 samples/ray, non-uniform visibility counts
    (the learnt `point_counter` that makes the aggregation occlusion-aware), non-rigid on.  A ray
    subset against the full CPU oracle, and the multi-pass (memory-bounded) route against one pass."""
    from occnerf_amd import synth
    net, ctx = build_network(0, True, S=192, non_rigid=True)
    frame = synth.make_frame(img_size=1024, pose72=synth.seeded_pose(3), orbit_frame=11)
    R = frame['rays'].shape[1]
    sel = np.sort(np.random.RandomState(4).choice(R, 200, replace=False))
    sub = dict(frame)
    sub['rays'], sub['near'], sub['far'] = frame['rays'][:, sel], frame['near'][sel], frame['far'][sel]
    with torch.no_grad():
        out = net(**frame_to_device(sub, DEV), iter_val=1e7)
        net.cfg.max_samples_per_pass = 192 * 64          # 4 passes of 64 rays
        out2 = net(**frame_to_device(sub, DEV), iter_val=1e7)
    want = stagewise_oracle_render(None, ctx, frame=sub, S=192, non_rigid=True)
    for k in ('rgb', 'alpha', 'depth'):
        assert np.abs(out[k].cpu().numpy() - want[k]).max() <= 1e-3, k     # amplified checkpoint
        assert torch.equal(out[k], out2[k]), k
    assert float(out['alpha'].max()) > 0.05                                  # a non-trivial field


def test_config4_full_frame(oracle):
    """BASELINE configs[3] as written: one full 1024x1024 x 192-sample frame (734 K rays, 141 M samples; three passes of
    the memory-bounded route == the default single pass, bit for bit), non-rigid on, seeded visibility counts (the occlusion-aware aggregation): finite,
    deterministic, a 4 096-ray slice rendered alone is bit-identical, 96 rays against the full CPU oracle."""
    from occnerf_amd import synth
    net, ctx = build_network(seed=0, amplify=False, S=192, non_rigid=True)
    rng = np.random.RandomState(4)
    pc = ctx['point_base']
    cnt = np.where(pc[:, 2] > 0, 1.0, 1.0 + rng.poisson(50, pc.shape[0])).astype(np.float32)
    net.point_counter.data.copy_(torch.from_numpy(cnt).to(DEV))
    frame = synth.make_frame(img_size=1024, pose72=synth.seeded_pose(1), orbit_frame=28)
    data = frame_to_device(frame, DEV)
    R = frame['rays'].shape[1]
    net.cfg.max_samples_per_pass = 1 << 26                            # the memory-bounded route: 3 passes of <= 64 M samples
    rays_per_pass = int(net.cfg.max_samples_per_pass) // 192
    assert -(-R // rays_per_pass) >= 3
    with torch.no_grad():
        out = net(**data, iter_val=1e7)
        net.cfg.max_samples_per_pass = 1 << 28                        # the default: the whole frame in one pass (63 GiB)
        assert R * 192 <= net.cfg.max_samples_per_pass
        one = net(**data, iter_val=1e7)
        for k in ('rgb', 'alpha', 'depth'):
            assert torch.equal(one[k], out[k]), k
        del one
        net.cfg.max_samples_per_pass = 1 << 26
        assert all(bool(torch.isfinite(out[k]).all()) for k in ('rgb', 'alpha', 'depth'))
        out2 = net(**data, iter_val=1e7)
        for k in ('rgb', 'alpha', 'depth'):
            assert torch.equal(out[k], out2[k]), k
        lo = R // 2
        part = dict(data)
        part['rays'], part['near'], part['far'] = data['rays'][:, lo:lo + 4096].contiguous(), data['near'][lo:lo + 4096], \
            data['far'][lo:lo + 4096]
        outp = net(**part, iter_val=1e7)
        for k in ('rgb', 'alpha', 'depth'):
            assert torch.equal(outp[k], out[k][lo:lo + 4096]), k
    sel = np.sort(np.random.RandomState(1).choice(R, 96, replace=False))
    sub = dict(frame)
    sub['rays'], sub['near'], sub['far'] = frame['rays'][:, sel], frame['near'][sel], frame['far'][sel]
    octx = dict(ctx)
    octx['counter'] = cnt
    want = stagewise_oracle_render(None, octx, frame=sub, S=192, non_rigid=True)
    for k in ('rgb', 'alpha', 'depth'):
        assert np.abs(out[k].cpu().numpy()[sel] - want[k]).max() <= 1e-4, k
    assert float(out['alpha'].max()) > 0.05


def test_config3_movement_sequence(tmp_path, oracle):
    """BASELINE configs[2]: the movement sequence (frames of the pose walk seen by one camera, create_dataset.py:28-33,
    run.py:137-186) at 512x512 x 128 samples through `run.py --type movement` itself -- device-generated rays, named
    camera, sharded renderer with one frame of lag, device image assembly, PNG writer:
      * one process, and two ranks (torchrun; gloo on this one GPU) write byte-identical images;
      * the same frames rendered in this process reproduce those images byte for byte;
      * 96 rays of every frame against the full CPU oracle within the BASELINE gate of 1e-4."""
    import subprocess
    import sys
    from PIL import Image
    from occnerf_amd import synth
    from occnerf_amd.image import assemble_uint8_device
    from occnerf_amd.rays import frame_rays
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n_frames, size, spp = 4, 512, 128
    cli = ['--cfg', os.path.join(root, 'configs/occnerf/synthetic/occnerf.yaml'), '--type', 'movement', 'render_frames',
           str(n_frames), 'render_size', str(size), 'N_samples', str(spp)]
    one, two = tmp_path / 'one', tmp_path / 'two'
    one.mkdir()
    two.mkdir()
    env = {**os.environ, 'PYTHONPATH': root, 'HSA_ENABLE_IPC_MODE_LEGACY': '0'}
    subprocess.check_call([sys.executable, os.path.join(root, 'run.py')] + cli, cwd=str(one), env=env)
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    subprocess.check_call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr',
                           '127.0.0.1', '--master-port', str(port), os.path.join(root, 'run.py')] + cli, cwd=str(two),
                          env={**env, 'OCC_DIST_BACKEND': 'gloo', 'OCC_FORCE_DEVICE': '0'}, timeout=900)
    sub = os.path.join('experiments', 'occnerf', 'synthetic', 'capsule_body', 'occnerf', 'seeded', 'movement')
    imgs = []
    for t in range(n_frames):
        a = np.asarray(Image.open(one / sub / f'{t:06d}.png'))
        b = np.asarray(Image.open(two / sub / f'{t:06d}.png'))
        assert a.shape == (size, size, 3) and np.array_equal(a, b), t
        imgs.append(a)
    assert len({im.tobytes() for im in imgs}) >= 3                       # the body really moves

    net, ctx = build_network(seed=0, amplify=False, S=spp, non_rigid=True)
    for t in range(n_frames):
        f = synth.make_frame(img_size=size, pose72=synth.movement_pose(t, n_frames), orbit_frame=0,
                             orbit_period=n_frames, bgcolor=[255., 255., 255.], with_rays=False)
        fr = frame_rays(f['camera_K'], f['camera_E'], size, size, f['dst_bbox_min'], f['dst_bbox_max'], DEV)
        data = {k: T(f[k]) for k in ('dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec')}
        data.update(rays=fr['rays'], near=fr['near'], far=fr['far'], bgcolor=f['bgcolor'],
                    cnl_bbox_min_xyz=f['cnl_bbox_min_xyz'], cnl_bbox_scale_xyz=f['cnl_bbox_scale_xyz'])
        R = int(fr['rays'].shape[1])
        with torch.no_grad():
            out = net(**data, iter_val=1e7, ray_order_key=('movement', R))
        ray_index = torch.nonzero(fr['ray_mask']).squeeze(1)
        img, _ = assemble_uint8_device(size, size, ray_index, np.array([1., 1., 1.]), out['rgb'], out['alpha'],
                                       want_alpha=False)
        assert np.array_equal(img.cpu().numpy(), imgs[t]), t
        sel = np.sort(np.random.RandomState(t).choice(R, 96, replace=False))
        frame = dict(f)
        rays_h = fr['rays'].cpu().numpy()
        frame['rays'], frame['near'], frame['far'] = rays_h[:, sel], fr['near'].cpu().numpy()[sel], fr['far'].cpu().numpy()[sel]
        want = stagewise_oracle_render(None, ctx, frame=frame, S=spp, non_rigid=True)
        for k in ('rgb', 'alpha', 'depth'):
            err = np.abs(out[k].cpu().numpy()[sel] - want[k]).max()
            assert err <= 1e-4, (t, k, err)
        assert float(out['alpha'].max()) > 0.05


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs 2 GPUs on the node (config 3: rays sharded over RCCL)')
def test_two_rank_sharded_render_over_rccl():
    """BASELINE configs[2]: rays of one frame sharded over 2 ranks (child processes started by the launcher; RCCL
    gather, pipelined) == the frame rendered by one rank, bit for bit; and bench.py --gpus 2 runs and reports strong
    scaling with the world size RCCL formed."""
    got = _torchrun(['tools/sharded_check.py'], 2)
    assert got['world_size_formed'] == 2 and got['bit_identical'], got
    line = _torchrun(['bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-alt', '--no-cpu-baseline'], 2)
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['config']['world_size_formed'] == 2
    assert line['value'] > 0


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher starts its two ranks itself (a child
    torch.distributed.run created before the parent touches the GPU), relays the JSON line and exits with the child's code.
    (gloo dry run: both ranks on this one GPU.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(OCC_DIST_BACKEND='gloo', OCC_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    res = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
                          '--no-alt'], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['config']['world_size_formed'] == 2 and line['value'] > 0
    # a launcher that started the wrong number of ranks is an error message, not an AssertionError
    res = subprocess.run([sys.executable, 'bench.py', '--gpus', '2'], cwd=root, env=dict(env, WORLD_SIZE='1', RANK='0'),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and 'WORLD_SIZE=1' in res.stderr and 'AssertionError' not in res.stderr


def test_two_process_sharded_render_one_gpu():
    """The sharded renderer with the real network and TWO ranks on this one GPU (RCCL refuses two ranks per device, so
    the blocks travel through the host with gloo): shard plans, 256-ray Morton-block dealing, buffer slots, the one-frame
    lag of the pipelined gather and the un-permutation are the production code; three frames must be bit-identical to
    rank 0 rendering them alone."""
    got = _torchrun(['tools/sharded_check.py'], 2, extra_env={'OCC_DIST_BACKEND': 'gloo', 'OCC_FORCE_DEVICE': '0'})
    assert got['world_size_formed'] == 2 and got['backend'] == 'gloo'
    assert got['bit_identical'] and got['max_abs_diff'] == 0.0, got
    # and bench.py's N > 1 leg end to end (sharding, pipelined gather, max-over-ranks timing, weak_frames side figure)
    line = _torchrun(['bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline'], 2,
                     extra_env={'OCC_DIST_BACKEND': 'gloo', 'OCC_FORCE_DEVICE': '0'})
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['config']['world_size_formed'] == 2
    assert line['value'] > 0 and line['weak_frames']['value'] > 0


def test_two_process_sharded_movement_at_size():
    """configs[2] at 512x512 x 128: three movement frames with device-generated rays and a named camera, rays sharded
    over two ranks (this one GPU, gloo), bit-identical to one rank rendering them alone."""
    got = _torchrun(['tools/sharded_check.py', '--kind', 'movement', '--size', '512', '--spp', '128', '--frames', '3'], 2,
                    extra_env={'OCC_DIST_BACKEND': 'gloo', 'OCC_FORCE_DEVICE': '0'})
    assert got['world_size_formed'] == 2 and got['kind'] == 'movement' and got['size'] == 512 and got['spp'] == 128
    assert got['bit_identical'] and got['max_abs_diff'] == 0.0, got
    assert min(got['rays']) > 100000


def test_eight_process_sharded_render_one_gpu():
    """The node size the path is built for: EIGHT ranks (sharing this one GPU, gloo through the host) render three frames with the
    cost-aware shard plan, bit-identical to one rank; `bench.py --gpus 8` runs end to end and its per-rank live-sample counts are
    within 1 % of their mean (the balance the plan exists for)."""
    env = {'OCC_DIST_BACKEND': 'gloo', 'OCC_FORCE_DEVICE': '0'}
    got = _torchrun(['tools/sharded_check.py'], 8, extra_env=env)
    assert got['world_size_formed'] == 8 and got['bit_identical'] and got['max_abs_diff'] == 0.0, got
    line = _torchrun(['bench.py', '--gpus', '8', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-alt'], 8, extra_env=env)
    assert line['n_gpus'] == 8 and line['config']['world_size_formed'] == 8 and line['scaling'] == 'strong'
    rays, live = line['config']['per_rank_rays'], line['config']['per_rank_live_samples']
    assert sum(rays) == line['config']['rays_per_frame'] and len(live) == 8
    assert max(rays) / (sum(rays) / 8) <= 1.01 and max(live) / (sum(live) / 8) <= 1.01, (rays, live)
