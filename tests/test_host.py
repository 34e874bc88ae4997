"""CPU: host-side logic -- config, synthetic inputs, checkpoint surface, ray sharding (gloo)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_merge_and_cli(tmp_path):
    from occnerf_amd.config import make_cfg
    y = tmp_path / 'exp.yaml'
    y.write_text('N_samples: 64\ncanonical_mlp:\n  mlp_width: 256\nfreeview:\n  frame_idx: 128\n')
    cfg, args = make_cfg(['--cfg', str(y), '--type', 'freeview', 'chunk', '1024', 'bgcolor', '[255.,255.,255.]'])
    assert args.type == 'freeview' and cfg.N_samples == 64 and cfg.chunk == 1024
    assert cfg.freeview.frame_idx == 128 and cfg.bgcolor == [255., 255., 255.]
    assert cfg.logdir == os.path.join('experiments', 'occnerf', 'zju_mocap', 'p387', 'occnerf')
    assert cfg.non_rigid_motion_mlp.kick_in_iter == 100000 and cfg.pose_decoder.kick_in_iter == 2000000


def test_synthetic_body_and_frame():
    from occnerf_amd import geometry, synth
    smpl = synth.SyntheticSMPL()
    v, j = smpl(np.zeros(72), np.zeros(10))
    assert v.shape == (6890, 3) and j.shape == (24, 3) and smpl.faces.max() == 6889
    n = geometry.vertex_normals(v, smpl.faces)
    assert n.dtype == np.float64 and np.allclose(np.linalg.norm(n, axis=1), 1)
    fps = geometry.farthest_point_sampling(v, 1 / 16)
    assert len(fps) == 431 and len(set(fps.tolist())) == 431 and fps[0] == 0
    f = synth.make_frame(64)
    R = f['rays'].shape[1]
    assert f['ray_mask'].sum() == R and f['near'].shape == (R, 1) and np.all(f['far'] > f['near'])
    assert f['motion_weights_priors'].shape == (25, 32, 32, 32)
    assert np.allclose(f['motion_weights_priors'].sum(0), 1, atol=1e-5)


def test_network_state_dict_surface_on_cpu():
    """Key names/shapes are the reference's: the seeded checkpoint loads strict."""
    from tests.gpu_util import build_network
    net, ctx = build_network(0, False, S=32, device='cpu')
    assert list(net.state_dict().keys()) == list(ctx['sd'].keys())
    assert net.cnl_mlp.module.encoder.embeddings.shape == (7755336, 2)
    assert hasattr(net, 'mweight_vol_decoder') and net.point_cloud.shape == (6890, 3)
    frame = {'rays': torch.zeros(2, 4, 3)}
    with pytest.raises(RuntimeError, match='GPU'), torch.no_grad():       # no silent CPU fallback
        net(rays=frame['rays'], dst_Rs=torch.zeros(24, 3, 3), dst_Ts=torch.zeros(24, 3),
            cnl_gtfms=torch.zeros(24, 4, 4), motion_weights_priors=torch.ones(25, 32, 32, 32),
            dst_posevec=torch.zeros(69), near=torch.zeros(4, 1), far=torch.ones(4, 1))


def test_shard_bounds():
    from occnerf_amd.parallel import shard_bounds
    for n, w in [(262144, 8), (183784, 8), (7, 8), (0, 2), (1000, 3)]:
        b = shard_bounds(n, w)
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


def test_shard_indices_partition():
    from occnerf_amd.parallel import shard_indices, shard_sizes
    for n, w, c in [(183784, 8, 256), (1001, 2, 64), (7, 8, 4), (0, 2, 16), (5000, 3, 4096), (734200, 8, 256)]:
        parts = [shard_indices(n, r, w, c) for r in range(w)]
        allidx = torch.cat(parts).sort().values
        assert torch.equal(allidx, torch.arange(n))
        sizes = [p.numel() for p in parts]
        assert sizes == shard_sizes(n, w, c)
        assert max(sizes) - min(sizes) <= c


def test_shard_plan_is_balanced():
    """4 096-ray chunks capped 8-GPU strong scaling at 0.935 before the first kernel ran.  With the
    default block the largest share is within 1 % of the mean at every N for the benchmark frame (183 784 rays) and a
    config-4 frame (734 200 rays), and a block stays a multiple of the 64-ray kNN tile."""
    from occnerf_amd.parallel import BLOCK, shard_sizes
    assert BLOCK % 64 == 0 and BLOCK <= 512
    for R in (183784, 734200, 262144):
        for N in (2, 4, 8):
            sizes = shard_sizes(R, N)
            assert sum(sizes) == R
            assert max(sizes) / (R / N) <= 1.01, (R, N, sizes)


def test_morton_plan_matches_positions():
    """The shard plan built from the Morton walk: every ray exactly once, rank r's rays are the walk's blocks r, r+N, ...
    and the un-permutation sends each ray's row of the receive buffer back to the caller's index."""
    from occnerf_amd.parallel import ShardedRenderer
    from occnerf_amd.rayorder import ray_patch_order
    g = torch.Generator().manual_seed(1)
    R, W, B = 3000, 4, 64
    rays = torch.randn(2, R, 3, generator=g)
    rays[1] += torch.tensor([0., 0., 8.])
    order = ray_patch_order(rays[1])
    assert torch.equal(order.sort().values, torch.arange(R))
    seen = []
    for rank in range(W):
        r = ShardedRenderer(None, 'cpu', block=B, single=True)
        r.world, r.rank, r.collective, r.verify_plan = W, rank, True, False                                    # plan arithmetic only: no process group needed
        plan = r._build_plan({'rays': rays})
        mine = plan['mine']['cpu']
        want = torch.cat([order[b * B:(b + 1) * B] for b in range(rank, -(-R // B), W)])
        assert torch.equal(mine, want) and mine.numel() == plan['sizes'][rank]
        seen.append(mine)
        if rank == 0:
            unperm, width = plan['unpermute'], plan['width']
    assert torch.equal(torch.cat(seen).sort().values, torch.arange(R))
    recv = torch.full((W * width,), -1, dtype=torch.long)
    for rank, mine in enumerate(seen):
        recv[rank * width:rank * width + mine.numel()] = mine         # each rank sends "its ray ids"
    assert torch.equal(recv[unperm], torch.arange(R))


def test_cost_aware_plan_balances_live_samples():
    """When the network can estimate what a ray costs (Network.live_samples_per_ray: live samples on a probe of the
    rays), blocks are dealt in descending cost order, serpentine: every ray still exactly once, ray counts within one
    block, and the cost per rank far closer than the static deal on a frame whose cost is concentrated in one region."""
    from occnerf_amd.parallel import ShardedRenderer
    g = torch.Generator().manual_seed(2)
    R, W, B = 40000 + 77, 8, 256
    rays = torch.randn(2, R, 3, generator=g) * 0.2
    rays[1] += torch.tensor([0., 0., 4.])
    true_cost = (100 * torch.exp(-((rays[1, :, 0] - 0.3) ** 2 + rays[1, :, 1] ** 2) / 0.05)).long() + 1   # a hot blob

    class CostNet:
        def live_samples_per_ray(self, rays, near, far, **_):
            return (100 * torch.exp(-((rays[1, :, 0] - 0.3) ** 2 + rays[1, :, 1] ** 2) / 0.05)).long() + 1
    data = {'rays': rays, 'near': torch.zeros(R, 1), 'far': torch.ones(R, 1)}
    result = {}
    for balance in (False, True):
        parts, sums = [], []
        for rank in range(W):
            r = ShardedRenderer(CostNet(), 'cpu', block=B, single=True, balance=balance)
            r.world, r.rank, r.collective, r.verify_plan = W, rank, True, False
            plan = r._build_plan(data)
            assert plan['cost_aware'] == balance
            mine = plan['mine']['cpu']
            assert mine.numel() == plan['sizes'][rank]
            parts.append(mine)
            sums.append(float(true_cost[mine].sum()))
            if rank == 0:
                unperm, width, sizes = plan['unpermute'], plan['width'], plan['sizes']
        assert torch.equal(torch.cat(parts).sort().values, torch.arange(R))
        assert max(sizes) - min(sizes) <= B
        recv = torch.full((W * width,), -1, dtype=torch.long)
        for rank, mine in enumerate(parts):
            recv[rank * width:rank * width + mine.numel()] = mine
        assert torch.equal(recv[unperm], torch.arange(R))
        result[balance] = max(sums) / (sum(sums) / W)
    assert result[True] <= 1.02 < result[False], result


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, n_rays, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from occnerf_amd.parallel import render_frame_sharded, shard_bounds

    class FakeNet:                                       # renders rgb = f(ray) per ray, no coupling
        def __call__(self, rays, near, far, iter_val=0, **_):
            o = rays[0]
            return {'rgb': o * 2.0, 'alpha': near[:, 0] + 1.0, 'depth': far[:, 0] * 3.0}

        def live_samples_per_ray(self, rays, near, far, **_):      # cost estimate -> the cost-aware (serpentine) plan
            return (rays[1, :, 0] * 50).long() + 1
    g = torch.Generator().manual_seed(0)
    data = {'rays': torch.rand(2, n_rays, 3, generator=g), 'near': torch.rand(n_rays, 1, generator=g),
            'far': torch.rand(n_rays, 1, generator=g)}
    def check(out, d):
        return (torch.equal(out['rgb'], d['rays'][0] * 2.0) and torch.equal(out['alpha'], d['near'][:, 0] + 1.0)
                and torch.equal(out['depth'], d['far'][:, 0] * 3.0))
    out = render_frame_sharded(FakeNet(), data, chunk=96)
    ok = check(out, data) if rank == 0 else out is None
    # pipelined: frame t's gather is waited for after frame t+1 has been submitted (two buffer slots, reused)
    from occnerf_amd.parallel import ShardedRenderer
    frames = []
    for t in range(6):                                    # frames of a sequence differ in ray count
        keep = n_rays - 7 * t
        f = {'rays': data['rays'][:, :keep] * (1.0 + 0.25 * t), 'near': data['near'][:keep] + t,
             'far': data['far'][:keep] * 2.0}
        frames.append((f, ('cam', keep)) if t % 2 else f)  # named camera (cached plan) and unnamed (plan per frame)
    frames.append(frames[1])                              # a cached plan is used again
    r = ShardedRenderer(FakeNet(), 'cpu', chunk=96)
    assert r.formed_world_size() == world
    outs = list(r.render_frames(frames))
    assert len(outs) == 7
    # results are the frames' own tensors, not views of the two slot buffers
    if rank == 0:
        assert len({o['packed'].data_ptr() for o in outs}) == len(outs)
    frames = [f[0] if isinstance(f, tuple) else f for f in frames]
    if n_rays == 300 and world == 8:
        assert r.last_shard_rays == (0 if rank >= 4 else r.last_shard_rays) and (rank < 4) == (r.last_shard_rays > 0)
    for o, d in zip(outs, frames):
        ok = ok and (check(o, d) if rank == 0 else o is None)
    if rank == 0:
        q.put(bool(ok))
    else:
        assert ok
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_rays,world', [(1001, 2), (64, 2), (5000, 8), (300, 8)])
def test_ray_sharding_gather_gloo(n_rays, world):
    """world 2 and world 8 (the node size the path is built for); 300 rays in blocks of 96 leave ranks 4..7 without a
    single ray: they skip the render and still join the gather."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, n_rays, q)) for r in range(world)]
    for p in procs:
        p.start()
    assert q.get(timeout=120) is True
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0



class _FakeNet:                                           # renders rgb = f(ray) per ray, no coupling
    def __init__(self):
        self.keys = []

    def __call__(self, rays, near, far, iter_val=0, ray_order_key=None, **_):
        self.keys.append(ray_order_key)
        return {'rgb': rays[0] * 2.0, 'alpha': near[:, 0] + 1.0, 'depth': far[:, 0] * 3.0}

    def live_samples_per_ray(self, rays, near, far, **_):
        return (rays[1, :, 0] * 50).long() + 1


def _plan_worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from occnerf_amd.parallel import ShardedRenderer
    g = torch.Generator().manual_seed(3)
    n = 1500
    data = {'rays': torch.rand(2, n, 3, generator=g), 'near': torch.rand(n, 1, generator=g), 'far': torch.rand(n, 1, generator=g)}
    ok = True
    if mode == 'force1':
        # a process group of ONE rank takes the N > 1 branch: plan, padded buffers, dist.gather, work.wait(), un-permutation
        r = ShardedRenderer(_FakeNet(), 'cpu', chunk=96, force_collective=True)
        assert r.collective and r.world == 1 and not ShardedRenderer(_FakeNet(), 'cpu', chunk=96).collective
        frames = [({k: v[..., :n - 11 * t, :] if k == 'rays' else v[:n - 11 * t] for k, v in data.items()},
                   ('cam', t % 2)) for t in range(5)]
        outs = list(r.render_frames(frames))
        for o, (d, _) in zip(outs, frames):
            ok = ok and torch.equal(o['rgb'], d['rays'][0] * 2.0) and torch.equal(o['alpha'], d['near'][:, 0] + 1.0) \
                and torch.equal(o['depth'], d['far'][:, 0] * 3.0)
        ok = ok and r.gathers_issued == 5 and r.plans_verified == 5       # (five different ray counts: five plans)
        # the network caches the Morton order of the rays it is handed per ray_order_key: a single-process renderer and a
        # collective one hand the same camera's rays over in different orders, so their keys must differ (round 4: they
        # collided at world 1 and the collective render ran with the other renderer's permutation -- right pixels, no locality)
        alone = ShardedRenderer(_FakeNet(), 'cpu', chunk=96, single=True)
        alone.finish(alone.submit(frames[0][0], ray_order_key=frames[0][1]))
        ok = ok and alone.net.keys[0] != r.net.keys[0] and alone.net.keys[0][0] == r.net.keys[0][0] == frames[0][1]
    else:
        mine = dict(data)
        if rank == 1 and mode == 'one_ray_less':
            mine = {'rays': data['rays'][:, :-1], 'near': data['near'][:-1], 'far': data['far'][:-1]}
        if rank == 1 and mode == 'one_ray_moved':
            mine['rays'] = data['rays'].clone()
            mine['rays'][1, 700] = torch.tensor([5., -5., 0.1])
        if rank == 1 and mode == 'costs_differ':
            class Other(_FakeNet):
                def live_samples_per_ray(self, rays, near, far, **_):
                    return (rays[1, :, 1] * 50).long() + 1
            net = Other()
        else:
            net = _FakeNet()
        r = ShardedRenderer(net, 'cpu', chunk=96)
        try:
            r.finish(r.submit(mine))
            ok = mode == 'same'
        except RuntimeError as e:
            want = {'one_ray_less': 'ray count', 'one_ray_moved': 'Morton walk', 'costs_differ': 'cost order'}[mode]
            ok = want in str(e) and 'differs from rank' in str(e)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('mode,world', [('force1', 1), ('same', 2), ('one_ray_less', 2), ('one_ray_moved', 2),
                                        ('costs_differ', 2)])
def test_forced_collective_and_plan_checksum_gloo(mode, world):
    """(a) with force_collective a ONE-rank process group runs the real gather branch (what a single-GPU box
    can execute of the RCCL path); (c) every rank all-gathers a checksum of each new shard plan and `finish` raises on ALL
    ranks when they disagree -- a frame that differs by one ray, a ray that sits elsewhere in the Morton walk, a
    different cost estimate -- instead of assembling a wrong image."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    assert got == {r: True for r in range(world)}, (mode, got)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0


@pytest.mark.parametrize('world', [2, 4, 8])
def test_emulated_ranks_cover_the_frame(world):
    """ShardedRenderer(emulate=(N, k)) -- what bench.py's `predicted_scaling` leg times while no multi-GPU node exists: every
    rank k of N emulated in this process renders exactly the share the N-rank plan gives it (the union is the frame, each ray
    once, shares balanced), writes it to ITS slot of the receive buffer, and rank 0's un-permutation puts those rays where the
    caller's order wants them; ranks k > 0 return None like real ones.  Pipelined over frames with two camera names."""
    from occnerf_amd.parallel import ShardedRenderer
    g = torch.Generator().manual_seed(5)
    n = 3000
    data = {'rays': torch.rand(2, n, 3, generator=g), 'near': torch.rand(n, 1, generator=g), 'far': torch.rand(n, 1, generator=g)}
    want = _FakeNet()(**data)
    seen = torch.zeros(n, dtype=torch.int32)
    sizes = []
    for k in range(world):
        r = ShardedRenderer(_FakeNet(), 'cpu', chunk=96, emulate=(world, k))
        assert r.collective and r.world == world and r.rank == k and not r.verify_plan
        outs = list(r.render_frames([(data, 'a'), (data, 'b'), (data, 'a')]))
        plan = r._get_plan(data, 'a')
        mine = plan['mine']['cpu']
        assert int(seen[mine].sum()) == 0
        seen[mine] += 1
        sizes.append(plan['sizes'][k])
        assert plan['sizes'][k] == mine.numel()
        if k == 0:
            for o in outs:      # rank 0 assembled the frame: its own rays are in place (the other slots are empty here)
                assert torch.equal(o['rgb'][mine], want['rgb'][mine]) and torch.equal(o['alpha'][mine], want['alpha'][mine])
                assert torch.equal(o['depth'][mine], want['depth'][mine])
        else:
            assert outs == [None, None, None]
            full = r._bufs[plan['width']]['recv'][1].view(world, plan['width'], 5)      # rank k's block sits in slot k
            blk = full[k, :mine.numel()]
            assert torch.equal(blk[:, :3], want['rgb'][mine]) and torch.equal(blk[:, 4], want['depth'][mine])
    assert bool((seen == 1).all())
    assert max(sizes) <= 1.1 * (n / world) + 96
    with pytest.raises(RuntimeError):
        ShardedRenderer(_FakeNet(), 'cpu', emulate=(4, 4))


def test_patch_ray_selection_is_the_reference_batch_shape():
    """The training batch of train.py / bench.py's `train` leg: n_patches random size x size pixel patches (default.yaml `patch`),
    as indices into the frame's bbox-hitting ray list.  full=True keeps only patches that lie wholly on such pixels, so the batch
    is exactly n_patches x size^2 rays; every patch is a size x size block of neighbouring pixels."""
    from occnerf_amd import synth
    from occnerf_amd.seeded import patch_ray_selection
    frame = synth.make_frame(img_size=128, pose72=synth.seeded_pose(1), orbit_frame=28)
    pix_of_ray = np.nonzero(np.asarray(frame['ray_mask']).reshape(-1))[0]
    sel = patch_ray_selection(frame, np.random.RandomState(3), n_patches=4, size=8, full=True)
    assert sel.shape == (4 * 64,) and sel.min() >= 0 and sel.max() < frame['rays'].shape[1]
    for p in range(4):
        pix = pix_of_ray[sel[p * 64:(p + 1) * 64]]
        ys, xs = pix // 128, pix % 128
        assert ys.max() - ys.min() == 7 and xs.max() - xs.min() == 7 and len(np.unique(pix)) == 64
    loose = patch_ray_selection(frame, np.random.RandomState(3), n_patches=4, size=8, full=False)
    assert 4 * 32 < loose.size <= 4 * 64
    again = patch_ray_selection(frame, np.random.RandomState(3), n_patches=4, size=8, full=True)
    assert np.array_equal(sel, again)

