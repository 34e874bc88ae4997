"""Multi-GPU: rays shard, one process per GPU, one gather per frame.

Rays are independent (the only cross-sample dependency is the scan inside one ray), so a frame's
`ray_mask`-compacted ray list is dealt to the ranks in chunks of 4 096 consecutive rays; every rank holds the
full (61 MiB) model, renders its rays with no data-path collective, and the `[R/N, 5]` (rgb, alpha, depth)
blocks are gathered on rank 0 over RCCL/xGMI (<= 5.2 MB per 512^2 frame: latency-bound, one collective).
This replaces the reference's nn.DataParallel over *samples* with its per-call weight broadcast
(network.py:68-72,142-146); it does not mirror it.

`ShardedRenderer` keeps everything that does not change from frame to frame -- the shard index lists, the
un-permutation that puts the gathered blocks back into the caller's ray order, the padded send / receive
buffers (two slots) -- and issues the gather asynchronously, so that frame t's gather runs under frame t+1's
kernels (`render_frames`).  A rank whose shard is empty (fewer rays than ranks x chunk) skips the render and
still takes part in the gather.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rays, world_size):
    """[lo, hi) of every rank's contiguous block; sizes differ by at most one."""
    base, rem = divmod(int(n_rays), int(world_size))
    bounds, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < rem else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def shard_indices(n_rays, rank, world_size, chunk=4096):
    """Ray indices of `rank` when rays are dealt in chunks of `chunk` consecutive rays (chunk c -> rank
    c % world_size).  Contiguous blocks of a pixel-ordered ray list are horizontal image bands with very
    different amounts of body in them; dealing chunks balances the work, and a chunk is still a compact
    set of pixels for the kNN tiles.  Every rank gets the same number of rays +- one chunk."""
    idx = torch.arange(int(n_rays))
    return idx[(idx // int(chunk)) % int(world_size) == int(rank)]


def shard_frame(data, rank, world_size):
    """Slice the per-ray entries of a frame dict (rays[2,R,3], near/far[R,1]) for `rank`."""
    R = data['rays'].shape[1]
    lo, hi = shard_bounds(R, world_size)[rank]
    out = dict(data)
    out['rays'] = data['rays'][:, lo:hi]
    out['near'], out['far'] = data['near'][lo:hi], data['far'][lo:hi]
    return out, (lo, hi)


def gather_rays(block, n_rays, dst=0, group=None):
    """Gather per-rank `[r_i, C]` blocks into `[n_rays, C]` on rank `dst` (None elsewhere).
    Blocks are padded to the largest shard so one fixed-size gather suffices."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return block
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = shard_bounds(n_rays, world)
    width = max(hi - lo for lo, hi in bounds)
    padded = block.new_zeros((width,) + tuple(block.shape[1:]))
    padded[:block.shape[0]] = block
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:hi - lo] for b, (lo, hi) in zip(bufs, bounds)], 0)


class _Pending:
    """One frame in flight: the gather's work handle, its plan (buffers) and the slot used."""
    __slots__ = ('work', 'slot', 'n_rays', 'plan')

    def __init__(self, work, slot, n_rays, plan):
        self.work, self.slot, self.n_rays, self.plan = work, slot, n_rays, plan


class ShardedRenderer:
    """Renders frames with their rays sharded over the ranks of `group` (see the module docstring)."""

    def __init__(self, net, device, group=None, chunk=4096, channels=5, single=False):
        """single: ignore the process group, this process renders whole frames by itself."""
        self.net, self.device, self.group, self.chunk, self.channels = net, torch.device(device), group, int(chunk), channels
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() and not single else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # gloo has no gather on device tensors: with that backend (tests: several processes sharing one GPU) the blocks
        # are exchanged through host buffers; with nccl (= RCCL) they stay on the device
        self.host_exchange = self.world > 1 and self.device.type == 'cuda' and dist.get_backend(group) == 'gloo'
        self._plans = {}          # rays per frame -> shard plan with its buffers (frames in flight keep theirs)
        self._turn = 0

    def formed_world_size(self):
        """World size the process group actually formed (what bench.py prints)."""
        return self.world

    def _get_plan(self, R):
        hit = self._plans.get(R)
        if hit is not None:
            return hit
        shards = [shard_indices(R, r, self.world, self.chunk) for r in range(self.world)]
        sizes = [int(s.numel()) for s in shards]
        width = max(max(sizes), 1)
        plan = {'R': R, 'mine_cpu': shards[self.rank], 'mine_dev': shards[self.rank].to(self.device), 'sizes': sizes,
                'width': width,
                'send': [torch.zeros(width, self.channels, device=self.device) for _ in range(2)]}
        if self.host_exchange:
            plan['send_host'] = [torch.zeros(width, self.channels).pin_memory() for _ in range(2)]
            if self.rank == 0:
                plan['recv_host'] = [torch.empty(self.world * width, self.channels).pin_memory() for _ in range(2)]
        if self.rank == 0 and self.world > 1:
            # position in the concatenated [world * width] receive buffer of every ray of the frame
            src = torch.empty(R, dtype=torch.long)
            for r, s in enumerate(shards):
                src[s] = r * width + torch.arange(sizes[r])
            plan['unpermute'] = src.to(self.device)
            plan['recv'] = [torch.empty(self.world * width, self.channels, device=self.device) for _ in range(2)]
        if len(self._plans) >= 4:                                 # a sequence's frames differ in ray count: keep a few plans
            self._plans.pop(next(iter(self._plans)))
        self._plans[R] = plan
        return plan

    def submit(self, data, iter_val=1e7, **net_kwargs):
        """Render this rank's share of the frame `data` and start the gather of (rgb, alpha, depth).  Per-ray entries
        (rays[2,R,3], near/far[R,1]) may live on the host (only the shard is then copied to the device) or on the
        device; every rank passes the same frame."""
        R = int(data['rays'].shape[1])
        plan = self._get_plan(R)
        slot = self._turn = self._turn ^ 1
        mine = plan['mine_dev'] if data['rays'].is_cuda else plan['mine_cpu']
        n_mine = plan['sizes'][self.rank]
        send = plan['send'][slot]
        if n_mine:
            local = dict(data)
            if n_mine == R:                                      # the whole frame is this rank's (one rank)
                sub = (data['rays'], data['near'], data['far'])
            elif data['rays'].is_cuda:
                sub = (data['rays'][:, mine], data['near'][mine], data['far'][mine])
            else:                                                # host frame: gather the shard into PINNED staging buffers, so
                st = plan.setdefault('stage', [None, None])      # that the copies below really are asynchronous
                if st[slot] is None:
                    gpu = self.device.type == 'cuda'
                    bufs = [torch.empty(2, n_mine, 3), torch.empty(n_mine, 1), torch.empty(n_mine, 1)]
                    st[slot] = tuple(b.pin_memory() if gpu else b for b in bufs) + (torch.cuda.Event() if gpu else None,)
                rs, ns, fs, ev = st[slot]
                if ev is not None:
                    ev.synchronize()                             # the copy issued from this slot two frames ago has left it
                torch.index_select(data['rays'], 1, mine, out=rs)
                torch.index_select(data['near'].reshape(-1, 1), 0, mine, out=ns)
                torch.index_select(data['far'].reshape(-1, 1), 0, mine, out=fs)
                sub = (rs, ns, fs)
            local['rays'] = sub[0].to(self.device, non_blocking=True)
            local['near'] = sub[1].to(self.device, non_blocking=True)
            local['far'] = sub[2].to(self.device, non_blocking=True)
            if not data['rays'].is_cuda and n_mine != R and plan['stage'][slot][3] is not None:
                plan['stage'][slot][3].record(torch.cuda.current_stream(self.device))
            for k, v in data.items():
                if k not in ('rays', 'near', 'far') and torch.is_tensor(v) and not v.is_cuda and v.numel() > 3:
                    local[k] = v.to(self.device, non_blocking=True)
            out = self.net(**local, iter_val=iter_val, **net_kwargs)
            send[:n_mine, :3] = out['rgb']
            send[:n_mine, 3] = out['alpha']
            send[:n_mine, 4] = out['depth']
        if self.world == 1:
            return _Pending(None, slot, R, plan)
        if self.host_exchange:
            send = plan['send_host'][slot].copy_(send)               # (synchronous: the gloo path is a test vehicle)
            rbuf = plan['recv_host'][slot] if self.rank == 0 else None
        else:
            rbuf = plan['recv'][slot] if self.rank == 0 else None
        recv = list(rbuf.view(self.world, plan['width'], self.channels).unbind(0)) if self.rank == 0 else None
        work = dist.gather(send, recv, dst=0, group=self.group, async_op=True)
        return _Pending(work, slot, R, plan)

    def finish(self, pending):
        """Wait for a frame's gather; -> {'rgb','alpha','depth'} in the caller's ray order on rank 0, None elsewhere."""
        plan = pending.plan
        if self.world == 1:
            full = plan['send'][pending.slot][:pending.n_rays]
        else:
            pending.work.wait()
            if self.rank != 0:
                return None
            if self.host_exchange:
                plan['recv'][pending.slot].copy_(plan['recv_host'][pending.slot], non_blocking=True)
            full = plan['recv'][pending.slot].index_select(0, plan['unpermute'])
        return {'rgb': full[:, :3], 'alpha': full[:, 3], 'depth': full[:, 4], 'packed': full}     # packed: contiguous [R,5]

    def render_frames(self, frames, iter_val=1e7, **net_kwargs):
        """Generator over `frames`: frame t's gather overlaps frame t+1's kernels (one frame of lag)."""
        prev = None
        for data in frames:
            cur = self.submit(data, iter_val=iter_val, **net_kwargs)
            if prev is not None:
                yield self.finish(prev)
            prev = cur
        if prev is not None:
            yield self.finish(prev)


_renderers = {}


def render_frame_sharded(net, data, iter_val=1e7, group=None, chunk=4096):
    """One frame, synchronously: this rank's share rendered, (rgb, alpha, depth) gathered on rank 0 -- the path's
    only collective.  Returns the full-frame dict in the caller's ray order on rank 0 and None on the other ranks."""
    dev = data['rays'].device
    if dev.type != 'cuda' and hasattr(net, 'point_base'):      # host frame: the shard is copied to the model's device
        dev = net.point_base.device
    key = (id(net), str(dev), id(group), int(chunk))
    r = _renderers.get(key)
    if r is None:
        r = _renderers[key] = ShardedRenderer(net, dev, group=group, chunk=chunk)
    return r.finish(r.submit(data, iter_val=iter_val))
