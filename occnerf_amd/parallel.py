"""Multi-GPU: rays shard, one process per GPU, one gather per frame.

Rays are independent (the only cross-sample dependency is the scan inside one ray), so a frame's
`ray_mask`-compacted ray list is dealt to the ranks in small blocks; every rank holds the full (61 MiB) model,
renders its rays with no data-path collective, and the `[R/N, 5]` (rgb, alpha, depth) blocks are gathered on rank 0
over RCCL/xGMI (<= 5.2 MB per 512^2 frame: latency-bound, one collective).  This replaces the reference's
nn.DataParallel over *samples* with its per-call weight broadcast (network.py:68-72,142-146); it does not mirror it.

**Which rays go where.**  The frame's rays are first walked along the renderer's 2-D Morton curve
(`rayorder.ray_patch_order`, the order `Network` renders in anyway), and that walk is dealt to the ranks in blocks of
256 consecutive rays (a ~16x16 pixel patch).  The shares differ by at most one block -- max/mean rays per rank <= 1.003 at
N = 8 for the 183 784-ray benchmark frame (the 4 096-ray chunks of round 2: 1.07) -- and a block is a multiple of the 64-ray
kNN tile and stays a compact pixel patch, so a rank's kernels see the same locality as a single GPU's.  WHICH blocks a rank
gets is decided by cost (`_block_costs`): a quarter of the samples are dead, unevenly over the image, and the static deal
(block b to rank b % N) left the live samples per rank 4.7 % apart at N = 8; the blocks are therefore sorted by the live
samples on every 16th of their rays and dealt serpentine, which brings the ranks within 0.1 % of each other
(tools/shard_balance.py).

The plan is computed by every rank for itself and must come out identical on all of them: it is a pure function of the frame
and the model (Morton keys, stable sorts, the deterministic sampler/warp kernel on identical devices).  No collective
distributes it, but one CHECKS it: whenever a plan is built, every rank all-gathers a 3 x 64-bit checksum of it (the ray
count and per-rank sizes, the Morton walk, the cost order of the blocks) and `finish()` of the first frame rendered with
the plan raises if any rank disagrees -- a mixed-firmware node or a host frame that differs by one ray is an error, not a
silently wrong image.  The exchange is asynchronous (device all-gather + pinned copy; `finish` waits on an event that was
recorded before the frame's kernels), one per NEW plan, not per frame of a named camera.  Ranks on dissimilar GPUs should
construct the renderer with `balance=False` (the static deal depends on the rays only).

`ShardedRenderer` keeps what does not change from frame to frame -- the shard index lists and the un-permutation
into the caller's ray order (per `ray_order_key`: a sequence shot by one camera names it, as `Network.forward`'s
Morton cache does), padded send / receive buffers in two slots -- and issues the gather asynchronously, so that frame
t's gather runs under frame t+1's kernels (`render_frames`).  A rank whose shard is empty (fewer blocks than ranks)
skips the render and still takes part in the gather.
"""
import os

import torch
import torch.distributed as dist

from .rayorder import ray_patch_order

BLOCK = 256          # rays per dealt block: 4 kNN tiles, a ~16x16 pixel patch
PROBE_STRIDE = 16     # the cost of a block is estimated from every 16th ray of the walk
WIDTH_QUANTUM = 4096  # buffers are sized in multiples of this many rays so that frames of a sequence share them


def shard_bounds(n_rays, world_size):
    """[lo, hi) of every rank's contiguous block; sizes differ by at most one."""
    base, rem = divmod(int(n_rays), int(world_size))
    bounds, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < rem else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def shard_sizes(n_rays, world_size, block=BLOCK):
    """Rays per rank when `n_rays` positions are dealt in blocks of `block` (block b -> rank b % world_size)."""
    n_rays, world_size, block = int(n_rays), int(world_size), int(block)
    full, tail = divmod(n_rays, block)
    sizes = [(full // world_size + (1 if r < full % world_size else 0)) * block for r in range(world_size)]
    if tail:
        sizes[full % world_size] += tail
    return sizes


def shard_positions(n_rays, rank, world_size, block=BLOCK, device=None):
    """Positions (in the dealt walk) of `rank`'s rays, ascending; pure arithmetic, no data-dependent shapes."""
    size = shard_sizes(n_rays, world_size, block)[int(rank)]
    j = torch.arange(size, device=device)
    return (torch.div(j, block, rounding_mode='floor') * world_size + int(rank)) * block + j % block


def shard_indices(n_rays, rank, world_size, chunk=BLOCK):
    """Ray indices of `rank` when the caller's own order is dealt (no Morton walk): chunk c -> rank c % world_size."""
    return shard_positions(n_rays, rank, world_size, chunk)


def shard_frame(data, rank, world_size):
    """Slice the per-ray entries of a frame dict (rays[2,R,3], near/far[R,1]) for `rank` (contiguous blocks)."""
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
    """One frame in flight: the gather's work handle, its plan and buffers and the slot used."""
    __slots__ = ('work', 'slot', 'n_rays', 'plan', 'bufs')

    def __init__(self, work, slot, n_rays, plan, bufs):
        self.work, self.slot, self.n_rays, self.plan, self.bufs = work, slot, n_rays, plan, bufs


class ShardedRenderer:
    """Renders frames with their rays sharded over the ranks of `group` (see the module docstring)."""

    def __init__(self, net, device, group=None, block=BLOCK, channels=5, single=False, morton=True, chunk=None,
                 balance=True, force_collective=None, verify_plan=True, emulate=None):
        """single: ignore the process group, this process renders whole frames by itself.  morton=False deals the
        caller's own ray order.  balance=False: static dealing (block b -> rank b % N) even when the network can
        estimate block costs.  chunk: older name of `block`.  force_collective (default: env OCC_FORCE_COLLECTIVE=1): a
        process group of ONE rank still takes the N > 1 branch -- Morton-block plan, padded send buffer, asynchronous
        `dist.gather` into the list-of-views receive buffer, `work.wait()`, un-permutation -- so that the RCCL path runs on
        a single-GPU box exactly as it does on a node.  verify_plan=False skips the plan checksum exchange.
        emulate=(N, k): act as rank k of a world of N inside THIS process (bench.py's `predicted_scaling`, while no multi-GPU
        node is available): the plan is the N-rank plan, this rank's share is gathered from the host frame / rendered / copied
        to the padded send buffer exactly as rank k would, and the frame's gather is issued on a ONE-rank process group into
        rank k's slot of the N-slot receive buffer (RCCL self-gather; a plain device copy when no group is initialised);
        rank 0 un-permutes the buffer (the other slots hold no pixels: timing only), ranks k > 0 return None like real ones."""
        self.net, self.device, self.group, self.channels = net, torch.device(device), group, channels
        self.block = int(chunk if chunk is not None else block)
        self.morton = bool(morton)
        self.balance = bool(balance)
        formed = dist.is_available() and dist.is_initialized() and not single
        self.world = dist.get_world_size(group) if formed else 1
        self.rank = dist.get_rank(group) if formed else 0
        if force_collective is None:
            force_collective = os.environ.get('OCC_FORCE_COLLECTIVE', '0') == '1'
        # collective: this renderer exchanges blocks through the process group (always when it has more than one rank)
        self.collective = bool(formed and (self.world > 1 or force_collective))
        self.verify_plan = bool(verify_plan)
        self.emulate = None
        if emulate is not None:
            n, k = int(emulate[0]), int(emulate[1])
            if not (n >= 1 and 0 <= k < n) or (formed and dist.get_world_size(group) != 1):
                raise RuntimeError(f'ShardedRenderer(emulate={emulate!r}): needs 0 <= k < N and at most a one-rank process group')
            self.emulate, self.world, self.rank, self.collective, self.verify_plan = (n, k), n, k, True, False
        self.backend = dist.get_backend(group) if formed else None
        # gloo has no gather on device tensors: with that backend (tests: several processes sharing one GPU) the blocks
        # are exchanged through host buffers; with nccl (= RCCL) they stay on the device
        self.host_exchange = self.collective and self.device.type == 'cuda' and self.backend == 'gloo'
        self.gathers_issued = 0          # dist.gather calls issued (what bench.py / the tests report)
        self.plans_verified = 0          # plan checksums exchanged and found equal on every rank
        self._plans = {}          # (rays per frame, ray_order_key) -> shard plan (index lists)
        self._bufs = {}           # padded width -> send / receive / staging buffers, two slots
        self._turn = 0
        self.last_shard_rays = None      # rays this rank rendered in the last submitted frame

    def formed_world_size(self):
        """World size the process group actually formed (what bench.py prints)."""
        return self.world

    # ------------------------------------------------------------------ plans and buffers
    def _block_costs(self, data, order, nb_full, iter_val=1e7):
        """Estimated cost of every full block of the walk: live samples on 16 of its rays (`Network.live_samples_per_ray`:
        the frame's preamble + the sampler/warp kernel on a sixteenth of the rays, ~0.3 ms for a 512^2 frame).  Every rank
        computes the same numbers from the same frame.  None when the network cannot tell (the plan is then static)."""
        probe = getattr(self.net, 'live_samples_per_ray', None)
        B = self.block
        if probe is None or not self.balance or nb_full < 2 * self.world or B % PROBE_STRIDE:
            return None
        per = B // PROBE_STRIDE
        p = (torch.arange(nb_full * per, device=order.device) * PROBE_STRIDE + PROBE_STRIDE // 2)
        idx = order[p]
        sub = {k: v for k, v in data.items() if k not in ('rays', 'near', 'far')}
        live = probe(rays=data['rays'][:, idx], near=data['near'].reshape(-1, 1)[idx], far=data['far'].reshape(-1, 1)[idx],
                     iter_val=iter_val, **sub)      # (the iteration decides whether the pose refiner is on: same warp as the render)
        return live.view(nb_full, per).sum(dim=1).to(order.device)      # (host frames: one read per plan)

    @staticmethod
    def _mix64(t):
        """Order-sensitive 64-bit checksum of an integer tensor (int64 arithmetic wraps): sum of (t_i + 1) * odd(i)."""
        t = t.to(torch.int64)
        i = torch.arange(t.numel(), device=t.device, dtype=torch.int64)
        return ((t + 1) * (i * -7046029254386353131 + 0x2545F4914F6CDD1D | 1)).sum()

    def _start_plan_check(self, plan, order, by_cost):
        """All-gather the plan's checksum [R / width / sizes, walk, cost order]; verified by the first `submit()` that uses the
        plan, before that frame's gather is issued."""
        if not (self.collective and self.verify_plan):
            return
        meta = torch.tensor([plan['R'], plan['width'], int(plan['cost_aware'])] + list(plan['sizes']), dtype=torch.int64)
        where = order.device
        mine = torch.stack([self._mix64(meta).to(where), self._mix64(order), self._mix64(by_cost)])
        W = self.world
        if self.backend == 'nccl':
            mine = mine.to(self.device)
            allc = torch.empty(W * 3, dtype=torch.int64, device=self.device)
            work = dist.all_gather_into_tensor(allc, mine, group=self.group, async_op=True)
            work.wait()                                   # stream-level: the copy below is ordered after the collective
            host = torch.empty(W * 3, dtype=torch.int64).pin_memory()
            host.copy_(allc, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            plan['check'] = (ev, host, allc)
        else:                                             # gloo (CPU tests, ranks sharing a GPU): host tensors, synchronous
            parts = [torch.empty(3, dtype=torch.int64) for _ in range(W)]
            dist.all_gather(parts, mine.cpu(), group=self.group)
            plan['check'] = (None, torch.cat(parts), None)

    def _verify_plan(self, plan):
        chk = plan.get('check')
        if chk is None:
            return
        ev, host, _ = chk
        if ev is not None:
            ev.synchronize()
        rows = host.view(self.world, 3)
        bad = [r for r in range(self.world) if not torch.equal(rows[r], rows[self.rank])]
        plan['check'] = None
        if bad:
            what = ['ray count / shard sizes', 'Morton walk', 'cost order of the blocks']
            diff = sorted({what[c] for r in bad for c in range(3) if rows[r, c] != rows[self.rank, c]})
            raise RuntimeError(f'shard plan of rank {self.rank} differs from rank(s) {bad} in: {", ".join(diff)} -- the '
                               'ranks were handed different frames or computed different block costs (dissimilar GPUs: '
                               'construct ShardedRenderer with balance=False)')
        self.plans_verified += 1

    def _build_plan(self, data, iter_val=1e7):
        rays = data['rays']
        R, W, B = int(rays.shape[1]), self.world, self.block
        nb_full, tail = divmod(R, B)
        nb = nb_full + (1 if tail else 0)
        if not self.collective:
            return {'R': R, 'sizes': [R], 'width': -(-max(R, 1) // WIDTH_QUANTUM) * WIDTH_QUANTUM}
        where = rays.device
        order = ray_patch_order(rays[1]) if self.morton else torch.arange(R, device=where)   # [R] walk position -> ray index
        cost = self._block_costs(data, order, nb_full, iter_val) if self.morton else None
        serp = cost is not None
        # Dealing position s of a block: its index in the walk (static plan), or its place in the descending order of the
        # estimated costs (cost-aware plan; the partial tail block always last).  Position s goes to rank s % W -- every
        # second round of W positions reversed in the cost-aware plan ("serpentine": ranks 0..W-1, W-1..0, ...), so that
        # every rank receives one block of each cost stratum and the sums even out -- and is that rank's slot s // W.

        def rank_of(sv):
            r = sv % W
            return torch.where((torch.div(sv, W, rounding_mode='floor') % 2) == 1, W - 1 - r, r) if serp else r
        slots = [0] * W
        for sv in range(nb):                                          # host arithmetic: a few hundred blocks
            r = sv % W
            slots[(W - 1 - r) if serp and (sv // W) % 2 else r] += 1
        last = (nb - 1) % W
        tail_rank = ((W - 1 - last) if serp and ((nb - 1) // W) % 2 else last) if tail else -1
        sizes = [slots[r] * B - ((B - tail) if r == tail_rank else 0) for r in range(W)]
        plan = {'R': R, 'sizes': sizes, 'width': -(-max(max(sizes), 1) // WIDTH_QUANTUM) * WIDTH_QUANTUM, 'cost_aware': serp}
        if serp:
            by_cost = torch.argsort(cost, descending=True, stable=True)                       # s -> block
            if tail:
                by_cost = torch.cat([by_cost, torch.tensor([nb_full], device=where)])
            spos = torch.empty_like(by_cost)
            spos[by_cost] = torch.arange(nb, device=where)                                    # block -> s
        else:
            by_cost = spos = torch.arange(nb, device=where)
        me = self.rank
        j = torch.arange(slots[me], device=where)
        s_mine = j * W + ((torch.where(j % 2 == 1, W - 1 - me, me)) if serp else me)
        i = torch.arange(sizes[me], device=where)
        pos = by_cost[s_mine][torch.div(i, B, rounding_mode='floor')] * B + i % B              # walk positions of my rays
        plan['mine'] = {where.type: order[pos]}
        if self.rank == 0:
            # row of the concatenated [world * width] receive buffer that holds walk position p, then per ray index
            p = torch.arange(R, device=where)
            sv = spos[torch.div(p, B, rounding_mode='floor')]
            dest = rank_of(sv) * plan['width'] + torch.div(sv, W, rounding_mode='floor') * B + p % B
            src = torch.empty_like(dest)
            src[order] = dest
            plan['unpermute'] = src.to(self.device)
        self._start_plan_check(plan, order, by_cost)
        return plan

    def _get_plan(self, data, key, iter_val=1e7):
        R = int(data['rays'].shape[1])
        if key is None and self.collective and self.morton:
            return self._build_plan(data, iter_val)               # unnamed camera: the walk is recomputed for this frame
        k = (R, key)
        hit = self._plans.get(k)
        if hit is None:
            if len(self._plans) >= 16:                            # a sequence's frames differ in ray count: keep a few plans
                self._plans.pop(next(iter(self._plans)))
            hit = self._plans[k] = self._build_plan(data, iter_val)
        return hit

    def _mine(self, plan, dev_type):
        m = plan['mine']
        if dev_type not in m:
            m[dev_type] = next(iter(m.values())).to(self.device if dev_type == 'cuda' else 'cpu')
        return m[dev_type]

    def _get_bufs(self, width):
        b = self._bufs.get(width)
        if b is None:
            if len(self._bufs) >= 4:
                self._bufs.pop(next(iter(self._bufs)))            # (frames in flight keep their own reference)
            gpu = self.device.type == 'cuda'
            b = {'send': [torch.zeros(width, self.channels, device=self.device) for _ in range(2)], 'stage': [None, None]}
            if self.host_exchange:
                b['send_host'] = [torch.zeros(width, self.channels).pin_memory() for _ in range(2)]
                if self.rank == 0:
                    b['recv_host'] = [torch.empty(self.world * width, self.channels).pin_memory() for _ in range(2)]
            if self.rank == 0 and self.collective:
                b['recv'] = [torch.empty(self.world * width, self.channels, device=self.device) for _ in range(2)]
            b['gpu'] = gpu
            self._bufs[width] = b
        return b

    # ------------------------------------------------------------------ one frame
    def submit(self, data, iter_val=1e7, ray_order_key=None, **net_kwargs):
        """Render this rank's share of the frame `data` and start the gather of (rgb, alpha, depth).  Per-ray entries
        (rays[2,R,3], near/far[R,1]) may live on the host (only the shard is then copied to the device) or on the
        device; every rank passes the same frame.  ray_order_key: hashable name of the camera -- frames with the same
        key and ray count reuse the shard plan (and, inside `Network`, the Morton order of the shard)."""
        rays = data['rays']
        R = int(rays.shape[1])
        plan = self._get_plan(data, ray_order_key, iter_val)
        bufs = self._get_bufs(plan['width'])
        slot = self._turn = self._turn ^ 1
        n_mine = plan['sizes'][self.rank]
        self.last_shard_rays = n_mine
        send = bufs['send'][slot]
        if n_mine:
            local = dict(data)
            if not self.collective:                              # the whole frame is this rank's, in the caller's order
                sub = (rays, data['near'], data['far'])
            elif rays.is_cuda:
                mine = self._mine(plan, 'cuda')
                sub = (rays[:, mine], data['near'].reshape(-1, 1)[mine], data['far'].reshape(-1, 1)[mine])
            else:                                                # host frame: gather the shard into PINNED staging buffers, so
                mine = self._mine(plan, 'cpu')                   # that the copies below really are asynchronous
                if bufs['stage'][slot] is None:
                    w = plan['width']
                    st = [torch.empty(2 * w * 3), torch.empty(w, 1), torch.empty(w, 1)]
                    bufs['stage'][slot] = tuple(b.pin_memory() if bufs['gpu'] else b for b in st) + \
                        (torch.cuda.Event() if bufs['gpu'] else None,)
                rs, ns, fs, ev = bufs['stage'][slot]
                if ev is not None:
                    ev.synchronize()                             # the copy issued from this slot two frames ago has left it
                # the rays stage is FLAT and viewed as [2, n_mine, 3]: a slice [:, :n_mine] of a [2, w, 3] buffer is not
                # contiguous, and .to(device) of a non-contiguous pinned tensor goes through a pageable temporary
                # (a synchronous copy)
                rs, ns, fs = rs[:2 * n_mine * 3].view(2, n_mine, 3), ns[:n_mine], fs[:n_mine]
                torch.index_select(rays, 1, mine, out=rs)
                torch.index_select(data['near'].reshape(-1, 1), 0, mine, out=ns)
                torch.index_select(data['far'].reshape(-1, 1), 0, mine, out=fs)
                sub = (rs, ns, fs)
            local['rays'] = sub[0].to(self.device, non_blocking=True)
            local['near'] = sub[1].to(self.device, non_blocking=True)
            local['far'] = sub[2].to(self.device, non_blocking=True)
            if self.collective and not rays.is_cuda and bufs['stage'][slot][3] is not None:
                bufs['stage'][slot][3].record(torch.cuda.current_stream(self.device))
            for k, v in data.items():
                if k not in ('rays', 'near', 'far') and torch.is_tensor(v) and not v.is_cuda and v.numel() > 3:
                    local[k] = v.to(self.device, non_blocking=True)
            if ray_order_key is not None:
                # (the network caches the Morton order of the rays IT is handed per key: the key names this renderer's way of
                # ordering / dealing them as well -- a single-process and a collective renderer of one process and camera
                # hand over the same rays in different orders)
                net_kwargs = dict(net_kwargs, ray_order_key=(ray_order_key, 'shard', self.rank, self.world, self.collective,
                                                             self.block, self.morton, self.balance))
            out = self.net(**local, iter_val=iter_val, **net_kwargs)
            send[:n_mine, :3] = out['rgb']
            send[:n_mine, 3] = out['alpha']
            send[:n_mine, 4] = out['depth']
        if not self.collective:
            return _Pending(None, slot, R, plan, bufs)
        # A new plan is verified BEFORE its first gather is enqueued: ranks that disagree on the frame could disagree on
        # `width`, and a gather with mismatched counts is undefined behaviour on RCCL, not the RuntimeError promised.  The
        # check's event was recorded ahead of the frame's kernels, so this waits for the checksum exchange only (once per
        # new plan; later frames of the plan find nothing pending).
        self._verify_plan(plan)
        if self.emulate is not None:
            # rank k's block into rank k's slot of the N-slot receive buffer, through the one-rank group when there is one
            if bufs.get('recv') is None:
                bufs['recv'] = [torch.zeros(self.world * plan['width'], self.channels, device=self.device) for _ in range(2)]
            mine = bufs['recv'][slot].view(self.world, plan['width'], self.channels)[self.rank]
            if self.backend is not None:
                work = dist.gather(send, [mine], dst=0, group=self.group, async_op=True)
                self.gathers_issued += 1
            else:
                mine.copy_(send)
                work = None
            return _Pending(work, slot, R, plan, bufs)
        if self.host_exchange:
            send = bufs['send_host'][slot].copy_(send)               # (synchronous: the gloo path is a test vehicle)
            rbuf = bufs['recv_host'][slot] if self.rank == 0 else None
        else:
            rbuf = bufs['recv'][slot] if self.rank == 0 else None
        recv = list(rbuf.view(self.world, plan['width'], self.channels).unbind(0)) if self.rank == 0 else None
        work = dist.gather(send, recv, dst=0, group=self.group, async_op=True)
        self.gathers_issued += 1
        return _Pending(work, slot, R, plan, bufs)

    def finish(self, pending):
        """Wait for a frame's gather; -> {'rgb','alpha','depth'} in the caller's ray order on rank 0, None elsewhere.
        The tensors are the frame's own (not views of the slot buffers, which the frame after next reuses)."""
        plan, bufs = pending.plan, pending.bufs
        if not self.collective:
            full = bufs['send'][pending.slot][:pending.n_rays].clone()
        else:
            if pending.work is not None:
                pending.work.wait()
            self._verify_plan(plan)
            if self.rank != 0:
                return None
            if self.host_exchange:
                bufs['recv'][pending.slot].copy_(bufs['recv_host'][pending.slot], non_blocking=True)
            full = bufs['recv'][pending.slot].index_select(0, plan['unpermute'])
        return {'rgb': full[:, :3], 'alpha': full[:, 3], 'depth': full[:, 4], 'packed': full}     # packed: contiguous [R,5]

    def render_frames(self, frames, iter_val=1e7, **net_kwargs):
        """Generator over `frames`: frame t's gather overlaps frame t+1's kernels (one frame of lag).  A frame may be a
        dict or a (dict, ray_order_key) pair."""
        prev = None
        for item in frames:
            data, key = item if isinstance(item, tuple) else (item, None)
            cur = self.submit(data, iter_val=iter_val, ray_order_key=key, **net_kwargs)
            if prev is not None:
                yield self.finish(prev)
            prev = cur
        if prev is not None:
            yield self.finish(prev)

    def shard_stats(self):
        """(rays, live samples) this rank rendered in the last submitted frame; live samples is None when the network
        does not report them.  One device read: call it outside timed regions."""
        live = getattr(self.net, 'last_live_count', None) if self.last_shard_rays else 0
        return int(self.last_shard_rays or 0), (None if live is None else int(live))


_renderers = {}


def get_renderer(net, device, group=None, block=BLOCK):
    key = (id(net), str(device), id(group), int(block))
    r = _renderers.get(key)
    if r is None:
        r = _renderers[key] = ShardedRenderer(net, device, group=group, block=block)
    return r


def render_frame_sharded(net, data, iter_val=1e7, group=None, chunk=BLOCK, ray_order_key=None):
    """One frame, synchronously: this rank's share rendered, (rgb, alpha, depth) gathered on rank 0 -- the path's
    only collective.  Returns the full-frame dict in the caller's ray order on rank 0 and None on the other ranks."""
    dev = data['rays'].device
    if dev.type != 'cuda' and hasattr(net, 'point_base'):      # host frame: the shard is copied to the model's device
        dev = net.point_base.device
    r = get_renderer(net, dev, group, chunk)
    return r.finish(r.submit(data, iter_val=iter_val, ray_order_key=ray_order_key))
