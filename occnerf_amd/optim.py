"""`FusedAdam`: the reference's optimiser step on the device in one multi-tensor pass.

The reference's trainer does `torch.nn.utils.clip_grad_norm_(network.parameters(), 1.0)` and then
`torch.optim.Adam.step()` with per-group learning rates (trainer.py:248-249, optimizer.py:12-43, betas (0.9, 0.999),
eps 1e-8, no weight decay).  This class keeps torch.optim.Adam's interface -- `param_groups` (so the reference's
exp_decay.update_lr edits work), `state_dict()` with the same per-parameter keys (`step`, `exp_avg`, `exp_avg_sq`),
`zero_grad` -- and evaluates clip + Adam with three launches of csrc/optim.hip (norm partials, norm, update)
instead of ~25 foreach kernels; the gradient norm never visits the host.
"""
import math

import numpy as np
import torch

from . import _lib, ops

CHUNK = 1 << 16          # elements per workgroup


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._plan = None

    def _params_with_grad(self):
        out = []
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.grad is not None:
                    out.append((gi, p))
        return out

    def _build_plan(self, items):
        dev = items[0][1].device
        chunks = []
        for ti, (_, p) in enumerate(items):
            for c in range((p.numel() + CHUNK - 1) // CHUNK):
                chunks.append((ti, c))
        row = int(_lib.lib().occnerf_adam_table_row_bytes())
        assert row == 56
        # p, g, m, v, n, (lr | bias_corr1), (bias_corr2_sqrt | pad)
        hosts = [torch.zeros(len(items), 7, dtype=torch.int64).pin_memory() for _ in range(2)]
        return {
            'key': tuple(id(p) for _, p in items), 'device': dev, 'hosts': hosts, 'np': [h.numpy() for h in hosts],
            'events': [None, None], 'turn': 0,
            'table': torch.zeros(len(items), 7, dtype=torch.int64, device=dev),
            'chunks': torch.tensor(chunks, dtype=torch.int32, device=dev).contiguous(),
            'scratch': torch.zeros(len(chunks) + 1, dtype=torch.float32, device=dev),
        }

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        """One Adam step over every parameter that has a gradient.  max_grad_norm: clip the global gradient norm
        first (what clip_grad_norm_(parameters, max_grad_norm) would do), inside the same pass."""
        loss = closure() if closure is not None else None
        items = self._params_with_grad()
        if not items:
            return loss
        for _, p in items:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.dtype == torch.float32):
                raise RuntimeError('FusedAdam: parameters and gradients must be contiguous fp32 GPU tensors')
        if self._plan is None or self._plan['key'] != tuple(id(p) for _, p in items):
            self._plan = self._build_plan(items)
        plan = self._plan
        turn = plan['turn'] = plan['turn'] ^ 1               # two pinned staging tables: the copy of step t-1 may still
        if plan['events'][turn] is not None:                 # be queued when step t fills its table
            plan['events'][turn].synchronize()
        host = plan['np'][turn]
        # validate BEFORE touching any state: a refused step must leave the optimiser as it was
        betas, eps = tuple(self.param_groups[items[0][0]]['betas']), float(self.param_groups[items[0][0]]['eps'])
        for gi, _ in items:
            group = self.param_groups[gi]
            if tuple(group['betas']) != betas or float(group['eps']) != eps:
                raise RuntimeError('FusedAdam: every parameter group must share betas and eps')
        b1, b2 = betas
        keep = []                                            # contiguous copies of strided gradients: alive until the launch
        for ti, (gi, p) in enumerate(items):
            group = self.param_groups[gi]
            st = self.state[p]
            if not st:
                st['step'] = torch.tensor(0.0)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['step'] += 1
            # a step count per parameter, as torch.optim.Adam: a parameter whose first gradient arrives late (the pose
            # refiner before pose_decoder.kick_in_iter, staged unfreezing, a resumed torch state) has its own bias corrections
            t = int(st['step'])
            g = p.grad
            if not g.is_contiguous():
                g = g.contiguous()
                keep.append(g)
            host[ti, 0], host[ti, 1] = p.data_ptr(), g.data_ptr()
            host[ti, 2], host[ti, 3] = st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
            host[ti, 4] = p.numel()
            host[ti, 5] = int(np.float32(group['lr']).view(np.uint32)) | \
                (int(np.float32(1.0 - b1 ** t).view(np.uint32)) << 32)          # low word lr, high word 1 - beta1^t
            host[ti, 6] = int(np.float32(math.sqrt(1.0 - b2 ** t)).view(np.uint32))
        plan['table'].copy_(plan['hosts'][turn], non_blocking=True)
        ev = plan['events'][turn] = plan['events'][turn] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream(plan['device']))
        with ops._guard_dev(plan['device']):
            rc = _lib.lib().occnerf_adam_step(
                plan['table'].data_ptr(), len(items), plan['chunks'].data_ptr(), plan['chunks'].shape[0], CHUNK,
                float(b1), float(b2), float(eps), float(max_grad_norm) if max_grad_norm else 0.0, plan['scratch'].data_ptr(),
                torch.cuda.current_stream(plan['device']).cuda_stream)
        _lib.check(rc, 'adam_step')
        del keep                                             # the launch is queued behind the copies on the same stream
        # the kernel wrote the parameters behind torch's back: move their version counters, which is what autograd's
        # saved-tensor checks and the renderer's per-version caches (packed MLP weights, decoded volume, point table) key on
        for _, p in items:
            torch.autograd.graph.increment_version(p)
        return loss

    def grad_norm(self):
        """Global gradient norm of the last step(max_grad_norm=...) call (device scalar, no sync)."""
        return None if self._plan is None else self._plan['scratch'][0].sqrt()
