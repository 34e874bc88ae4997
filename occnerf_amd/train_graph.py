"""The per-step "static" part of the training step as two hipGraphs.

What a training step evaluates before and after the per-sample kernels -- the pose refiner MLP, Rodrigues, the corrected
rotations, forward kinematics + inverse (motion bases), the motion-weight volume decoder with its softmax against the prior,
and the per-point SDF block -- is ~250 tiny launches forward and ~450 backward (rocprofv3, profiles/r05_train_*): a few
hundred KFLOP spread over a millisecond of GPU time and several milliseconds of launch latency, during which the GPU idles.
The reference has the same structure (trainer.py:239-249 over network.py:558-596, 263-284, network_util.py:98-200).

Here the whole part is captured ONCE, forward and backward, with `torch.cuda.make_graphed_callables` (hipGraph on ROCm) and
replayed every step: static input buffers, one graph launch forward, one backward, parameter gradients handed to autograd
as usual.  The capture is keyed on the data pointers of the parameters it reads and on the input shapes: an optimiser that
updates in place (occnerf_amd/optim.FusedAdam, torch.optim.*) never triggers a re-capture; `Network.invalidate_cache()` -- which
`load_state_dict`, `.to()` / `_apply` and a new point cloud all run -- drops the captured graphs together with the device-side
constants they baked addresses of (the normals of `Network._context()`), so the next step captures afresh; an entry also keeps
references to those constants, so nothing a live graph reads can be freed under it.

A captured callable owns ONE set of static outputs and saved activations.  A second grad-enabled forward before the first one's
backward (gradient accumulation over frames, a loss over two frames) would overwrite them: such a call is detected (the previous
outputs are still alive and no gradient has reached them) and served by the eager modules instead -- correct, slower.

Nothing numerical changes: the graph holds the very kernels the eager modules launch (the only edit for capturability is
`torch.linalg.inv_ex` in place of `torch.inverse`, whose error check reads `info` on the host -- same rocSOLVER kernels).
`cfg.train_graph = False` or any failure to capture (reported once through `warnings`) falls back to the eager modules.
"""
import warnings
import weakref

import torch
import torch.nn as nn


class _StaticPart(nn.Module):
    """(posevec[1,69], dst_Rs[1,24,3,3], dst_Ts[1,24,3], cnl_gtfms[1,24,4,4], prior[1,25,G,G,G])
    -> (Rs[24,3,3], Ts[24,3], vol[25,G,G,G], knn_base[P,3] f64, sdf[P,1])."""

    def __init__(self, net, refine):
        super().__init__()
        # plain attribute access (not registered twice under the network: this wrapper is never attached to it as a submodule)
        self.pose_decoder = net.pose_decoder
        self.motion_basis_computer = net.motion_basis_computer
        self.mweight_vol_decoder = net.mweight_vol_decoder
        self.point_dist = net.point_dist
        self.__dict__['net'] = net
        # the graph bakes in the addresses of these per-model constants (train_path.point_sdf_block reads them): keep them alive
        ctx = net._context()
        self.__dict__['held'] = tuple(ctx[k] for k in ('normals', 'unit') if k in ctx)
        self.refine = bool(refine)
        self.total_bones = int(net.cfg.total_bones)

    def forward(self, posevec, dst_Rs, dst_Ts, cnl_gtfms, prior):
        from . import train_path
        Rs, Ts = train_path.motion_bases(self.net, self.refine, posevec, dst_Rs, dst_Ts, cnl_gtfms)
        vol = self.mweight_vol_decoder(motion_weights_priors=prior)[0]
        knn_base, sdf = train_path.point_sdf_block(self.net)
        return Rs, Ts, vol, knn_base, sdf


class PerStepGraph:
    """Owner of the captured callables of one Network (kept in the network's __dict__, outside nn.Module registration)."""

    def __init__(self, net):
        self.net = net
        self.entries = {}            # (refine, shapes) -> (callable, parameter data pointers)
        self.captures = 0            # how many times a graph was captured (tests: stays 1 over many steps)
        self.replays = 0
        self.eager_fallbacks = 0     # grad-enabled calls served by the eager modules because a replay was still outstanding
        self.failed = None
        self._pending = None         # weakref to the last replay's first output while its backward has not run

    def _param_key(self, mod):
        return tuple(p.data_ptr() for p in mod.parameters()) + (self.net.point_base.data_ptr(),)

    def __call__(self, refine, posevec, dst_Rs, dst_Ts, cnl_gtfms, prior):
        """-> (Rs, Ts, vol, knn_base, sdf), differentiable w.r.t. the parameters; None when graphs are unavailable."""
        if self.failed is not None:
            return None
        if self._pending is not None and self._pending() is not None:
            self.eager_fallbacks += 1                         # the last replay's outputs are alive and not yet backpropagated
            return None
        args = tuple(t.detach().float().contiguous() for t in (posevec, dst_Rs, dst_Ts, cnl_gtfms, prior))
        key = (bool(refine),) + tuple(tuple(a.shape) for a in args)
        hit = self.entries.get(key)
        if hit is not None and hit[1] != self._param_key(hit[2]):
            hit = None                                        # a parameter moved (load_state_dict / .to()): capture again
        if hit is None:
            mod = _StaticPart(self.net, refine)
            # (make_graphed_callables warms up on a side stream, so the parameters' AccumulateGrad nodes belong to that
            # stream while the step's gradients arrive on the current one: intended here, one event wait per parameter)
            quiet = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
            if quiet is not None:
                quiet(False)
            try:
                with torch.enable_grad():
                    fn = torch.cuda.make_graphed_callables(mod, tuple(a.clone() for a in args), num_warmup_iters=3,
                                                           allow_unused_input=True)
            except Exception as e:                            # noqa: BLE001  (capture is an optimisation, never a requirement)
                self.failed = f'{type(e).__name__}: {e}'
                warnings.warn('occnerf_amd: the per-step hipGraph could not be captured, falling back to eager per-frame '
                              f'modules ({self.failed})')
                torch.cuda.synchronize()
                return None
            self.captures += 1
            hit = self.entries[key] = (fn, self._param_key(mod), mod)
        self.replays += 1
        outs = hit[0](*args)
        if torch.is_grad_enabled():
            live = [o for o in outs if torch.is_tensor(o) and o.requires_grad]
            if live:
                self._pending = weakref.ref(live[0])

                def done(grad, owner=self):
                    owner._pending = None
                    return grad
                for o in live:
                    o.register_hook(done)
        return outs


def get(net):
    g = net.__dict__.get('_per_step_graph')
    if g is None:
        g = net.__dict__['_per_step_graph'] = PerStepGraph(net)
    return g
