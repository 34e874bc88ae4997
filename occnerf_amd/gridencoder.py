"""Multi-resolution hash-grid encoder: host-side layout + the torch-facing module.

Mirrors the reference's operator surface for this path (core/nets/occnerf/gridencoder/
grid.py): ``grid_offsets`` reproduces the level layout of ``GridEncoder.__init__``
(:102-131); ``GridEncoder`` keeps the constructor arguments, parameter names
(``embeddings``, ``offsets``) and forward semantics (:146-170); the three functions
``grid_encode_forward / grid_encode_backward / grad_total_variation`` keep the pybind
signatures of src/bindings.cpp:5-9 and forward to the C-ABI library (include/occnerf_hip.h).
"""
import numpy as np


def grid_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size,
                 desired_resolution=None, align_corners=False, level_dim=2):
    """-> (offsets int32 [L+1], per_level_scale float64).  grid.py:102-131."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    max_params = 2 ** log2_hashmap_size
    offsets, offset = [], 0
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), per_level_scale


# ---------------------------------------------------------------------------------------
# torch-facing operator + module (imports torch lazily so grid_offsets stays numpy-only)
# ---------------------------------------------------------------------------------------
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from torch.autograd import Function  # noqa: E402

from . import ops  # noqa: E402

# the reference's pybind names (src/bindings.cpp:5-9); `import occnerf_amd.gridencoder as
# _gridencoder` is a drop-in for its `_gridencoder` extension module
grid_encode_forward = ops.grid_encode_forward
grid_encode_backward = ops.grid_encode_backward
grad_total_variation = ops.grad_total_variation

_gridtype_to_id = {'hash': 0, 'tiled': 1}
_interp_to_id = {'linear': 0, 'smoothstep': 1}


class _GridEncode(Function):
    """grid.py:24-90: forward writes [L,B,C] and returns [B, L*C]; backward scatters into the
    embedding gradient (atomics) and, when inputs need it, contracts with dy_dx.  Under autocast the
    embeddings (and with them outputs and gradients) are torch.half, inputs stay float (grid.py:42-45)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda')
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution,
                calc_grad_inputs=False, gridtype=0, align_corners=False, interpolation=0):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        Cc = embeddings.shape[1]
        S = float(np.log2(per_level_scale))
        H = int(base_resolution)
        # grid.py:42-45: "manually handle autocast (only use half precision embeddings, inputs must be float for enough
        # precision); if C % 2 != 0, force float, since half for atomicAdd is very slow"
        if torch.is_autocast_enabled('cuda') and Cc % 2 == 0:
            embeddings = embeddings.to(torch.half)
        outputs = torch.empty(L, B, Cc, device=inputs.device, dtype=embeddings.dtype)
        dy_dx = torch.empty(B, L * D * Cc, device=inputs.device, dtype=embeddings.dtype) \
            if calc_grad_inputs else None
        ops.grid_encode_forward(inputs, embeddings.contiguous(), offsets, outputs, B, D, Cc, L, S, H,
                                dy_dx, gridtype, align_corners, interpolation)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = (B, D, Cc, L, S, H, gridtype, interpolation)
        ctx.align_corners = align_corners
        return outputs.permute(1, 0, 2).reshape(B, L * Cc)

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, Cc, L, S, H, gridtype, interpolation = ctx.dims
        if dy_dx is None and grad.dtype == torch.float32 and inputs.dtype == torch.float32 and B >= 4096:
            # large batches without an input gradient (the training step's sample encoding): the transposition to [L,B,C]
            # merges runs of bitwise identical inputs on the way (csrc/grid_encode.hip grid_grad_runs_kernel)
            grad = ops.grid_grad_runs(grad.contiguous(), inputs, B, D, L, Cc)
        else:
            grad = grad.view(B, L, Cc).permute(1, 0, 2).contiguous()
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None
        ops.grid_encode_backward(grad, inputs, embeddings.contiguous(), offsets, grad_embeddings, B, D,
                                 Cc, L, S, H, dy_dx, grad_inputs, gridtype, ctx.align_corners,
                                 interpolation)
        if grad_inputs is not None:
            grad_inputs = grad_inputs.to(inputs.dtype)
        return grad_inputs, grad_embeddings, None, None, None, None, None, None, None


grid_encode = _GridEncode.apply


class GridEncoder(nn.Module):
    """Same constructor arguments, parameter/buffer names and forward semantics as the
    reference module (grid.py:97-170)."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype='hash', align_corners=False,
                 interpolation='linear'):
        super().__init__()
        offsets, per_level_scale = grid_offsets(input_dim, num_levels, per_level_scale,
                                                base_resolution, log2_hashmap_size, desired_resolution,
                                                align_corners, level_dim)
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale = per_level_scale
        self.log2_per_level_scale = float(np.log2(per_level_scale))
        self.log2_hashmap_size, self.base_resolution = log2_hashmap_size, base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype, self.gridtype_id = gridtype, _gridtype_to_id[gridtype]
        self.interpolation, self.interp_id = interpolation, _interp_to_id[interpolation]
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size
        self.register_buffer('offsets', torch.from_numpy(offsets))
        self.n_params = int(offsets[-1]) * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def __repr__(self):
        return (f'GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} '
                f'level_dim={self.level_dim} resolution={self.base_resolution} -> '
                f'{int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} '
                f'per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} '
                f'gridtype={self.gridtype} align_corners={self.align_corners} '
                f'interpolation={self.interpolation}')

    def forward(self, inputs, bound=1):
        if bound is not None:
            if torch.is_tensor(bound):
                bound = bound.to(inputs.device).float()
                inputs = (inputs - bound[:, :3]) / (bound[:, 3:] - bound[:, :3])
            else:
                inputs = (inputs + bound) / (2 * bound)
        prefix = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        out = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale,
                          self.base_resolution, inputs.requires_grad, self.gridtype_id,
                          self.align_corners, self.interp_id)
        return out.view(prefix + [self.output_dim])

    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        raise RuntimeError('grad_total_variation: not implemented (never called by the reference '
                           'trainer; SURVEY.md section 8 row a20)')
