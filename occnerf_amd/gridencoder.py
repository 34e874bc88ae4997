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
