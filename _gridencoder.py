"""`_gridencoder`: the module name the reference's operator seam imports.

/root/reference/core/nets/occnerf/gridencoder/grid.py:9-12 does ``import _gridencoder as _backend`` (a pybind11 extension built
from src/bindings.cpp:5-9) and calls, positionally,

    _backend.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners,
                                 interpolation)                                                                  (grid.py:55)
    _backend.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                                  gridtype, align_corners, interpolation)                                         (grid.py:83)
    _backend.grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners)

With this file's directory on sys.path the reference's grid.py binds to the gfx950 kernels of occnerf_amd/liboccnerf_hip.so
(C ABI: include/occnerf_hip.h section 1) without an edit.  Same argument order and meaning, outputs written in place, dispatch
on the embeddings' / grad's dtype (float32, float16, float64) as AT_DISPATCH_FLOATING_TYPES_AND_HALF does at gridencoder.cu:467,500.
No fallback: importing this module loads the HIP library or fails.
"""
from occnerf_amd.gridencoder import grad_total_variation, grid_encode_backward, grid_encode_forward  # noqa: F401

__all__ = ['grid_encode_forward', 'grid_encode_backward', 'grad_total_variation']
