"""ctypes binding of liboccnerf_hip.so (the C ABI declared in include/occnerf_hip.h).

There is no CPU or eager-torch fallback: if the library is missing the import of any
op fails with an explicit error, and every op refuses non-GPU tensors.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OCCNERF_HIP_LIB=<path>: load another build of the same library (A/B timing of kernel variants); never a different backend
LIB_PATH = os.environ.get('OCCNERF_HIP_LIB') or os.path.join(_HERE, 'liboccnerf_hip.so')

ABI_VERSION = 5          # include/occnerf_hip.h OCCNERF_ABI_VERSION this binding mirrors

_vp, _i32, _i64, _u32, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_float

# name -> (restype, argtypes); mirrors include/occnerf_hip.h one to one
SIGNATURES = {
    'occnerf_abi_version': (C.c_int, []),
    'occnerf_last_error': (C.c_char_p, []),
    'occnerf_experiment_knob': (C.c_int, [C.c_char_p, C.c_int]),
    'occnerf_grid_encode_forward': (C.c_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32,
                                               _vp, _u32, C.c_int, _u32, _vp]),
    'occnerf_grid_encode_backward': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32,
                                                _u32, _vp, _vp, _u32, C.c_int, _u32, _vp]),
    'occnerf_grid_grad_runs': (C.c_int, [_vp, _vp, _i64, _u32, _u32, _u32, _vp, _vp]),
    'occnerf_grid_encode_forward_h': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32,
                                                 _vp, _u32, C.c_int, _u32, _vp]),
    'occnerf_grid_encode_backward_h': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32,
                                                  _u32, _vp, _vp, _u32, C.c_int, _u32, _vp, _i64, _vp]),
    'occnerf_grid_encode_forward_f16': (C.c_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32,
                                                   _vp, _u32, C.c_int, _u32, _vp]),
    'occnerf_grid_encode_backward_f16': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32,
                                                    _u32, _vp, _vp, _u32, C.c_int, _u32, _vp]),
    'occnerf_grid_encode_forward_f64': (C.c_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32,
                                                   _vp, _u32, C.c_int, _u32, _vp]),
    'occnerf_grid_encode_backward_f64': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32,
                                                    _u32, _vp, _vp, _u32, C.c_int, _u32, _vp]),
    'occnerf_trunks_packed_bytes': (_i64, []),
    'occnerf_trunks_pack_bf16': (C.c_int, [_vp, _vp, _vp]),
    'occnerf_trunks_forward_bf16': (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_grad_total_variation': (C.c_int, [_vp, _vp, _vp, _vp, _f32, _u32, _u32, _u32, _u32, _f32,
                                                _u32, _u32, C.c_int, _vp]),
    'occnerf_sample_warp': (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp,
                                       _vp, _vp, _vp, _vp, _vp]),
    'occnerf_bone_boxes': (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    'occnerf_sample_warp_culled': (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_nonrigid_packed_floats': (_i64, []),
    'occnerf_nonrigid_pack': (C.c_int, [_vp, _vp, _vp, _vp]),
    'occnerf_nonrigid': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_nonrigid_direct': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_nonrigid_packed_bf16_bytes': (_i64, []),
    'occnerf_nonrigid_pack_bf16': (C.c_int, [_vp, _vp, _vp]),
    'occnerf_nonrigid_bf16x3': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_nonrigid_bf16x3_rows': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_msknn': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    'occnerf_msknn_clustered': (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32,
                                           _vp, _vp, _vp, _vp, _vp]),
    'occnerf_msknn_clustered_centered': (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32,
                                           _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_knn_center': (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    'occnerf_knn_small': (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    'occnerf_unit_normals': (C.c_int, [_vp, _i32, _vp, _vp]),
    'occnerf_point_sdf': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    'occnerf_ray_order_temp_bytes': (_i64, [_i64]),
    'occnerf_ray_order': (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _vp]),
    'occnerf_point_sdf_backward': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    'occnerf_gen_rays': (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_agg_forward': (C.c_int, [_vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp]),
    'occnerf_agg_backward_slices': (_i32, [_i64]),
    'occnerf_agg_backward_scratch_bytes': (_i64, [_i64, _i32]),
    'occnerf_agg_backward': (C.c_int, [_vp, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    'occnerf_live_rows_temp_bytes': (_i64, [_i64]),
    'occnerf_live_rows': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    'occnerf_scatter_raw': (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    'occnerf_repeat_heads_temp_bytes': (_i64, [_i64]),
    'occnerf_repeat_heads': (C.c_int, [_vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    'occnerf_unique_heads_temp_bytes': (_i64, [_i64]),
    'occnerf_unique_heads': (C.c_int, [_vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    'occnerf_scatter_raw_heads': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'occnerf_canonical_mlp_rows': (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    'occnerf_canonical_mlp_counted': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    'occnerf_nonrigid_rows': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_point_table_stride': (_i32, []),
    'occnerf_point_table': (C.c_int, [_vp, _vp, _vp, _i32, _f32, _f32, _vp, _vp, _vp, _u32, _f32, _u32, _vp,
                                       _vp]),
    'occnerf_point_pack': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    'occnerf_sample_features': (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _vp,
                                           _vp, _vp, _u32, _f32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                                           _vp, _vp]),
    'occnerf_sample_features_centered': (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _vp,
                                                    _vp, _vp, _u32, _f32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                                                    _vp, _vp, _vp, _vp]),
    'occnerf_canonical_mlp_packed_floats': (_i64, []),
    'occnerf_canonical_mlp_pack': (C.c_int, [_vp, _vp, _vp, _vp]),
    'occnerf_canonical_mlp': (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    'occnerf_canonical_mlp_direct': (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    'occnerf_canonical_mlp_packed_bf16_bytes': (_i64, []),
    'occnerf_canonical_mlp_pack_bf16': (C.c_int, [_vp, _vp, _vp]),
    'occnerf_canonical_mlp_bf16x3': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i32, _vp]),
    'occnerf_canonical_mlp_pack_f16': (C.c_int, [_vp, _vp, _vp]),
    'occnerf_canonical_mlp_f16x3': (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_nonrigid_pack_f16': (C.c_int, [_vp, _vp, _vp]),
    'occnerf_nonrigid_f16x3': (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_canonical_mlp_bf16x3_rows': (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _vp]),
    'occnerf_composite': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_linear_pack': (C.c_int, [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    'occnerf_linear_forward': (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _i32,
                                          _i32, _vp, _i32, _i64, _i64, _i32, _i32, _vp]),
    'occnerf_linear_wgrad_slices': (_i32, [_i64]),
    'occnerf_linear_wgrad': (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i64, _i32, _vp, _vp, _vp]),
    'occnerf_linear_wgrad_reduce': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp]),
    'occnerf_composite_backward': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_warp_backward_slices': (_i32, [_i64]),
    'occnerf_warp_backward': (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_agg_weights': (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    'occnerf_pose_motion_bases': (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_pose_motion_bases_backward': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'occnerf_prior_softmax': (C.c_int, [_vp, _vp, _i32, _i64, _vp, _vp]),
    'occnerf_pack_rays': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'occnerf_assemble_image': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    'occnerf_convt3d_col2im': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    'occnerf_convt3d_im2col': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    'occnerf_adam_table_row_bytes': (_i32, []),
    'occnerf_adam_step': (C.c_int, [_vp, _i32, _vp, _i32, _i32] + [C.c_double] * 4 + [_vp, _vp]),
}

_lib = None


def lib():
    """Load the HIP library once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'{LIB_PATH} not found: the gfx950 kernels are not built. Run '
                '`python -c "import __graft_entry__ as g; g.build()"` (or `make -C '
                'occnerf_amd/csrc`). There is no CPU fallback for this path.')
        # torch first: it ships its own libamdhip64 and the kernels must run on the HIP runtime that owns
        # torch's streams and allocations.  Loading this library first would bring in /opt/rocm's copy and
        # leave two runtimes in the process ("no ROCm-capable device is detected" on the first launch).
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)            # AttributeError if the ABI drifted
            fn.restype, fn.argtypes = res, args
        if handle.occnerf_abi_version() != ABI_VERSION:
            raise ImportError(f'liboccnerf_hip.so: ABI version {handle.occnerf_abi_version()}, this binding was written '
                              f'for {ABI_VERSION} (include/occnerf_hip.h OCCNERF_ABI_VERSION): rebuild the library')
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().occnerf_last_error().decode(errors='replace')
        raise RuntimeError(f'{what} failed ({rc}): {msg}')
