"""Torch-facing wrappers of the C ABI (include/occnerf_hip.h).

Each function validates its tensors (GPU, contiguous, dtype -- the reference's
CHECK_CUDA / CHECK_CONTIGUOUS / CHECK_IS_* of gridencoder.cu:15-18, raised as
RuntimeError), allocates outputs with torch, and launches on torch's current stream of
the tensors' device.  PyTorch is plumbing here: memory, streams, device guard.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_f32p = C.POINTER(C.c_float)


def _chk(t, dtype, name):
    if not torch.is_tensor(t):
        raise RuntimeError(f'{name} must be a tensor')
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a CUDA(HIP) tensor; there is no CPU path')
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be a contiguous tensor')
    if t.dtype != dtype:
        raise RuntimeError(f'{name} must be {dtype}, got {t.dtype}')
    return t.data_ptr()


def _guard_dev(dev):
    if torch.device(dev).type != 'cuda':
        raise RuntimeError(f'tensors are on {dev}: this path runs on a GPU only, there is no CPU path')
    return torch.cuda.device(dev)


def _guard(t):
    if not torch.is_tensor(t):
        raise RuntimeError('expected a tensor')
    return _guard_dev(t.device)


def _opt(t, dtype, name):
    return None if t is None else _chk(t, dtype, name)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _host_f32(vals, n):
    a = np.ascontiguousarray(np.asarray(vals, dtype=np.float32).ravel())
    assert a.size == n, (a.size, n)
    return a, a.ctypes.data_as(C.c_void_p)


def _host_i32(vals):
    a = np.ascontiguousarray(np.asarray(vals, dtype=np.int32).ravel())
    return a, a.ctypes.data_as(C.c_void_p)


class _HostCopies:
    """Host copies of small constant device tensors (level offsets, per-frame float[3] constants), fetched with
    one blocking copy the first time a tensor OBJECT is seen.  Keyed by object identity (checked through a weak
    reference) and version counter -- never by device address, which the caching allocator recycles."""

    def __init__(self, limit=256):
        self._items, self._limit = {}, limit

    def get(self, t, convert):
        import weakref
        hit = self._items.get(id(t))
        if hit is not None and hit[0]() is t and hit[1] == t._version:
            return hit[2]
        if len(self._items) > self._limit:
            self._items = {k: v for k, v in self._items.items() if v[0]() is not None}
            if len(self._items) > self._limit:
                self._items.clear()
        val = convert(t)
        self._items[id(t)] = (weakref.ref(t), t._version, val)
        return val


_host_copies = _HostCopies()


def _host_offsets(offsets):
    """Host copy of a (constant) level-offset tensor, fetched once per tensor object."""
    def convert(t):
        a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.int32))
        return a, a.ctypes.data_as(C.c_void_p)
    return _host_copies.get(offsets, convert)[1]


def host_float3(v):
    """float[3] per-frame constant (bbox min / scale, background colour) as host float32: free for host arrays,
    one blocking copy per tensor object for device tensors."""
    if torch.is_tensor(v):
        return _host_copies.get(v, lambda t: np.asarray(t.detach().cpu().numpy(), dtype=np.float32).reshape(3).copy())
    return np.asarray(v, dtype=np.float32).reshape(3)


def _ptr_table(tensors, name):
    ptrs = [_chk(t, torch.float32, f'{name}[{i}]') for i, t in enumerate(tensors)]
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    return arr


# ------------------------------------------------------------------ grid encoder (section 1)
def _encoder_dtype(t, what):
    """gridencoder.cu:467,500 dispatch on the tensor dtype over float / double / half (grid.py:42-45 feeds half embeddings
    under autocast): all three cases of AT_DISPATCH_FLOATING_TYPES_AND_HALF are built (csrc/grid_encode{,_f16,_f64}.hip)."""
    if t.dtype not in (torch.float32, torch.float16, torch.float64):
        raise RuntimeError(f'{what}: tensors are {t.dtype}; gridencoder.cu:467 dispatches float32, float64 and float16')
    return t.dtype


def grid_grad_runs(grad_rows, inputs, B, D, L, Cc):
    """grad_rows[B, L*C] fp32 -> [L,B,C] with the rows of runs of bitwise identical inputs summed into the run's first sample."""
    out = torch.empty(L, B, Cc, device=grad_rows.device, dtype=torch.float32)
    with _guard(grad_rows):
        rc = _lib.lib().occnerf_grid_grad_runs(_chk(grad_rows, torch.float32, 'grad_rows'), _chk(inputs, torch.float32, 'inputs'),
                                               int(B), int(D), int(L), int(Cc), out.data_ptr(), _stream(grad_rows))
    _lib.check(rc, 'grid_grad_runs')
    return out


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, Cc, L, S, H, dy_dx=None,
                        gridtype=0, align_corners=False, interp=0):
    """Same positional signature as the reference's `_gridencoder.grid_encode_forward`
    (bindings.cpp:6); writes `outputs[L,B,C]` (and `dy_dx`) in place.  Dispatches on embeddings.dtype like the reference."""
    dt = _encoder_dtype(embeddings, 'grid_encode_forward')
    with _guard(inputs):
        if dt == torch.float64:
            rc = _lib.lib().occnerf_grid_encode_forward_f64(
                _chk(inputs, torch.float32, 'inputs'), _chk(embeddings, torch.float64, 'embeddings'),
                _chk(offsets, torch.int32, 'offsets'), _chk(outputs, torch.float64, 'outputs'),
                int(B), int(D), int(Cc), int(L), float(S), int(H), _opt(dy_dx, torch.float64, 'dy_dx'),
                int(gridtype), int(bool(align_corners)), int(interp), _stream(inputs))
        elif dt == torch.float16:
            rc = _lib.lib().occnerf_grid_encode_forward_f16(
                _chk(inputs, torch.float32, 'inputs'), _chk(embeddings, torch.float16, 'embeddings'),
                _chk(offsets, torch.int32, 'offsets'), _chk(outputs, torch.float16, 'outputs'),
                int(B), int(D), int(Cc), int(L), float(S), int(H), _opt(dy_dx, torch.float16, 'dy_dx'),
                int(gridtype), int(bool(align_corners)), int(interp), _stream(inputs))
        else:
            rc = _lib.lib().occnerf_grid_encode_forward_h(
                _chk(inputs, torch.float32, 'inputs'), _chk(embeddings, torch.float32, 'embeddings'),
                _chk(offsets, torch.int32, 'offsets'), _host_offsets(offsets), _chk(outputs, torch.float32, 'outputs'),
                int(B), int(D), int(Cc), int(L), float(S), int(H), _opt(dy_dx, torch.float32, 'dy_dx'),
                int(gridtype), int(bool(align_corners)), int(interp), _stream(inputs))
    _lib.check(rc, 'grid_encode_forward')


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, Cc, L, S, H,
                         dy_dx=None, grad_inputs=None, gridtype=0, align_corners=False, interp=0):
    """`_gridencoder.grid_encode_backward` (bindings.cpp:7); dispatches on grad.dtype like the reference (:500)."""
    dt = _encoder_dtype(grad, 'grid_encode_backward')
    if dt == torch.float64:
        with _guard(inputs):
            rc = _lib.lib().occnerf_grid_encode_backward_f64(
                _chk(grad, torch.float64, 'grad'), _chk(inputs, torch.float32, 'inputs'),
                _chk(embeddings, torch.float64, 'embeddings'), _chk(offsets, torch.int32, 'offsets'),
                _chk(grad_embeddings, torch.float64, 'grad_embeddings'), int(B), int(D), int(Cc), int(L), float(S), int(H),
                _opt(dy_dx, torch.float64, 'dy_dx'), _opt(grad_inputs, torch.float64, 'grad_inputs'), int(gridtype),
                int(bool(align_corners)), int(interp), _stream(inputs))
        _lib.check(rc, 'grid_encode_backward')
        return
    if dt == torch.float16:
        with _guard(inputs):
            rc = _lib.lib().occnerf_grid_encode_backward_f16(
                _chk(grad, torch.float16, 'grad'), _chk(inputs, torch.float32, 'inputs'),
                _chk(embeddings, torch.float16, 'embeddings'), _chk(offsets, torch.int32, 'offsets'),
                _chk(grad_embeddings, torch.float16, 'grad_embeddings'), int(B), int(D), int(Cc), int(L), float(S), int(H),
                _opt(dy_dx, torch.float16, 'dy_dx'), _opt(grad_inputs, torch.float16, 'grad_inputs'), int(gridtype),
                int(bool(align_corners)), int(interp), _stream(inputs))
        _lib.check(rc, 'grid_encode_backward')
        return
    # room for the tile-set pre-pass of the tiled backward (large D = 4, C = 2 batches only)
    scratch = torch.empty(int(L) * int(B), device=inputs.device, dtype=torch.int64) if (int(D) == 4 and int(Cc) == 2
                                                                                      and int(B) >= 32768) else None
    with _guard(inputs):
        rc = _lib.lib().occnerf_grid_encode_backward_h(
            _chk(grad, torch.float32, 'grad'), _chk(inputs, torch.float32, 'inputs'),
            _chk(embeddings, torch.float32, 'embeddings'), _chk(offsets, torch.int32, 'offsets'),
            _host_offsets(offsets), _chk(grad_embeddings, torch.float32, 'grad_embeddings'), int(B), int(D), int(Cc),
            int(L), float(S), int(H), _opt(dy_dx, torch.float32, 'dy_dx'),
            _opt(grad_inputs, torch.float32, 'grad_inputs'), int(gridtype),
            int(bool(align_corners)), int(interp), None if scratch is None else scratch.data_ptr(),
            0 if scratch is None else scratch.numel() * 8, _stream(inputs))
    _lib.check(rc, 'grid_encode_backward')


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, Cc, L, S, H, gridtype=0,
                         align_corners=False):
    """`_gridencoder.grad_total_variation` (bindings.cpp:8): exported, not implemented."""
    rc = _lib.lib().occnerf_grad_total_variation(None, None, None, None, float(weight), int(B), int(D),
                                                 int(Cc), int(L), float(S), int(H), int(gridtype),
                                                 int(bool(align_corners)), None)
    _lib.check(rc, 'grad_total_variation')


# ------------------------------------------------------------------ per-frame ray generation
def _host_f64(vals, n):
    a = np.ascontiguousarray(np.asarray(vals, dtype=np.float64).ravel())
    assert a.size == n, (a.size, n)
    return a, a.ctypes.data_as(C.c_void_p)


def gen_rays(K, E, H, W, bbox_min, bbox_max, device):
    """All H*W pixel rays of a camera on the device -> rays8[H*W,8] (o, d, near, far), mask[H*W] uint8."""
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise RuntimeError(f'gen_rays: needs a GPU device, got {dev}')
    f32 = int(np.asarray(K).dtype == np.float32 and np.asarray(E).dtype == np.float32)   # numpy's result dtype
    E = np.asarray(E, dtype=np.float64)
    k0, pk = _host_f64(np.linalg.inv(np.asarray(K)).astype(np.float64), 9)      # inverse in K's own dtype
    r0, pr = _host_f64(E[:3, :3], 9)
    t0, pt = _host_f64(E[:3, 3], 3)
    l0, pl = _host_f64(bbox_min, 3)
    h0, ph = _host_f64(bbox_max, 3)
    rays8 = torch.empty(H * W, 8, device=dev, dtype=torch.float32)
    mask = torch.empty(H * W, device=dev, dtype=torch.uint8)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_gen_rays(pk, pr, pt, f32, int(H), int(W), pl, ph, rays8.data_ptr(),
                                         mask.data_ptr(), _stream(rays8))
    _lib.check(rc, 'gen_rays')
    return rays8, mask


# ------------------------------------------------------------------ sample pipeline (section 2)
def bone_boxes(vol, nb):
    """int32[nb,6] = {x0,x1,y0,y1,z0,z1}: index box of the non-zero voxels of every bone's motion-weight channel."""
    boxes = torch.empty(nb, 6, device=vol.device, dtype=torch.int32)
    with _guard(vol):
        rc = _lib.lib().occnerf_bone_boxes(_chk(vol, torch.float32, 'vol'), int(nb), int(vol.shape[-1]), boxes.data_ptr(), _stream(vol))
    _lib.check(rc, 'bone_boxes')
    return boxes


def sample_warp(rays8, S, t_vals, Rs, Ts, vol, bbox_min, bbox_scale, t_rand=None, want_pts=False, boxes=None):
    """rays8[n,8] -> z_vals[n,S], x_skel[n*S,3], mask[n*S] (, pts[n*S,3]).  boxes (ops.bone_boxes of the same vol; render only:
    no jitter, no pts): bones that cannot reach a wave's samples are skipped, same bits."""
    n = rays8.shape[0]
    dev = rays8.device
    if boxes is not None and t_rand is None and not want_pts:
        z = torch.empty(n, S, device=dev, dtype=torch.float32)
        xs = torch.empty(n * S, 3, device=dev, dtype=torch.float32)
        mk = torch.empty(n * S, device=dev, dtype=torch.float32)
        _kmin, pmin = _host_f32(bbox_min, 3)
        _ksc, psc = _host_f32(bbox_scale, 3)
        with _guard_dev(dev):
            rc = _lib.lib().occnerf_sample_warp_culled(
                _chk(rays8, torch.float32, 'rays'), n, int(S), _chk(t_vals, torch.float32, 't_vals'),
                _chk(Rs, torch.float32, 'Rs'), _chk(Ts, torch.float32, 'Ts'), _chk(vol, torch.float32, 'vol'), int(Rs.shape[0]),
                int(vol.shape[-1]), _chk(boxes, torch.int32, 'boxes'), pmin, psc, z.data_ptr(), xs.data_ptr(), mk.data_ptr(),
                _stream(rays8))
        _lib.check(rc, 'sample_warp_culled')
        return z, xs, mk, None
    z = torch.empty(n, S, device=dev, dtype=torch.float32)
    xs = torch.empty(n * S, 3, device=dev, dtype=torch.float32)
    mk = torch.empty(n * S, device=dev, dtype=torch.float32)
    pts = torch.empty(n * S, 3, device=dev, dtype=torch.float32) if want_pts else None
    _kmin, pmin = _host_f32(bbox_min, 3)
    _ksc, psc = _host_f32(bbox_scale, 3)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_sample_warp(
            _chk(rays8, torch.float32, 'rays'), n, int(S), _chk(t_vals, torch.float32, 't_vals'),
            _opt(t_rand, torch.float32, 't_rand'), _chk(Rs, torch.float32, 'Rs'),
            _chk(Ts, torch.float32, 'Ts'), _chk(vol, torch.float32, 'vol'), int(Rs.shape[0]),
            int(vol.shape[-1]), pmin, psc, z.data_ptr(), None if pts is None else pts.data_ptr(),
            xs.data_ptr(), mk.data_ptr(), _stream(rays8))
    _lib.check(rc, 'sample_warp')
    return z, xs, mk, pts


def nonrigid_pack(weights, biases):
    dev = weights[0].device
    n = _lib.lib().occnerf_nonrigid_packed_floats()
    packed = torch.zeros(n, device=dev, dtype=torch.float32)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_nonrigid_pack(_ptr_table(weights, 'W'), _ptr_table(biases, 'b'),
                                              packed.data_ptr(), _stream(packed))
    _lib.check(rc, 'nonrigid_pack')
    return packed


def nonrigid(xyz, cond, hann, W0, b0, packed, out=None, direct=False):
    """direct=True: the 32-sample-wave direct-load kernel (cross-check / A-B timing)."""
    out = torch.empty_like(xyz) if out is None else out
    _kh, ph = _host_f32(hann, 6)
    fn = _lib.lib().occnerf_nonrigid_direct if direct else _lib.lib().occnerf_nonrigid
    with _guard(xyz):
        rc = fn(
            _chk(xyz, torch.float32, 'xyz'), xyz.shape[0], _chk(cond, torch.float32, 'cond'), ph,
            _chk(W0, torch.float32, 'W0'), _chk(b0, torch.float32, 'b0'),
            _chk(packed, torch.float32, 'packed'), _chk(out, torch.float32, 'xyz_out'), _stream(xyz))
    _lib.check(rc, 'nonrigid')
    return out


def nonrigid_rows(xyz, rows, count, cond, hann, W0, b0, packed):
    """Non-rigid offsets, in place, for the samples rows[0 .. count) only (list and count on the device)."""
    _kh, ph = _host_f32(hann, 6)
    with _guard(xyz):
        rc = _lib.lib().occnerf_nonrigid_rows(
            _chk(xyz, torch.float32, 'xyz'), rows.shape[0], _chk(rows, torch.int32, 'rows'),
            _chk(count, torch.int32, 'count'), _chk(cond, torch.float32, 'cond'), ph,
            _chk(W0, torch.float32, 'W0'), _chk(b0, torch.float32, 'b0'), _chk(packed, torch.float32, 'packed'),
            _stream(xyz))
    _lib.check(rc, 'nonrigid_rows')
    return xyz


def live_rows(mask):
    """-> rows int32[N] (first count entries valid: ascending indices with mask != 0), count int32[1]; no sync."""
    N = mask.shape[0]
    dev = mask.device
    nbytes = int(_lib.lib().occnerf_live_rows_temp_bytes(N))
    if nbytes < 0:
        raise RuntimeError('live_rows: temp size query failed')
    temp = torch.empty(max(nbytes, 16), device=dev, dtype=torch.uint8)
    rows = torch.empty(N, device=dev, dtype=torch.int32)
    count = torch.empty(1, device=dev, dtype=torch.int32)
    with _guard(mask):
        rc = _lib.lib().occnerf_live_rows(_chk(mask, torch.float32, 'mask'), N, rows.data_ptr(), count.data_ptr(),
                                          temp.data_ptr(), nbytes, _stream(mask))
    _lib.check(rc, 'live_rows')
    return rows, count


def scatter_raw(raw_c, rows, count, raw_full):
    with _guard(raw_c):
        rc = _lib.lib().occnerf_scatter_raw(_chk(raw_c, torch.float32, 'raw_c'), _chk(rows, torch.int32, 'rows'),
                                            _chk(count, torch.int32, 'count'), rows.shape[0],
                                            _chk(raw_full, torch.float32, 'raw_full'), _stream(raw_c))
    _lib.check(rc, 'scatter_raw')
    return raw_full


def repeat_heads(keys, key_cols, count, rows=None, want_mask=False):
    """Run-length elimination of repeated samples (see the header): keys float32/int32 [M, C] (row-contiguous), the first
    key_cols columns compared as bit patterns; entry m is row rows[m] (rows given) or m, for m < count[0] (device).
    -> scan int32[cap], heads int32[cap], head_count int32[1], head_mask float32[M] or None; cap = len(rows) or M."""
    dev = keys.device
    if keys.dim() != 2 or not keys.is_contiguous() or keys.element_size() != 4:
        raise RuntimeError('repeat_heads: keys must be a contiguous [M, C] tensor of 4-byte elements')
    cap = keys.shape[0] if rows is None else rows.shape[0]
    scan = torch.empty(cap, device=dev, dtype=torch.int32)
    heads = torch.empty(cap, device=dev, dtype=torch.int32)
    head_count = torch.empty(1, device=dev, dtype=torch.int32)
    head_mask = torch.zeros(keys.shape[0], device=dev, dtype=torch.float32) if want_mask else None
    if cap == 0:
        head_count.zero_()
        return scan, heads, head_count, head_mask
    nbytes = int(_lib.lib().occnerf_repeat_heads_temp_bytes(cap))
    if nbytes < 0:
        raise RuntimeError('repeat_heads: temp size query failed')
    temp = torch.empty(max(nbytes, 16), device=dev, dtype=torch.uint8)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_repeat_heads(
            keys.data_ptr(), keys.shape[1], int(key_cols), _opt(rows, torch.int32, 'rows'),
            _chk(count, torch.int32, 'count'), cap, scan.data_ptr(), heads.data_ptr(), head_count.data_ptr(),
            None if head_mask is None else head_mask.data_ptr(), temp.data_ptr(), nbytes, _stream(keys))
    _lib.check(rc, 'repeat_heads')
    return scan, heads, head_count, head_mask


def unique_heads(keys, key_cols, heads, count, scan=None, scan_count=None):
    """Distinct entries of a list (see the header): heads int32[cap] (or None: entry j is row j of keys, cap = rows of
    keys), count int32[1] on the device -> (heads_out int32[cap], count_out int32[1]); scan (optional, with scan_count) is
    rewritten in place from 1-based entry numbers to 1-based positions in heads_out."""
    dev = keys.device
    if keys.dim() != 2 or not keys.is_contiguous() or keys.element_size() != 4:
        raise RuntimeError('unique_heads: keys must be a contiguous [M, C] tensor of 4-byte elements')
    cap = keys.shape[0] if heads is None else heads.shape[0]
    heads_out = torch.empty(cap, device=dev, dtype=torch.int32)
    count_out = torch.empty(1, device=dev, dtype=torch.int32)
    if cap == 0:
        count_out.zero_()
        return heads_out, count_out
    nbytes = int(_lib.lib().occnerf_unique_heads_temp_bytes(cap))
    if nbytes <= 0:
        raise RuntimeError('unique_heads: temp size query failed')
    temp = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_unique_heads(
            keys.data_ptr(), keys.shape[1], int(key_cols), _opt(heads, torch.int32, 'heads'),
            _chk(count, torch.int32, 'count'), cap, heads_out.data_ptr(), count_out.data_ptr(),
            _opt(scan, torch.int32, 'scan'), _opt(scan_count, torch.int32, 'scan_count'),
            0 if scan is None else scan.shape[0], temp.data_ptr(), nbytes, _stream(keys))
    _lib.check(rc, 'unique_heads')
    return heads_out, count_out


def scatter_raw_heads(raw_h, raw_c, rows, count, scan_a, scan_b, raw_full):
    with _guard(raw_c):
        rc = _lib.lib().occnerf_scatter_raw_heads(
            _chk(raw_h, torch.float32, 'raw_h'), _chk(raw_c, torch.float32, 'raw_c'), _chk(rows, torch.int32, 'rows'),
            _chk(count, torch.int32, 'count'), _opt(scan_a, torch.int32, 'scan_a'), _opt(scan_b, torch.int32, 'scan_b'),
            rows.shape[0], _chk(raw_full, torch.float32, 'raw_full'), _stream(raw_c))
    _lib.check(rc, 'scatter_raw_heads')
    return raw_full


def nonrigid_pack_bf16(weights):
    dev = weights[0].device
    n = _lib.lib().occnerf_nonrigid_packed_bf16_bytes()
    packed = torch.zeros(n // 2, device=dev, dtype=torch.bfloat16)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_nonrigid_pack_bf16(_ptr_table(weights[:6], 'W'), packed.data_ptr(), _stream(packed))
    _lib.check(rc, 'nonrigid_pack_bf16')
    return packed


def nonrigid_pack_f16(weights):
    """Split-fp16 operand stream of the non-rigid MLP (cfg.mlp_precision = 'f16x3': csrc/split.h F16x3)."""
    dev = weights[0].device
    n = _lib.lib().occnerf_nonrigid_packed_bf16_bytes()               # same layout, 2-byte elements
    packed = torch.zeros(n // 2, device=dev, dtype=torch.float16)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_nonrigid_pack_f16(_ptr_table(weights[:6], 'W'), packed.data_ptr(), _stream(packed))
    _lib.check(rc, 'nonrigid_pack_f16')
    return packed


def _nonrigid_f16x3(xyz, rows, count, cond, hann, W0, b0, packed, packed_f16, out, domain_flag=None):
    _kh, ph = _host_f32(hann, 6)
    n_max = xyz.shape[0] if rows is None else rows.shape[0]
    with _guard(xyz):
        rc = _lib.lib().occnerf_nonrigid_f16x3(
            _chk(xyz, torch.float32, 'xyz'), n_max, _opt(rows, torch.int32, 'rows'), _opt(count, torch.int32, 'count'),
            _chk(cond, torch.float32, 'cond'), ph, _chk(W0, torch.float32, 'W0'), _chk(b0, torch.float32, 'b0'),
            _chk(packed, torch.float32, 'packed'), _chk(packed_f16, torch.float16, 'packed_f16'),
            _chk(out, torch.float32, 'xyz_out'), _opt(domain_flag, torch.int32, 'domain_flag'), _stream(xyz))
    _lib.check(rc, 'nonrigid_f16x3')
    return out


def nonrigid_bf16x3(xyz, cond, hann, W0, b0, packed, packed_bf16, out=None, domain_flag=None):
    """Split-operand non-rigid MLP; the dtype of the packed stream selects the split (bfloat16: bf16x3, float16: f16x3).
    domain_flag (int32[1] on the device, f16x3 only): bit 0 is set when a hidden activation reached the mode's clamp (4 094)."""
    out = torch.empty_like(xyz) if out is None else out
    if packed_bf16.dtype == torch.float16:
        return _nonrigid_f16x3(xyz, None, None, cond, hann, W0, b0, packed, packed_bf16, out, domain_flag)
    _kh, ph = _host_f32(hann, 6)
    with _guard(xyz):
        rc = _lib.lib().occnerf_nonrigid_bf16x3(
            _chk(xyz, torch.float32, 'xyz'), xyz.shape[0], _chk(cond, torch.float32, 'cond'), ph,
            _chk(W0, torch.float32, 'W0'), _chk(b0, torch.float32, 'b0'), _chk(packed, torch.float32, 'packed'),
            _chk(packed_bf16, torch.bfloat16, 'packed_bf16'), _chk(out, torch.float32, 'xyz_out'), _stream(xyz))
    _lib.check(rc, 'nonrigid_bf16x3')
    return out


def nonrigid_bf16x3_rows(xyz, rows, count, cond, hann, W0, b0, packed, packed_bf16, domain_flag=None):
    """In place on the listed samples (the split-operand counterpart of nonrigid_rows): xyz[rows[i]] += offset, i < count[0]."""
    if packed_bf16.dtype == torch.float16:
        return _nonrigid_f16x3(xyz, rows, count, cond, hann, W0, b0, packed, packed_bf16, xyz, domain_flag)
    _kh, ph = _host_f32(hann, 6)
    with _guard(xyz):
        rc = _lib.lib().occnerf_nonrigid_bf16x3_rows(
            _chk(xyz, torch.float32, 'xyz'), rows.shape[0], _chk(rows, torch.int32, 'rows'), _chk(count, torch.int32, 'count'),
            _chk(cond, torch.float32, 'cond'), ph, _chk(W0, torch.float32, 'W0'), _chk(b0, torch.float32, 'b0'),
            _chk(packed, torch.float32, 'packed'), _chk(packed_bf16, torch.bfloat16, 'packed_bf16'), _stream(xyz))
    _lib.check(rc, 'nonrigid_bf16x3_rows')
    return xyz


def msknn(xyz, points, index_map, scale_begin, seed_from_coarser):
    N = xyz.shape[0]
    nscale = len(scale_begin) - 1
    out = torch.empty(N, nscale, 10, device=xyz.device, dtype=torch.int32)
    _kb, pb = _host_i32(scale_begin)
    _ks, ps = _host_i32(seed_from_coarser)
    with _guard(xyz):
        rc = _lib.lib().occnerf_msknn(_chk(xyz, torch.float32, 'xyz'), N,
                                      _chk(points, torch.float32, 'points'),
                                      _chk(index_map, torch.int32, 'index_map'), pb, ps, nscale,
                                      out.data_ptr(), _stream(xyz))
    _lib.check(rc, 'msknn')
    return out


def knn_center(c, points, index_map, scale_begin):
    """c[3] (device) against the brute-force multi-scale layout (ops.msknn's points / index_map / scale_begin) ->
    (center[4] = (c, r^2), idx[nscale, 10]): c's neighbours and the squared radius inside which every query provably has the same
    neighbours in the same order (0: no such radius).  For msknn_clustered(center=...)."""
    nscale = len(scale_begin) - 1
    center = torch.empty(4, device=c.device, dtype=torch.float32)
    idx = torch.empty(nscale, 10, device=c.device, dtype=torch.int32)
    _kb, pb = _host_i32(scale_begin)
    with _guard(c):
        rc = _lib.lib().occnerf_knn_center(_chk(c, torch.float32, 'c'), _chk(points, torch.float32, 'points'),
                                           _chk(index_map, torch.int32, 'index_map'), pb, nscale, center.data_ptr(), idx.data_ptr(),
                                           _stream(c))
    _lib.check(rc, 'knn_center')
    return center, idx


def msknn_clustered(xyz, n_rays, S, cl, seed_from_coarser, mask=None, rows=None, count=None, out=None, center=None):
    """cl: device-side cluster layout (dict, see Network._context / geometry.build_knn_clusters).
    mask[n_rays*S] (optional): samples with mask == 0 are skipped, their output rows left unwritten.
    rows / count (optional, instead of mask): ascending int32 list of the samples to query and its length on the device.
    center (optional): ops.knn_center's pair -- queries inside its radius take the cached indices (same results)."""
    nscale = int(cl['ranges'].shape[0]) + 1
    if out is None:
        out = torch.empty(n_rays * S, nscale, 10, device=xyz.device, dtype=torch.int32)
    if (rows is None) != (count is None) or (rows is not None and mask is not None):
        raise RuntimeError('msknn_clustered: rows and count come together, instead of mask')
    ray_start = torch.empty(n_rays + 1, device=xyz.device, dtype=torch.int32) if rows is not None else None
    _kc, pc = _host_i32(cl['coarse_rows'])
    _ks, ps = _host_i32(seed_from_coarser)
    with _guard(xyz):
        rc = _lib.lib().occnerf_msknn_clustered_centered(
            _chk(xyz, torch.float32, 'xyz'), _opt(mask, torch.float32, 'mask'), int(n_rays), int(S),
            _chk(cl['points'], torch.float32, 'points'), _chk(cl['centers'], torch.float32, 'centers'), _chk(cl['ranges'], torch.int32, 'cluster_ranges'),
            _chk(cl['radius'], torch.float32, 'cluster_radius'), int(cl['ncl']),
            _opt(cl.get('group_centers'), torch.float32, 'group_centers'), _opt(cl.get('group_ranges'), torch.int32, 'group_ranges'),
            _opt(cl.get('group_radius'), torch.float32, 'group_radius'), int(cl.get('ngrp', 0)), pc, ps, nscale,
            _opt(rows, torch.int32, 'rows'), _opt(count, torch.int32, 'count'),
            None if ray_start is None else ray_start.data_ptr(),
            None if center is None else _chk(center[0], torch.float32, 'center'),
            None if center is None else _chk(center[1], torch.int32, 'center_idx'), out.data_ptr(), _stream(xyz))
    _lib.check(rc, 'msknn_clustered')
    return out


def knn_small(q, s, k):
    out = torch.empty(q.shape[0], k, device=q.device, dtype=torch.int32)
    with _guard(q):
        rc = _lib.lib().occnerf_knn_small(_chk(q, torch.float32, 'q'), q.shape[0],
                                          _chk(s, torch.float32, 's'), s.shape[0], int(k),
                                          out.data_ptr(), _stream(q))
    _lib.check(rc, 'knn_small')
    return out


def unit_normals(normals):
    out = torch.empty_like(normals)
    with _guard(normals):
        rc = _lib.lib().occnerf_unit_normals(_chk(normals, torch.float64, 'normals'), normals.shape[0],
                                             out.data_ptr(), _stream(normals))
    _lib.check(rc, 'unit_normals')
    return out


def point_sdf(point_cloud, point_base, normals, unit, kidx):
    P = point_cloud.shape[0]
    kb = torch.empty(P, 3, device=point_cloud.device, dtype=torch.float64)
    dist = torch.empty(P, device=point_cloud.device, dtype=torch.float32)
    with _guard(point_cloud):
        rc = _lib.lib().occnerf_point_sdf(
            _chk(point_cloud, torch.float32, 'point_cloud'), _chk(point_base, torch.float32, 'point_base'),
            _chk(normals, torch.float64, 'normals'), _chk(unit, torch.float64, 'unit_normals'),
            _chk(kidx, torch.int32, 'kidx'), P, kb.data_ptr(), dist.data_ptr(), _stream(point_cloud))
    _lib.check(rc, 'point_sdf')
    return kb, dist


def table_stride():
    """Row pitch (floats) of the per-point feature table."""
    return int(_lib.lib().occnerf_point_table_stride())


def point_table(knn_base, sdf, learnable, bound32, two_bound32, embeddings, offsets, S, H):
    P = knn_base.shape[0]
    table = torch.empty(P, table_stride(), device=knn_base.device, dtype=torch.float32)
    with _guard(knn_base):
        rc = _lib.lib().occnerf_point_table(
            _chk(knn_base, torch.float64, 'knn_base'), _chk(sdf, torch.float32, 'point_sdf'),
            _chk(learnable, torch.float32, 'learnable'), P, float(bound32), float(two_bound32),
            _chk(embeddings, torch.float32, 'embeddings'), _chk(offsets, torch.int32, 'offsets'),
            _host_offsets(offsets), int(offsets.shape[0] - 1), float(S), int(H), table.data_ptr(),
            _stream(knn_base))
    _lib.check(rc, 'point_table')
    return table


def point_pack(point_base, normals, unit, counter, table):
    """Per-point records of the 8-lanes-per-sample feature kernel: (geo[P,16], tail[P,4]) -- see the header."""
    P, dev = point_base.shape[0], point_base.device
    geo = torch.empty(P, 16, device=dev, dtype=torch.float32)
    tail = torch.empty(P, 4, device=dev, dtype=torch.float32)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_point_pack(
            _chk(point_base, torch.float32, 'point_base'), _chk(normals, torch.float64, 'normals'),
            _chk(unit, torch.float64, 'unit_normals'), _chk(counter, torch.float32, 'counter'),
            _chk(table, torch.float32, 'table'), P, geo.data_ptr(), tail.data_ptr(), _stream(point_base))
    _lib.check(rc, 'point_pack')
    return geo, tail


def center_row(mlp_in_row, enc_in_row):
    """The 72 floats sample_features(center_agg=...) takes: [columns 0..35 of the centre sample's mlp_in | its encoder input x[4] |
    its columns 36..67 (the 32 encoded features of x)]."""
    return torch.cat([mlp_in_row[:36], enc_in_row[:4], mlp_in_row[36:68]]).contiguous()


def sample_features(xyz, knn_idxs, point_base, normals, unit, counter, table, bound32, two_bound32,
                    embeddings, offsets, S, H, raw=None, want_enc_in=False, geo_idxs=None,
                    att_in=None, rows=None, count=None, pack=None, center=None, center_agg=None):
    """rows (int32[M], optional): compact list of samples to evaluate; outputs then have M rows.
    center / center_agg (optional, renderer's path): ops.knn_center's [4] and ops.center_row of a sample with the centre's
    neighbour lists -- groups of samples inside the radius copy its aggregate columns instead of gathering rows, and its encoded
    columns too when their encoder input is bitwise the centre's (same bits).  FRESHNESS CONTRACT (not checkable here):
    `center` must come from ops.knn_center on the SAME point set the kNN indices were searched in, and `center_agg` from this
    function on a sample at that centre with the SAME table, counter, embeddings and pack -- stale ones give silently wrong
    features.  Network computes both once per frame from the frame's own table (`_knn_center`) and checks the table's
    identity and version before every use (`_stage_features`).
    count (int32[1] on the device, optional, with rows): the list's real length; M is then a capacity.
    pack: point_pack(...) of the same per-point inputs (built here when the renderer's kernel applies and the caller
    did not cache it)."""
    N = xyz.shape[0] if rows is None else rows.shape[0]
    dev = xyz.device
    if (pack is None and N > 0 and knn_idxs.shape[1] == 4 and geo_idxs is None and att_in is None
            and counter is not None and table.dim() == 2 and table.shape[1] == table_stride()):
        pack = point_pack(point_base, normals, unit, counter, table)
    geo, tail = pack if pack is not None else (None, None)
    if table.dim() != 2 or table.shape[1] != table_stride():
        raise RuntimeError(f'sample_features: table must be [P,{table_stride()}] (ops.point_table), got {tuple(table.shape)}')
    mlp_in = torch.empty(N, 68, device=dev, dtype=torch.float32)
    raw = torch.empty(N, 5, device=dev, dtype=torch.float32) if raw is None else raw
    enc_in = torch.empty(N, 4, device=dev, dtype=torch.float32) if want_enc_in else None
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_sample_features_centered(
            _chk(xyz, torch.float32, 'xyz'), N, _chk(knn_idxs, torch.int32, 'knn_idxs'),
            int(knn_idxs.shape[1]), _chk(point_base, torch.float32, 'point_base'),
            _chk(normals, torch.float64, 'normals'), _chk(unit, torch.float64, 'unit_normals'),
            _opt(counter, torch.float32, 'counter'), _chk(table, torch.float32, 'table'),
            float(bound32), float(two_bound32), _chk(embeddings, torch.float32, 'embeddings'),
            _chk(offsets, torch.int32, 'offsets'), _host_offsets(offsets), int(offsets.shape[0] - 1),
            float(S), int(H), _opt(geo_idxs, torch.int32, 'geo_idxs'), _opt(att_in, torch.float32, 'att_in'),
            _opt(rows, torch.int32, 'rows'), _opt(count, torch.int32, 'count'),
            _opt(geo, torch.float32, 'point_geo'), _opt(tail, torch.float32, 'point_tail'),
            0 if geo is None else int(geo.shape[0]),
            _opt(center, torch.float32, 'center'), _opt(center_agg, torch.float32, 'center_agg'),
            mlp_in.data_ptr(), _chk(raw, torch.float32, 'raw'),
            None if enc_in is None else enc_in.data_ptr(), _stream(xyz))
    _lib.check(rc, 'sample_features')
    return mlp_in, raw, enc_in


def canonical_mlp_pack(weights, biases):
    """weights/biases: the 10 Linear layers in module order (see the header)."""
    dev = weights[0].device
    n = _lib.lib().occnerf_canonical_mlp_packed_floats()
    packed = torch.zeros(n, device=dev, dtype=torch.float32)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_canonical_mlp_pack(_ptr_table(weights, 'W'), _ptr_table(biases, 'b'),
                                                   packed.data_ptr(), _stream(packed))
    _lib.check(rc, 'canonical_mlp_pack')
    return packed


def canonical_mlp(mlp_in, packed, raw, direct=False, count=None, in_rows=None):
    """fp32 MLP trunks.  direct=True: the 32-sample-wave direct-load kernel (cross-check / A-B timing).
    count (int32[1] on the device): only the first count rows exist; the host does not know the number.
    in_rows (int32, with count): output row n is computed from input row in_rows[n]."""
    if in_rows is not None:
        with _guard(mlp_in):
            rc = _lib.lib().occnerf_canonical_mlp_rows(
                _chk(mlp_in, torch.float32, 'mlp_in'), _chk(in_rows, torch.int32, 'in_rows'), in_rows.shape[0],
                _chk(count, torch.int32, 'count'), _chk(packed, torch.float32, 'packed'),
                _chk(raw, torch.float32, 'raw'), _stream(mlp_in))
        _lib.check(rc, 'canonical_mlp_rows')
        return raw
    if count is not None:
        with _guard(mlp_in):
            rc = _lib.lib().occnerf_canonical_mlp_counted(
                _chk(mlp_in, torch.float32, 'mlp_in'), mlp_in.shape[0], _chk(count, torch.int32, 'count'),
                _chk(packed, torch.float32, 'packed'), _chk(raw, torch.float32, 'raw'), _stream(mlp_in))
        _lib.check(rc, 'canonical_mlp_counted')
        return raw
    fn = _lib.lib().occnerf_canonical_mlp_direct if direct else _lib.lib().occnerf_canonical_mlp
    with _guard(mlp_in):
        rc = fn(_chk(mlp_in, torch.float32, 'mlp_in'), mlp_in.shape[0], _chk(packed, torch.float32, 'packed'),
                _chk(raw, torch.float32, 'raw'), _stream(mlp_in))
    _lib.check(rc, 'canonical_mlp')
    return raw


def trunks_pack_bf16(weights, out=None):
    """Plain-bf16 operand stream of the nine MFMA layers for the training step's fused forward (csrc/trunks.hip)."""
    dev = weights[0].device
    if out is None:
        out = torch.zeros(_lib.lib().occnerf_trunks_packed_bytes() // 2, device=dev, dtype=torch.bfloat16)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_trunks_pack_bf16(_ptr_table(weights[:9], 'W'), out.data_ptr(), _stream(out))
    _lib.check(rc, 'trunks_pack_bf16')
    return out


def trunks_forward_bf16(agg, var, enc, packed_f32, packed_bf16):
    """Both trunks forward in one kernel, the activations the backward needs written on the way (csrc/trunks.hip).
    -> X0[M,96], [A1..A4][M,256], GEO[M,96], [B1..B4][M,256] (bf16, row-major), raw4[M,4] fp32."""
    M, dev, bf = agg.shape[0], agg.device, torch.bfloat16
    X0 = torch.empty(M, 96, device=dev, dtype=bf)
    A = [torch.empty(M, 256, device=dev, dtype=bf) for _ in range(4)]
    GEO = torch.empty(M, 96, device=dev, dtype=bf)
    B = [torch.empty(M, 256, device=dev, dtype=bf) for _ in range(4)]
    raw4 = torch.empty(M, 4, device=dev, dtype=torch.float32)
    pa = (C.c_void_p * 4)(*[t.data_ptr() for t in A])
    pb = (C.c_void_p * 4)(*[t.data_ptr() for t in B])
    with _guard(agg):
        rc = _lib.lib().occnerf_trunks_forward_bf16(
            _chk(agg, torch.float32, 'agg'), _chk(var, torch.float32, 'var'), _chk(enc, torch.float32, 'enc'), M,
            _chk(packed_f32, torch.float32, 'packed_f32'), _chk(packed_bf16, torch.bfloat16, 'packed_bf16'), X0.data_ptr(), pa,
            GEO.data_ptr(), pb, raw4.data_ptr(), _stream(agg))
    _lib.check(rc, 'trunks_forward_bf16')
    return X0, A, GEO, B, raw4


def canonical_mlp_pack_bf16(weights):
    """hi/lo bf16 split of the 10 weight matrices in bf16-MFMA operand order."""
    dev = weights[0].device
    n = _lib.lib().occnerf_canonical_mlp_packed_bf16_bytes()
    packed = torch.zeros(n // 2, device=dev, dtype=torch.bfloat16)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_canonical_mlp_pack_bf16(_ptr_table(weights, 'W'), packed.data_ptr(),
                                                        _stream(packed))
    _lib.check(rc, 'canonical_mlp_pack_bf16')
    return packed


def canonical_mlp_pack_f16(weights):
    """Split-fp16 operand stream of the canonical MLP (cfg.mlp_precision = 'f16x3': csrc/split.h F16x3): Wh = f16(W),
    Wl' = f16((W - Wh) 2^11), in fp16-MFMA operand order."""
    dev = weights[0].device
    n = _lib.lib().occnerf_canonical_mlp_packed_bf16_bytes()          # same layout, 2-byte elements
    packed = torch.zeros(n // 2, device=dev, dtype=torch.float16)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_canonical_mlp_pack_f16(_ptr_table(weights, 'W'), packed.data_ptr(), _stream(packed))
    _lib.check(rc, 'canonical_mlp_pack_f16')
    return packed


def canonical_mlp_bf16x3(mlp_in, packed, packed_bf16, raw, variant=0, count=None, in_rows=None, domain_flag=None):
    """Split-operand canonical MLP; the dtype of the packed stream selects the split (bfloat16: bf16x3, float16: f16x3).
    count (int32[1] on the device): entries to evaluate, read by the kernel (the launch covers the worst case);
    in_rows (int32, optional): entry n takes input row in_rows[n]; results are compact (raw[n]).
    domain_flag (int32[1] on the device, f16x3 only): bit 0 is set when a value reached the mode's clamp (csrc/split.h)."""
    if packed_bf16.dtype == torch.float16:
        n_max = mlp_in.shape[0] if in_rows is None else in_rows.shape[0]
        with _guard(mlp_in):
            rc = _lib.lib().occnerf_canonical_mlp_f16x3(
                _chk(mlp_in, torch.float32, 'mlp_in'), _opt(in_rows, torch.int32, 'in_rows'), n_max,
                _opt(count, torch.int32, 'count'), _chk(packed, torch.float32, 'packed'),
                _chk(packed_bf16, torch.float16, 'packed_f16'), _chk(raw, torch.float32, 'raw'),
                _opt(domain_flag, torch.int32, 'domain_flag'), _stream(mlp_in))
        _lib.check(rc, 'canonical_mlp_f16x3')
        return raw
    with _guard(mlp_in):
        if count is None:
            assert in_rows is None
            rc = _lib.lib().occnerf_canonical_mlp_bf16x3(
                _chk(mlp_in, torch.float32, 'mlp_in'), mlp_in.shape[0], _chk(packed, torch.float32, 'packed'),
                _chk(packed_bf16, torch.bfloat16, 'packed_bf16'), _chk(raw, torch.float32, 'raw'),
                int(variant), _stream(mlp_in))
        else:
            n_max = mlp_in.shape[0] if in_rows is None else in_rows.shape[0]
            rc = _lib.lib().occnerf_canonical_mlp_bf16x3_rows(
                _chk(mlp_in, torch.float32, 'mlp_in'), _opt(in_rows, torch.int32, 'in_rows'), n_max,
                _chk(count, torch.int32, 'count'), _chk(packed, torch.float32, 'packed'),
                _chk(packed_bf16, torch.bfloat16, 'packed_bf16'), _chk(raw, torch.float32, 'raw'), int(variant),
                _stream(mlp_in))
    _lib.check(rc, 'canonical_mlp_bf16x3')
    return raw


def composite(raw, mask, z_vals, rays8, bgcolor, want_weights=False, want_term=False, out=None, out_rows=None):
    """out_rows (int64[n], optional): results of ray r are written to row out_rows[r] of `out` = (rgb, acc, depth)
    (caller-allocated, at least max(out_rows)+1 rows; without out_rows: exactly n rows)."""
    n, S = z_vals.shape
    dev = raw.device
    if out is None:
        if out_rows is not None:
            raise RuntimeError('composite: out_rows needs caller-allocated outputs')
        out = (torch.empty(n, 3, device=dev, dtype=torch.float32), torch.empty(n, device=dev, dtype=torch.float32),
               torch.empty(n, device=dev, dtype=torch.float32))
    rgb, acc, dep = out
    for t, nm in ((rgb, 'rgb'), (acc, 'acc'), (dep, 'depth')):
        _chk(t, torch.float32, nm)
    w = torch.empty(n, S, device=dev, dtype=torch.float32) if want_weights else None
    tp = torch.empty(n, device=dev, dtype=torch.int32) if want_term else None
    _kbg, pbg = _host_f32(bgcolor, 3)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_composite(
            _chk(raw, torch.float32, 'raw'), _chk(mask, torch.float32, 'mask'),
            _chk(z_vals, torch.float32, 'z_vals'), _chk(rays8, torch.float32, 'rays'), pbg, n, int(S),
            rgb.data_ptr(), acc.data_ptr(), dep.data_ptr(), None if w is None else w.data_ptr(),
            None if tp is None else tp.data_ptr(), _opt(out_rows, torch.int64, 'out_rows'), _stream(raw))
    _lib.check(rc, 'composite')
    return rgb, acc, dep, w, tp


# ------------------------------------------------------------------ per-frame preamble
def pose_motion_bases(pose_decoder, posevec, refine, dst_Rs, dst_Ts, cnl_gtfms):
    """a2 + a3 in one launch -> Rs[24,3,3], Ts[24,3] (float32, on the device)."""
    import torch.nn as nn
    dev = dst_Rs.device
    lin = [m for m in pose_decoder.block_mlps if isinstance(m, nn.Linear)]
    if len(lin) != 5 or lin[0].in_features != 69 or lin[0].out_features != 256 or lin[4].out_features != 69:
        raise RuntimeError('pose_motion_bases: the fused kernel is built for the 69 -> 256 x4 -> 69 refiner of occnerf.yaml')
    Rs = torch.empty(24, 3, 3, device=dev, dtype=torch.float32)
    Ts = torch.empty(24, 3, device=dev, dtype=torch.float32)
    with _guard_dev(dev):
        rc = _lib.lib().occnerf_pose_motion_bases(
            _ptr_table([m.weight.detach() for m in lin], 'W'), _ptr_table([m.bias.detach() for m in lin], 'b'),
            _chk(posevec, torch.float32, 'posevec'), int(bool(refine)), _chk(dst_Rs, torch.float32, 'dst_Rs'),
            _chk(dst_Ts, torch.float32, 'dst_Ts'), _chk(cnl_gtfms, torch.float32, 'cnl_gtfms'), Rs.data_ptr(),
            Ts.data_ptr(), _stream(dst_Rs))
    _lib.check(rc, 'pose_motion_bases')
    return Rs, Ts


def prior_softmax(decoded, prior):
    """softmax over channels of decoded[C,...] + log(prior[C,...]) -> vol, same shape."""
    Cn = decoded.shape[0]
    V = decoded.numel() // Cn
    vol = torch.empty_like(decoded)
    with _guard(decoded):
        rc = _lib.lib().occnerf_prior_softmax(_chk(decoded, torch.float32, 'decoded'), _chk(prior, torch.float32, 'prior'),
                                              int(Cn), int(V), vol.data_ptr(), _stream(decoded))
    _lib.check(rc, 'prior_softmax')
    return vol


def ray_order(dirs):
    """dirs: [R,3] float32 GPU tensor (any row stride, unit element stride: a slice of rays8 works) -> int64[R], the Morton
    walk of the rays (csrc/rays.hip; occnerf_amd/rayorder.py is the same construction in torch ops).  No sync."""
    R = dirs.shape[0]
    order = torch.empty(R, device=dirs.device, dtype=torch.int64)
    if R == 0:
        return order
    if not dirs.is_cuda or dirs.dtype != torch.float32 or dirs.dim() != 2 or dirs.shape[1] != 3 or dirs.stride(1) != 1:
        raise RuntimeError('ray_order: dirs must be a float32 [R,3] GPU tensor with unit element stride')
    nbytes = int(_lib.lib().occnerf_ray_order_temp_bytes(R))
    if nbytes < 0:
        raise RuntimeError('ray_order: temp size query failed')
    temp = torch.empty(nbytes, device=dirs.device, dtype=torch.uint8)
    with _guard(dirs):
        rc = _lib.lib().occnerf_ray_order(dirs.data_ptr(), R, int(dirs.stride(0)), order.data_ptr(), temp.data_ptr(), nbytes,
                                          _stream(dirs))
    _lib.check(rc, 'ray_order')
    return order


def pack_rays(rays, near, far, order=None):
    """rays[2,R,3], near[R,1], far[R,1] -> rays8[R,8] in `order` (int64 permutation, optional)."""
    R = rays.shape[1]
    rays8 = torch.empty(R, 8, device=rays.device, dtype=torch.float32)
    with _guard(rays):
        rc = _lib.lib().occnerf_pack_rays(_chk(rays, torch.float32, 'rays'), _chk(near, torch.float32, 'near'),
                                          _chk(far, torch.float32, 'far'), _opt(order, torch.int64, 'order'), R,
                                          rays8.data_ptr(), _stream(rays))
    _lib.check(rc, 'pack_rays')
    return rays8


# ------------------------------------------------------------------ training path: neighbour aggregation
def agg_forward(feats, knn, atts):
    """agg[n] = sum_j atts[n,j] * feats[knn[n,j]]  (feats[P,F] fp32, knn[N,K] int32, atts[N,K] fp32)."""
    N, K = knn.shape
    F = feats.shape[1]
    agg = torch.empty(N, F, device=feats.device, dtype=torch.float32)
    with _guard(feats):
        rc = _lib.lib().occnerf_agg_forward(_chk(feats, torch.float32, 'feats'), int(F), _chk(knn, torch.int32, 'knn'),
                                            _chk(atts, torch.float32, 'atts'), N, int(K), agg.data_ptr(), _stream(feats))
    _lib.check(rc, 'agg_forward')
    return agg


def agg_backward(grad_agg, knn, atts, P):
    N, K = knn.shape
    F = grad_agg.shape[1]
    W = int(_lib.lib().occnerf_agg_backward_slices(N))
    partial = torch.empty(W, P, F, device=grad_agg.device, dtype=torch.float32)
    nbytes = int(_lib.lib().occnerf_agg_backward_scratch_bytes(N, int(F)))
    scratch = torch.empty(max(nbytes, 16), device=grad_agg.device, dtype=torch.uint8)
    with _guard(grad_agg):
        rc = _lib.lib().occnerf_agg_backward(_chk(grad_agg, torch.float32, 'grad_agg'), int(F),
                                             _chk(knn, torch.int32, 'knn'), _chk(atts, torch.float32, 'atts'), N,
                                             int(K), int(P), partial.data_ptr(), scratch.data_ptr(), nbytes,
                                             _stream(grad_agg))
    _lib.check(rc, 'agg_backward')
    return partial.sum(0)


class _Aggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, knn, atts):
        feats = feats.contiguous()
        ctx.save_for_backward(knn, atts)
        ctx.P = feats.shape[0]
        return agg_forward(feats, knn, atts)

    @staticmethod
    def backward(ctx, grad_agg):
        knn, atts = ctx.saved_tensors
        return agg_backward(grad_agg.contiguous(), knn, atts, ctx.P), None, None


def aggregate(feats, knn, atts):
    """Differentiable (w.r.t. feats) weighted neighbour sum; knn and atts carry no gradient.  Runs of consecutive samples whose
    ids AND weights are bitwise identical are evaluated / scattered once (exact for arbitrary atts).  Limits of the backward,
    refused by name: K <= 64 neighbours, F <= 64 columns, P <= 32 x min(1024, 18432 // F) rows (16 832 at F = 35)."""
    return _Aggregate.apply(feats, knn.contiguous(), atts.contiguous())
