"""numpy/ctypes front end of the C oracle (oracle/occnerf_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Each wrapper takes and returns
plain numpy arrays; shapes follow SURVEY.md section 8(a).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'liboccnerf_oracle.so')
_lib = None

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)


def build():
    """Compile the oracle with gcc (oracle/Makefile)."""
    subprocess.check_call(['make', '-s', '-C', _HERE])


def effective_cpus():
    """CPUs this process may really use: the affinity mask, cut by the cgroup's CPU quota when there is one (libgomp and torch
    size their pools from the mask alone; under a quota that is oversubscription with spinning barriers)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def set_threads(n):
    """Size the oracle's OpenMP team (libgomp is in the process once the oracle is loaded)."""
    lib()
    try:
        C.CDLL('libgomp.so.1').omp_set_num_threads(int(n))
    except OSError:
        pass
    return int(n)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        # (before libgomp initialises: idle team members sleep instead of spinning -- the tests alternate between OpenMP
        # regions, torch's own pool and GPU waits, and a spinning team of every host core starves all three)
        os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')
        os.environ.setdefault('OMP_NUM_THREADS', str(effective_cpus()))
        _lib = C.CDLL(_LIB_PATH)
        _lib.oc_version.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _ptr_array(arrs):
    """float** from a list of float32 arrays (keeps them alive via the return)."""
    arrs = [_f32(a) for a in arrs]
    tab = (_f32p * len(arrs))(*[_p(a, _f32p) for a in arrs])
    return tab, arrs


def grid_level_params(L, S, H):
    scale = np.zeros(L, np.float32)
    res = np.zeros(L, np.uint32)
    lib().oc_grid_level_params(C.c_uint32(L), C.c_float(S), C.c_uint32(H), _p(scale, _f32p),
                               _p(res, _u32p))
    return scale, res


def grid_encode_forward(inputs, embeddings, offsets, S, H, want_dy_dx=False, gridtype=0,
                        align_corners=False, interp=0):
    """-> outputs[L,B,C] (level-major like the reference op), dy_dx[B,L*D*C] or None."""
    inputs, embeddings, offsets = _f32(inputs), _f32(embeddings), _i32(offsets)
    B, D = inputs.shape
    Cc = embeddings.shape[1]
    L = offsets.shape[0] - 1
    out = np.empty((L, B, Cc), np.float32)
    dy = np.empty((B, L * D * Cc), np.float32) if want_dy_dx else None
    lib().oc_grid_encode_forward(_p(inputs, _f32p), _p(embeddings, _f32p), _p(offsets, _i32p),
                                 _p(out, _f32p), C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc),
                                 C.c_uint32(L), C.c_float(S), C.c_uint32(H), _p(dy, _f32p),
                                 C.c_uint32(gridtype), C.c_int(int(align_corners)),
                                 C.c_uint32(interp))
    return out, dy


def grid_encode_backward(grad, inputs, offsets, n_emb, Cc, S, H, dy_dx=None, gridtype=0,
                         align_corners=False, interp=0):
    """grad[L,B,C] -> grad_embeddings[n_emb,C], grad_inputs[B,D] or None."""
    grad, inputs, offsets = _f32(grad), _f32(inputs), _i32(offsets)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    ge = np.zeros((n_emb, Cc), np.float32)
    gi = np.zeros((B, D), np.float32) if dy_dx is not None else None
    dy = _f32(dy_dx) if dy_dx is not None else None
    lib().oc_grid_encode_backward(_p(grad, _f32p), _p(inputs, _f32p), _p(offsets, _i32p),
                                  _p(ge, _f32p), C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc),
                                  C.c_uint32(L), C.c_float(S), C.c_uint32(H), _p(dy, _f32p),
                                  _p(gi, _f32p), C.c_uint32(gridtype),
                                  C.c_int(int(align_corners)), C.c_uint32(interp))
    return ge, gi


def grid_encode_forward_f64(inputs, embeddings, offsets, S, H, want_dy_dx=False, gridtype=0, align_corners=False, interp=0):
    """gridencoder.cu:467's double dispatch case: float64 embeddings -> float64 outputs[L,B,C] (, dy_dx[B,L*D*C]); float inputs."""
    inputs, embeddings, offsets = _f32(inputs), _f64(embeddings), _i32(offsets)
    B, D = inputs.shape
    Cc = embeddings.shape[1]
    L = offsets.shape[0] - 1
    out = np.empty((L, B, Cc), np.float64)
    dy = np.empty((B, L * D * Cc), np.float64) if want_dy_dx else None
    lib().oc_grid_encode_forward_f64(_p(inputs, _f32p), _p(embeddings, _f64p), _p(offsets, _i32p), _p(out, _f64p),
                                     C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc), C.c_uint32(L), C.c_float(S),
                                     C.c_uint32(H), _p(dy, _f64p), C.c_uint32(gridtype), C.c_int(int(align_corners)),
                                     C.c_uint32(interp))
    return out, dy


def grid_encode_backward_f64(grad, inputs, offsets, n_emb, Cc, S, H, dy_dx=None, gridtype=0, align_corners=False, interp=0):
    """float64 grad[L,B,C] -> float64 grad_embeddings[n_emb,C], grad_inputs[B,D] or None."""
    grad, inputs, offsets = _f64(grad), _f32(inputs), _i32(offsets)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    ge = np.zeros((n_emb, Cc), np.float64)
    gi = np.zeros((B, D), np.float64) if dy_dx is not None else None
    dy = _f64(dy_dx) if dy_dx is not None else None
    lib().oc_grid_encode_backward_f64(_p(grad, _f64p), _p(inputs, _f32p), _p(offsets, _i32p), _p(ge, _f64p), C.c_uint32(B),
                                      C.c_uint32(D), C.c_uint32(Cc), C.c_uint32(L), C.c_float(S), C.c_uint32(H),
                                      _p(dy, _f64p), _p(gi, _f64p), C.c_uint32(gridtype), C.c_int(int(align_corners)),
                                      C.c_uint32(interp))
    return ge, gi


_u16p = C.POINTER(C.c_uint16)


def _f16(a):
    return np.ascontiguousarray(a, dtype=np.float16)


def grid_encode_forward_f16(inputs, embeddings, offsets, S, H, want_dy_dx=False, gridtype=0, align_corners=False, interp=0):
    """gridencoder.cu:467's at::Half dispatch case: float16 embeddings -> float16 outputs[L,B,C] (, dy_dx[B,L*D*C])."""
    inputs, embeddings, offsets = _f32(inputs), _f16(embeddings), _i32(offsets)
    B, D = inputs.shape
    Cc = embeddings.shape[1]
    L = offsets.shape[0] - 1
    out = np.empty((L, B, Cc), np.float16)
    dy = np.empty((B, L * D * Cc), np.float16) if want_dy_dx else None
    lib().oc_grid_encode_forward_f16(_p(inputs, _f32p), _p(embeddings, _u16p), _p(offsets, _i32p), _p(out, _u16p),
                                     C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc), C.c_uint32(L), C.c_float(S),
                                     C.c_uint32(H), _p(dy, _u16p), C.c_uint32(gridtype), C.c_int(int(align_corners)),
                                     C.c_uint32(interp))
    return out, dy


def grid_encode_backward_f16(grad, inputs, offsets, n_emb, Cc, S, H, dy_dx=None, gridtype=0, align_corners=False, interp=0):
    """float16 grad[L,B,C] -> float16 grad_embeddings[n_emb,C] (sequential half accumulation), grad_inputs[B,D] or None."""
    grad, inputs, offsets = _f16(grad), _f32(inputs), _i32(offsets)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    ge = np.zeros((n_emb, Cc), np.float16)
    gi = np.zeros((B, D), np.float16) if dy_dx is not None else None
    dy = _f16(dy_dx) if dy_dx is not None else None
    lib().oc_grid_encode_backward_f16(_p(grad, _u16p), _p(inputs, _f32p), _p(offsets, _i32p), _p(ge, _u16p), C.c_uint32(B),
                                      C.c_uint32(D), C.c_uint32(Cc), C.c_uint32(L), C.c_float(S), C.c_uint32(H),
                                      _p(dy, _u16p), _p(gi, _u16p), C.c_uint32(gridtype), C.c_int(int(align_corners)),
                                      C.c_uint32(interp))
    return ge, gi


def half_roundtrip(values):
    """float32 -> float16 -> float32 through the oracle's own conversions (checked against numpy's in the tests)."""
    f = _f32(values).ravel()
    h = np.empty(f.shape, np.uint16)
    lib().oc_float_to_half(_p(f, _f32p), _p(h, _u16p), C.c_int64(f.size))
    back = np.empty(f.shape, np.float32)
    lib().oc_half_to_float(_p(h, _u16p), _p(back, _f32p), C.c_int64(f.size))
    return h.view(np.float16), back


def sample_rays(rays8, t_vals, t_rand=None):
    rays8, t_vals = _f32(rays8), _f32(t_vals)
    n, S = rays8.shape[0], t_vals.shape[0]
    tr = _f32(t_rand) if t_rand is not None else None
    z = np.empty((n, S), np.float32)
    pts = np.empty((n, S, 3), np.float32)
    lib().oc_sample_rays(_p(rays8, _f32p), _p(t_vals, _f32p), _p(tr, _f32p), C.c_int64(n),
                         C.c_int(S), _p(z, _f32p), _p(pts, _f32p))
    return z, pts


def motion_field(pts, Rs, Ts, vol, bbox_min, bbox_scale):
    """pts[N,3]; Rs[nb,3,3]; Ts[nb,3]; vol[>=nb,G,G,G] -> x_skel[N,3], mask[N]."""
    pts = _f32(pts).reshape(-1, 3)
    Rs, Ts, vol = _f32(Rs), _f32(Ts), _f32(vol)
    nb, G = Rs.shape[0], vol.shape[-1]
    N = pts.shape[0]
    xs = np.empty((N, 3), np.float32)
    mk = np.empty((N,), np.float32)
    bmin, bsc = _f32(bbox_min), _f32(bbox_scale)
    lib().oc_motion_field(_p(pts, _f32p), C.c_int64(N), _p(Rs, _f32p), _p(Ts, _f32p),
                          _p(vol, _f32p), C.c_int(nb), C.c_int(G), _p(bmin, _f32p),
                          _p(bsc, _f32p), _p(xs, _f32p), _p(mk, _f32p))
    return xs, mk


def nonrigid(xyz, cond, hann, weights, biases, skip_layer=4):
    """weights/biases: torch-layout lists, last entry is the 3-wide output layer."""
    xyz, cond, hann = _f32(xyz), _f32(cond).ravel(), _f32(hann)
    N = xyz.shape[0]
    depth = len(weights) - 1
    width = weights[0].shape[0]
    Wt, _kw = _ptr_array(weights)
    Bt, _kb = _ptr_array(biases)
    out = np.empty((N, 3), np.float32)
    lib().oc_nonrigid(_p(xyz, _f32p), C.c_int64(N), _p(cond, _f32p), C.c_int(cond.shape[0]),
                      _p(hann, _f32p), C.c_int(hann.shape[0]), Wt, Bt, C.c_int(width),
                      C.c_int(depth), C.c_int(skip_layer), _p(out, _f32p))
    return out


def knn(q, s, k, return_dist=False):
    q, s = _f32(q), _f32(s)
    nq, ns = q.shape[0], s.shape[0]
    idx = np.empty((nq, k), np.int32)
    dist = np.empty((nq, k), np.float32) if return_dist else None
    lib().oc_knn(_p(q, _f32p), C.c_int64(nq), _p(s, _f32p), C.c_int(ns), C.c_int(k),
                 _p(idx, _i32p), _p(dist, _f32p))
    return (idx, dist) if return_dist else idx


def msknn(xyz, base, fps_list, k=10):
    xyz, base = _f32(xyz), _f32(base)
    fps = [_i32(f) for f in fps_list]
    N = xyz.shape[0]
    ns = len(fps) + 1
    tab = (_i32p * len(fps))(*[_p(f, _i32p) for f in fps])
    nf = _i32([f.shape[0] for f in fps])
    out = np.empty((N, ns, k), np.int32)
    lib().oc_msknn(_p(xyz, _f32p), C.c_int64(N), _p(base, _f32p), C.c_int(base.shape[0]), tab,
                   _p(nf, _i32p), C.c_int(ns), C.c_int(k), _p(out, _i32p))
    return out


def point_sdf(point_cloud, point_base, normals):
    pc, pb, nr = _f32(point_cloud), _f32(point_base), _f64(normals)
    P = pc.shape[0]
    kb = np.empty((P, 3), np.float64)
    dist = np.empty((P,), np.float32)
    lib().oc_point_sdf(_p(pc, _f32p), _p(pb, _f32p), _p(nr, _f64p), C.c_int(P), _p(kb, _f64p),
                       _p(dist, _f32p))
    return kb, dist


def point_table(knn_base, point_sdf_, learnable, bound, embeddings, offsets, S, H):
    kb, sd, le = _f64(knn_base), _f32(point_sdf_).ravel(), _f32(learnable)
    emb, off = _f32(embeddings), _i32(offsets)
    P = kb.shape[0]
    L, Cc = off.shape[0] - 1, emb.shape[1]
    tab = np.empty((P, L * Cc + 3), np.float32)
    b32, tb32 = np.float32(bound), np.float32(2 * np.float64(bound))
    lib().oc_point_table(_p(kb, _f64p), _p(sd, _f32p), _p(le, _f32p), C.c_int(P), C.c_float(b32),
                         C.c_float(tb32), _p(emb, _f32p), _p(off, _i32p), C.c_uint32(L),
                         C.c_uint32(Cc), C.c_float(S), C.c_uint32(H), _p(tab, _f32p))
    return tab


def canonical_mlp(xyz, knn_idxs, point_base, normals, counter, table, bound, embeddings,
                  offsets, S, H, Wg, Bg, Wc, Bc, want_mlp_in=False):
    """Wg/Bg: hidden layers + geo_linear; Wc/Bc: hidden layers + output_linear."""
    xyz, idx = _f32(xyz), _i32(knn_idxs)
    pb, nr, cnt, tab = _f32(point_base), _f64(normals), _f32(counter), _f32(table)
    emb, off = _f32(embeddings), _i32(offsets)
    N, nscale, k = idx.shape
    L, Cc = off.shape[0] - 1, emb.shape[1]
    depth, width = len(Wg) - 1, Wg[0].shape[0]
    Wgt, _k1 = _ptr_array(Wg)
    Bgt, _k2 = _ptr_array(Bg)
    Wct, _k3 = _ptr_array(Wc)
    Bct, _k4 = _ptr_array(Bc)
    raw = np.empty((N, 5), np.float32)
    mi = np.empty((N, L * Cc * 2 + 4), np.float32) if want_mlp_in else None
    b32, tb32 = np.float32(bound), np.float32(2 * np.float64(bound))
    lib().oc_canonical_mlp(_p(xyz, _f32p), C.c_int64(N), _p(idx, _i32p), C.c_int(nscale),
                           C.c_int(k), _p(pb, _f32p), _p(nr, _f64p), _p(cnt, _f32p),
                           _p(tab, _f32p), C.c_float(b32), C.c_float(tb32), _p(emb, _f32p),
                           _p(off, _i32p), C.c_uint32(L), C.c_uint32(Cc), C.c_float(S),
                           C.c_uint32(H), Wgt, Bgt, Wct, Bct, C.c_int(depth), C.c_int(width),
                           _p(raw, _f32p), _p(mi, _f32p))
    return (raw, mi) if want_mlp_in else raw


def raw2outputs(raw, mask, z_vals, rays_d, bgcolor):
    raw, mask, z = _f32(raw), _f32(mask), _f32(z_vals)
    d, bg = _f32(rays_d), _f32(bgcolor)
    n, S = z.shape
    rgb = np.empty((n, 3), np.float32)
    acc = np.empty((n,), np.float32)
    dep = np.empty((n,), np.float32)
    w = np.empty((n, S), np.float32)
    tp = np.empty((n,), np.int32)
    lib().oc_raw2outputs(_p(raw, _f32p), _p(mask, _f32p), _p(z, _f32p), _p(d, _f32p), C.c_int(3),
                         _p(bg, _f32p), C.c_int64(n), C.c_int(S), _p(rgb, _f32p), _p(acc, _f32p),
                         _p(dep, _f32p), _p(w, _f32p), _p(tp, _i32p))
    return rgb, acc, w, dep, tp


# ----------------------------------------------------------------------------- ray generation
def gen_rays(K, E, H, W, bbox_min, bbox_max):
    """camera_util.py:133-160 (get_rays_from_KRT) + :163-212 (rays_intersect_3d_bbox), as chained at
    tpose.py:155-172: all pixels -> rays_o[H*W,3], rays_d[H*W,3] (clamped in place like the reference),
    near[R], far[R], mask[H*W].  Dtypes are left to numpy exactly as in the reference: a float32 camera
    (tpose.py:66-84) gives float32 rays, a float64 one float64 rays; the box stage is float64 either way."""
    K, E = np.asarray(K), np.asarray(E)
    R, T = E[:3, :3], E[:3, 3]
    rays_o = -np.dot(R.T, T).ravel()                                            # :149
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    xy1 = np.stack([i, j, np.ones_like(i)], axis=2)                             # :151-154
    pixel_camera = np.dot(xy1, np.linalg.inv(K).T)                              # :155
    pixel_world = np.dot(pixel_camera - T.ravel(), R)                           # :156
    ray_d = (pixel_world - rays_o[None, None]).reshape(-1, 3).copy()              # :158
    ray_o = np.broadcast_to(rays_o, ray_d.shape)
    bounds = np.stack([np.asarray(bbox_min, np.float64), np.asarray(bbox_max, np.float64)], 0)
    bounds = bounds + np.array([-0.01, 0.01])[:, None]                          # :181
    nominator = bounds[None] - ray_o[:, None]
    ray_d[np.abs(ray_d) < 1e-5] = 1e-5                                          # :184 (in place in the reference)
    d_intersect = (nominator / ray_d[:, None]).reshape(-1, 6)
    p_intersect = d_intersect[..., None] * ray_d[:, None] + ray_o[:, None]
    lo, hi = bounds[0] - 1e-6, bounds[1] + 1e-6                                 # :190-197
    at_box = np.all((p_intersect >= lo) & (p_intersect <= hi), axis=-1)
    mask = at_box.sum(-1) == 2                                                  # :199
    p_iv = p_intersect[mask][at_box[mask]].reshape(-1, 2, 3)
    ro, rd = ray_o[mask], ray_d[mask]
    nrm = np.linalg.norm(rd, axis=1)
    d0 = np.linalg.norm(p_iv[:, 0] - ro, axis=1) / nrm
    d1 = np.linalg.norm(p_iv[:, 1] - ro, axis=1) / nrm
    return ray_o, ray_d, np.minimum(d0, d1), np.maximum(d0, d1), mask
