"""Import the UNMODIFIED reference (/root/reference) on CPU in the build container.

TEST INFRASTRUCTURE, build-container only: /root/reference does not exist on the GPU
box and nothing here is imported by tests marked gpu, smoke() or bench.py.  Its one
job is to run the reference's own Python so that make_golden.py can record golden
inputs/outputs.

What is shimmed, and with what:
  * third-party packages absent from this image: cv2.Rodrigues, trimesh vertex
    normals, torch_cluster.fps (deterministic start), torchvision, pytorch3d,
    termcolor -- small stand-ins defined below / in occnerf_amd.synth;
  * pykeops.torch.LazyTensor: only the expressions knn.py:46-83 use; the reduction is
    executed by the C oracle's exact kNN (oracle.knn);
  * _gridencoder: the reference's CUDA extension cannot run without a GPU; the C
    oracle's restatement of gridencoder.cu is bound in its place;
  * third_parties.smpl.smpl_numpy.SMPL: the licensed pickle is absent -> SyntheticSMPL.
"""
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from occnerf_amd import synth  # noqa: E402
from occnerf_amd.geometry import farthest_point_sampling  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# ---------------------------------------------------------------- pykeops shim
class LazyTensor:
    """Just enough of pykeops.torch.LazyTensor for knn.py:46-83."""

    def __init__(self, x=None, kind='var', a=None, b=None):
        self.kind, self.x, self.a, self.b = kind, x, a, b
        self.ranges = None

    def __sub__(self, other):
        return LazyTensor(kind='sub', a=self, b=other)

    def norm2(self):
        assert self.kind == 'sub'
        return LazyTensor(kind='norm2', a=self.a, b=self.b)

    def Kmin_argKmin(self, k, dim=1):
        assert self.kind == 'norm2' and dim == 1
        q = self.a.x.reshape(-1, self.a.x.shape[-1])          # (N,1,C) -> (N,C)
        s = self.b.x.reshape(-1, self.b.x.shape[-1])          # (1,M,C) -> (M,C)
        if q.dtype == torch.float64:                          # the float64 "truth" pass of make_golden.py
            return _kmin_f64(q.detach(), s.detach(), k, self.ranges)
        qn, sn = q.detach().cpu().numpy(), s.detach().cpu().numpy()
        if self.ranges is None:
            idx, dist = orc.knn(qn, sn, k, return_dist=True)
        else:
            ranges_x, _, ranges_y = self.ranges[0], self.ranges[1], self.ranges[2]
            idx = np.zeros((qn.shape[0], k), np.int32)
            dist = np.zeros((qn.shape[0], k), np.float32)
            for (x0, x1), (y0, y1) in zip(ranges_x.tolist(), ranges_y.tolist()):
                i, d = orc.knn(qn[x0:x1], sn[y0:y1], k, return_dist=True)
                idx[x0:x1] = i + y0                           # global support-row index
                dist[x0:x1] = d
        return torch.from_numpy(dist), torch.from_numpy(idx.astype(np.int64))


def _kmin_f64(q, s, k, ranges):
    """Kmin_argKmin of norm2() in float64 (direct differences, no matmul trick), ascending, ties to the lower row."""
    def block(qb, sb):
        idx = torch.empty(qb.shape[0], k, dtype=torch.int64)
        dist = torch.empty(qb.shape[0], k, dtype=torch.float64)
        for lo in range(0, qb.shape[0], 4096):
            d = (qb[lo:lo + 4096, None, :] - sb[None]).square().sum(-1).sqrt()
            dd, ii = torch.sort(d, dim=1, stable=True)
            dist[lo:lo + 4096], idx[lo:lo + 4096] = dd[:, :k], ii[:, :k]
        return dist, idx
    if ranges is None:
        return block(q, s)
    idx = torch.zeros(q.shape[0], k, dtype=torch.int64)
    dist = torch.zeros(q.shape[0], k, dtype=torch.float64)
    for (x0, x1), (y0, y1) in zip(ranges[0].tolist(), ranges[2].tolist()):
        d, i = block(q[x0:x1], s[y0:y1])
        dist[x0:x1], idx[x0:x1] = d, i + y0
    return dist, idx


# ------------------------------------------------------------ gridencoder shim
def grid_encode_forward_f64(x, emb, offsets, S, H):
    """gridencoder.cu:87-245 (kernel_grid, hash grid type, align_corners=False, linear interpolation) evaluated in float64:
    the SAME function -- each level's scale is the float constant the kernel computes (`exp2f(level*S)*H - 1.0f`, :139: it
    defines the grid, and the reference keeps it a float even in its double dispatch case), resolution ceil(scale)+1, uint32
    index arithmetic with the kernel's primes and wrap-around, the dense / hashed decision per level, zeros for out-of-range
    inputs -- with the cell coordinates and interpolation weights in float64.  Used only by make_golden.py's truth pass."""
    x = np.asarray(x, np.float64)
    emb = np.asarray(emb, np.float64)
    B, D = x.shape
    L, Cc = len(offsets) - 1, emb.shape[1]
    primes = np.array([1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737], np.uint64)
    out = np.zeros((L, B, Cc), np.float64)
    oob = ((x < 0) | (x > 1)).any(1)
    M32 = np.uint64(0xffffffff)
    scales, ress = orc.grid_level_params(L, float(S), int(H))
    for lvl in range(L):
        size = int(offsets[lvl + 1] - offsets[lvl])
        scale, res = np.float64(scales[lvl]), int(ress[lvl])
        pos = x * scale + 0.5
        pg = np.floor(pos)
        fr = pos - pg
        pg = pg.astype(np.int64)
        acc = np.zeros((B, Cc), np.float64)
        n_dense, stride = 0, 1                               # gridencoder.cu:68-77: dims walked while stride <= hashmap_size
        strides = []
        while n_dense < D and stride <= size:
            strides.append(stride)
            stride *= res + 1                                # align_corners == False
            n_dense += 1
        hashed = stride > size
        for corner in range(1 << D):
            w = np.ones(B, np.float64)
            idx = np.zeros(B, np.uint64)
            hsh = np.zeros(B, np.uint64)
            for d in range(D):
                bit = (corner >> d) & 1
                w = w * (fr[:, d] if bit else 1.0 - fr[:, d])
                c = (pg[:, d] + bit).astype(np.uint64) & M32
                if d < n_dense:
                    idx = (idx + c * np.uint64(strides[d])) & M32
                hsh = hsh ^ ((c * primes[d]) & M32)
            index = (hsh if hashed else idx) % np.uint64(size)
            acc += w[:, None] * emb[int(offsets[lvl]) + index.astype(np.int64)]
        acc[oob] = 0.0
        out[lvl] = acc
    return out


def _grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx,
                         gridtype, align_corners, interp):
    if embeddings.dtype == torch.float64:                      # truth pass (make_golden.py): the same function in float64
        assert dy_dx is None and gridtype == 0 and not align_corners and interp == 0
        outputs.copy_(torch.from_numpy(grid_encode_forward_f64(inputs.detach().numpy(), embeddings.detach().numpy(),
                                                               offsets.numpy(), float(S), int(H))))
        return
    out, dy = orc.grid_encode_forward(inputs.detach().numpy(), embeddings.detach().numpy(),
                                      offsets.numpy(), float(S), int(H), dy_dx is not None,
                                      gridtype, align_corners, interp)
    outputs.copy_(torch.from_numpy(out))
    if dy_dx is not None:
        dy_dx.copy_(torch.from_numpy(dy))


def _grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H,
                          dy_dx, grad_inputs, gridtype, align_corners, interp):
    ge, gi = orc.grid_encode_backward(grad.detach().numpy(), inputs.detach().numpy(),
                                      offsets.numpy(), embeddings.shape[0], int(C), float(S),
                                      int(H), None if dy_dx is None else dy_dx.numpy(), gridtype,
                                      align_corners, interp)
    grad_embeddings.copy_(torch.from_numpy(ge))
    if grad_inputs is not None:
        grad_inputs.copy_(torch.from_numpy(gi))


class _Trimesh:
    def __init__(self, vertices=None, faces=None, **_):
        self.vertices, self.faces = np.asarray(vertices), np.asarray(faces)

    @property
    def vertex_normals(self):
        return synth.vertex_normals(self.vertices, self.faces)


def _fps(x, batch=None, ratio=0.5, random_start=True):
    return torch.from_numpy(farthest_point_sampling(x.detach().cpu().numpy(), ratio))


def install(argv):
    """Register the stand-ins, chdir into the reference and import its config."""
    os.chdir(REF)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    sys.argv = list(argv)

    _mod('cv2', Rodrigues=lambda v: (synth.rodrigues_exact(np.asarray(v).ravel())
                                     if np.asarray(v).size == 3 else None, None))
    _mod('trimesh', Trimesh=_Trimesh)
    tv = _mod('torchvision')
    tv.models = _mod('torchvision.models')
    p3 = _mod('pytorch3d')
    p3.ops = _mod('pytorch3d.ops')
    p3.ops.points_normals = _mod('pytorch3d.ops.points_normals',
                                 estimate_pointcloud_normals=lambda *a, **k: None)
    _mod('termcolor', colored=lambda s, *a, **k: s)
    _mod('torch_cluster', fps=_fps)
    pk = _mod('pykeops')
    pk.torch = _mod('pykeops.torch', LazyTensor=LazyTensor)
    _mod('_gridencoder', grid_encode_forward=_grid_encode_forward,
         grid_encode_backward=_grid_encode_backward,
         grad_total_variation=lambda *a, **k: None)
    _mod('_shencoder', sh_encode_forward=lambda *a, **k: None,
         sh_encode_backward=lambda *a, **k: None)

    import third_parties.smpl  # noqa: F401  (namespace package of the reference)
    _mod('third_parties.smpl.smpl_numpy', SMPL=synth.SyntheticSMPL)

    real_count = torch.cuda.device_count
    torch.cuda.device_count = lambda: 1
    try:
        import configs  # parses sys.argv at import (configs/config.py:65-72)
    finally:
        torch.cuda.device_count = real_count
    cfg = configs.cfg
    cfg.primary_gpus = ['cpu']
    cfg.secondary_gpus = ['cpu']
    return cfg
