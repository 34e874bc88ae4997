"""Host-side geometry set-up that the reference delegates to third-party packages
(torch_cluster.fps, trimesh) in Network.generate_neural_points (network.py:90-129).
Runs once per model, on the CPU, in numpy."""
import numpy as np

from .synth import vertex_normals  # noqa: F401  (re-exported)


def farthest_point_sampling(points, ratio):
    """Greedy farthest-point sampling, ceil(ratio * N) indices in selection order.

    Stands in for ``torch_cluster.fps(x, ratio=ratio)`` (network.py:113-118) with a
    deterministic start (index 0) instead of its default random start, so the coarse
    point scales are reproducible across launches."""
    pts = np.asarray(points, dtype=np.float64)
    n = pts.shape[0]
    m = int(np.ceil(ratio * n))
    sel = np.empty(m, dtype=np.int64)
    sel[0] = 0
    d = np.sum((pts - pts[0]) ** 2, axis=1)
    for i in range(1, m):
        j = int(np.argmax(d))
        sel[i] = j
        d = np.minimum(d, np.sum((pts - pts[j]) ** 2, axis=1))
    return sel
