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


def build_knn_clusters(point_base, scale_indices, clusters_per_group=8):
    """Cluster layout for occnerf_msknn_clustered (include/occnerf_hip.h).

    point_base[P,3] float32; scale_indices = [arange(P), fps0, fps1, fps2] (rows of each
    scale in the reference's order, network.py:239-241).  The points of every scale but the
    coarsest are grouped by their nearest coarsest-scale point and stored cluster by cluster
    (ascending original row inside a cluster), each segment padded to a multiple of 4 rows
    with +inf points; float4.w carries (as int bits) ORIGINAL row within the scale << 16 | base-point
    index: the row breaks distance ties, the base index is what the search reports.

    The clusters are listed group by group: a group = the clusters nearest to one of the first ceil(ncl / 8) coarsest points
    (farthest-point order: well spread), with a bounding sphere per scale around everything its clusters hold, so that the
    search tests ~14 group spheres and the clusters of the few groups in reach instead of all 108 cluster spheres per scale.

    Returns dict of numpy arrays: points[M,4] f32, centers[ncl,4] f32, ranges[nscale-1,ncl,2] i32,
    radius[nscale-1,ncl] f32, coarse_rows (begin, end), group_centers[G,4] f32, group_ranges[G,2] i32 (cluster index
    range), group_radius[nscale-1,G] f32 (< 0: no points at that scale)."""
    base = np.ascontiguousarray(point_base, dtype=np.float32)
    sets = [np.asarray(s, dtype=np.int64) for s in scale_indices]
    nscale = len(sets)
    cidx = sets[-1]
    centers64 = base[cidx].astype(np.float64)
    ncl = len(cidx)
    rows, ranges = [], np.zeros((nscale - 1, ncl, 2), np.int32)
    radius = np.zeros((nscale - 1, ncl), np.float32)
    cursor = 0

    assert base.shape[0] < 65536 and max(len(t) for t in sets) < 32768, 'row << 16 | base index must fit 31 bits'

    def emit(pts, orig_rows, base_rows):
        nonlocal cursor
        n = len(orig_rows)
        pad = (-n) % 4
        blk = np.full((n + pad, 4), np.inf, np.float32)
        blk[:n, :3] = pts
        w = np.zeros(n + pad, np.int32)
        w[:n] = (np.asarray(orig_rows, np.int64) << 16) | np.asarray(base_rows, np.int64)
        blk[:, 3] = w.view(np.float32)
        rows.append(blk)
        begin = cursor
        cursor += n + pad
        return begin, cursor

    for l in range(nscale - 1):
        pts = base[sets[l]]
        d = np.linalg.norm(pts.astype(np.float64)[:, None, :] - centers64[None, :, :], axis=-1)
        owner, dmin = d.argmin(1), d.min(1)
        for k in range(ncl):
            members = np.nonzero(owner == k)[0]
            if len(members) == 0:
                ranges[l, k] = (cursor, cursor)
                continue
            ranges[l, k] = emit(pts[members], members, sets[l][members])
            radius[l, k] = np.float32(dmin[members].max() * (1 + 1e-6) + 1e-7)
    coarse = emit(base[cidx], np.arange(ncl), cidx)
    centers = np.zeros((ncl, 4), np.float32)
    centers[:, :3] = base[cidx]
    # groups of clusters (the index arrays are permuted; the point rows stay where they are)
    ngrp = max(1, -(-ncl // int(clusters_per_group)))
    seeds = centers64[:ngrp]
    dg = np.linalg.norm(centers64[:, None, :] - seeds[None, :, :], axis=-1)
    group_of = dg.argmin(1)
    perm = np.argsort(group_of, kind='stable')
    ranges, radius, centers = ranges[:, perm], radius[:, perm], centers[perm]
    gsorted = group_of[perm]
    group_ranges = np.zeros((ngrp, 2), np.int32)
    group_centers = np.zeros((ngrp, 4), np.float32)
    group_radius = np.full((nscale - 1, ngrp), -1.0, np.float32)
    for gi in range(ngrp):
        members = np.nonzero(gsorted == gi)[0]
        group_ranges[gi] = (members[0], members[-1] + 1) if len(members) else (0, 0)
        group_centers[gi, :3] = seeds[gi]
        c32 = group_centers[gi, :3].astype(np.float64)
        for l in range(nscale - 1):
            reach = [np.linalg.norm(centers[k, :3].astype(np.float64) - c32) + float(radius[l, k]) for k in members
                     if ranges[l, k, 0] < ranges[l, k, 1]]
            if reach:
                group_radius[l, gi] = np.float32(max(reach) * (1 + 1e-6) + 1e-7)
    return {'points': np.concatenate(rows, 0), 'centers': np.ascontiguousarray(centers), 'ranges': np.ascontiguousarray(ranges),
            'radius': np.ascontiguousarray(radius), 'coarse_rows': np.array(coarse, np.int32), 'ncl': ncl,
            'group_centers': group_centers, 'group_ranges': group_ranges, 'group_radius': group_radius, 'ngrp': ngrp}
