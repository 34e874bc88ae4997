"""How many DISTINCT per-point table rows do the samples that share a wave of the feature kernel gather?
  Benchmark frame, the renderer's own sample order (Morton rays, live samples only): for groups
of G consecutive listed samples, distinct neighbour ids / (G * 40), overall and per scale.
    python tools/row_sharing.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from occnerf_amd import ops, synth  # noqa: E402
from occnerf_amd.seeded import build_network, frame_to_device  # noqa: E402


def main():
    dev = 'cuda:0'
    net = build_network(seed=0, amplify=False, S=128, non_rigid=True, device=dev)
    net.cfg.dedup_repeated_samples = False
    frame = synth.make_frame(img_size=512, pose72=synth.seeded_pose(1), orbit_frame=28)
    data = frame_to_device(frame, dev)
    grabbed = {}
    real = ops.msknn_clustered

    def spy(xyz, *a, **k):
        out = real(xyz, *a, **k)
        grabbed['knn'], grabbed['rows'], grabbed['count'] = out, k.get('rows'), k.get('count')
        return out
    ops.msknn_clustered = spy
    with torch.no_grad():
        net(**data, iter_val=1e7)
    ops.msknn_clustered = real
    n = int(grabbed['count'])
    rows = grabbed['rows'][:n].long()
    knn = grabbed['knn']
    knn = knn[rows] if knn.shape[0] != n else knn[:n]          # [n,4,10] ids of the listed samples, list order
    knn = knn.reshape(n, 4, 10).long()
    report = {'listed_samples': n}
    for G in (8, 16, 64):
        m = n // G
        k = knn[:m * G].reshape(m, G, 4, 10)
        per_scale = []
        for l in range(4):
            ids = k[:, :, l].reshape(m, G * 10)
            s = ids.sort(dim=1).values
            distinct = 1 + (s[:, 1:] != s[:, :-1]).sum(dim=1)
            per_scale.append(float(distinct.float().mean()) / (G * 10))
        ids = (k + torch.arange(4, device=k.device).view(1, 1, 4, 1) * 0).reshape(m, G * 40)
        s = ids.sort(dim=1).values
        distinct = 1 + (s[:, 1:] != s[:, :-1]).sum(dim=1)
        report[f'G{G}'] = {'distinct_fraction': float(distinct.float().mean()) / (G * 40),
                           'distinct_rows_mean': float(distinct.float().mean()), 'distinct_rows_max': int(distinct.max()),
                           'distinct_rows_p99': float(distinct.float().quantile(0.99)),
                           'per_scale_fraction': per_scale}
    print(json.dumps(report))


if __name__ == '__main__':
    main()
