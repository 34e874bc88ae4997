"""How typical is the collapse fraction?  

Round 4's exact shortcuts -- the kNN / feature centre cache -- pay in proportion to the share of a frame's live samples that lie
inside the provable radius around the frame's collapse point (wherever a sample's motion-weight sum is far below the warp's
1e-4 clamp, network.py:388, its canonical position lands on offset(0)).  This tool renders frames over poses x checkpoints
x cameras at the benchmark size and prints, per frame: live samples, the share inside the radius, the radius, and the frame
time with the shortcuts on and off (identical pixels).
    python3 tools/collapse_fraction.py > profiles/rNN_collapse_fraction.md"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
from occnerf_amd import ops, seeded, synth  # noqa: E402

IMG, SPP = 512, 128


def frame_ms(net, data, n=3):
    with torch.no_grad():
        net(**data, iter_val=1e7, ray_order_key='c')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            net(**data, iter_val=1e7, ray_order_key='c')
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    cases = [(f'pose {s}', dict(pose72=synth.seeded_pose(s), orbit_frame=(13 * s) % 100)) for s in range(1, 11)]
    cases += [(f'pose {s}, 3x amplitude', dict(pose72=synth.seeded_pose(s, sigma=0.9), orbit_frame=(13 * s) % 100)) for s in (1, 2, 3)]
    cases += [('T-pose', dict(pose72=None, orbit_frame=0))]
    cases += [(f'pose {s}, near camera (radius 3.0, focal 625)', dict(pose72=synth.seeded_pose(s), orbit_frame=(13 * s) % 100, camera_radius=3.0,
                                                                       camera_focal=625.0)) for s in (1, 2)]
    print('# Collapse fraction over poses x checkpoints x cameras (512x512 x 128, every live sample evaluated)\n')
    print('share = live samples with |p - c|^2 < r^2 (served by the centre cache); on / off = frame ms with the two exact shortcuts of round 4 '
          '(centre cache, bone-box culling) on / off, device-resident frame, named camera\n')
    shares = []
    for level, lname in ((0, 'random-init'), (1, 'amplified'), (2, 'trained-like')):
        net = seeded.build_network(0, level, S=SPP, non_rigid=True)
        net.cfg.dedup_repeated_samples = False
        print(f'## checkpoint: {lname}\n')
        print('| frame | rays | live samples | share inside r | r (m) | ms on | ms off | saved |')
        print('|---|---|---|---|---|---|---|---|')
        grabbed = {}
        real = ops.msknn_clustered

        def grab(xyz, n_rays, S, cl, seed, mask=None, rows=None, count=None, out=None, center=None):
            if center is not None and rows is not None:
                grabbed.update(xyz=xyz, rows=rows, count=count, center=center[0])
            return real(xyz, n_rays, S, cl, seed, mask=mask, rows=rows, count=count, out=out, center=center)
        for name, kw in cases:
            frame = synth.make_frame(img_size=IMG, **kw)
            data = seeded.frame_to_device(frame, 'cuda:0')
            for k in ('cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor'):
                data[k] = data[k].cpu()
            net._ray_orders.clear()
            ops.msknn_clustered = grab
            try:
                with torch.no_grad():
                    net(**data, iter_val=1e7)
            finally:
                ops.msknn_clustered = real
            n = int(grabbed['count'])
            p = grabbed['xyz'][grabbed['rows'][:n].long()]
            c = grabbed['center']
            share = float((((p - c[:3]) ** 2).sum(1) < c[3]).float().mean())
            r = float(c[3].clamp_min(0).sqrt())
            net._ray_orders.clear()
            on = frame_ms(net, data)
            net.cfg.knn_center_cache, net.cfg.warp_bone_culling = False, False
            net._ray_orders.clear()
            off = frame_ms(net, data)
            net.cfg.knn_center_cache, net.cfg.warp_bone_culling = True, True
            shares.append(share)
            print(f'| {name} | {frame["rays"].shape[1]} | {n} | {100 * share:.1f} % | {r:.2e} | {on:.1f} | {off:.1f} | {off - on:.1f} |')
        print()
        del net
        torch.cuda.empty_cache()
    s = np.array(shares)
    print(f'share inside the radius over all {s.size} frames: median {100 * np.median(s):.1f} %, min {100 * s.min():.1f} %, max {100 * s.max():.1f} %')


if __name__ == '__main__':
    main()
