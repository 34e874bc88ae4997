#!/bin/bash
# Usage (GPU box, repo root): tools/pmc_hbm.sh <out.json>
# HBM traffic per launch of every occ:: kernel of the bench frame: two separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE; never combined with trace domains), per MI355X_MICROARCH.md's HBM section:
# counter values are KiB; FETCH_SIZE x2 for wide coalesced reads on gfx950; WRITE_SIZE uncorrected.  The x2 holds for wide
# coalesced streams only (tools/fetch_calib.hip, profiles/archive/r03_fetch_calibration.md: a gather's L2 miss is one 64-byte fabric
# request and is reported at face value), so every kernel gets ONE traffic figure with the factor of its access pattern:
# x1 for the gather kernels (hash corners, table rows, kNN points, motion-volume taps), x2 for the streaming ones.
out=$1
export TMPDIR=/tmp
d=gpurun_out/pmc_hbm
mkdir -p $d
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $d/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/write -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $d/write.log 2>&1
python3 - $d $out <<'PY'
import csv, glob, json, sys, collections
d, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for which in ('fetch', 'write'):
    for f in glob.glob(f'{d}/{which}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if 'occ::' in k:
                acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
line = [l for l in open(f'{d}/fetch.log').read().splitlines() if l.startswith('{')]
n = None
if line:
    j = json.loads(line[-1]); n = int(j['config'].get('samples_evaluated_per_launch', j['config'].get('samples_evaluated_per_frame', j['config']['rays_per_frame'] * j['config']['samples_per_ray'])))
GATHER = ('sample_features', 'msknn', 'sample_warp', 'knn_small', 'point_table', 'point_sdf', 'grid_forward', 'grid_backward',
          'assemble_image')
res = {'source': 'tools/pmc_hbm.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), '
                 'bench.py --steps 2 --warmup 1 on MI355X',
       'units': 'counter values are KiB per launch (mean over the launches); hbm_bytes_corrected = fetch_factor x FETCH_SIZE + '
                'WRITE_SIZE with fetch_factor 2 for kernels that stream wide coalesced reads (MI355X_MICROARCH.md, HBM section) '
                'and 1 for the gather kernels (profiles/archive/r03_fetch_calibration.md: 64-byte gather misses are counted at face value)',
       'samples_per_launch': n, 'kernels': {}}
for k, c in acc.items():
    f = sum(c['FETCH_SIZE']) / max(len(c['FETCH_SIZE']), 1)
    w = sum(c['WRITE_SIZE']) / max(len(c['WRITE_SIZE']), 1)
    factor = 1 if any(t in k for t in GATHER) else 2
    res['kernels'][k] = {'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w, 'fetch_factor': factor,
                         'hbm_bytes_corrected': (factor * f + w) * 1024, 'launches': len(c['FETCH_SIZE'])}
json.dump(res, open(out, 'w'), indent=1)
for k, v in res['kernels'].items():
    print(f"{k:50s} {v['hbm_bytes_corrected'] / 1e9:8.3f} GB/launch")
PY
