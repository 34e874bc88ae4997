#!/bin/bash
# Usage (GPU box, repo root): tools/pmc_train.sh <out.json>
# HBM traffic of one TRAINING step (BASELINE configs[4]: 6 144 rays x 128 samples, bf16 trunks), kernel by kernel: two separate
# --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with trace domains) over tools/train_step_trace.py, corrected as
# tools/pmc_hbm.sh does (counter values are KiB; FETCH_SIZE x2 for kernels that stream wide coalesced reads, x1 for the gather
# kernels; WRITE_SIZE at face value).  Per step = the sum over every dispatch of the run / the steps it ran (3 warm-up + 4).
out=$1
export TMPDIR=/tmp
d=gpurun_out/pmc_train
mkdir -p $d
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -o p -- python3 tools/train_step_trace.py 4 > $d/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/write -o p -- python3 tools/train_step_trace.py 4 > $d/write.log 2>&1
python3 - $d $out <<'PY'
import csv, glob, json, sys, collections
d, out = sys.argv[1], sys.argv[2]
STEPS = 7
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for which in ('fetch', 'write'):
    for f in glob.glob(f'{d}/{which}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            if which == 'fetch':
                cnt[k] += 1
GATHER = ('sample_features', 'msknn', 'sample_warp', 'knn_small', 'knn_center', 'point_table', 'point_sdf', 'grid_forward',
          'grid_backward', 'agg_forward', 'agg_backward', 'agg_weights', 'warp_backward')
rows = {}
for k, c in acc.items():
    factor = 1 if any(t in k for t in GATHER) else 2
    rows[k] = {'launches_per_step': cnt[k] / STEPS, 'FETCH_SIZE_KiB_per_step': c['FETCH_SIZE'] / STEPS,
               'WRITE_SIZE_KiB_per_step': c['WRITE_SIZE'] / STEPS, 'fetch_factor': factor,
               'hbm_bytes_per_step': (factor * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / STEPS}
rows = dict(sorted(rows.items(), key=lambda kv: -kv[1]['hbm_bytes_per_step']))
total = sum(v['hbm_bytes_per_step'] for v in rows.values())
log = open(f'{d}/fetch.log').read().splitlines()
res = {'source': 'tools/pmc_train.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over '
                 '`python3 tools/train_step_trace.py 4` (3 warm-up + 4 steps) on MI355X',
       'units': 'KiB counters summed over every dispatch of the run / 7 steps; hbm_bytes_per_step = fetch_factor x FETCH_SIZE + '
                'WRITE_SIZE (factor 2 for streaming kernels, 1 for gather kernels: MI355X_MICROARCH.md HBM section, '
                'profiles/archive/r03_fetch_calibration.md)',
       'step_line_of_the_fetch_pass': [l for l in log if l.startswith('train step')][-1:],
       'hbm_bytes_per_step_total': total,
       'kernels': {k: v for k, v in list(rows.items())[:40]}}
json.dump(res, open(out, 'w'), indent=1)
print(f'total {total / 1e9:.2f} GB per step')
for k, v in list(rows.items())[:25]:
    print(f"{k[:70]:70s} {v['launches_per_step']:6.1f} {v['hbm_bytes_per_step'] / 1e9:8.3f} GB/step")
PY
