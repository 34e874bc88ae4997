#!/bin/bash
# Usage (GPU box, repo root): tools/xcd_levels.sh <outdir>  -- timing, then FETCH_SIZE / WRITE_SIZE of each form in its own --pmc pass
o=$1
export TMPDIR=/tmp
mkdir -p $o
python3 tools/xcd_levels.py > $o/timing.txt 2>&1
for mode in shipped xcd; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 300 rocprofv3 --pmc $ctr --output-format csv -d $o/pmc_${mode}_$ctr -o p -- python3 tools/xcd_levels.py --mode $mode --reps 3 > $o/pmc_${mode}_$ctr.log 2>&1
  done
done
python3 - $o <<'PY'
import csv, glob, sys
o = sys.argv[1]
print(open(o + '/timing.txt').read())
for mode in ('shipped', 'xcd'):
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        fs = glob.glob(f'{o}/pmc_{mode}_{ctr}/**/*counter_collection.csv', recursive=True)
        if not fs:
            print(mode, ctr, 'no counter file'); continue
        vals = [float(r['Counter_Value']) for r in csv.DictReader(open(fs[0])) if 'grid_forward_d4c2' in r['Kernel_Name'] and r['Counter_Name'] == ctr]
        if vals:
            # counter values are KiB per launch (MI355X_MICROARCH.md, HBM section); gathers: FETCH_SIZE at face value
            print(f'{mode:8s} {ctr}: {len(vals)} launches, mean {sum(vals) / len(vals) * 1024 / 1e9:.3f} GB per launch')
PY
