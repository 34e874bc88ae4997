#!/bin/bash
# Usage: tools/pmc_clock.sh <outdir> <python script + args...>: GRBM_GUI_ACTIVE per dispatch -> effective clock
out=$1; shift
export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out" -o p -- python3 "$@" > "$out.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'occ::' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            dt = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
            acc[r['Kernel_Name'].split('(')[0]].append((float(r['Counter_Value']), dt))
for k, v in acc.items():
    cyc = sum(a for a, _ in v) / len(v); dt = sum(b for _, b in v) / len(v)
    print(f'{k:50s} n={len(v)} cycles={cyc:.4g} time={dt*1e3:.3f} ms clock={cyc/dt/1e9:.3f} GHz')
PY
