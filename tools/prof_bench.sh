#!/bin/bash
# Usage (GPU box, repo root): tools/prof_bench.sh <tag>  -> gpurun_out/prof_<tag>/ + top kernels on stdout
tag=$1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > gpurun_out/prof_${tag}_line.json 2> gpurun_out/prof_${tag}.err
python3 - gpurun_out/prof_$tag <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(__import__("os").environ.get("TOPN", "9"))]:
    print(f"{r['Name'][:60]:60s} {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:9.3f} ms  min {float(r['MinNs'])/1e6:9.3f}  {float(r['Percentage']):6.2f} %")
PY
