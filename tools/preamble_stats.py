"""Per-frame preamble statistics from a rocprofv3 kernel trace of bench.py (tools/prof_bench.sh <tag>):
python tools/preamble_stats.py gpurun_out/prof_<tag>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ci = [i for i, r in enumerate(rows) if 'composite_kernel' in r['Kernel_Name']]
wi = [i for i, r in enumerate(rows) if 'sample_warp_kernel' in r['Kernel_Name']]
a, b = ci[-3], wi[-2]
span = (int(rows[b]['Start_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e6
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[a + 1:b]) / 1e6
print(f'preamble: {b - a - 1} kernels, span {span:.3f} ms, busy {busy:.3f} ms')
c, t = collections.Counter(), collections.Counter()
for r in rows[a + 1:b]:
    n = r['Kernel_Name'][:60]
    c[n] += 1
    t[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for n, v in t.most_common(10):
    print(f'  {n:62s} {c[n]:4d} {v:9.1f} us')
