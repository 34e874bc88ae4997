"""Every kernel launch of the last frame of a rocprofv3 --kernel-trace run, in order, as a markdown table.

    python tools/frame_trace.py <prof_dir> <first kernel of a frame, substring> > profiles/rNN_frame_kernel_trace.md
"""
import csv
import glob
import sys

prof, first = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(prof + '/**/*kernel_trace.csv', recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]
lo = starts[-1]
# back up over the harness copies / fills that precede the frame's first kernel
while lo > 0 and int(rows[lo]['Start_Timestamp']) - int(rows[lo - 1]['End_Timestamp']) < 400_000 and \
        ('copyBuffer' in rows[lo - 1]['Kernel_Name'] or 'elementwise' in rows[lo - 1]['Kernel_Name']):
    lo -= 1
frame = rows[lo:]
t0 = int(frame[0]['Start_Timestamp'])
print('| start (ms) | duration (us) | kernel |')
print('|---|---|---|')
for r in frame:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"| {(s - t0) / 1e6:.3f} | {(e - s) / 1e3:.1f} | `{r['Kernel_Name'][:80]}` |")
print(f'\n{len(frame)} launches.')
