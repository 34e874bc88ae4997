"""Last frame of a rocprofv3 --kernel-trace run of tools/overlap_frame.py: per launch start, end, queue, and how much of its
duration another queue's kernel was running beside it.
    python tools/overlap_trace.py <prof_dir> [min_us]"""
import csv
import glob
import sys

prof = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
f = sorted(glob.glob(prof + '/**/*kernel_trace.csv', recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'pose_motion_bases' in r['Kernel_Name']]
frame = rows[starts[-1]:]
t0 = int(frame[0]['Start_Timestamp'])
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name']) for r in frame]
print('| start ms | end ms | dur us | queue | beside other queue us | kernel |')
print('|---|---|---|---|---|---|')
for s, e, q, name in iv:
    if (e - s) / 1e3 < min_us:
        continue
    ov = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, q2, _ in iv if q2 != q)
    print(f'| {(s - t0) / 1e6:.3f} | {(e - t0) / 1e6:.3f} | {(e - s) / 1e3:.0f} | {q} | {ov / 1e3:.0f} | `{name[:60]}` |')
print(f'\nframe span {(max(e for _, e, _, _ in iv) - t0) / 1e6:.3f} ms, {len(iv)} launches')
