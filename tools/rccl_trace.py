import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# last two frames: find pose_motion_bases starts
starts = [i for i, r in enumerate(rows) if 'pose_motion_bases' in r['Kernel_Name']]
lo, hi = starts[-3], starts[-1]
t0 = int(rows[lo]['Start_Timestamp'])
prev_end = t0
for r in rows[lo:hi]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if (e - s) > 100_000 or 'ccl' in r['Kernel_Name'].lower() or s - prev_end > 200_000:
        print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} dur {(e - s) / 1e3:9.1f} us gap {(s - prev_end) / 1e3:8.1f} q{r['Queue_Id']} {r['Kernel_Name'][:70]}")
    prev_end = max(prev_end, e)
