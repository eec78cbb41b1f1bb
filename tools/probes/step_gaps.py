"""timeline of one steady-state train step from a rocprofv3 kernel trace:  python tools/probes/step_gaps.py <kernel_trace.csv> [anchor kernel substring]
prints, for the LAST complete step, every launch with its start offset, duration and the gap to the previous launch's end"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else 'k_solo_fwd'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
mid = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) // 2
a, b = idx[mid], idx[mid + 1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = None
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print('%9.1f us  dur %8.1f us  gap %7.1f us  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, r['Kernel_Name'][:70]))
    prev_end = e
print('step period %.1f us' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
