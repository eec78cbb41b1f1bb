"""where one graph-replayed train step's wall time goes, from a rocprofv3 kernel trace of tools/probes/step_only.py (which
synchronises after every step: steps are the dispatch clusters separated by > 200 us of idle).
    python tools/probes/step_timeline.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]) for r in rows)
steps, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur[-4:]) > 200000:
        steps.append(cur); cur = []
    cur.append(e)
steps.append(cur)
print('%d clusters; sizes %s' % (len(steps), [len(s) for s in steps][-8:]))
st = steps[-2]
wall = (max(e[1] for e in st) - st[0][0]) / 1e3
busy, gaps, end = 0.0, collections.Counter(), st[0][0]
per = collections.Counter(); cnt = collections.Counter()
for s, e, n in st:
    per[n] += (e - s) / 1e3; cnt[n] += 1
    if s > end:
        gaps[n] += (s - end) / 1e3
    busy += (max(e, end) - max(s, end)) / 1e3 if e > end else 0.0
    end = max(end, e)
print('step: %d dispatches, wall %.1f us, device busy %.1f us, idle between dispatches %.1f us' % (len(st), wall, busy, wall - busy))
print('-- kernels by time')
for n, t in per.most_common(22):
    print('%9.1f us %5d x %7.2f  %s' % (t, cnt[n], t / cnt[n], n[-90:]))
print('-- idle time charged to the kernel that follows it')
for n, t in gaps.most_common(12):
    print('%9.1f us %5d x %7.2f  %s' % (t, cnt[n], t / cnt[n], n[-90:]))
