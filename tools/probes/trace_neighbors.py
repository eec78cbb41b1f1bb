"""where do the dispatches of one kernel sit in a rocprofv3 *_kernel_trace.csv: index range, histogram of predecessors"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
pat = sys.argv[2]
idx = [i for i, r in enumerate(rows) if pat in r['Kernel_Name']]
print(len(rows), 'dispatches;', len(idx), 'match; first/last index', idx[:1], idx[-1:])
hist = collections.Counter(i * 20 // len(rows) for i in idx)
print('distribution over 20 time bins:', [hist.get(b, 0) for b in range(20)])
prev = collections.Counter(rows[i - 1]['Kernel_Name'].split('(')[0][-50:] for i in idx if i > 0)
nxt = collections.Counter(rows[i + 1]['Kernel_Name'].split('(')[0][-50:] for i in idx if i + 1 < len(rows))
print('predecessors:', prev.most_common(8))
print('successors:', nxt.most_common(8))
