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
if len(steps[-1]) > 2 * len(steps[-2]) and any('k_adam_step' in e[2] for e in steps[-1]):
    # (the host came back within 200 us: the replays form one cluster -- cut it behind every optimizer launch instead)
    cut, cur = [], []
    for e in steps[-1]:
        cur.append(e)
        if 'k_adam_step' in e[2]:
            cut.append(cur); cur = []
    steps = cut + [cur]
    print('cut behind k_adam_step: sizes %s' % [len(s) for s in steps][-8:])
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
# the step's serial spine is its chain launches: what sits in front of the first, between them, and behind the last
ch = [(s, e, n) for s, e, n in st if 'convnet_chain' in n]
if ch:
    print('-- chain launches: %d, %.1f us; before the first %.1f us, after the last %.1f us' % (
        len(ch), sum(e - s for s, e, _ in ch) / 1e3, (ch[0][0] - st[0][0]) / 1e3, (max(e[1] for e in st) - ch[-1][1]) / 1e3))
    between = collections.Counter(); bt = 0.0
    for (s0, e0, _), (s1, e1, _) in zip(ch[:-1], ch[1:]):
        bt += (s1 - e0) / 1e3
    print('   between consecutive chain launches %.1f us in total (heads, glue; the side stream runs under all of it)' % bt)
    tail = [(s, e, n) for s, e, n in st if s >= ch[-1][1]]
    agg, cnt2 = collections.Counter(), collections.Counter()
    for s, e, n in tail:
        agg[n] += (e - s) / 1e3; cnt2[n] += 1
    for n, t in agg.most_common(16):
        print('   tail %8.1f us %4d x %7.2f  %s' % (t, cnt2[n], t / cnt2[n], n[-80:]))
    # one typical interval per (kind of chain launch): everything that starts between two consecutive launches of the same kernel
    q = {(int(r['Start_Timestamp']), r['Kernel_Name'].split('(')[0]): r.get('Queue_Id', '?') for r in rows}
    shown = set()
    for i in range(len(ch) // 2 - 40, len(ch) - 1):
        (s0, e0, n0), (s1, e1, n1) = ch[i], ch[i + 1]
        if n0 != n1 or n0 in shown or i < 3 or ch[i - 1][2] != n0:
            continue
        shown.add(n0)
        print('-- %s: launch %.1f us, then %.1f us to the next' % (n0[-60:], (e0 - s0) / 1e3, (s1 - e0) / 1e3))
        for s, e, n in st:
            if s0 < s < s1:
                print('      +%6.1f .. %6.1f us  q%s  %s' % ((s - e0) / 1e3, (e - e0) / 1e3, q.get((s, n), '?'), n[-70:]))
    for i in range(3, len(ch) // 2 - 1):
        (s0, e0, n0), (s1, e1, n1) = ch[i], ch[i + 1]
        if n0 != n1 or n0 in shown or ch[i - 1][2] != n0:
            continue
        shown.add(n0)
        print('-- %s: launch %.1f us, then %.1f us to the next' % (n0[-60:], (e0 - s0) / 1e3, (s1 - e0) / 1e3))
        for s, e, n in st:
            if s0 < s < s1:
                print('      +%6.1f .. %6.1f us  q%s  %s' % ((s - e0) / 1e3, (e - e0) / 1e3, q.get((s, n), '?'), n[-70:]))
    big = sorted(range(len(ch) - 1), key=lambda i: ch[i][1] - ch[i + 1][0])[:6]
    for i in sorted(big):
        (s0, e0, n0), (s1, e1, n1) = ch[i], ch[i + 1]
        print('== interval %d: %.1f us between %s and %s' % (i, (s1 - e0) / 1e3, n0[-42:], n1[-42:]))
        for s, e, n in st:
            if s0 < s < s1:
                print('      +%6.1f .. %6.1f us  q%s  %s' % ((s - e0) / 1e3, (e - e0) / 1e3, q.get((s, n), '?'), n[-80:]))
