# HBM traffic and duration of the deferred weight-gradient launches of a C4 step (FETCH_SIZE / WRITE_SIZE in separate passes, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pw_$ctr
  NF_STEPS=3 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pw_$ctr -o p -- python $GRAFT_REPO_ROOT/tools/step_kernels.py c4 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('/tmp/pw_%s/**/p_counter_collection.csv' % ctr, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0]
        if 'wgrad' not in n and 'chain_bwd<4' not in n: continue
        a = agg[(n, r.get('Grid_Size', ''))]
        if ctr == 'FETCH_SIZE': a[0] += 1; a[1] += float(r['Counter_Value'])
        else: a[2] += float(r['Counter_Value'])
for (n, gsz), (c, fe, wr) in sorted(agg.items()):
    print('%-70s grid %-9s calls %4d  fetch %9.1f MB (x2 on gfx950: %9.1f)  write %9.1f MB per launch' % (n[-70:], gsz, c, fe / c / 1024, 2 * fe / c / 1024, wr / c / 1024))
PY
