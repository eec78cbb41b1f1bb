"""pure kernel times (eager launches under rocprofv3 --kernel-trace) of the image heads at the CIFAR Glow's levels:
   python tools/probes/head_time.py ;  python tools/probes/head_time.py --report <kernel_trace.csv>"""
import collections, csv, importlib, os, sys
if len(sys.argv) > 2 and sys.argv[1] == '--report':
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[2])):
        n = r['Kernel_Name'].split('(')[0]
        if 'head' in n or 'invconv' in n or 'chan_affine' in n or 'half_move' in n:
            acc[(n[-44:], r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k in sorted(acc):
        v = sorted(acc[k])
        print('%-46s grid %7s  n %4d  median %7.2f us  min %7.2f' % (k[0], k[1], len(v), v[len(v) // 2], v[0]))
    sys.exit(0)
import torch
from types import SimpleNamespace as NS
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
pkg = importlib.import_module(bench.PKG)
net = pkg.Glow((3, 32, 32), 'image', NS(layers=2, mixtures=None)).cuda().train()
y = torch.rand(64, 3, 32, 32, device='cuda')
for _ in range(12):
    z, ld = net(y)
    (z.sum() + ld.sum()).backward()
torch.cuda.synchronize()
print('done')
