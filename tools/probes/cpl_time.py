"""kernel times of the image coupling steps at the three conditioner levels of the CIFAR Glow, fused into the chain launches or not:
run under rocprofv3 --kernel-trace, then   python tools/probes/cpl_time.py --report <kernel_trace.csv>"""
import collections, csv, importlib, os, sys
if len(sys.argv) > 2 and sys.argv[1] == '--report':
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r['Start_Timestamp']))
    acc = collections.defaultdict(list)
    phase = 'fused  '
    for r in rows:
        n = r['Kernel_Name'].split('(')[0]
        if 'affine_img' in n:
            phase = 'unfused'                   # the first stand-alone coupling kernel: the second half of the run
        if 'chain' in n or 'affine_img' in n or 'half_move' in n:
            acc[(phase, n[-40:], r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k in sorted(acc):
        v = sorted(acc[k])
        print('%s %-42s grid %7s  n %4d  median %7.2f us  min %7.2f' % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[0]))
    sys.exit(0)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
pkg = importlib.import_module(bench.PKG)
fc = importlib.import_module(bench.PKG + '.fused_conv')
DEV = 'cuda'
B = 64
for fused in (True, False):
    fc.CONV_COUPLING_ON = fused
    for dims, masking in [((3, 32, 32), 'checkerboard'), ((12, 16, 16), 'channelwise'), ((12, 16, 16), 'checkerboard'),
                          ((48, 8, 8), 'channelwise'), ((48, 8, 8), 'checkerboard')]:
        torch.manual_seed(3)
        k1 = pkg.AffineCoupling(dims, masking=masking, odd=False).to(DEV)
        k1.net.fused = True
        z = torch.randn((B, ) + dims, device=DEV).requires_grad_(True)
        for _ in range(12):
            y, l = k1(z, torch.zeros(B, device=DEV))
            (y.sum() + l.sum()).backward()
        torch.cuda.synchronize()
print('done')
