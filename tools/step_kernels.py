"""per-train-step kernel census of a bench config: run under rocprofv3, then summarise

    rocprofv3 --kernel-trace --stats --output-format csv -d out -o st -- python tools/step_kernels.py c4
    python tools/step_kernels.py --census out/st_kernel_trace.csv          # steady-state window between the marker launches
    python tools/step_kernels.py --by-grid out/st_kernel_trace.csv chain   # per grid size

STEPS identical eager train steps (the launches of the hipGraph replay, one by one) after one initialising step."""
import csv
import importlib
import os
import sys
from types import SimpleNamespace as NS

STEPS = int(os.environ.get("NF_STEPS", 10))


def summarise(path):
    rows = list(csv.DictReader(open(path)))
    tot_calls = sum(int(r['Calls']) for r in rows)
    tot_ns = sum(float(r['TotalDurationNs']) for r in rows)
    n = STEPS + 1
    print('per step: %.0f launches, %.2f ms of kernel time' % (tot_calls / n, tot_ns / n / 1e6))
    print('%-72s %9s %9s %9s' % ('kernel', 'calls/st', 'avg us', 'ms/step'))
    for r in rows[:int(os.environ.get('TOP', 40))]:
        print('%-72s %9.1f %9.1f %9.3f' % (r['Name'][:72], int(r['Calls']) / n, float(r['AverageNs']) / 1e3,
                                          float(r['TotalDurationNs']) / n / 1e6))


def window_rows(path):
    """rows of a *_kernel_trace.csv between the first and the last marker launch (main() brackets the measured steps with
    torch.cuda._sleep: start-up work -- the model upload is one copy per tensor, GradBucket's flattening one per parameter, ~14 k
    copyBuffer launches for the CIFAR Glow -- stays outside the per-step census)"""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'sleep' in r['Kernel_Name'].lower() or 'spin' in r['Kernel_Name'].lower()]
    if len(marks) >= 2:
        return rows[marks[0] + 1:marks[-1]], True
    return rows, False


def census(path):
    """per-step launch census of the steady-state window of a kernel trace: the committed profiles/rNN_cX_step_kernels.txt"""
    rows, windowed = window_rows(path)
    agg = {}
    for r in rows:
        name = r['Kernel_Name']
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    n = STEPS if windowed else STEPS + 1
    tot_calls, tot_ns = sum(a[0] for a in agg.values()), sum(a[1] for a in agg.values())
    print('%s: per step %.0f launches, %.2f ms of kernel time' % ('steady-state window (%d eager steps)' % n if windowed else 'whole process', tot_calls / n, tot_ns / n / 1e6))
    print('%-84s %9s %9s %9s' % ('kernel', 'calls/st', 'avg us', 'ms/step'))
    for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOP', 40))]:
        print('%-84s %9.1f %9.1f %9.3f' % (name[:84], c / n, t / c / 1e3, t / n / 1e6))


def by_grid(path, pattern):
    """average duration of the kernels matching `pattern` per grid size, from the *_kernel_trace.csv of the same run"""
    agg = {}
    rows, windowed = window_rows(path)
    for r in rows:
        name = r['Kernel_Name']
        if pattern not in name:
            continue
        key = (name.split('(')[0][-40:], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])))
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    n = STEPS if windowed else STEPS + 1
    print('%-42s %8s %9s %9s %9s' % ('kernel', 'blocks', 'calls/st', 'avg us', 'ms/step'))
    for (k, g), (c, t) in sorted(agg.items()):
        print('%-42s %8d %9.1f %9.1f %9.3f' % (k, g, c / n, t / c / 1e3, t / n / 1e6))


def main(name):
    import numpy as np
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    pkg = importlib.import_module(bench.PKG)
    nftrain = importlib.import_module(bench.PKG + '.train')
    nfdata = importlib.import_module(bench.PKG + '.data')
    cfg = dict(bench.CONFIGS[name])
    if os.environ.get('NF_BATCH'):           # per-GPU batch override (e.g. config 4's literal 512 on one GPU)
        cfg['batch'] = int(os.environ['NF_BATCH'])
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    np.random.seed(0)
    net = getattr(pkg, cfg['cls'])(cfg['dims'], cfg['datatype'], NS(layers=cfg['layers'], mixtures=cfg['mixtures'])).to(dev)
    trainer = nftrain.FlowTrainer(net, graph=False)
    y = nfdata.sample(cfg['data'], cfg['batch'], 1234)
    if cfg['data'] == 'cifar':
        y = y.reshape((cfg['batch'], ) + cfg['dims'])
    y = y.to(dev)
    for _ in range(2):                       # initialising step + one fused step: outside the window
        trainer.train_on_batch(y)
    torch.cuda.synchronize()
    torch.cuda._sleep(1000)                  # marker launch: the census window opens
    for _ in range(STEPS):
        trainer.train_on_batch(y)
    torch.cuda._sleep(1000)                  # marker launch: the window closes
    torch.cuda.synchronize()


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--summarise':
        summarise(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[1] == '--census':
        census(sys.argv[2])
    elif len(sys.argv) > 3 and sys.argv[1] == '--by-grid':
        by_grid(sys.argv[2], sys.argv[3])
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else 'c2')
