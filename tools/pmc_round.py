"""
HBM traffic of the kernels bench.py reports a roofline for, from rocprofv3 PMC passes -- collected exactly as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE runs with
--kernel-trace only; FETCH_SIZE doubled on gfx950 (it tallies 128-byte requests at 64 bytes; calibrated in the same run
on a 256 MiB device copy), WRITE_SIZE as reported (KiB).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o fetch -- python tools/pmc_round.py c4 c1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -o write -- python tools/pmc_round.py c4 c1
    python tools/pmc_round.py --json out/.../fetch_counter_collection.csv out/.../write_counter_collection.csv profiles/rNN_pmc.json

The workload is the bench's own train step (eager launches of the trainer bench.py times, via bench.dominant_kernel_roofline), so the kernel names
and shapes are the bench's.  The JSON maps kernel name (up to '<' / '(') -> {batch: {fetch_kib, write_kib, traffic_bytes}}.
"""
import csv
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIDECAR = os.environ.get('NF_PMC_SIDECAR', '/tmp/nf_pmc_kernels.json')
sys.path.insert(0, ROOT)


def to_json(fetch_csv, write_csv, out_path):
    import bench
    agg = {}
    for path in (fetch_csv, write_csv):
        for r in csv.DictReader(open(path)):
            name = (r.get('Kernel_Name') or r.get('Kernel Name') or '')
            key = (name, r['Counter_Name'])
            a = agg.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    out = {'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only); FETCH doubled per MI355X_MICROARCH.md '
                   '(gfx950 tallies 128-byte requests at 64 bytes), calibrated in-run on a 256 MiB copy; per launch, averaged over the launches '
                   'of tools/pmc_round.py (= bench.dominant_kernel_roofline)'}
    cal = {}
    for (name, ctr), (n, v) in agg.items():
        if 'copyBuffer' in name or 'copy_kernel' in name.lower():
            cal.setdefault(name.split('(')[0][:40], {})[ctr] = round(v / n, 1)
    out['calibration_kib'] = cal
    # which kernel does each bench config report?  (name prefix before the first space of roofline.kernel)
    for cfg_name, batch in [(a.split(':')[0], a.split(':')[1] if ':' in a else None) for a in os.environ.get('NF_PMC_CONFIGS', 'c4,c1').split(',')]:
        cfg = bench.CONFIGS[cfg_name]
        B = int(batch) if batch else cfg['batch']
        try:
            named = json.load(open(SIDECAR)).get(cfg_name + (':' + batch if batch else ''))
        except (OSError, ValueError):
            named = None
        for (name, ctr), (n, v) in agg.items():
            short = name.split('(')[0].replace('void ', '')
            base = short.split('<')[0]
            if not base.startswith('k_') or (named is not None and base not in named):
                continue                                 # only the kernels this config's roofline object names
            e = out.setdefault(base, {}).setdefault(str(B) + ':' + short, {})
            e['fetch_kib' if ctr == 'FETCH_SIZE' else 'write_kib'] = round(v / n, 2)
            e['calls'] = n
    for k, sub in list(out.items()):
        if not isinstance(sub, dict) or k == 'calibration_kib':
            continue
        for b, e in sub.items():
            if 'fetch_kib' in e and 'write_kib' in e:
                e['traffic_bytes'] = int(round((2.0 * e['fetch_kib'] + e['write_kib']) * 1024))
    with open(out_path, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True)[:3000])


def main(names):
    import torch
    import bench
    pkg = importlib.import_module(bench.PKG)
    pkg._native.load()
    dev = torch.device('cuda', 0)
    x = torch.randn(2 ** 26, device=dev)                       # 256 MiB read + 256 MiB write: calibration
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    del x, y
    import re
    from types import SimpleNamespace as NS
    import numpy as np
    nftrain = importlib.import_module(bench.PKG + '.train')
    nfdata = importlib.import_module(bench.PKG + '.data')
    which = {}
    for name in names:
        name, _, batch = name.partition(':')                   # 'c4:512' = config 4 at its literal batch on one GPU
        cfg = bench.CONFIGS[name]
        B = int(batch) if batch else cfg['batch']
        torch.manual_seed(0)
        np.random.seed(0)
        net = getattr(pkg, cfg['cls'])(cfg['dims'], cfg['datatype'], NS(layers=cfg['layers'], mixtures=cfg['mixtures'])).to(dev)
        trainer = nftrain.FlowTrainer(net, graph=False)
        y = nfdata.sample(cfg['data'], B, 1234)
        if cfg['data'] == 'cifar':
            y = y.reshape((B, ) + cfg['dims'])
        y = y.to(dev)
        for _ in range(3):
            trainer.train_on_batch(y)
        r = bench.dominant_kernel_roofline(pkg, name, cfg, B, dev, trainer, y)     # three more eager steps: the counted launches
        print(name, r['kernel'], r['us_per_launch'])
        which[name + (':' + batch if batch else '')] = sorted(set(re.findall(r'k_[a-z0-9_]+', r['kernel'])))
        del trainer, net
    torch.cuda.synchronize()
    with open(SIDECAR, 'w') as f:                       # config -> the kernels its roofline object names (read back by --json)
        json.dump(which, f)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--json':
        to_json(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        main(sys.argv[1:] or ['c4', 'c1'])
