"""
Workload for the rocprofv3 --pmc passes of the dominant kernels of C3 and C4 exactly as bench.py launches them for its
roofline figure (bench.dominant_kernel_roofline), after a calibration copy of known size.  FETCH_SIZE / WRITE_SIZE in
SEPARATE runs, kernel-trace only:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o fetch -- python tools/pmc_roofline_kernels.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -o write -- python tools/pmc_roofline_kernels.py
    python tools/pmc_probe.py --summarise out/fetch_counter_collection.csv out/write_counter_collection.csv
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
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
    for name in sys.argv[1:] or ['c3', 'c4']:
        cfg = bench.CONFIGS[name]
        r = bench.dominant_kernel_roofline(pkg, cfg, cfg['batch'], dev)
        print(name, r['kernel'], r['us_per_launch'])
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
