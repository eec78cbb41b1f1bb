"""
Model-level asymptotic sweep: the train step of the bench configs at batches far beyond the latency regime --
Glow on CIFAR-shape batches B = 512, 2048, 4096 (8192 samples of 32 x 32 x 3 keep ~ 0.27 GB per saved 16 x 16 activation, ~ 10 of them per
conditioner, 64 such conditioners: past what 288 GB hold next to the 8 x 8 and 4 x 4 levels -- run on request), the 2-D models up to 2^22 rows
(2^24 rows of a 32-step model keep 24.6 KB per row for the backward: 412 GB) -- samples/s and, for the image model, the whole-step matrix-pipe and HBM fractions.

    python tools/model_sweep.py [--big]   > profiles/rNN_model_sweep.txt
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = [('c4', 512, 6), ('c4', 2048, 3), ('c4', 4096, 2), ('c1', 65536, 10), ('c1', 1 << 20, 5), ('c2', 65536, 10), ('c2', 1 << 20, 5), ('c2', 1 << 22, 3),
        ('c3', 1 << 20, 5), ('c5', 1 << 20, 5), ('c5', 1 << 22, 3)]
if '--big' in sys.argv:
    RUNS.insert(3, ('c4', 8192, 2))
print('%-6s %10s %14s %12s %10s %10s   %s' % ('config', 'batch', 'samples/s', 'ms/step', 'mfma', 'hbm', 'dominant kernel (us per launch)'))
for cfg, B, steps in RUNS:
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', cfg, '--batch', str(B), '--steps', str(steps), '--warmup', '2',
                        '--skip-cpu'], capture_output=True, text=True, timeout=1500)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if r.returncode != 0 or not line:
        print('%-6s %10d   failed: %s' % (cfg, B, (r.stderr.strip().splitlines() or ['?'])[-1][:120]))
        continue
    d = json.loads(line[-1])
    w = d.get('whole_step') or {}
    roof = d.get('roofline') or {}
    print('%-6s %10d %14.1f %12.3f %10s %10s   %s (%.1f)' % (cfg, B, d['value'], d['ms_per_step'],
          ('%.3f' % w['mfma_frac']) if 'mfma_frac' in w else '-', ('%.3f' % w['hbm_frac']) if 'hbm_frac' in w else '-',
          (roof.get('kernel') or '-').split(' (')[0], roof.get('us_per_launch') or 0.0))
    sys.stdout.flush()
