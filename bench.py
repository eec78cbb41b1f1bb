#!/usr/bin/env python
"""
bench.py -- samples/sec (+ bits/dim) of one training step of the flow hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2] [--no-graph] [--skip-cpu]

A "step" is one full pass of the hot path over one batch of synthetic input: forward flow + log-det, NLL,
autograd backward through every transform kernel, gradient all-reduce (N > 1), Adam -- main.py:78-92.
Default workloads = the two configs BASELINE.json's metric is quoted on: Glow CIFAR-10 (3,32,32) L=3 K=32, batch 64 per GPU
(= 512 over 8, configs[3]) on the top level of the line, and RealNVP moons-2D K=32 batch 256 (configs[0]) under "also".
`--config c1..c5` measures one workload.  `--gpus N` outside torchrun re-executes itself under torch.distributed.run.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`, `cpu_baseline` and `parity` objects.

Timed region: inputs already resident in HBM; barrier + synchronize on both sides; max over ranks.
"""
import argparse
import ctypes
import importlib
import json
import math
import os
import sys
import time
from types import SimpleNamespace as NS

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = 'normalizing-flows-pytorch_amd'

CONFIGS = {
    # name: model class, oracle kind, dims, datatype, layers, mixtures, data, per-GPU batch
    # `batch` = per-GPU batch of the weak-scaling runs (the default); `global_batch` = the fixed job batch of --scaling strong (BASELINE.json's
    # literal figure where it names one: C4 "batch 512 ... over 8", C5 "batch 131072, 8 x"), per GPU = global_batch / N there
    'c1': dict(cls='RealNVP', kind='realnvp', dims=(2, ), datatype='2d', layers=32, mixtures=None, data='moons', batch=256, global_batch=256,
               desc='RealNVP moons-2D K=32 batch 256', cpu_threads=1),
    'c2': dict(cls='Glow', kind='glow', dims=(2, ), datatype='2d', layers=32, mixtures=None, data='moons', batch=4096, global_batch=4096,
               desc='Glow moons-2D K=32 batch 4096 per GPU'),
    'c3': dict(cls='Flowpp', kind='flowpp', dims=(2, ), datatype='2d', layers=32, mixtures=8, data='circles', batch=65536, global_batch=65536,
               desc='Flow++ circles-2D K=32 mixtures=8 batch 65536 per GPU'),
    'c4': dict(cls='Glow', kind='glow', dims=(3, 32, 32), datatype='image', layers=32, mixtures=None, data='cifar',
               batch=64, global_batch=512, desc='Glow CIFAR-shape (3,32,32) L=3 K=32 batch 64 per GPU (512 over 8)', cpu_threads=16),
    'c5': dict(cls='MAF', kind='maf', dims=(2, ), datatype='2d', layers=10, mixtures=None, data='normals', batch=16384, global_batch=131072,
               desc='MAF normals-2D 10 AR layers batch 16384 per GPU (131072 over 8)'),
    # not BASELINE.json configs: the two other models north_star names on CIFAR-shape batches, at the reference's default depth
    # (configs/default.yaml: network.layers = 32): flows/flowpp.py:17-62 and flows/realnvp.py:17-47.  Measured in the default line under "also"
    'fpp_img': dict(cls='Flowpp', kind='flowpp', dims=(3, 32, 32), datatype='image', layers=32, mixtures=8, data='cifar', batch=64,
                    global_batch=512, desc='Flow++ CIFAR-shape (3,32,32) layers=32 mixtures=8 batch 64 per GPU', cpu_threads=16),
    'rnvp_img': dict(cls='RealNVP', kind='realnvp', dims=(3, 32, 32), datatype='image', layers=32, mixtures=None, data='cifar', batch=64,
                     global_batch=512, desc='RealNVP CIFAR-shape (3,32,32) layers=32 batch 64 per GPU', cpu_threads=16),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default=None, choices=sorted(CONFIGS),
                    help='one workload; default: c4 (Glow CIFAR-10) on the top level + c1 (RealNVP moons) under "also"')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch override (asymptotic sweeps)')
    ap.add_argument('--layers', type=int, default=None, help='flow steps per level override (e.g. the reference default 32 for fpp_img)')
    ap.add_argument('--scaling', default='weak', choices=('weak', 'strong'),
                    help='weak (default): the per-GPU batch is fixed as N grows; strong: the GLOBAL batch is fixed (C4 512, C5 131072, ...), '
                         'per GPU = global / N')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--skip-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    return ap.parse_args()


def event_time_ms(fn, reps, stream):
    """average duration of `fn` over `reps` launches, HIP events recorded on `stream` (the launch stream)."""
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        fn()
    start.record(stream)
    for _ in range(reps):
        fn()
    stop.record(stream)
    stop.synchronize()
    return start.elapsed_time(stop) / reps


def graph_time_us(fn, dev, per_graph=50, replays=10, reset=None):
    """average duration of one launch of `fn` inside a hipGraph of `per_graph` back-to-back (dependent) launches: no host
    launch cost in the number, HIP events recorded on the stream the graph is replayed on.  reset(): re-zeroes the exchange
    workspaces of persistent kernels after the untimed warm replay (a timed replay must not find the previous replay's
    generation stamps in the slots: its exchanges would not wait for anybody)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(per_graph):
            fn()
    g.replay()
    torch.cuda.synchronize()
    if reset is not None:
        assert replays == 1, 'one timed replay per set of zeroed workspaces'
        reset()
        torch.cuda.synchronize()
    stream = torch.cuda.current_stream()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record(stream)
    for _ in range(replays):
        g.replay()
    stop.record(stream)
    stop.synchronize()
    return start.elapsed_time(stop) * 1e3 / (per_graph * replays)


def pmc_traffic(kernel, B, contains=''):
    """HBM bytes per launch of `kernel` at batch B from THIS round's committed rocprofv3 PMC passes (the newest profiles/rNN_pmc.json,
    written by tools/pmc_round.py from separate FETCH_SIZE / WRITE_SIZE runs of the same train step), or None.  PMC collection needs
    its own profiler runs, so it cannot happen inside the timed bench process; the kernel's DURATION in the same object is measured
    live, and tests/test_cabi.py checks that the file belongs to the current round's kernels."""
    import glob
    try:
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc.json')))
        with open(files[-1]) as f:
            sub = json.load(f).get(kernel, {})
        for key, e in sub.items():
            if key.split(':')[0] == str(B) and contains in key and 'traffic_bytes' in e:
                return e['traffic_bytes']
    except (OSError, ValueError, IndexError):
        pass
    return None


def pmc_file():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc.json')))
    return os.path.relpath(files[-1], ROOT) if files else None


MFMA_F32_TFLOPS = 157.3            # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector peak
MFMA_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak (the 2:1-sparsity headline figure is never used)


def gpu_delay(ms):
    """keep the device busy for ~ms milliseconds (calibrated spin kernel) so that the host runs AHEAD of it: the launches enqueued
    behind the delay sit back to back in the queue, and HIP events around one of them see device time, not host launch latency."""
    if not hasattr(gpu_delay, 'cycles_per_ms'):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
        a.record()
        torch.cuda._sleep(2000000)
        b.record()
        b.synchronize()
        gpu_delay.cycles_per_ms = 2000000 / max(a.elapsed_time(b), 1e-3)
    torch.cuda._sleep(int(ms * gpu_delay.cycles_per_ms))


def conv_glow_step_counts(cfg, B):
    """(flops, algorithmic HBM bytes) of one train step of an image Glow / RealNVP stack: 2 x MACs of every conditioner convolution
    (forward, data gradient, weight gradient: 3 x; the 1x1 flow convolutions and the elementwise transforms are < 1 % of the MACs)
    and SURVEY.md section 8(d)'s ideal fused traffic (20 B per element of z per flow step forward, training ~ 3 x forward)."""
    C, H, W = cfg['dims']
    K = cfg['layers']
    mac, elem_steps = 0, 0
    mid = (C, H, W)

    def cond(c_in, c_out, h, w):
        return h * w * (9 * c_in * 32 + 4 * 9 * 32 * 32 + 32 * c_out)
    while max(mid[1], mid[2]) > 8:
        mac += K * cond(2 * mid[0], 4 * mid[0], mid[1] // 2, mid[2] // 2)            # checkerboard: halves are (2C, H/2, W/2)
        mid = (mid[0] * 4, mid[1] // 2, mid[2] // 2)
        mac += K * cond(mid[0] // 2, mid[0], mid[1], mid[2])                          # channel-wise: halves are (C/2, H, W)
        elem_steps += 2 * K
    mac += (K + 1) * cond(2 * mid[0], 4 * mid[0], mid[1] // 2, mid[2] // 2)
    elem_steps += K + 1
    return 3 * 2 * mac * B, 3 * elem_steps * 20 * C * H * W * B


def dominant_kernel_roofline(pkg, name, cfg, B, dev, trainer, y, reps=3):
    """roofline of the kernel that dominates the timed region (rocprofv3 summaries under profiles/), measured ON THE RUN: `reps` more
    train steps of the very trainer that was timed -- same weights, same resident batch -- are launched eagerly behind a device-side
    delay (so the host is ahead of the device and the launches queue back to back, as in the hipGraph replay), with HIP events on the
    launch stream around every launch of the dominant entry point (_native.timed_launches).
      c1      -> nf_realnvp_flow_vec_bwd: k_glow_flow_bwd<2>, the backward of all 32 RealNVP flow steps in one launch (csrc/mlp_chain.hip)
      c2      -> nf_glow_flow_vec_bwd_deferred: k_glow_flow_bwd<1>, the backward of all 32 Glow flow steps in one launch (+ the one fold launch);
                 with NF_GLOW_FLOW=steps: nf_glow_flow_steps_bwd, 32 x k_mlp_chain_bwd<1> + 1 x k_glow_fold_all per call (average over the 33)
      c3      -> nf_flowpp_vec_step_bwd: k_flowpp_cond_bwd<2, true> (gated-attention conditioner + mixture coupling, backward)
      c4      -> nf_convnet_chain_bwd at 16 x 16: the data gradient of a whole image conditioner + coupling backward, one persistent launch
      c5      -> nf_maf_step_bwd_partial: k_maf_step_bwd, the backward of a whole MAF flow step
    achieved = algorithmic flops per launch / the measured average duration."""
    N = pkg._native
    dims, S = cfg['dims'], int(cfg['layers'])
    D = dims[0]
    extra = {}
    if len(dims) == 3 and cfg['kind'] in ('glow', 'realnvp'):
        Hh, Ww = dims[1] // 2, dims[2] // 2
        I0, O = 2 * dims[0], 4 * dims[0]
        entry, match = 'nf_convnet_chain_bwd', (lambda a: int(a[4]) == Hh and int(a[5]) == Ww)
        M = B * Hh * Ww
        flop = 2 * M * (4 * 9 * 32 * 32 + 9 * 32 * I0 + 32 * O)          # five transposed convolutions, data gradient only
        nbytes = 4 * (M * 32 * (5 + 5 + 2) + 2 * M * O + 3 * B * dims[0] * dims[1] * dims[2])
        wgs = int(N.load().nf_convnet_chain_blocks(B, I0, O, Hh, Ww))     # one 1024-thread workgroup per tile
        kname = ('k_convnet_chain_bwd (data gradient of the whole ConvNet conditioner + coupling backward in one persistent launch: '
                 '%d -> 32 x 5 -> %d channels, %d x %d)' % (I0, O, Hh, Ww))
        pmc = ('k_convnet_chain_bwd', '18, 181')
        per_call = 1
        if wgs == 0:
            # batches beyond the persistent chain's co-residency limit at this level (B = 512 on one GPU): the conditioners run as one
            # launch per layer; the 3 x 3 layers on the large-batch kernels of csrc/conv_bulk.hip (independent waves, three-way bf16
            # split).  The dominant one by time per step is the data-gradient pass of a 32 -> 32 channel layer at 16 x 16
            entry, match = 'nf_conv_bn_bwd', (lambda a: int(a[2]) == 32 and int(a[3]) == 32 and int(a[4]) == Hh and int(a[6]) == 3)
            flop = 2 * M * 9 * 32 * 32
            nbytes = int(4 * M * 32 * 4.5)          # gn_src, out, in read + gn_out written (+ g_skip read, g_store written: every fourth launch)
            kname = ('k_conv3_bulk_bwd (data gradient of one 32 -> 32 channel 3 x 3 layer: BatchNorm backward on load, transposed '
                     'convolution on the bf16 pipe, ReLU mask, batch sums; %d x %d)' % (Hh, Ww))
            pmc = ('k_conv3_bulk_bwd', '<2, ')
            extra['bf16_split'] = True
            extra['hbm'] = True
        note = ('%s of 256 compute units hold the launch; five dependent convolutions inside five grid-wide BatchNorm exchanges '
                '(DESIGN.md section 3.16)' % wgs) if wgs else 'per-layer launches: the batch exceeds the persistent chain at this level'
        extra['workgroups'] = wgs or None
        if wgs:
            extra['bf16_split'] = True
    elif cfg['kind'] in ('glow', 'realnvp') and len(dims) == 1:
        glow = cfg['kind'] == 'glow'
        F = importlib.import_module(PKG + '.fused')
        per_step = glow and F._glow_steps_on(y)
        if per_step:
            entry, per_call = 'nf_glow_flow_steps_bwd', S + 1
            flop = 12 * 2 * 32 * 32 * B                                   # ALGORITHMIC: 6 data-gradient + 6 weight-gradient 32x32 products (the 5 recomputed forward ones are not counted)
            kname = 'k_mlp_chain_bwd<1> (whole Glow flow step, one launch, gradient fold deferred; average over %d step launches + 1 k_glow_fold_all)' % S
            pmc = ('k_mlp_chain_bwd', '')
            nbytes = B * (3 * D + 1) * 4
        else:
            entry = 'nf_glow_flow_vec_bwd' if glow else 'nf_realnvp_flow_vec_bwd'
            if F.FLOW_DEFER_FOLD:
                entry += '_deferred'                         # (+ the one k_glow_fold_all launch behind it, inside the bracket)
            per_call = 1
            solo = (not glow) and D == 2 and B <= N.header_constant('NF_FLOW_SOLO_MAX_ROWS') and F.FLOW_DEFER_FOLD
            if solo:
                # C1: the one-workgroup backward (csrc/flow_solo.hip) reads the activations the forward stashed -- 12 products of 32 x 32 per
                # row and step (6 data-gradient + 6 weight-gradient), no recomputed ones
                flop = S * 12 * 2 * 32 * 32 * B
                kname = 'k_solo_bwd (backward of all %d RealNVP flow steps in ONE workgroup, activations from the forward\'s stash; + k_glow_fold_all)' % S
                pmc = ('k_solo_bwd', '')
                nbytes = S * B * ((3 * D + 1) * 4 + 5 * 32 * 4)       # rows in / out + the stash read back
                extra['workgroups'] = 1
            else:
                flop = S * 12 * 2 * 32 * 32 * B                  # ALGORITHMIC: 6 data-gradient + 6 weight-gradient products (the kernel also recomputes 5 forward ones: not counted)
                kname = 'k_glow_flow_bwd<%d> (backward of all %d %s flow steps, one launch)' % (1 if glow else 2, S, 'Glow' if glow else 'RealNVP')
                pmc = ('k_glow_flow_bwd', '<1>' if glow else '<2>')      # (C1 and C2 run two instantiations of the same kernel)
                nbytes = S * B * (3 * D + 1) * 4
        match = None
        note = ('neither MFMA- nor HBM-bound at this batch: per flow step six grid-wide (or workgroup-wide) BatchNorm reductions and '
                'single-tile issue latency serialise the launch (DESIGN.md sections 2 and 3.11)')
    elif cfg['kind'] == 'maf':
        entry, match, per_call = 'nf_maf_step_bwd_partial', None, 1
        flop = 2 * 2 * 3 * (32 * D + 1024 + 1024 + 32 * D) * B             # two nets x (recompute + data + weight gradients)
        nbytes = B * (3 * D + 1) * 4
        kname = 'k_maf_step_bwd (whole MAF flow step, one launch; gradient fold deferred to one k_maf_fold_all per pass)'
        pmc = ('k_maf_step_bwd', '')
        note = 'neither MFMA- nor HBM-bound at this batch: four grid-wide BatchNorm exchanges serialise the launch (DESIGN.md 3.13)'
    elif cfg['kind'] == 'flowpp' and len(dims) == 1:
        K = cfg['mixtures']
        O, I0 = (2 + 3 * K) * (D - D // 2), D // 2
        entry, match, per_call = 'nf_flowpp_vec_step_bwd', None, 1
        mac = (2048 + 1024 + 2048) + 2 * (O * 32 + 2048 + 1024 + 2048) + 32 * I0      # recompute + data + weight gradients
        flop = 2 * mac * B
        nbytes = B * (2 * I0 + O) * 4
        kname = 'k_flowpp_cond_bwd<2, true> (gated-attention conditioner + mixture coupling, backward; slab finalize deferred)'
        pmc = ('k_flowpp_cond_bwd', '')
        note = ('fp32-input MFMA (exact fp32, 1/16 of the bf16 rate); the rest is transcendental VALU work and the LDS transposes of '
                'the weight-gradient operands (DESIGN.md section 3.12)')
    elif cfg['kind'] == 'flowpp' and len(dims) == 3:
        # image Flow++ (csrc/flowpp_img.hip, flowpp_img_att.hip): the longest launch is the softmax backward of the 16 x 16 level -- per
        # (sample, head) workgroup the two sweeps over the 256 x 256 scores, VALU work (the fp32 vector peak equals the fp32 matrix peak)
        Np = (dims[1] // 2) * (dims[2] // 2)
        fpi = importlib.import_module(PKG + '.fused_flowpp_img')
        if B < fpi.SPLIT_BELOW:
            entry, match, per_call = 'nf_flowpp_img_att_bwd', (lambda a: int(a[-3]) * int(a[-2]) == Np), 1
            mac = 4 * Np * Np * (24 + 32) + 4 * Np * (2 * 24 * 32 + 24 * 32)          # two sweeps; conv1 rows forward, data and weight gradient
            nbytes = 4 * B * Np * (32 * 2 + 32 * 2 + 4 + 4 * 32)
            kname = ('k_fi_att_bwd<256> (softmax backward of one attention head over %d positions + the head\'s conv1 rows, one '
                     'workgroup per (sample, head))' % Np)
            pmc = ('k_fi_att_bwd', '')
            wgs = 4 * B
        else:
            entry, match, per_call = 'nf_flowpp_img_mid_bwd', (lambda a: int(a[-4]) * int(a[-3]) == Np), 1
            mac = 3 * Np * (96 * 32 + 64 * 32) + 4 * Np * Np * (16 + 16 + 32)
            nbytes = 4 * B * 32 * Np * 5
            kname = 'k_fi_mid<256, true> (gate + LayerNorm + attention over %d positions + LayerNorm, backward, one workgroup per sample)' % Np
            pmc = ('k_fi_mid', '')
            wgs = B
        flop = 2 * mac * B
        note = ('vector-ALU kernel (softmax sweeps with one exp per score, 8-wide dot products from LDS): %d workgroups on 256 compute '
                'units at this batch' % wgs)
    else:
        return None
    with N.timed_launches(entry, match) as t:
        for _ in range(reps):
            torch.cuda.synchronize()
            gpu_delay(250.0 if len(dims) == 3 else 40.0)
            trainer._forward_backward(y)
        d = t.durations_us()
    if not d:
        return None
    us = sum(d) / len(d) / per_call
    tf = flop / (us * 1e-6) / 1e12
    out = {'bound': 'mfma', 'kernel': kname, 'achieved': round(tf, 3), 'peak': MFMA_F32_TFLOPS, 'unit': 'TFLOP/s',
           'frac': round(tf / MFMA_F32_TFLOPS, 5), 'traffic': pmc_traffic(pmc[0], B, pmc[1]), 'flop_per_launch': int(flop),
           'traffic_source': 'rocprofv3 PMC passes of the same train step, committed as %s (separate profiler runs; not measured in this '
                             'process)' % (pmc_file() or 'profiles/rNN_pmc.json -- none found'),
           'bytes_per_launch': int(nbytes), 'us_per_launch': round(us, 3), 'launches_timed': len(d) * per_call,
           'us_min_max': [round(min(d) / per_call, 2), round(max(d) / per_call, 2)],
           'how': 'HIP events on the launch stream around every %s call of %d eager train steps of the timed trainer (real weights, '
                  'real activations), enqueued behind a device-side delay so that they queue back to back' % (entry, reps),
           'note': note}
    if extra.get('bf16_split'):
        # the kernel forms every fp32 product as SIX bf16 x bf16 MFMA products of three-way splits: the work the matrix
        # pipe actually executes is 6 x the algorithmic flops, priced against the dense bf16 peak (2.5 PFLOP/s)
        out['bf16_pipe'] = {'hardware_tflops': round(6.0 * tf, 2), 'peak': MFMA_BF16_TFLOPS, 'frac': round(6.0 * tf / MFMA_BF16_TFLOPS, 5),
                            'note': 'six v_mfma_f32_32x32x16_bf16 per fp32 product: hardware flops = 6 x algorithmic, against the dense bf16 peak'}
        extra.pop('bf16_split')
    if extra.pop('hbm', None):
        # the large-batch kernels move whole activation tensors per launch: the same launch against the HBM roof (algorithmic bytes)
        gbs = nbytes / (us * 1e-6) / 1e9
        out['hbm'] = {'achieved_gbs': round(gbs, 1), 'peak_gbs': HBM_PEAK_GBS, 'frac': round(gbs / HBM_PEAK_GBS, 5),
                      'note': 'algorithmic bytes per launch (DESIGN.md 3.24) over the measured duration: this launch sits between the two roofs'}
    out.update({k: v for k, v in extra.items() if v is not None})
    return out


def cpu_baseline(cfg, state, y_cpu, seconds):
    """the oracle (our CPU restatement of the reference, validated against it) timed on this box's host cores on the
    same workload and weights: full train step incl. Adam, bounded to ~`seconds` of CPU work.  Returns (object, first):
    ``first`` = (z, loss) of the FIRST step from the initial weights -- the same step the GPU trainer's first call performs
    (data-dependent ActNorm initialisation included), which is what the bench line's ``parity`` object compares."""
    from oracle import models as om
    from oracle import transforms as tf
    # the flow step is thousands of tiny ops: torch's intra-op pool stops scaling (and then collapses) beyond a few threads.
    # Measured once per round on the GPU box's 256 host cores (tools/cpu_threads.py, profiles/r06_cpu_threads.txt): C4 1297 / 1239 /
    # 2703 / 7002 ms per step at 8 / 16 / 32 / 64 threads, C1 32 / 47 / 55 / 60 / 128 ms at 1 / 4 / 8 / 16 / 32 -- the baseline runs
    # at the FASTEST setting of its config (C4: 16, C1: 1; the others 8, what the reference was probed with in BASELINE.md);
    # `cores` reports what ran
    cores = min(os.cpu_count() or 1, int(cfg.get('cpu_threads', 8)))
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu().clone() for k, v in state.items()}
    ora = om.FlowOracle(cfg['kind'], cfg['dims'], cfg['datatype'], cfg['layers'], sd, mixtures=cfg['mixtures'],
                        training=True).requires_grad_(True)
    params = list(ora.parameters().values())
    opt = torch.optim.Adam(params, lr=1.0e-4)
    B = y_cpu.shape[0]
    last = {}

    def step():
        opt.zero_grad()
        z, ld = ora.forward(y_cpu)
        loss = tf.nll_loss(z, ld)
        loss.backward()
        opt.step()
        last['z'] = z.detach()
        return float(loss.detach())

    loss1 = step()                                           # ActNorm init + warm caches (untimed); parity is taken here
    first = (last['z'].clone(), loss1)
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 200:
            break
    model = 'unknown'
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return ({'value': round(B * n / el, 1), 'unit': 'samples/s', 'cores': cores, 'kind': 'port',
             'sample': '%d train steps of the same workload (batch %d, same initial weights) in %.1f s on %s' % (n, B, el, model),
             'ms_per_step': round(1e3 * el / n, 2), 'host_cores': os.cpu_count(),
             'threads_note': 'fastest of the per-round thread sweep for this config (profiles/r06_cpu_threads.txt)'}, first)


def resolve_batch(cfg, scaling, batch_override, world):
    """(per-GPU batch, config, strong?) of one workload on `world` ranks.  weak: the per-GPU batch is the config's `batch` whatever N is (the
    job's batch grows with N); strong: the config's `global_batch` is the JOB's batch and every rank takes global / N rows of it
    (the minibatch is sharded, no data-path collective; per-replica batch statistics as in DESIGN.md section 5)."""
    strong = scaling == 'strong'
    if strong and not batch_override:
        G = cfg['global_batch']
        if G % world:
            raise SystemExit('--scaling strong: the global batch %d does not divide over %d ranks' % (G, world))
        B = G // world
        cfg = dict(cfg, desc=cfg['desc'].split(' batch ')[0] + ' GLOBAL batch %d, %d per GPU (strong scaling)' % (G, B))
        return B, cfg, True
    return batch_override or cfg['batch'], cfg, strong


def run_workload(name, args, pkg, rank, world, dev, steps, warmup, cpu_seconds):
    """one workload: W warm-up steps, K timed steps (barrier + synchronize on both sides, max over ranks), then on rank 0
    the dominant kernel's roofline, the CPU baseline and the step-1 parity; returns the JSON object (None off rank 0)."""
    nfdist = importlib.import_module(PKG + '.dist')
    nftrain = importlib.import_module(PKG + '.train')
    nfdata = importlib.import_module(PKG + '.data')
    cfg = CONFIGS[name]
    if getattr(args, 'layers', None):
        cfg = dict(cfg, layers=args.layers, desc=cfg['desc'] + ' -- measured with layers=%d' % args.layers)
    B, cfg, strong = resolve_batch(cfg, getattr(args, 'scaling', 'weak'), args.batch, world)

    torch.manual_seed(0)                                     # identical initial weights on every rank
    np.random.seed(0)
    net = getattr(pkg, cfg['cls'])(cfg['dims'], cfg['datatype'], NS(layers=cfg['layers'], mixtures=cfg['mixtures']))
    state0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to(dev)
    nfdist.broadcast_parameters(net)
    trainer = nftrain.FlowTrainer(net, graph=not args.no_graph, warmup=2)

    y_cpu = nfdata.sample(cfg['data'], B, 1234 + rank)
    if cfg['data'] == 'cifar':
        y_cpu = y_cpu.reshape((B, ) + cfg['dims'])
    y = y_cpu.to(dev)                                        # resident in HBM before the timed region

    z1, loss1 = trainer.train_on_batch(y)                    # step 1 (the parity object below compares THIS step with the CPU)
    z1, loss1 = z1.detach().cpu().clone(), float(loss1)
    for _ in range(max(warmup, 4) - 1):                      # >= 4 so that graph capture happens before timing
        trainer.train_on_batch(y)
    torch.cuda.synchronize()
    if world > 1 or torch.distributed.is_initialized():   # (initialised at one rank: NF_DP_FORCE_COLLECTIVE=1, the DP control flow on one GPU)
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        z, loss = trainer.train_on_batch(y)
    torch.cuda.synchronize()
    if world > 1 or torch.distributed.is_initialized():   # (initialised at one rank: NF_DP_FORCE_COLLECTIVE=1, the DP control flow on one GPU)
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1 or torch.distributed.is_initialized():   # (initialised at one rank: NF_DP_FORCE_COLLECTIVE=1, the DP control flow on one GPU)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss)

    # the MEDIAN of hipEvent-timed steps next to the wall-clock mean above -- the same K steps again, one event pair per
    # step on the launch stream (outside the contract's timed region, so that the event records cannot perturb `value`)
    stream = torch.cuda.current_stream()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(steps, 20))]
    for a, b in evs:
        a.record(stream)
        trainer.train_on_batch(y)
        b.record(stream)
    torch.cuda.synchronize()
    ev_ms = sorted(a.elapsed_time(b) for a, b in evs)
    ev_median, ev_min = ev_ms[len(ev_ms) // 2], ev_ms[0]

    # forward-only and inverse-only rates (eval mode, no autograd), informational
    with torch.no_grad():
        net.eval()
        zz, _ = net(y)

        def replay_ms(fn):
            """one pass as a hipGraph replay (like the training step); eager launches if capture is refused"""
            try:
                return graph_time_us(fn, dev, per_graph=1, replays=10) / 1e3
            except Exception:
                return event_time_ms(fn, 5, torch.cuda.current_stream())
        fwd_ms = replay_ms(lambda: net(y))
        inv_ms = replay_ms(lambda: net.backward(zz))
        net.train()

    timeouts = pkg._native.persistent_timeouts()
    assert timeouts == 0, 'persistent kernels timed out %d times: the GPU was shared, results invalid' % timeouts
    if rank != 0:
        return None
    roof = dominant_kernel_roofline(pkg, name, cfg, B, dev, trainer, y)
    ms_step = 1e3 * elapsed / steps
    whole = None
    if cfg['datatype'] == 'image' and cfg['kind'] in ('glow', 'realnvp'):
        flops, nbytes = conv_glow_step_counts(cfg, B)
        whole = {'flop_per_step': int(flops), 'mfma_tflops': round(flops / (ms_step * 1e-3) / 1e12, 3),
                 'mfma_frac': round(flops / (ms_step * 1e-3) / 1e12 / MFMA_F32_TFLOPS, 5),
                 'hbm_bytes_per_step': int(nbytes), 'hbm_gbs': round(nbytes / (ms_step * 1e-3) / 1e9, 2),
                 'hbm_frac': round(nbytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                 'note': 'whole train step against both roofs: 2 x MACs of every conditioner convolution (forward + data gradient + '
                         'weight gradient) over the fp32-MFMA peak, and SURVEY 8(d)\'s ideal fused traffic over 8 TB/s'}
    elif len(cfg['dims']) == 1:
        # (products of 32 x 32 per row and flow step: 6 forward + 6 data-gradient + 6 weight-gradient; the kernels that recompute the
        #  forward in their backward execute 5 more, which are not algorithmic work)
        per_row = {'glow': 18 * 2 * 32 * 32, 'realnvp': 18 * 2 * 32 * 32, 'maf': 2 * 2 * 3 * (64 * cfg['dims'][0] + 2048)}.get(cfg['kind'])
        if cfg['kind'] == 'flowpp':
            # gated-attention conditioner of one coupling (coupling.py:159-166 at one position per sample): Linear(I0,32), the gate (32 -> 64),
            # the attention's value / output rows that survive (32 -> 32 + 32 -> 64), Linear(32, O); forward + data gradient + weight gradient
            Dd, Kk = cfg['dims'][0], cfg['mixtures']
            Oo, Ii = (2 + 3 * Kk) * (Dd - Dd // 2), Dd // 2
            per_row = 3 * 2 * (32 * Ii + 2048 + 1024 + 2048 + 32 * Oo)
        if per_row is not None:
            flops = per_row * B * cfg['layers']
            whole = {'flop_per_step': int(flops), 'mfma_tflops': round(flops / (ms_step * 1e-3) / 1e12, 3),
                     'mfma_frac': round(flops / (ms_step * 1e-3) / 1e12 / MFMA_F32_TFLOPS, 5)}
            if cfg['kind'] == 'flowpp':
                # the mixture coupling itself is transcendental VALU work on (2 + 3K) parameters per element: its HBM side (SURVEY 8(d):
                # parameters read once per direction, 4 B each, three passes) is the larger roof fraction of this config
                nb = 3 * 4 * (Oo + 4 * Dd) * B * cfg['layers']
                whole.update({'hbm_bytes_per_step': int(nb), 'hbm_gbs': round(nb / (ms_step * 1e-3) / 1e9, 2),
                              'hbm_frac': round(nb / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)})
    out = {
        'metric': 'samples/sec (train step: forward flow + log-det + NLL + backward + Adam)',
        'value': round(B * world * steps / elapsed, 1),
        'unit': 'samples/s',
        'n_gpus': world,
        'steps': steps,
        'warmup': warmup,
        'ms_per_step': round(1e3 * elapsed / steps, 4),
        'ms_per_step_event_median': round(ev_median, 4),
        'ms_per_step_event_min': round(ev_min, 4),
        'samples_per_s_event_median': round(B * world / (ev_median * 1e-3), 1),
        'higher_is_better': True,
        'scaling': 'strong' if strong else 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic (seeded %s restatement, random-init weights)' % cfg['data'],
        'config': {'workload': cfg['desc'] if not args.batch else cfg['desc'] + ' -- measured at per-GPU batch %d' % B, 'name': name, 'per_gpu_batch': B, 'global_batch': B * world,
                   'scaling': 'strong (global batch fixed, per GPU = global / N)' if strong else 'weak (per-GPU batch fixed)', 'parallelism': 'dp%d' % world, 'hipgraph': trainer._g_fb is not None, 'deterministic': bool(pkg._native.deterministic()), 'dp_one_graph': bool(getattr(trainer, '_g_whole', False)) and trainer.bucket.collective, 'collective': collective_info(world)},
        'loss_nats': round(loss_val, 5),
        'bits_per_dim': round(nftrain.bits_per_dim(loss_val, cfg['dims']), 5),
        'forward_samples_per_s': round(B * world / (fwd_ms * 1e-3), 1),
        'inverse_samples_per_s': round(B * world / (inv_ms * 1e-3), 1),
        'grad_bucket_bytes': trainer.bucket.nbytes(),
        'roofline': roof,
        'whole_step': whole,
    }
    if cfg['datatype'] == 'image':
        out['bits_per_dim_plus_log2_255'] = round(out['bits_per_dim'] + math.log2(255.0), 5)
    if world == 1 and not args.skip_cpu:
        out['cpu_baseline'], (zc, lc) = cpu_baseline(cfg, state0, y_cpu, cpu_seconds)
        D = float(np.prod(cfg['dims']))
        # the SAME first train step from the SAME initial weights and batch on both sides (forward incl. the data-dependent
        # ActNorm initialisation); the full-size parity tests (tests/test_gpu_fullsize_parity.py) bound these differences by the
        # distance of the fp32 CPU path from float64 arithmetic
        out['parity'] = {'loss_gpu_step1': round(loss1, 6), 'loss_cpu_step1': round(lc, 6),
                         'abs_dloss_per_dim': float('%.3e' % (abs(loss1 - lc) / D)),
                         'max_abs_dz': float('%.3e' % float((z1 - zc).abs().max())),
                         'max_abs_z': float('%.3e' % float(zc.abs().max())),
                         'note': 'step 1 from identical initial weights and batch: GPU trainer vs the oracle on the host cores'}
    else:
        out['cpu_baseline'] = None
    del trainer, net
    torch.cuda.empty_cache()
    return out


def summary_of(out):
    """the LAST key of the line: one short row per measured workload -- [samples/s, ms per step, roofline fraction of the dominant kernel,
    whole-step fp32-MFMA fraction, CPU baseline samples/s, |dloss| per dim at step 1, max |dz| at step 1] -- so that a record which keeps only
    the end of the line still holds every workload's numbers (the full objects are under "also")."""
    def row(o):
        if not o:
            return None
        r, w, c, p = o.get('roofline') or {}, o.get('whole_step') or {}, o.get('cpu_baseline') or {}, o.get('parity') or {}
        return [o['value'], o['ms_per_step'], r.get('frac'), w.get('mfma_frac'), c.get('value'), p.get('abs_dloss_per_dim'), p.get('max_abs_dz')]
    rows = {out['config']['name']: row(out)}
    for k, o in (out.get('also') or {}).items():
        rows[k] = row(o)
    return {'columns': ['samples_per_s', 'ms_per_step', 'roofline_frac', 'whole_step_mfma_frac', 'cpu_samples_per_s', 'dloss_per_dim', 'max_abs_dz'],
            'rows': rows}


LINE_CAP = 4000             # bytes of the ONE stdout line: the round-5 line (23 150 B) was not parsed by the driver, the round-4 one (15 189 B) was;
                            # the driver keeps an 8 081-character tail of stdout, so the whole line has to fit inside that with room to spare
HEAD_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
             'data', 'config', 'roofline', 'cpu_baseline', 'parity', 'whole_step', 'loss_nats', 'bits_per_dim', 'summary', 'detail')
ROOF_KEYS = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'flop_per_launch', 'bytes_per_launch', 'us_per_launch',
             'launches_timed', 'workgroups', 'traffic_source')
CPU_KEYS = ('value', 'unit', 'cores', 'kind', 'sample', 'ms_per_step', 'host_cores')
PARITY_KEYS = ('loss_gpu_step1', 'loss_cpu_step1', 'abs_dloss_per_dim', 'max_abs_dz', 'max_abs_z')
WHOLE_KEYS = ('flop_per_step', 'mfma_tflops', 'mfma_frac', 'hbm_bytes_per_step', 'hbm_gbs', 'hbm_frac')
CONFIG_KEYS = ('workload', 'name', 'per_gpu_batch', 'global_batch', 'parallelism', 'hipgraph', 'deterministic', 'dp_one_graph', 'collective')


def _clip(v, n):
    """strings of the headline are phrases, not paragraphs (the paragraphs are in the detail file)"""
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 3].rstrip() + '...'


def _pick(obj, keys, n=100):
    return None if obj is None else {k: _clip(obj[k], n) for k in keys if k in obj}


def headline_of(out, detail_path=None):
    """the ONE stdout line: the primary workload's contract keys + `roofline`, `cpu_baseline`, `parity`, `whole_step` cut to their numbers
    and a phrase each, + one `summary` row per other workload measured in the same run.  The complete objects (every note, every `also`
    workload) are the DETAIL, written to a side file and to stderr -- never to stdout.  Raises if the line would pass LINE_CAP."""
    head = {k: out[k] for k in HEAD_KEYS if k in out}
    head['metric'] = _clip(head['metric'], 90)
    head['data'] = _clip(head.get('data'), 70)
    head['config'] = _pick(out.get('config'), CONFIG_KEYS, 110)
    head['roofline'] = _pick(out.get('roofline'), ROOF_KEYS, 120)
    head['cpu_baseline'] = _pick(out.get('cpu_baseline'), CPU_KEYS, 130)
    head['parity'] = _pick(out.get('parity'), PARITY_KEYS)
    head['whole_step'] = _pick(out.get('whole_step'), WHOLE_KEYS)
    head['summary'] = summary_of(out)
    if detail_path:
        head['detail'] = detail_path
    line = json.dumps(head, separators=(', ', ': '))
    if len(line.encode()) >= LINE_CAP:
        raise AssertionError('bench.py: the stdout line is %d bytes (cap %d): trim headline_of, do not grow the line' % (len(line.encode()), LINE_CAP))
    json.loads(line)
    return line


def write_detail(out):
    """the full objects of every workload of this run: gpurun_out/bench_detail.json (merged back from the GPU box) and, when that
    directory cannot be made, next to bench.py; returns the path relative to the repo root (None if nothing could be written)"""
    text = json.dumps(out, indent=1)
    for rel in (os.path.join('gpurun_out', 'bench_detail.json'), 'bench_detail.json'):
        path = os.path.join(ROOT, rel)
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, 'w') as f:
                f.write(text + '\n')
            return rel
        except OSError:
            continue
    return None


def collective_info(world):
    """what the gradient exchange ran on: the process group's backend and size as torch.distributed reports them (N = 1: none)"""
    if not torch.distributed.is_initialized():
        return {'ranks': 1, 'backend': None}
    info = {'ranks': torch.distributed.get_world_size(), 'backend': torch.distributed.get_backend()}
    try:
        info['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        info['rccl_version'] = None
    return info


def self_spawn(args):
    """`python bench.py --gpus N` outside torchrun: re-exec under torch.distributed.run with N ranks on this node."""
    import socket
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < args.gpus:
        raise SystemExit('bench.py --gpus %d: this node exposes %d GPU(s); the bench measures the MI355X path only '
                         '(no CPU / gloo stand-in is timed)' % (args.gpus, n_dev))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # the host driver only supports dmabuf IPC (RCCL needs it)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execve(sys.executable, cmd, env)


def _claim_stdout():
    """stdout carries exactly ONE line, the JSON: RCCL prints its start-up banner (version, host name, library path) through C stdio on
    stdout, and when stdout is a pipe or file that buffer is flushed at process exit -- AFTER the JSON line python printed (measured on
    the GPU box with a one-rank group: five banner lines behind the line).  Everything written to fd 1 from here on goes to stderr; the
    line itself is written to the saved descriptor at the very end."""
    sys.stdout.flush()
    fd = os.dup(1)
    os.dup2(2, 1)
    return fd


def _emit_line(fd, line):
    sys.stdout.flush()
    sys.stderr.flush()
    try:
        ctypes.CDLL(None).fflush(None)                       # every C stdio buffer out (to stderr) before the line: it is the last output
    except Exception:
        pass
    os.write(fd, (line + '\n').encode())


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_spawn(args)                                     # does not return
    json_fd = _claim_stdout()
    pkg = importlib.import_module(PKG)
    nfdist = importlib.import_module(PKG + '.dist')
    rank, world, local_rank = nfdist.init_from_env()
    if args.gpus != world:
        raise SystemExit('--gpus %d but the launcher started %d rank(s)' % (args.gpus, world))
    assert torch.cuda.is_available(), 'bench.py measures the MI355X path; no GPU visible'
    torch.cuda.set_device(local_rank)                        # BEFORE the library is loaded / armed: one device per process
    pkg._native.load()
    dev = torch.device('cuda', local_rank)
    # BASELINE.json quotes the metric on RealNVP moons-2D (c1) and Glow CIFAR-10 (c4): the default run measures both -- the
    # line's top level is c4 (the larger one), c1 rides along as a second object of the same shape under "also"
    primary = args.config or 'c4'
    out = run_workload(primary, args, pkg, rank, world, dev, args.steps, args.warmup, args.cpu_seconds)
    if args.config is None and args.batch is None and args.layers is None:
        # (200 steps for the configs whose step is 0.7 - 2.5 ms: a 50-step region is 35 - 125 ms, where one host or clock hiccup moves the
        #  mean by several per cent -- wall-clock means of 1.36 .. 1.51 ms were seen for C1 next to an event median of 1.31)
        also = run_workload('c1', args, pkg, rank, world, dev, max(args.steps, 200), args.warmup, min(args.cpu_seconds, 6.0))
        # config 4's literal batch (512) on ONE GPU: informational -- the 4 x 4 level runs the persistent chain, the 8 x 8 and 16 x 16
        # levels exceed its co-residency limit at this batch and run one launch per layer (single-GPU runs only: the DP runs shard 512)
        b512 = None
        if world == 1 and args.scaling == 'weak':                # (--scaling strong at N = 1 IS this workload: the top level of the line)
            args.batch = 512
            b512 = run_workload('c4', args, pkg, rank, world, dev, min(args.steps, 8), args.warmup, min(args.cpu_seconds, 12.0))
            args.batch = None
        # the three other BASELINE.json configs (each < 3 ms per step): complete objects of the same shape, CPU leg bounded to a few seconds
        more = {}
        for extra_cfg in ('c2', 'c3', 'c5'):
            more[extra_cfg] = run_workload(extra_cfg, args, pkg, rank, world, dev, max(args.steps, 200), args.warmup, min(args.cpu_seconds, 5.0))
        # row (g): the two other models north_star names on CIFAR-shape batches, at the reference's default depth (layers = 32)
        for extra_cfg in ('rnvp_img', 'fpp_img'):
            more[extra_cfg] = run_workload(extra_cfg, args, pkg, rank, world, dev, min(args.steps, 10), args.warmup, min(args.cpu_seconds, 8.0))
        if rank == 0:
            out['also'] = {'c1': also}
            out['also'].update(more)
            if b512 is not None:
                out['also']['c4_b512'] = b512
    if world > 1 or torch.distributed.is_initialized():   # (initialised at one rank: NF_DP_FORCE_COLLECTIVE=1, the DP control flow on one GPU)
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        detail = write_detail(out)
        sys.stderr.write('bench detail (every workload, full notes):\n' + json.dumps(out) + '\n')
        _emit_line(json_fd, headline_of(out, detail))


if __name__ == '__main__':
    main()
