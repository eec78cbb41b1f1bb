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
    'c1': dict(cls='RealNVP', kind='realnvp', dims=(2, ), datatype='2d', layers=32, mixtures=None, data='moons', batch=256,
               desc='RealNVP moons-2D K=32 batch 256'),
    'c2': dict(cls='Glow', kind='glow', dims=(2, ), datatype='2d', layers=32, mixtures=None, data='moons', batch=4096,
               desc='Glow moons-2D K=32 batch 4096 per GPU'),
    'c3': dict(cls='Flowpp', kind='flowpp', dims=(2, ), datatype='2d', layers=32, mixtures=8, data='circles', batch=65536,
               desc='Flow++ circles-2D K=32 mixtures=8 batch 65536 per GPU'),
    'c4': dict(cls='Glow', kind='glow', dims=(3, 32, 32), datatype='image', layers=32, mixtures=None, data='cifar',
               batch=64, desc='Glow CIFAR-shape (3,32,32) L=3 K=32 batch 64 per GPU (512 over 8)'),
    'c5': dict(cls='MAF', kind='maf', dims=(2, ), datatype='2d', layers=10, mixtures=None, data='normals', batch=16384,
               desc='MAF normals-2D 10 AR layers batch 16384 per GPU (131072 over 8)'),
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


def pmc_traffic(kernel, B):
    """HBM bytes per launch of `kernel` at batch B from THIS round's committed rocprofv3 PMC passes (the newest profiles/rNN_pmc.json,
    written by tools/pmc_round.py from separate FETCH_SIZE / WRITE_SIZE runs of these very launches), or None.  PMC collection needs
    its own profiler runs, so it cannot happen inside the timed bench process; the kernel's DURATION in the same object is measured
    live, and tests/test_cabi.py checks that the file belongs to the current round's kernels."""
    import glob
    try:
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc.json')))
        with open(files[-1]) as f:
            sub = json.load(f).get(kernel, {})
        for key, e in sub.items():
            if key.split(':')[0] == str(B) and 'traffic_bytes' in e:
                return e['traffic_bytes']
    except (OSError, ValueError, IndexError):
        pass
    return None


def dominant_kernel_roofline(pkg, cfg, B, dev):
    """roofline of the kernel that dominates the timed region (rocprofv3 summaries under profiles/):
      c1 / c2 -> k_mlp_chain_bwd, the one-launch backward of the whole MLP conditioner (six 32-wide linears, five
                 BatchNorms; csrc/mlp_chain.hip) -- fp32 MFMA work, bound by the five grid-wide BatchNorm exchanges;
      c5      -> k_maf_step_bwd, the one-launch backward of a whole MAF flow step (flow BatchNorm, MADE pair, transform);
      c3      -> k_flowpp_cond_bwd, the one-launch backward of the gated-attention conditioner (fp32 MFMA);
      c4      -> k_convnet_chain_bwd, the data gradient of a whole image conditioner (+ coupling backward) in one persistent launch.
    achieved = algorithmic bytes or flops per launch / average launch duration at the workload's
    shape, measured live with HIP events around hipGraph replays of that launch on the launch stream."""
    N, NF = pkg._native, pkg.functional
    F = importlib.import_module(PKG + '.fused')
    dims = cfg['dims']
    g = torch.Generator(device='cpu').manual_seed(7)
    extra = {}
    MFMA_F32_TFLOPS = 157.3                                      # MI355X_MICROARCH.md: fp32-input MFMA = the fp32 vector peak
    if cfg['kind'] in ('glow', 'realnvp') and len(dims) == 1 and dims[0] in (2, 4) and B <= N.mlp_max_rows():
        D = dims[0]
        glow = cfg['kind'] == 'glow'
        k = pkg.AffineCoupling((D, )).to(dev).train()
        if glow:
            a, c = pkg.ActNorm((D, )).to(dev), pkg.InvertibleConv1x1(D).to(dev)
            a.initialized = True
            head = [a.log_scale, a.bias, c.P, c.L, c.U, c.L_mask, c.U_mask, c.sign_s, c.log_s, k.s_log_scale, k.s_bias]
            lh = [head[0], head[1], head[3], head[4], head[8], head[9], head[10]]
        else:
            bn = pkg.BatchNorm((D, ), affine=False).to(dev).train()
            head = [bn.log_gamma, bn.beta, bn.batch_mean, bn.batch_var, bn.running_mean, bn.running_var, k.s_log_scale, k.s_bias]
            lh = [head[6], head[7]]
        mts = F._mlp_tensors(k.net)
        lm = list(mts[:18]) + [t for j in range(5) for t in mts[18 + 5 * j:18 + 5 * j + 2]]
        dh, dm = [torch.zeros_like(t) for t in lh], [torch.zeros_like(t) for t in lm]
        htab, mtab = F._ptr_table([t.detach() for t in head]), F._ptr_table([t.detach() for t in mts])
        hg, mg = F._ptr_table(dh), F._ptr_table(dm)
        z = torch.randn(B, D, generator=g).to(dev)
        gy = torch.randn(B, D, generator=g).to(dev)
        y, ld, gz = torch.empty_like(z), torch.zeros(B, device=dev), torch.empty_like(z)
        nws = N.header_constant('NF_MLP_WS_FLOATS')
        save = torch.empty(N.header_constant('NF_REALNVP_SAVE_FLOATS'), device=dev)
        ws0 = torch.zeros(nws, device=dev)
        st = N.stream()
        if glow:
            N.call('nf_glow_step_vec_fwd', z.data_ptr(), y.data_ptr(), ld.data_ptr(), ctypes.addressof(htab), ctypes.addressof(mtab),
                   save.data_ptr(), ws0.data_ptr(), B, D, 0, 1, 1.0e-5, 0.1, 1.0e-5, st)
        else:
            N.call('nf_realnvp_step_vec_fwd', z.data_ptr(), y.data_ptr(), ld.data_ptr(), ctypes.addressof(htab),
                   ctypes.addressof(mtab), save.data_ptr(), ws0.data_ptr(), B, D, 0, 1.0e-5, 0.1, 1.0e-5, 0.1, 1.0e-5, st)
        S = int(cfg['layers'])
        per_step = glow and F._glow_steps_on(z)
        if (F._flow_on(z) or per_step) and S >= 2:
            # small batches train through the whole-flow launch (all S steps in one kernel per direction), larger Glow batches
            # through S single-step launches with the gradient folds deferred to one more launch: that is what the timed region
            # is made of, so that is what is measured here
            steps, sinks, keep = [], [], []
            for i in range(S):
                ki = pkg.AffineCoupling((D, ), odd=bool(i & 1)).to(dev).train()
                mi = F._mlp_tensors(ki.net)
                if glow:
                    ai, ci = pkg.ActNorm((D, )).to(dev), pkg.InvertibleConv1x1(D).to(dev)
                    hi = [ai.log_scale, ai.bias, ci.P, ci.L, ci.U, ci.L_mask, ci.U_mask, ci.sign_s, ci.log_s, ki.s_log_scale, ki.s_bias]
                    steps.append((int(i & 1), hi, mi))
                    learn = F._glow_step_learnables(hi, mi)
                else:
                    bi = pkg.BatchNorm((D, ), affine=False).to(dev).train()
                    hi = [bi.log_gamma, bi.beta, bi.batch_mean, bi.batch_var, bi.running_mean, bi.running_var, ki.s_log_scale, ki.s_bias]
                    steps.append((int(i & 1), 1.0e-5, 0.1, hi, mi))
                    learn = F._realnvp_step_learnables(hi, mi)
                sinks.append([torch.zeros_like(t) for t in learn])
                keep.append((ki, hi))
            table = (F._glow_flow_table if glow else F._realnvp_flow_table)(steps, sinks, D, dev)
            ys, gzs = torch.empty(S, B, D, device=dev), torch.empty(S, B, D, device=dev)
            saves = torch.empty(S, N.header_constant('NF_REALNVP_SAVE_FLOATS'), device=dev)
            ws1 = torch.zeros(S * nws, device=dev)
            slabs2 = F._glow_flow_slabs(dev)
            if glow:
                N.call('nf_glow_flow_vec_fwd', table.data_ptr(), S, z.data_ptr(), ys.data_ptr(), ld.data_ptr(), saves.data_ptr(),
                       ws1.data_ptr(), B, D, 1, 1.0e-5, 0.1, 1.0e-5, st)
            else:
                N.call('nf_realnvp_flow_vec_fwd', table.data_ptr(), S, z.data_ptr(), ys.data_ptr(), ld.data_ptr(), saves.data_ptr(),
                       ws1.data_ptr(), B, D, 1.0e-5, 0.1, 1.0e-5, st)
            wsf = torch.zeros(10, S * nws, device=dev)
            itf = [0]
            if per_step:
                host = ctypes.addressof(F._GLOW_FLOW_HOST[table.data_ptr()])
                slabs_all, rec = F._glow_steps_scratch(S, (B + 127) // 128, dev)
                N.call('nf_glow_flow_steps_fwd', host, S, z.data_ptr(), ys.data_ptr(), ld.data_ptr(), saves.data_ptr(), ws1.data_ptr(),
                       B, D, 1, 1.0e-5, 0.1, 1.0e-5, st)

            def fn_flow():
                ws = wsf[itf[0] % 10]
                itf[0] += 1
                if per_step:
                    N.call('nf_glow_flow_steps_bwd', host, table.data_ptr(), S, z.data_ptr(), ys.data_ptr(), gy.data_ptr(), None,
                           gzs.data_ptr(), saves.data_ptr(), 1, ws.data_ptr(), slabs_all.data_ptr(), rec.data_ptr(), B, D, 1, 1.0e-5,
                           1.0e-5, N.stream())
                elif glow:
                    N.call('nf_glow_flow_vec_bwd', table.data_ptr(), S, z.data_ptr(), ys.data_ptr(), gy.data_ptr(), None,
                           gzs.data_ptr(), saves.data_ptr(), 1, ws.data_ptr(), slabs2.data_ptr(), B, D, 1, 1.0e-5, 1.0e-5, N.stream())
                else:
                    N.call('nf_realnvp_flow_vec_bwd', table.data_ptr(), S, z.data_ptr(), ys.data_ptr(), gy.data_ptr(), None,
                           gzs.data_ptr(), saves.data_ptr(), 1, ws.data_ptr(), slabs2.data_ptr(), B, D, 1.0e-5, 1.0e-5, N.stream())
            us = graph_time_us(fn_flow, dev, per_graph=10, replays=1, reset=wsf.zero_)
            if per_step:
                # S launches of k_mlp_chain_bwd<1> + the one fold launch: the average over the S + 1 back-to-back launches
                us /= S + 1
                flop = 17 * 2 * 32 * 32 * B                      # 5 recomputed + 6 data-gradient + 6 weight-gradient 32x32 products
                tf = flop / (us * 1e-6) / 1e12
                return {'bound': 'mfma', 'kernel': 'k_mlp_chain_bwd<1> (whole Glow flow step, one launch, gradient fold deferred)',
                        'achieved': round(tf, 3), 'peak': MFMA_F32_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / MFMA_F32_TFLOPS, 5),
                        'traffic': pmc_traffic('k_mlp_chain_bwd', B), 'flop_per_launch': int(flop),
                        'bytes_per_launch': int(B * (3 * D + 1) * 4), 'us_per_launch': round(us, 3),
                        'note': 'average over the %d step launches + 1 k_glow_fold_all launch of a backward pass, back to back in a '
                                'hipGraph; neither MFMA- nor HBM-bound at this batch: five grid-wide BatchNorm exchanges (~1.6 us '
                                'each at 32 workgroups) and single-tile issue latency serialise the launch (DESIGN.md sections 2 '
                                'and 3.11; tools/probes/mlp_chain_prof.py)' % S}
            flop = S * 17 * 2 * 32 * 32 * B
            tf = flop / (us * 1e-6) / 1e12
            name = 'k_glow_flow_bwd<%d> (backward of all %d %s flow steps, one launch)' % (1 if glow else 2, S, 'Glow' if glow else 'RealNVP')
            return {'bound': 'mfma', 'kernel': name, 'achieved': round(tf, 3), 'peak': MFMA_F32_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(tf / MFMA_F32_TFLOPS, 5), 'traffic': pmc_traffic('k_glow_flow_bwd', B), 'flop_per_launch': int(flop),
                    'bytes_per_launch': int(S * B * (3 * D + 1) * 4), 'us_per_launch': round(us, 3),
                    'note': 'neither MFMA- nor HBM-bound at this batch: per step six grid-wide exchanges and single-tile issue latency '
                            'serialise the launch (DESIGN.md sections 2 and 3.11)'}
        slabs = F._mlp_slabs(dev)
        wss = torch.zeros(50, nws, device=dev)                  # a fresh zero workspace per launch inside the timing graph
        it = [0]

        def fn():
            ws = wss[it[0] % 50]
            it[0] += 1
            if glow:
                N.call('nf_glow_step_vec_bwd', z.data_ptr(), gy.data_ptr(), None, gz.data_ptr(), ctypes.addressof(htab),
                       ctypes.addressof(mtab), save.data_ptr(), ctypes.addressof(hg), ctypes.addressof(mg), 1, ws.data_ptr(),
                       slabs.data_ptr(), B, D, 0, 1, 1.0e-5, 1.0e-5, N.stream())
            else:
                N.call('nf_realnvp_step_vec_bwd', z.data_ptr(), gy.data_ptr(), None, gz.data_ptr(), ctypes.addressof(htab),
                       ctypes.addressof(mtab), save.data_ptr(), dh[0].data_ptr(), dh[1].data_ptr(), ctypes.addressof(mg), 1,
                       ws.data_ptr(), slabs.data_ptr(), B, D, 0, 1.0e-5, 1.0e-5, N.stream())
        us = graph_time_us(fn, dev, per_graph=50, replays=1, reset=wss.zero_)   # 50 launches = 50 distinct zero workspaces
        flop = 17 * 2 * 32 * 32 * B                              # 5 recomputed + 6 data-gradient + 6 weight-gradient 32x32 products
        tf = flop / (us * 1e-6) / 1e12
        name = 'k_mlp_chain_bwd<%d> (whole %s flow step, one launch)' % (1 if glow else 2, 'Glow' if glow else 'RealNVP')
        return {'bound': 'mfma', 'kernel': name, 'achieved': round(tf, 3), 'peak': MFMA_F32_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(tf / MFMA_F32_TFLOPS, 5), 'traffic': pmc_traffic('k_mlp_chain_bwd', B),
                'flop_per_launch': int(flop), 'bytes_per_launch': int(B * (3 * D + 1) * 4), 'us_per_launch': round(us, 3),
                'note': 'neither MFMA- nor HBM-bound at this batch: six grid-wide exchanges (five BatchNorm reductions + the fenced '
                        'barrier in front of the fold, ~1.6 us each at 32 workgroups) and single-tile issue latency serialise the '
                        'launch (DESIGN.md section 2; tools/probes/mlp_chain_prof.py)'}
    if cfg['kind'] == 'maf' and len(dims) == 1 and dims[0] <= 4 and B <= N.maf_max_rows():
        D = dims[0]
        bn = pkg.BatchNorm((D, ), affine=False).to(dev).train()
        ar = pkg.AutoregressiveTransfrom(D).to(dev).train()
        ms, mt = ar.net_s.draw_masks(dev), ar.net_t.draw_masks(dev)
        head = [bn.log_gamma, bn.beta, bn.batch_mean, bn.batch_var, bn.running_mean, bn.running_var, ar.perm, ar.s_log_scale,
                ar.s_bias]
        made = F._made_tensors(ar.net_s, ms) + F._made_tensors(ar.net_t, mt)
        z = torch.randn(B, D, generator=g).to(dev)
        gy = torch.randn(B, D, generator=g).to(dev)
        with torch.no_grad():
            F.maf_step_vec(z, torch.zeros(B, device=dev), bn, ar)
        save = torch.empty(N.header_constant('NF_MAF_SAVE_FLOATS'), device=dev)
        y, ld = torch.empty_like(z), torch.zeros(B, device=dev)
        htab, mtab = F._ptr_table([t.detach() for t in head]), F._ptr_table([t.detach() for t in made])
        nws = N.header_constant('NF_MAF_WS_FLOATS')
        ws0 = torch.zeros(nws, device=dev)
        N.call('nf_maf_step_fwd', z.data_ptr(), y.data_ptr(), ld.data_ptr(), ctypes.addressof(htab), ctypes.addressof(mtab),
               save.data_ptr(), ws0.data_ptr(), B, D, 1.0e-5, 0.1, 1.0e-5, N.stream())
        learn = F._made_learnables(made[:27]) + F._made_learnables(made[27:])
        dst = [torch.zeros_like(t) for t in learn]
        gtab = F._ptr_table(dst)
        ga, gc, gz = torch.zeros(1, device=dev), torch.zeros(1, device=dev), torch.empty_like(z)
        slabs = F._maf_slabs(dev)
        S = int(cfg['layers'])
        nwss = max(32, 2 * S)
        wss = torch.zeros(nwss, nws, device=dev)                # a fresh zero workspace per launch inside the timing graph
        it = [0]

        deferred = F.MAF_FLOW and S >= 2        # the train step runs the steps in deferred-fold mode (fused.maf_flow_vec)
        if deferred:
            slabs_all, rec = F._maf_steps_scratch(S, (B + 127) // 128, dev)
            nsl = ((B + 127) // 128) * N.header_constant('NF_MAF_SLAB_WG_FLOATS')
            nrec = ((B + 127) // 128) * N.header_constant('NF_MAF_HEAD_REC_WG')
            pm, pg = F._ptr_table([t.detach() for t in made] * S), F._ptr_table(dst * S)
            pa, pc = F._ptr_table([ga] * S), F._ptr_table([gc] * S)

        def fn():
            ws = wss[it[0] % nwss]
            i = it[0] % S
            it[0] += 1
            if deferred:
                N.call('nf_maf_step_bwd_partial', z.data_ptr(), gy.data_ptr(), None, gz.data_ptr(), ctypes.addressof(htab),
                       ctypes.addressof(mtab), save.data_ptr(), ctypes.addressof(gtab), ws.data_ptr(),
                       slabs_all.data_ptr() + 4 * i * nsl, rec.data_ptr() + 4 * i * nrec, B, D, N.stream())
                if i == S - 1:
                    N.call('nf_maf_fold_all', ctypes.addressof(pm), ctypes.addressof(pg), ctypes.addressof(pa), ctypes.addressof(pc), S,
                           slabs_all.data_ptr(), rec.data_ptr(), (B + 127) // 128, D, N.stream())
                return
            N.call('nf_maf_step_bwd', z.data_ptr(), gy.data_ptr(), None, gz.data_ptr(), ctypes.addressof(htab),
                   ctypes.addressof(mtab), save.data_ptr(), ctypes.addressof(gtab), ga.data_ptr(), gc.data_ptr(), ws.data_ptr(),
                   slabs.data_ptr(), B, D, N.stream())
        if deferred:                                             # 2 backward passes: 2 S step launches + 2 fold launches
            us = graph_time_us(fn, dev, per_graph=2 * S, replays=1, reset=wss.zero_) * (2 * S) / (2 * S + 2)
        else:
            us = graph_time_us(fn, dev, per_graph=25, replays=1, reset=wss.zero_)
        mac = 2 * 3 * (32 * D + 1024 + 1024 + 32 * D)           # two nets x (recompute + data + weight gradients)
        flop = 2 * mac * B
        tf = flop / (us * 1e-6) / 1e12
        name = 'k_maf_step_bwd (whole MAF flow step, one launch%s)' % (
            '; gradient fold deferred: average over the %d step launches + 1 k_maf_fold_all launch of a backward pass' % S if deferred
            else '')
        return {'bound': 'mfma', 'kernel': name, 'achieved': round(tf, 3),
                'peak': MFMA_F32_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / MFMA_F32_TFLOPS, 5), 'traffic': pmc_traffic('k_maf_step_bwd', B),
                'flop_per_launch': int(flop), 'bytes_per_launch': int(B * (3 * D + 1) * 4), 'us_per_launch': round(us, 3),
                'note': 'neither MFMA- nor HBM-bound at this batch: four grid-wide BatchNorm exchanges over 128 workgroups '
                        'serialise the launch (DESIGN.md sections 2 and 3.13)'}
    if cfg['kind'] == 'flowpp' and len(dims) == 1:
        K = cfg['mixtures']
        layer = pkg.MixLogAttnCoupling(dims, n_mixtures=K).to(dev)
        ts, F_ = F._flowpp_tensors(layer.net)
        I0 = ts[0].shape[1]
        O = ts[13].shape[0]
        x = torch.randn(B, I0, generator=g).to(dev)
        gout = torch.randn(B, O, generator=g).to(dev)
        gx = torch.empty_like(x)
        dst = [torch.zeros_like(t) for t in ts]
        d = [t.data_ptr() for t in dst]
        d[7] += 4 * 2 * F_ * 32
        d[8] += 4 * 2 * F_
        wsb = F.flowpp_bwd_workspace(dev)
        args = F._flowpp_fwd_args(ts, F_)

        fused_step = dims[0] == 2 and K <= 8 and F.FLOWPP_FUSED_BWD
        if fused_step:   # what the train step launches: the coupling's backward rides inside the conditioner's
            zz = torch.randn(B, 2, generator=g).to(dev)
            prm = torch.empty(B, O, device=dev)
            yy, ldd = torch.empty_like(zz), torch.zeros(B, device=dev)
            aa, cc = torch.full((1, ), 0.5, device=dev), torch.zeros(1, device=dev)
            gaa, gcc = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
            gyy, gld = torch.randn(B, 2, generator=g).to(dev), torch.full((B, ), -1.0 / B, device=dev)
            gzz = torch.empty_like(zz)
            N.call('nf_flowpp_vec_step_fwd', zz.data_ptr(), *args, aa.data_ptr(), cc.data_ptr(), None, None, prm.data_ptr(),
                   yy.data_ptr(), ldd.data_ptr(), K, 1.0e-5, 0, B, N.stream())

            def fn():
                N.call('nf_flowpp_vec_step_bwd', gyy.data_ptr(), gld.data_ptr(), zz.data_ptr(), prm.data_ptr(), *args, aa.data_ptr(),
                       cc.data_ptr(), None, None, gzz.data_ptr(), *d, gaa.data_ptr(), gcc.data_ptr(), None, None, wsb.data_ptr(), K,
                       1.0e-5, 0, B, 0, N.stream())
        else:
            def fn():
                N.call('nf_flowpp_cond_bwd', x.data_ptr(), *args, gout.data_ptr(), gx.data_ptr(), *d, wsb.data_ptr(), I0, 1, I0, 1, 0, B,
                       I0, O, N.stream())
        us = graph_time_us(fn, dev, per_graph=20, replays=5) * 1.0
        mac = (2048 + 1024 + 2048) + 2 * (O * 32 + 2048 + 1024 + 2048) + 32 * I0      # recompute + data + weight gradients
        flop = 2 * mac * B
        tf = flop / (us * 1e-6) / 1e12
        kname = ('k_flowpp_cond_bwd<2, true> + k_flowpp_cond_finalize (gated-attention conditioner + mixture coupling, backward)'
                 if fused_step else 'k_flowpp_cond_bwd + k_flowpp_cond_finalize (gated-attention conditioner)')
        return {'bound': 'mfma', 'kernel': kname,
                'achieved': round(tf, 3), 'peak': MFMA_F32_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / MFMA_F32_TFLOPS, 5),
                'traffic': pmc_traffic('k_flowpp_cond_bwd', B), 'flop_per_launch': int(flop), 'bytes_per_launch': int(B * (2 * I0 + O) * 4),
                'us_per_launch': round(us, 3),
                'note': 'fp32-input MFMA (exact fp32, 1/16 of the bf16 rate); the rest is transcendental VALU work and the '
                        'LDS transposes of the weight-gradient operands (DESIGN.md section 3)'}
    if len(dims) == 3 and cfg['kind'] in ('glow', 'realnvp'):
        # image flows: 72 % of the train step is the conditioner's persistent chain kernels (profiles/rNN_c4_kernel_stats.csv); the
        # largest single-shape share is the data-gradient chain of the 16 x 16 conditioners (64 of the 161 per step; with the coupling's
        # backward fused in, exactly as the step launches it): 6 -> 32 -> ... -> 32 -> 12 channels on (B, ., 16, 16)
        FC = importlib.import_module(PKG + '.fused_conv')
        Hh, Ww = dims[1] // 2, dims[2] // 2
        I0, O = 2 * dims[0], 4 * dims[0]
        if FC.CONV_CHAIN_ON and FC.CONV_CHAIN_BWD_ON and N.load().nf_convnet_chain_usable(B, I0, O, Hh, Ww):
            R = FC.R
            T = lambda *sh: torch.randn(*sh, generator=g).to(dev)          # noqa: E731
            w = [T(32, I0, 3, 3) * 0.1] + [T(32, 32, 3, 3) * 0.06 for _ in range(4)] + [T(O, 32, 1, 1) * 0.1]
            ones, zeros = torch.ones(32, device=dev), torch.zeros(32, device=dev)
            acts = [T(B, 32, Hh, Ww) for _ in range(5)]
            gn = [torch.empty(B, 32, Hh, Ww, device=dev) for _ in range(5)]
            stores = [torch.empty(B, 32, Hh, Ww, device=dev) for _ in range(2)]
            sums = torch.zeros(5, 2, R * 32, device=dev)
            z, gy = T(B, dims[0], dims[1], dims[2]), T(B, dims[0], dims[1], dims[2])
            out = torch.cat([torch.exp(0.1 * T(B, O // 2, Hh, Ww)), torch.tanh(T(B, O // 2, Hh, Ww))], 1).contiguous()
            gld = torch.full((B, ), -1.0, device=dev)
            gz, gout = torch.empty_like(z), torch.empty_like(out)
            a, c, gac = torch.full((1, ), 0.5, device=dev), torch.zeros(1, device=dev), torch.zeros(2, device=dev)
            nws = int(N.load().nf_convnet_chain_ws_floats(B, I0, O, Hh, Ww))
            split = nws > N.header_constant('NF_CONVNET_WS_FLOATS')     # a sample over two workgroups (halo rows handed over per layer)
            per = 20
            wsf = torch.zeros(per + 3, nws, device=dev)
            it = [0]
            d = FC.ConvNetBwdDesc()
            for i in range(6):
                d.w[i] = w[i].data_ptr()
            for j in range(5):
                d.gamma[j], d.beta[j], d.save_mean[j], d.save_invstd[j] = ones.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), ones.data_ptr()
                d.acts[j], d.gn[j] = acts[j].data_ptr(), gn[j].data_ptr()
                d.sum_g[j], d.sum_gx[j] = sums[j, 0].data_ptr(), sums[j, 1].data_ptr()
            d.g_store[0], d.g_store[1] = stores[0].data_ptr(), stores[1].data_ptr()
            d.cp_g_y, d.cp_g_ld, d.cp_z, d.cp_out = gy.data_ptr(), gld.data_ptr(), z.data_ptr(), out.data_ptr()
            d.cp_a, d.cp_c, d.cp_g_z, d.cp_g_out = a.data_ptr(), c.data_ptr(), gz.data_ptr(), gout.data_ptr()
            d.cp_g_a, d.cp_g_c = gac.data_ptr(), gac.data_ptr() + 4
            d.cp_mode, d.cp_odd, d.cp_C = N.SPLIT_CHECKER, 0, dims[0]

            def fn():                      # every launch of a graph gets its own zeroed exchange slots (generation stamps)
                d.ws_zero = wsf[it[0] % (per + 3)].data_ptr()
                it[0] += 1
                N.call('nf_convnet_chain_bwd', ctypes.addressof(d), B, I0, O, Hh, Ww, 1, N.stream())

            def reset():
                wsf.zero_()
                it[0] = 0
            us = graph_time_us(fn, dev, per_graph=per, replays=1, reset=reset)
            M = B * Hh * Ww
            flop = 2 * M * (4 * 9 * 32 * 32 + 9 * 32 * I0 + 32 * O)          # five transposed convolutions, data gradient only
            tf = flop / (us * 1e-6) / 1e12
            bytes_alg = 4 * (M * 32 * (5 + 5 + 2) + 2 * M * O + 3 * z.numel())
            wgs = int((M + (127 if split else 255)) // (128 if split else 256))
            return {'bound': 'mfma', 'kernel': 'k_convnet_chain_bwd<%s> (data gradient of the whole ConvNet conditioner + coupling backward in '
                                               'one persistent launch: %d -> 32 x 5 -> %d channels, %d x %d)' % ('4, 4, true, true, 18, 181' if split and Ww == 16 else ('4, 4, true, true, 0, 0' if split else '8, 2, false, true, 0, 0'), I0, O, Hh, Ww),
                    'achieved': round(tf, 3), 'peak': MFMA_F32_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / MFMA_F32_TFLOPS, 5),
                    'traffic': pmc_traffic('k_convnet_chain_bwd', B), 'flop_per_launch': int(flop), 'bytes_per_launch': int(bytes_alg),
                    'us_per_launch': round(us, 3), 'workgroups': wgs,
                    'note': '%s: %d of 256 compute units hold the whole launch (peak reachable by it: %d / 256 of the chip), five dependent '
                            'convolutions inside five grid-wide BatchNorm exchanges (DESIGN.md section 3.16)'
                            % ('two 1024-thread workgroups per sample, boundary rows handed over per layer' if split
                               else 'one 1024-thread workgroup per sample', wgs, wgs)}
    if cfg['kind'] in ('maf', 'glow', 'realnvp') and len(dims) == 1:              # multi-launch linear + BatchNorm chain
        nets = 2 if cfg['kind'] == 'maf' else 1
        T = lambda *sh: torch.randn(*sh, generator=g).to(dev)          # noqa: E731
        descs, keep = [], []
        slabs = F.bwd_slabs(B)
        for _ in range(nets):
            x, gn_src, out, gn_out = T(B, 32), T(B, 32), T(B, 32), torch.empty(B, 32, device=dev)
            Wt, gam, bet = T(32, 32) * 0.2, torch.rand(32, generator=g).to(dev) + 0.5, T(32) * 0.1
            ws = torch.zeros(64, 32, device=dev)
            ws[24] += 1.0
            wg = None if cfg['kind'] == 'maf' else torch.rand(32, generator=g).to(dev) + 0.5
            mask = torch.ones(32, 32, device=dev) if cfg['kind'] == 'maf' else None
            gweff = torch.empty(slabs * 1024, device=dev)
            keep += [x, gn_src, out, gn_out, Wt, gam, bet, ws, wg, mask, gweff]
            descs.append(F._desc(F.LinearBwdDesc, in_=x, weight=Wt, weight_g=wg, mask=mask, bn_gamma=gam, bn_beta=bet,
                                 bn_save_mean=ws[16], bn_save_invstd=ws[24], gn_src=gn_src, out=out, cbn_gamma=gam,
                                 cbn_save_mean=ws[16], cbn_save_invstd=ws[24], cbn_sum_g=ws[32], cbn_sum_gx=ws[40],
                                 g_bias=ws[48], g_weff=gweff, gn_out=gn_out, sum_g=ws[0], sum_gx=ws[8]))

        def fn():
            F._launch_bwd(descs, B, 32, 32)
        name = 'k_linear_bn_bwd (32x32 hidden layer, %d net%s)' % (nets, 's' if nets > 1 else '')
        nbytes = nets * B * 32 * 4 * 4                           # in, gn_src, out read; gn_out written
        extra = {'flop_per_launch': int(nets * 2 * 2 * B * 32 * 32), 'mfma_peak_tflops': 157.3}
    elif cfg['kind'] == 'flowpp':
        K = cfg['mixtures']
        z = torch.randn((B, ) + dims, generator=g).to(dev)
        params = (torch.randn(B, 2 + 3 * K, generator=g) * 0.5).to(dev)
        a, c = torch.full((1, ), 0.5, device=dev), torch.zeros(1, device=dev)
        ld, y = torch.zeros(B, device=dev), torch.empty_like(z)

        def fn():
            N.call('nf_mixlog_coupling_fwd', z.data_ptr(), params.data_ptr(), a.data_ptr(), c.data_ptr(), y.data_ptr(),
                   ld.data_ptr(), K, 1.0e-5, N.SPLIT_1D, 0, B, dims[0], 1, 1, N.stream())
        name = 'k_mixlog_rows_fwd'
        nbytes = B * ((4 + 3 * K) * 4 + 8 + 8)                   # (4+3K)*4 + pass-through r/w + ld rmw
    else:
        if len(dims) == 1:
            shape, mode, pshape = (B, dims[0]), N.SPLIT_1D, (B, dims[0])
        else:                                                    # first-resolution checkerboard step of the image stack
            shape, mode = (B, ) + dims, N.SPLIT_CHECKER
            pshape = (B, 4 * dims[0], dims[1] // 2, dims[2] // 2)
        z = torch.randn(shape, generator=g).to(dev)
        params = (torch.randn(pshape, generator=g) * 0.5).to(dev)
        a, c = torch.full((1, ), 0.5, device=dev), torch.zeros(1, device=dev)
        ld, y = torch.zeros(B, device=dev), torch.empty_like(z)
        n_half = z[0].numel() // 2
        C, H, W = (dims[0], 1, 1) if len(dims) == 1 else dims

        def fn():
            N.call('nf_affine_coupling_fwd', z.data_ptr(), params.data_ptr(), params.data_ptr() + 4 * n_half, 2 * n_half,
                   a.data_ptr(), c.data_ptr(), y.data_ptr(), ld.data_ptr(), mode, 0, 0, B, C, H, W, N.stream())
        name = 'k_affine_rows_fwd' if n_half <= 16 else 'k_affine_slab_fwd'
        nbytes = z.numel() * 12 + B * 8                          # 12 B / element of z + ld rmw
    us = graph_time_us(fn, dev)
    gbs = nbytes / (us * 1e-6) / 1e9
    traffic = pmc_traffic(name.split(' ')[0], B)   # HBM bytes per launch from this round's rocprofv3 PMC passes, same kernel + shape
    if traffic is not None and 'net' in name:
        traffic *= (2 if '2 nets' in name else 1)
    out = {'bound': 'hbm', 'kernel': name, 'achieved': round(gbs, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
           'frac': round(gbs / HBM_PEAK_GBS, 5), 'traffic': traffic, 'bytes_per_launch': int(nbytes),
           'us_per_launch': round(us, 3),
           'note': 'latency-bound at this batch size (DESIGN.md section 2); asymptotic rates in profiles/'}
    if extra:
        out['tflops'] = round(extra['flop_per_launch'] / (us * 1e-6) / 1e12, 3)
        out.update(extra)
    return out


def cpu_baseline(cfg, state, y_cpu, seconds):
    """the oracle (our CPU restatement of the reference, validated against it) timed on this box's host cores on the
    same workload and weights: full train step incl. Adam, bounded to ~`seconds` of CPU work.  Returns (object, first):
    ``first`` = (z, loss) of the FIRST step from the initial weights -- the same step the GPU trainer's first call performs
    (data-dependent ActNorm initialisation included), which is what the bench line's ``parity`` object compares."""
    from oracle import models as om
    from oracle import transforms as tf
    # the flow step is thousands of tiny ops: torch's intra-op pool stops scaling (and then collapses) beyond a few
    # threads, so the baseline uses 8 (what the reference was probed with) -- `cores` reports what ran
    cores = min(os.cpu_count() or 1, int(os.environ.get('NF_CPU_THREADS', '8')))
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
             'ms_per_step': round(1e3 * el / n, 2)}, first)


def run_workload(name, args, pkg, rank, world, dev, steps, warmup, cpu_seconds):
    """one workload: W warm-up steps, K timed steps (barrier + synchronize on both sides, max over ranks), then on rank 0
    the dominant kernel's roofline, the CPU baseline and the step-1 parity; returns the JSON object (None off rank 0)."""
    nfdist = importlib.import_module(PKG + '.dist')
    nftrain = importlib.import_module(PKG + '.train')
    nfdata = importlib.import_module(PKG + '.data')
    cfg = CONFIGS[name]
    B = args.batch or cfg['batch']

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
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        z, loss = trainer.train_on_batch(y)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss)

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
    out = {
        'metric': 'samples/sec (train step: forward flow + log-det + NLL + backward + Adam)',
        'value': round(B * world * steps / elapsed, 1),
        'unit': 'samples/s',
        'n_gpus': world,
        'steps': steps,
        'warmup': warmup,
        'ms_per_step': round(1e3 * elapsed / steps, 4),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic (seeded %s restatement, random-init weights)' % cfg['data'],
        'config': {'workload': cfg['desc'], 'name': name, 'per_gpu_batch': B, 'global_batch': B * world,
                   'parallelism': 'dp%d' % world, 'hipgraph': trainer._g_fb is not None},
        'loss_nats': round(loss_val, 5),
        'bits_per_dim': round(nftrain.bits_per_dim(loss_val, cfg['dims']), 5),
        'forward_samples_per_s': round(B * world / (fwd_ms * 1e-3), 1),
        'inverse_samples_per_s': round(B * world / (inv_ms * 1e-3), 1),
        'grad_bucket_bytes': trainer.bucket.nbytes(),
        'roofline': dominant_kernel_roofline(pkg, cfg, B, dev),
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


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_spawn(args)                                     # does not return
    pkg = importlib.import_module(PKG)
    pkg._native.load()
    nfdist = importlib.import_module(PKG + '.dist')
    rank, world, local_rank = nfdist.init_from_env()
    if args.gpus != world:
        raise SystemExit('--gpus %d but the launcher started %d rank(s)' % (args.gpus, world))
    assert torch.cuda.is_available(), 'bench.py measures the MI355X path; no GPU visible'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if os.environ.get('NF_MIOPEN_FIND', '0') == '1':
        torch.backends.cudnn.benchmark = True                # MIOpen find mode for the image conditioner's convolutions
    # BASELINE.json quotes the metric on RealNVP moons-2D (c1) and Glow CIFAR-10 (c4): the default run measures both -- the
    # line's top level is c4 (the larger one), c1 rides along as a second object of the same shape under "also"
    primary = args.config or 'c4'
    out = run_workload(primary, args, pkg, rank, world, dev, args.steps, args.warmup, args.cpu_seconds)
    if args.config is None and args.batch is None:
        also = run_workload('c1', args, pkg, rank, world, dev, max(args.steps, 50), args.warmup, min(args.cpu_seconds, 6.0))
        if rank == 0:
            out['also'] = {'c1': also}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
