"""
Tight oracle parity of GRADIENTS on the trainer's launch paths at the full BASELINE batch sizes: two-step slices of the
full-size models at several depths.

Why slices: a full-depth gradient cannot meet a max-norm bar against any other fp32 implementation (the loss is only
piecewise smooth; a single ReLU decision that falls on the other side of its kink is amplified ~1.3x per flow step on the
way back: tests/test_gpu_fullsize_parity.py measures and documents that).  A two-step slice has nothing to amplify with, so
here the bar is the strict one.

Per config: the full model is built (ActNorm initialised from the oracle's first forward, see below) under the FlowTrainer that
bench.py drives.  For each slice [a, b) of ``net.layers`` the oracle computes, in float64, the activations that reach layer a
from the real batch, runs the slice and the NLL of the slice's output, and differentiates: input gradient + every parameter
gradient of the slice.  The GPU runs the SAME slice of the SAME modules (``net.forward_slice`` -- identical Compose peepholes,
so identical kernels: whole-flow / per-step launches with deferred folds, fused Flow++ steps, fused heads + conv conditioners
with deferred weight gradients) inside ``FlowTrainer._run_step`` (flat gradient bucket, zero arena, deferred-work queues) on
the float64 activations cast to fp32.  Bars, max norm:
    z, ld, loss       1e-5 * scale + SLACK * |cpu32 - cpu64|          (cpu32: the same slice in float32 on the same input)
    input gradient    2e-5 * max|g| + SLACK * gap, row by row; at most FLIPS rows (samples) may miss it -- a ReLU pre-activation
                      within rounding of zero is masked differently by two correct implementations (measured: about one slice in
                      twenty at B = 4096)
    parameter grads   2e-5 * max|g| + SLACK * gap (+ 8 / B * max|g| per flipped sample found above: its footprint)
Needs a real MI355X.
"""
import importlib
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import trajectory as traj

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = 1.0e-5
SLACK = 4.0
FLIPS = 2

# the CIFAR Glow's layer list: [Logit] + 32 x 3 @ (3,32,32) checker + [Squeeze] + 32 x 3 @ (12,16,16) channel + 32 x 3 checker +
# [Squeeze] + 32 x 3 @ (48,8,8) channel + 33 x 3 checker + 2 x [Unsqueeze]
CONFIGS = [
    # name, oracle kind, class, dims, datatype, layers, mixtures, per-GPU batch, data, slices [a, b)
    ('c1_realnvp_moons', 'realnvp', 'RealNVP', (2, ), '2d', 32, None, 256, 'moons', [(0, 4), (28, 32), (60, 64)]),
    ('c2_glow_moons', 'glow', 'Glow', (2, ), '2d', 32, None, 4096, 'moons', [(0, 6), (42, 48), (90, 96)]),
    ('c3_flowpp_circles', 'flowpp', 'Flowpp', (2, ), '2d', 32, 8, 65536, 'circles', [(0, 4), (60, 64)]),
    ('c4_glow_cifar', 'glow', 'Glow', (3, 32, 32), 'image', 32, None, 64, 'cifar',
     [(0, 7), (98, 104), (194, 200), (291, 297), (480, 488)]),
    ('c5_maf_normals', 'maf', 'MAF', (2, ), '2d', 10, None, 16384, 'normals', [(0, 4), (16, 20)]),
]


def _maxerr(a, b):
    return float((a.detach().double().cpu().reshape(-1) - b.detach().double().reshape(-1)).abs().max())


@pytest.mark.parametrize('cfg', CONFIGS, ids=[c[0] for c in CONFIGS])
def test_slice_gradients_match_oracle_at_full_batch(pkg, cfg):
    name, kind, cls, dims, datatype, layers, mix, B, data, slices = cfg
    nfdata = importlib.import_module(pkg.__name__ + '.data')
    nftrain = importlib.import_module(pkg.__name__ + '.train')
    torch.manual_seed(0)
    np.random.seed(0)
    net = getattr(pkg, cls)(dims, datatype, NS(layers=layers, mixtures=mix))
    y = nfdata.sample(data, B, 1234)
    if data == 'cifar':
        y = y.reshape((B, ) + dims)
    # Deterministic weights: the data-dependent ActNorm initialisation is taken from the oracle's first forward on the CPU and
    # loaded, instead of running a GPU training step (whose atomics make the last bits of the weights -- and with them WHICH
    # ReLU pre-activations sit within rounding of zero -- vary from run to run).  The forward kernels of the slices are
    # deterministic, so the outcome below is a fixed function of the seeds: with these there is no kink event on any slice.
    # (A flipped ReLU in a 4 x 4 conditioner moves the BatchNorm backward's batch sums by 1 / 1024, i.e. EVERY row by ~1e-3 of
    # the largest gradient: measured while this test still trained on the GPU first, one run in three.)
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    _, ora = traj.run(kind, dims, datatype, layers, sd0, y, 1, mixtures=mix)
    init = {k: v.detach().clone() for k, v in ora.sd.items() if k.endswith(('log_scale', 'bias')) and k.count('.') == 3}
    sd0.update({k: v for k, v in init.items() if k in sd0})
    net.load_state_dict(sd0)
    for m in net.modules():
        if hasattr(m, 'initialized'):
            m.initialized = True
    net = net.to(DEV)
    trainer = nftrain.FlowTrainer(net, graph=False)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    params = dict(net.named_parameters())
    problems = []
    for a, b in slices:
        r64 = traj.run_slice(kind, dims, datatype, layers, sd, a, b, None, None, mixtures=mix, dtype=torch.float64, y=y)
        r32 = traj.run_slice(kind, dims, datatype, layers, sd, a, b, r64['z_in'], r64['ld_in'], mixtures=mix, dtype=torch.float32)
        z_in = r64['z_in'].float().to(DEV).requires_grad_(True)
        ld_in = r64['ld_in'].float().to(DEV)
        net.train()

        def forward_loss():
            z, ld = net.forward_slice(z_in, ld_in.clone(), a, b)
            forward_loss.ld = ld.detach()
            return z, nftrain.nll_loss(z, ld)

        z, loss = trainer._run_step(z_in.device, forward_loss)
        torch.cuda.synchronize()
        tag = '%s[%d:%d]' % (name, a, b)
        D = float(np.prod(dims))
        for what, got, scale in (('z', z, None), ('ld', forward_loss.ld, None),
                                 ('loss', loss, max(1.0, abs(float(r64['loss'])) / D))):
            s = scale if scale is not None else max(1.0, float(r64[what].abs().max()))
            err, gap = _maxerr(got, r32[what]), _maxerr(r32[what], r64[what])
            ulp = 4.0 * 1.2e-7 * float(r64[what].abs().max())     # the value itself is an fp32 number (loss ~ 1.5e4 for CIFAR)
            if err > TOL * s + SLACK * gap + ulp:
                problems.append((tag, what, err, gap))
        # input gradient, sample by sample
        g_gpu, g32, g64 = (t.detach().double().cpu().reshape(B, -1) for t in (z_in.grad, r32['g_in'], r64['g_in']))
        s = max(float(g64.abs().max()), 1e-30)
        gap = float((g32 - g64).abs().max())
        row_err = (g_gpu - g32).abs().max(dim=1).values
        bar = 2.0 * TOL * s + SLACK * gap
        # a KINK EVENT: some sample has a ReLU pre-activation within rounding of zero and the GPU masks it the other way -- that
        # row's gradient is off by O(1) (>= 1e-3 of the largest entry here), and through the batch sums of the BatchNorm backward
        # every other row and every parameter gradient of the slice moves a little (1 / (samples x pixels) of it).  With the
        # deterministic weights above the events are a fixed property of (seed, slice); one exists at c5[16:20] (one sample in
        # 16384, identical on all three GPU launch paths: tools/probes/slice_dbg2.py).  Such a slice keeps the bound on the
        # number of hit rows and gets a 2e-2 bar for the batch-coupled remainder; every other slice gets the strict bar.
        flipped = int((row_err > max(100.0 * bar, 1.0e-3 * s)).sum())
        kink = flipped > 0
        if flipped > FLIPS:
            problems.append((tag, 'input gradient: %d rows hit by a kink' % flipped, float(row_err.max()) / s, gap / s))
        rest = row_err[row_err <= max(100.0 * bar, 1.0e-3 * s)]
        if rest.numel() and float(rest.max()) > (2.0e-2 * s if kink else bar):
            problems.append((tag, 'input gradient: rows outside the bar', float(rest.max()) / s, gap / s))
        n = 0
        G_all = max(float(w_.abs().max()) for w_ in r64['grads'].values())      # the largest gradient entry of the slice
        for k, want in r64['grads'].items():
            p = params[k]
            assert p.grad is not None, k
            sk = max(1.0e-6, float(want.abs().max()))            # the tensor's OWN largest entry (no floor at 1.0)
            err, gk = _maxerr(p.grad, r32['grads'][k]), _maxerr(r32['grads'][k], want)
            if float(want.abs().max()) < 1.0e-9 * G_all:
                # an ANALYTICALLY ZERO gradient (the bias of a linear layer in front of a BatchNorm: float64 leaves 1e-17): what any fp32
                # path returns is the rounding of B cancelling terms, ~1e-7 of their size -- bounded against the slice's gradient scale
                if err > 2.0 * TOL * G_all:
                    problems.append((tag, k + ' (analytically zero)', err / G_all, gk / G_all))
            elif err > (2.0e-2 * sk if kink else 2.0 * TOL * sk + SLACK * gk):
                problems.append((tag, k, err / sk, gk / sk))
            n += 1
        assert n >= 4, (tag, n)
        print('%-26s z %.2e  ld %.2e  g_in %.2e (gap %.2e, %d flipped rows)  %d parameter gradients' % (
            tag, _maxerr(z, r32['z']), _maxerr(forward_loss.ld, r32['ld']), float(row_err.max()) / s, gap / s, flipped, n) + (' KINK EVENT' if kink else ''))
    assert pkg._native.persistent_timeouts() == 0
    assert not problems, '%d quantities outside their bar: %s' % (len(problems), problems[:8])
