"""GPU parity of the large-batch 3x3 convolution kernels (csrc/conv_bulk.hip: independent waves, three-way bf16 split on the matrix
pipe) behind nf_conv_bn_fwd / nf_conv_bn_bwd: against a float64 torch restatement of the launch's contract (flows/modules.py:416-438:
BatchNorm2d + ReLU on load, convolution, bias, residual, batch sums; the BatchNorm backward assembled on load, transposed convolution,
ReLU mask, batch sums), against the per-layer kernels of csrc/conv_bn.hip on the same operands, and through the whole conditioner at a
batch beyond the persistent chain."""
import copy
import importlib

import pytest
import torch
import torch.nn.functional as TF

from tests import _golden as G

pytestmark = pytest.mark.gpu
DEV = 'cuda'
R = 8            # NF_STAT_REPL


def _cfg(pkg, on, min_px=-1, nblk=-1):
    pkg._native.call('nf_conv_bulk_config', int(on), int(min_px), int(nblk))


@pytest.fixture()
def bulk(pkg):
    pkg._native.load()
    _cfg(pkg, 1, 0, 0)
    yield pkg
    _cfg(pkg, 1, 16385, 0)


def _replicas(n=32):
    return torch.zeros(R * 32, device=DEV)


# B, I, H, W, nblk (0 = automatic), packed weights, residual
FWD = [(72, 32, 16, 16, 0, True, True), (72, 32, 16, 16, 1, False, False), (72, 32, 16, 16, 2, True, True), (70, 6, 16, 16, 2, True, False),
       (300, 24, 8, 8, 0, True, False), (37, 24, 8, 8, 2, False, True), (37, 32, 8, 8, 1, True, True), (133, 32, 4, 4, 1, False, True),
       (9, 17, 32, 32, 1, False, False), (3, 32, 16, 16, 2, True, True), (1, 32, 8, 8, 2, True, False)]


@pytest.mark.parametrize('B,I,H,W,nblk,packed,res', FWD)
def test_bulk_forward_matches_float64_and_the_per_layer_kernel(bulk, B, I, H, W, nblk, packed, res):
    fc = importlib.import_module(bulk.__name__ + '.fused_conv')
    N = bulk._native
    torch.manual_seed(B + I)
    has_bn = I == 32
    x = torch.randn(B, I, H, W, device=DEV) * 1.5 + 0.3
    w = torch.randn(32, I, 3, 3, device=DEV) * 0.08
    bias = torch.randn(32, device=DEV) * 0.2
    resid = torch.randn(B, 32, H, W, device=DEV) if res else None
    gamma, beta = torch.rand(I, device=DEV) + 0.5, torch.randn(I, device=DEV) * 0.3
    center = torch.randn(I, device=DEV) * 0.1
    n = B * H * W
    # the input BatchNorm's batch sums, as the producing launch leaves them: shifted by `center`, spread over the replicas
    xs = (x - center.view(1, -1, 1, 1)).double()
    s1 = torch.zeros(R, 32, device=DEV)
    s2 = torch.zeros(R, 32, device=DEV)
    s1[0, :I] = xs.sum((0, 2, 3)).float() * 0.25
    s1[3, :I] = xs.sum((0, 2, 3)).float() * 0.75
    s2[5, :I] = (xs * xs).sum((0, 2, 3)).float()
    pack = None
    if packed:
        nimg = int(N.load().nf_conv_weight_pack_images(32, I, 3))
        pack = torch.empty(nimg * N.header_constant('NF_CONV_PACK_IMAGE_FLOATS'), device=DEV)
        import ctypes
        d = fc.ConvPackDesc(w.data_ptr(), pack.data_ptr(), 32, I, 3, 0)
        N.call('nf_conv_weight_pack', ctypes.addressof(d), 1, N.stream())

    def run(on):
        _cfg(bulk, on, 0, nblk)
        out = torch.empty(B, 32, H, W, device=DEV)
        st1, st2 = _replicas(), _replicas()
        rm, rv = torch.zeros(I, device=DEV), torch.ones(I, device=DEV)
        sm, si = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
        kw = dict(in_=x, weight=w, bias=bias, residual=resid, out=out, stat_sum=st1, stat_sqsum=st2, wpk=pack if on else None)
        if has_bn:
            kw.update(bn_gamma=gamma, bn_beta=beta, bn_sum=s1.view(-1), bn_sqsum=s2.view(-1), bn_center=center, bn_running_mean=rm,
                      bn_running_var=rv, bn_save_mean=sm, bn_save_invstd=si)
        fc._fwd((B, H, W), I, 32, 3, True, **kw)
        torch.cuda.synchronize()
        return out, st1.view(R, 32).sum(0), st2.view(R, 32).sum(0), rm, rv, sm, si

    got = run(1)
    old = run(0)
    # float64 restatement
    xd = x.double()
    if has_bn:
        mean = xd.mean((0, 2, 3))
        var = xd.var((0, 2, 3), unbiased=False)
        act = torch.relu((xd - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5) * gamma.double().view(1, -1, 1, 1)
                         + beta.double().view(1, -1, 1, 1))
    else:
        act = xd
    conv = TF.conv2d(act, w.double(), None, padding=1)
    dv = conv + (resid.double() if res else 0.0)
    want = dv + bias.double().view(1, -1, 1, 1)
    scale = max(1.0, float(want.abs().max()))
    # one-pass batch statistics of the per-layer contract (E[x^2] - E[x]^2 of shifted sums) limit how well the normalised input of
    # EITHER kernel agrees with the two-pass float64 restatement; the two kernels see the same constants and agree much better
    tol64 = 3e-5 * scale if has_bn else 1e-5 * scale
    assert float((got[0].double() - want).abs().max()) <= tol64, ('vs float64', float((got[0].double() - want).abs().max()), tol64)
    assert float((got[0] - old[0]).abs().max()) <= 1e-5 * scale, ('vs per-layer kernel', float((got[0] - old[0]).abs().max()))
    G.assert_close(got[1], dv.sum((0, 2, 3)).float(), 2e-5 * max(1.0, float(dv.abs().sum((0, 2, 3)).max())), what='stat_sum')
    G.assert_close(got[2], (dv * dv).sum((0, 2, 3)).float(), 2e-5 * float((dv * dv).sum((0, 2, 3)).max()), what='stat_sqsum')
    if has_bn:
        for a, b, what in zip(got[3:], old[3:], ('running_mean', 'running_var', 'save_mean', 'save_invstd')):
            G.assert_close(a, b, 1e-6, rtol=1e-6, what=what)
    assert n == B * H * W


BWD = [(72, 32, 16, 16, 0, True, True), (72, 32, 16, 16, 2, False, False), (72, 32, 16, 16, 1, True, True), (70, 6, 16, 16, 2, True, True),
       (300, 24, 8, 8, 0, True, True), (37, 32, 8, 8, 2, False, True), (37, 32, 8, 8, 1, True, False), (133, 32, 4, 4, 1, False, True),
       (3, 32, 16, 16, 2, True, True), (5, 32, 32, 32, 1, True, False)]


@pytest.mark.parametrize('B,I,H,W,nblk,packed,skip', BWD)
def test_bulk_backward_data_pass_matches_float64_and_the_per_layer_kernel(bulk, B, I, H, W, nblk, packed, skip):
    fc = importlib.import_module(bulk.__name__ + '.fused_conv')
    N = bulk._native
    torch.manual_seed(B * 3 + I)
    has_bn = I == 32
    n = B * H * W
    x = torch.randn(B, I, H, W, device=DEV)                  # forward input of the layer (pre-BatchNorm when has_bn)
    w = torch.randn(32, I, 3, 3, device=DEV) * 0.08
    out = torch.randn(B, 32, H, W, device=DEV) * 2.0 + 0.5     # forward output = the consumer BatchNorm's input
    gn_src = torch.randn(B, 32, H, W, device=DEV)
    g_skip = torch.randn(B, 32, H, W, device=DEV) if skip else None
    gamma, beta = torch.rand(I, device=DEV) + 0.5, torch.randn(I, device=DEV) * 0.3
    cgamma = torch.rand(32, device=DEV) + 0.5
    xd, od, gd = x.double(), out.double(), gn_src.double()
    mean = xd.mean((0, 2, 3)) if has_bn else None
    invstd = 1.0 / torch.sqrt(xd.var((0, 2, 3), unbiased=False) + 1e-5) if has_bn else None
    cmean = od.mean((0, 2, 3))
    cinvstd = 1.0 / torch.sqrt(od.var((0, 2, 3), unbiased=False) + 1e-5)
    xh = (od - cmean.view(1, -1, 1, 1)) * cinvstd.view(1, -1, 1, 1)
    sum_g_c = gd.sum((0, 2, 3))
    sum_gx_c = (gd * xh).sum((0, 2, 3))
    cs1, cs2 = torch.zeros(R, 32, device=DEV), torch.zeros(R, 32, device=DEV)
    cs1[1] = sum_g_c.float() * 0.5
    cs1[6] = sum_g_c.float() * 0.5
    cs2[2] = sum_gx_c.float()
    pack = None
    if packed:
        import ctypes
        nimg = int(N.load().nf_conv_weight_pack_images(32, I, 3))
        pack = torch.empty(nimg * N.header_constant('NF_CONV_PACK_IMAGE_FLOATS'), device=DEV)
        d = fc.ConvPackDesc(w.data_ptr(), pack.data_ptr(), 32, I, 3, 0)
        N.call('nf_conv_weight_pack', ctypes.addressof(d), 1, N.stream())

    def run(on):
        _cfg(bulk, on, 0, nblk)
        g_store = torch.zeros(B, 32, H, W, device=DEV)
        gn_out = torch.zeros(B, I, H, W, device=DEV)
        sg, sgx = _replicas(), _replicas()
        kw = dict(in_=x, weight=w, gn_src=gn_src, out=out, g_skip=g_skip, g_store=g_store, gn_out=gn_out, cbn_gamma=cgamma,
                  cbn_save_mean=cmean.float(), cbn_save_invstd=cinvstd.float(), cbn_sum_g=cs1.view(-1), cbn_sum_gx=cs2.view(-1),
                  wpk=pack if on else None)
        if has_bn:
            kw.update(bn_gamma=gamma, bn_beta=beta, bn_save_mean=mean.float(), bn_save_invstd=invstd.float(), sum_g=sg, sum_gx=sgx)
        fc._bwd((B, H, W), I, 32, 3, **kw)
        torch.cuda.synchronize()
        return g_store, gn_out, sg.view(R, 32).sum(0), sgx.view(R, 32).sum(0)

    got = run(1)
    old = run(0)
    Gd = cgamma.double().view(1, -1, 1, 1) * cinvstd.view(1, -1, 1, 1) * (gd - (sum_g_c / n).view(1, -1, 1, 1) - xh * (sum_gx_c / n).view(1, -1, 1, 1))
    if skip:
        Gd = Gd + g_skip.double()
    gin = TF.conv_transpose2d(Gd, w.double(), None, padding=1)
    if has_bn:
        xhat = (xd - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
        pre = xhat * gamma.double().view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)
        risky = pre.abs() < 1e-5                         # ReLU decisions within fp32 rounding of the kink: either side is correct
        gn = torch.where(pre > 0, gin, torch.zeros_like(gin))
    else:
        risky = torch.zeros_like(gin, dtype=torch.bool)
        gn = gin
    sG = max(1.0, float(Gd.abs().max()))
    assert float((got[0].double() - Gd).abs().max()) <= 1e-5 * sG, ('g_store vs float64', float((got[0].double() - Gd).abs().max()))
    assert float((got[0] - old[0]).abs().max()) <= 1e-5 * sG
    sg_ = max(1.0, float(gn.abs().max()))
    err = (got[1].double() - gn).abs()
    err[risky] = 0.0
    assert float(err.max()) <= 1e-5 * sg_, ('gn_out vs float64', float(err.max()), sg_)
    err = (got[1] - old[1]).abs()
    err[risky] = 0.0
    assert float(err.max()) <= 1e-5 * sg_, ('gn_out vs per-layer kernel', float(err.max()))
    if has_bn:
        nr = int(risky.sum())
        slack = nr * sg_ * 4.0
        want_sg, want_sgx = gn.sum((0, 2, 3)), (gn * xhat).sum((0, 2, 3))
        G.assert_close(got[2][:I], want_sg.float(), 2e-5 * max(1.0, float(gn.abs().sum((0, 2, 3)).max())) + slack, what='sum_g')
        G.assert_close(got[3][:I], want_sgx.float(), 2e-5 * max(1.0, float((gn * xhat).abs().sum((0, 2, 3)).max())) + slack, what='sum_gx')


@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('I,O,H,W,B', [(6, 12, 16, 16, 160), (24, 48, 8, 8, 520), (12, 24, 16, 8, 300)])
def test_convnet_beyond_the_chain_matches_modules(bulk, I, O, H, W, B, training):
    """the whole conditioner at a batch beyond the persistent chain (per-layer launches: the 3x3 layers on conv_bulk.hip), forward,
    input gradient, every parameter gradient and the running statistics against the module stack"""
    from tests.test_gpu_convnet import _close_but_for, _nets, _risky_samples
    fc = importlib.import_module(bulk.__name__ + '.fused_conv')
    _cfg(bulk, 1, 16385, 0)
    a, b = _nets(bulk, I, O)
    a.train(training)
    b.train(training)
    # No ReLU input near zero, by construction: with ~5 M ReLU decisions per pass EVERY random batch holds units inside rounding of their
    # kink (round 4 multiplied the parameter tolerance by 2500 for that).  Here every BatchNorm in front of a ReLU gets gamma in [0.5, 1]
    # and beta = +8 on the even channels (always on) / -8 on the odd ones (always off): both mask values and every code path are exercised,
    # no decision is ambiguous, and every gradient of the two paths must agree to rounding.
    x1 = torch.randn(B, I, H, W, device=DEV)
    with torch.no_grad():
        for net in (a, b):
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.copy_(torch.linspace(0.5, 1.0, m.weight.numel(), device=DEV))
                    m.bias.copy_(torch.where(torch.arange(m.bias.numel(), device=DEV) % 2 == 0, 8.0, -8.0))
        if not training:
            # evaluation mode normalises with the RUNNING statistics: give them this batch's (momentum 1 for one training-mode pass of
            # the module stack), so that the normalised values are O(1) here as well
            bns = [m for m in b.modules() if isinstance(m, torch.nn.BatchNorm2d)]
            for m in bns:
                m.momentum = 1.0
            b.train(True)
            b.forward_reference(x1)
            b.train(False)
            for m in bns:
                m.momentum = 0.1
    state = copy.deepcopy(b.state_dict())
    a.load_state_dict(state)
    risky = _risky_samples(b, x1)
    b.load_state_dict(state)
    assert not bool(risky.any()), 'a pre-activation within rounding of zero despite |beta| = 8'
    x1.requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    assert fc.convnet_usable(a, x1) and not fc._chain_usable(B, I, O, H, W)
    y1, y2 = a(x1), b(x2)
    G.assert_close(y1, y2, 2e-4, rtol=1e-4, what='output')
    wgt = torch.randn_like(y2)
    (y1 * wgt).sum().backward()
    (y2 * wgt).sum().backward()
    tol = lambda t: 2e-4 * max(1.0, float(t.abs().max()))
    _close_but_for(x1.grad, x2.grad, tol(x2.grad), risky, 'input grad')
    loose = 5.0
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for name, p in pb.items():
        assert pa[name].grad is not None, name
        pre_bn_bias = training and name.endswith('module.bias') and 'out_block' not in name
        t = 5e-3 + 3e-6 * B * H * W if pre_bn_bias else loose * tol(p.grad)
        G.assert_close(pa[name].grad, p.grad, t, what='grad ' + name)
    ba, bb = dict(a.named_buffers()), dict(b.named_buffers())
    for name in bb:
        G.assert_close(ba[name].float(), bb[name].float(), 1e-5, rtol=1e-5, what='buffer ' + name)


# B, I, H, W, layers in the launch, skip gradient, direct gradient, consumer BatchNorm (gn_src)
WGRAD = [(72, 32, 16, 16, 3, True, False, True), (70, 6, 16, 16, 1, False, False, True), (301, 24, 8, 8, 2, True, False, True),
         (37, 32, 8, 8, 16, False, True, False), (9, 17, 32, 32, 2, True, True, True), (3, 32, 16, 16, 1, True, False, True),
         (130, 12, 16, 8, 2, False, False, True), (1, 32, 8, 8, 1, True, False, True)]


@pytest.mark.parametrize('B,I,H,W,nl,skip,direct,src', WGRAD)
def test_bulk_weight_gradient_matches_float64_and_the_per_layer_kernel(bulk, B, I, H, W, nl, skip, direct, src):
    """nf_conv_bn_wgrad_multi on the pixel-contraction kernel (k_conv3_bulk_wgrad: bf16 planes [channel][row][pixel], dx taps by register
    shifts, walking / filling waves): weight-gradient slabs (summed by nf_slab_sum into (O, I, 3, 3)) and bias sums of every layer of the
    launch against float64 and against the per-layer kernel of conv_bn.hip on the same descriptors."""
    import ctypes
    fc = importlib.import_module(bulk.__name__ + '.fused_conv')
    N = bulk._native
    torch.manual_seed(B * 7 + I + nl)
    has_bn = I == 32
    n = B * H * W
    layers = []
    for _ in range(nl):
        x = torch.randn(B, I, H, W, device=DEV)
        out = torch.randn(B, 32, H, W, device=DEV) * 2.0 + 0.5
        gn_src = torch.randn(B, 32, H, W, device=DEV) if src else None
        g_skip = torch.randn(B, 32, H, W, device=DEV) if skip else None
        g_direct = torch.randn(B, 32, H, W, device=DEV) if direct else None
        gamma, beta = torch.rand(I, device=DEV) + 0.5, torch.randn(I, device=DEV) * 0.3
        cgamma = torch.rand(32, device=DEV) + 0.5
        xd, od = x.double(), out.double()
        mean = xd.mean((0, 2, 3))
        invstd = 1.0 / torch.sqrt(xd.var((0, 2, 3), unbiased=False) + 1e-5)
        cmean = od.mean((0, 2, 3))
        cinvstd = 1.0 / torch.sqrt(od.var((0, 2, 3), unbiased=False) + 1e-5)
        Gd = torch.zeros(B, 32, H, W, dtype=torch.float64, device=DEV)
        cs1, cs2 = torch.zeros(R, 32, device=DEV), torch.zeros(R, 32, device=DEV)
        if src:
            gd = gn_src.double()
            xh = (od - cmean.view(1, -1, 1, 1)) * cinvstd.view(1, -1, 1, 1)
            sg_c, sgx_c = gd.sum((0, 2, 3)), (gd * xh).sum((0, 2, 3))
            cs1[2] = sg_c.float() * 0.5
            cs1[7] = sg_c.float() * 0.5
            cs2[4] = sgx_c.float()
            Gd = cgamma.double().view(1, -1, 1, 1) * cinvstd.view(1, -1, 1, 1) * (gd - (sg_c / n).view(1, -1, 1, 1) - xh * (sgx_c / n).view(1, -1, 1, 1))
        if skip:
            Gd = Gd + g_skip.double()
        if direct:
            Gd = Gd + g_direct.double()
        act = torch.relu((xd - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1) * gamma.double().view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)) if has_bn else xd
        want_w = torch.nn.grad.conv2d_weight(act, (32, I, 3, 3), Gd, padding=1)
        want_b = Gd.sum((0, 2, 3))
        kw = dict(in_=x, weight=torch.zeros(32, I, 3, 3, device=DEV), out=out, g_skip=g_skip, g_direct=g_direct)
        if src:
            kw.update(gn_src=gn_src, cbn_gamma=cgamma, cbn_save_mean=cmean.float(), cbn_save_invstd=cinvstd.float(), cbn_sum_g=cs1.view(-1), cbn_sum_gx=cs2.view(-1))
        if has_bn:
            kw.update(bn_gamma=gamma, bn_beta=beta, bn_save_mean=mean.float(), bn_save_invstd=invstd.float())
        layers.append((kw, want_w, want_b, float((act.abs().sum() * 0 + 1)), act, Gd))

    def run(on):
        _cfg(bulk, on, 0, 0)
        slabs = int(N.load().nf_conv_wgrad_slabs(B, H, W, nl))
        arr = (fc.ConvBwdDesc * nl)()
        regions, gbs, keep = [], [], []
        for i, (kw, *_rest) in enumerate(layers):
            region = torch.full((slabs * 32 * I * 9, ), float('nan'), device=DEV)
            gb = torch.zeros(R * 256, device=DEV)
            d = fc._desc(fc.ConvBwdDesc, g_weff=region, g_bias=gb, **kw)
            ctypes.memmove(ctypes.addressof(arr) + i * ctypes.sizeof(fc.ConvBwdDesc), ctypes.addressof(d), ctypes.sizeof(fc.ConvBwdDesc))
            regions.append(region)
            gbs.append(gb)
            keep.append(d)
        N.call('nf_conv_bn_wgrad_multi', ctypes.addressof(arr), nl, B, I, 32, H, W, 3, N.stream())
        outs = []
        for region, gb in zip(regions, gbs):
            g_w = torch.empty(32, I, 3, 3, device=DEV)
            fc._slab_sum([(region, g_w, g_w.numel(), g_w.numel(), slabs, False, 9)])
            outs.append((g_w, gb.view(R, 256).sum(0)[:32].clone()))
        torch.cuda.synchronize()
        return outs

    got, old = run(1), run(0)
    for (kw, want_w, want_b, _, act, Gd), (gw, gb), (ow, ob) in zip(layers, got, old):
        # the bar of a sum of n products: 1e-5 of the largest sum of |G| |act| over a tap
        mag = float(torch.nn.grad.conv2d_weight(act.abs(), (32, I, 3, 3), Gd.abs(), padding=1).max())
        assert bool(torch.isfinite(gw).all())
        assert float((gw.double() - want_w).abs().max()) <= 1e-5 * max(1.0, mag), ('g_weff vs float64', float((gw.double() - want_w).abs().max()), mag)
        assert float((gw - ow).abs().max()) <= 1e-5 * max(1.0, mag), ('g_weff vs per-layer kernel', float((gw - ow).abs().max()), mag)
        magb = float(Gd.abs().sum((0, 2, 3)).max())
        assert float((gb.double() - want_b).abs().max()) <= 2e-5 * max(1.0, magb), ('g_bias vs float64', float((gb.double() - want_b).abs().max()), magb)
        assert float((gb - ob).abs().max()) <= 2e-5 * max(1.0, magb)


ONE = [(72, 12, 16, 16), (300, 48, 8, 8), (37, 24, 8, 8), (5, 12, 32, 32), (3, 64, 16, 16), (1, 7, 8, 8)]


@pytest.mark.parametrize('B,O,H,W', ONE)
def test_bulk_1x1_forward_and_data_gradient_match_float64_and_the_per_layer_kernel(bulk, B, O, H, W):
    """the conditioner's 1x1 output convolution at large batches (k_conv1_bulk_fwd / _bwd: BatchNorm + ReLU and the split in registers,
    operands straight from global memory): forward against float64 and the per-layer kernel; the data pass (G = g_direct) -- gn_out with
    its ReLU mask and the two batch sums -- likewise."""
    fc = importlib.import_module(bulk.__name__ + '.fused_conv')
    torch.manual_seed(B + O)
    I = 32
    n = B * H * W
    x = torch.randn(B, I, H, W, device=DEV) * 1.5 + 0.3
    w = torch.randn(O, I, 1, 1, device=DEV) * 0.2
    bias = torch.randn(O, device=DEV) * 0.2
    gamma, beta = torch.rand(I, device=DEV) + 0.5, torch.randn(I, device=DEV) * 0.3
    center = torch.randn(I, device=DEV) * 0.1
    xs = (x - center.view(1, -1, 1, 1)).double()
    s1, s2 = torch.zeros(R, 32, device=DEV), torch.zeros(R, 32, device=DEV)
    s1[2] = xs.sum((0, 2, 3)).float()
    s2[5] = (xs * xs).sum((0, 2, 3)).float()
    xd = x.double()
    mean = xd.mean((0, 2, 3))
    var = xd.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    pre = (xd - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1) * gamma.double().view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)
    act = torch.relu(pre)
    want = TF.conv2d(act, w.double(), bias.double())

    def fwd(on):
        _cfg(bulk, on, 0, 0)
        out = torch.full((B, O, H, W), float('nan'), device=DEV)
        rm, rv = torch.zeros(I, device=DEV), torch.ones(I, device=DEV)
        sm, si = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
        fc._fwd((B, H, W), I, O, 1, True, in_=x, weight=w, bias=bias, out=out, bn_gamma=gamma, bn_beta=beta, bn_sum=s1.view(-1),
                bn_sqsum=s2.view(-1), bn_center=center, bn_running_mean=rm, bn_running_var=rv, bn_save_mean=sm, bn_save_invstd=si)
        torch.cuda.synchronize()
        return out, rm, rv, sm, si

    got, old = fwd(1), fwd(0)
    scale = max(1.0, float(want.abs().max()))
    assert float((got[0].double() - want).abs().max()) <= 3e-5 * scale, float((got[0].double() - want).abs().max())
    assert float((got[0] - old[0]).abs().max()) <= 1e-5 * scale
    for a, b in zip(got[1:], old[1:]):
        G.assert_close(a, b, 1e-6, rtol=1e-6)

    g_direct = torch.randn(B, O, H, W, device=DEV)

    def bwd(on):
        _cfg(bulk, on, 0, 0)
        gn_out = torch.full((B, I, H, W), float('nan'), device=DEV)
        sg, sgx = _replicas(), _replicas()
        fc._bwd((B, H, W), I, O, 1, in_=x, weight=w, g_direct=g_direct, gn_out=gn_out, bn_gamma=gamma, bn_beta=beta, bn_save_mean=mean.float(),
                bn_save_invstd=invstd.float(), sum_g=sg, sum_gx=sgx)
        torch.cuda.synchronize()
        return gn_out, sg.view(R, 32).sum(0), sgx.view(R, 32).sum(0)

    gb, ob = bwd(1), bwd(0)
    gin = TF.conv_transpose2d(g_direct.double(), w.double())
    risky = pre.abs() < 1e-5
    gn = torch.where(pre > 0, gin, torch.zeros_like(gin))
    sg_ = max(1.0, float(gn.abs().max()))
    err = (gb[0].double() - gn).abs()
    err[risky] = 0.0
    assert float(err.max()) <= 1e-5 * sg_, float(err.max())
    err = (gb[0] - ob[0]).abs()
    err[risky] = 0.0
    assert float(err.max()) <= 1e-5 * sg_
    xhat = (xd - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    slack = int(risky.sum()) * sg_ * 4.0
    G.assert_close(gb[1], gn.sum((0, 2, 3)).float(), 2e-5 * max(1.0, float(gn.abs().sum((0, 2, 3)).max())) + slack, what='sum_g')
    G.assert_close(gb[2], (gn * xhat).sum((0, 2, 3)).float(), 2e-5 * max(1.0, float((gn * xhat).abs().sum((0, 2, 3)).max())) + slack, what='sum_gx')
    assert n == B * H * W
