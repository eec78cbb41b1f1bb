"""GPU parity of the fused image conditioner (csrc/conv_bn.hip) against the module-by-module ConvNet
(flows/modules.py:416-438: WeightNorm convolutions, BatchNorm2d in train / eval mode, residual blocks)."""
import copy
import importlib

import pytest
import torch

from tests import _golden as G

pytestmark = pytest.mark.gpu
DEV = 'cuda'

# (in_channels, out_channels, H, W): the five conditioner shapes of Glow / RealNVP / Flow++ on CIFAR (checkerboard halves
# are (C, H, W/2), channel halves (C/2, H, W)); the levels of the 28 x 28 (MNIST-shape) pyramid, which the kernels run in
# power-of-two storage with a dead border (nf_conv_desc.valid_h / valid_w); and one map nothing fits (module path)
SHAPES = [(3, 6, 32, 16), (6, 12, 16, 16), (12, 24, 16, 8), (24, 48, 8, 8), (48, 96, 8, 4), (96, 192, 4, 4),
          (1, 2, 28, 14), (2, 4, 14, 14), (4, 8, 14, 7), (8, 16, 7, 7), (16, 32, 7, 7), (1, 2, 6, 200)]


def _risky_samples(net, x):
    """samples in which some ReLU input of the module path lies within rounding of zero: the two paths may then take
    different sides of the kink there (the forward value is unaffected, the gradient of that sample's neighbourhood and the
    parameter gradients are not).  With ~5 M ReLU decisions per pass about one in five runs has such a unit."""
    pre = []
    hooks = [m.register_forward_hook(lambda mod, i, o: pre.append(o.detach().clone()))
             for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    with torch.no_grad():
        net.forward_reference(x.detach())
    for h in hooks:
        h.remove()
    risky = torch.zeros(x.shape[0], dtype=torch.bool, device=x.device)
    for p in pre:
        risky |= (p.abs() < 2e-5 * max(1.0, float(p.abs().max()))).flatten(1).any(1)
    return risky


def _close_but_for(a, b, atol, skip, what):
    keep = ~skip
    G.assert_close(a[keep], b[keep], atol, rtol=1e-4, what=what)


def _nets(pkg, I, O):
    cond = importlib.import_module(pkg.__name__ + '.conditioners')
    torch.manual_seed(I * 100 + O)
    a = cond.ConvNet(I, O).to(DEV)
    with torch.no_grad():
        for m in a.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
    b = copy.deepcopy(a)
    a.fused = True                  # the HIP conditioner (the default: NF_FUSED_CONV=0 selects the module path); b = the module stack
    b.fused = False
    return a, b


@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('B', [64, 5])
@pytest.mark.parametrize('I,O,H,W', SHAPES)
def test_convnet_fused_vs_modules(pkg, I, O, H, W, B, training):
    fc = importlib.import_module(pkg.__name__ + '.fused_conv')
    a, b = _nets(pkg, I, O)
    a.train(training)
    b.train(training)
    x1 = torch.randn(B, I, H, W, device=DEV).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    if (H, W) == (6, 200):
        assert not fc.convnet_usable(a, x1)            # wider than a tile even in power-of-two storage: the module path serves it
        G.assert_close(a(x1), b(x2), 1e-6, what='fallback')
        return
    assert fc.convnet_usable(a, x1)
    state = copy.deepcopy(b.state_dict())
    risky = _risky_samples(b, x1)                      # (touches the running statistics in train mode: restored)
    b.load_state_dict(state)
    y1, y2 = a(x1), b(x2)
    G.assert_close(y1, y2, 2e-4, rtol=1e-4, what='output')
    w = torch.randn_like(y2)
    (y1 * w).sum().backward()
    (y2 * w).sum().backward()
    tol = lambda t: 2e-4 * max(1.0, float(t.abs().max()))
    _close_but_for(x1.grad, x2.grad, tol(x2.grad), risky, 'input grad')
    loose = 2500.0 if bool(risky.any()) else 5.0      # a flipped unit moves parameter gradients by its whole share
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for name, p in pb.items():
        assert pa[name].grad is not None, name
        # a bias in front of a training-mode BatchNorm has gradient exactly 0: both sides return cancellation noise
        pre_bn_bias = training and name.endswith('module.bias') and 'out_block' not in name
        t = 5e-3 + 3e-6 * B * H * W if pre_bn_bias else loose * tol(p.grad)
        G.assert_close(pa[name].grad, p.grad, t, what='grad ' + name)
    ba, bb = dict(a.named_buffers()), dict(b.named_buffers())
    for name in bb:
        G.assert_close(ba[name].float(), bb[name].float(), 1e-5, rtol=1e-5, what='buffer ' + name)


def test_glow_image_coupling_uses_the_fused_conditioner(pkg):
    """AffineCoupling on image data: conditioner through the fused kernels == module path, forward and backward."""
    torch.manual_seed(3)
    k1 = pkg.AffineCoupling((12, 16, 16), masking='channelwise').to(DEV)
    k2 = copy.deepcopy(k1)
    k1.net.fused = True
    k2.net.fused = False
    z1 = torch.randn(16, 12, 16, 16, device=DEV).requires_grad_(True)
    z2 = z1.detach().clone().requires_grad_(True)
    y1, l1 = k1(z1, torch.zeros(16, device=DEV))
    y2, l2 = k2(z2, torch.zeros(16, device=DEV))
    G.assert_close(y1, y2, 2e-4, rtol=1e-4, what='y')
    G.assert_close(l1, l2, 2e-3, rtol=1e-4, what='log-det')
    (y1.sum() + l1.sum()).backward()
    (y2.sum() + l2.sum()).backward()
    bad = ((z1.grad - z2.grad).abs() > 2e-4 * max(1.0, float(z2.grad.abs().max()))).flatten(1).any(1)
    assert int(bad.sum()) <= 1, 'grad z differs in %d samples' % int(bad.sum())      # one ReLU flip at most


@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('dims,masking,odd,B', [((12, 16, 16), 'channelwise', False, 64), ((12, 16, 16), 'channelwise', True, 5),
                                                ((3, 32, 32), 'checkerboard', False, 64), ((12, 16, 16), 'checkerboard', True, 64),
                                                ((48, 8, 8), 'channelwise', False, 64), ((48, 8, 8), 'checkerboard', True, 64),
                                                ((48, 8, 8), 'checkerboard', False, 3)])
def test_coupling_fused_into_the_conditioner_launch(pkg, dims, masking, odd, B, training):
    """AffineCoupling of an image model with the transform, merge and log-det inside the conditioner's chain launches (one launch
    per direction) == conditioner launch + coupling kernel + gather / scatter, forward, backward and inverse."""
    fc = importlib.import_module(pkg.__name__ + '.fused_conv')
    torch.manual_seed(11)
    k1 = pkg.AffineCoupling(dims, masking=masking, odd=odd).to(DEV)
    with torch.no_grad():
        for p in k1.parameters():
            p.add_(0.05 * torch.randn_like(p))
        k1.s_log_scale.fill_(0.7)
        k1.s_bias.fill_(-0.2)
    k2 = copy.deepcopy(k1)
    k1.net.fused = k2.net.fused = True
    k1.train(training)
    k2.train(training)
    z1 = torch.randn((B, ) + dims, device=DEV).requires_grad_(True)
    z2 = z1.detach().clone().requires_grad_(True)
    ld0 = torch.randn(B, device=DEV)
    assert fc.coupling_fusable(k1.net, z1, k1.mode)
    y1, l1 = k1(z1, ld0.clone())
    assert type(y1.grad_fn).__name__.startswith('_FusedConvCoupling'), type(y1.grad_fn).__name__
    old = fc.CONV_COUPLING_ON
    fc.CONV_COUPLING_ON = False
    try:
        y2, l2 = k2(z2, ld0.clone())
        assert not type(y2.grad_fn).__name__.startswith('_FusedConvCoupling')
        G.assert_close(y1, y2, 2e-5, rtol=1e-5, what='y')
        G.assert_close(l1, l2, 1e-4, rtol=1e-5, what='log-det')
        w, wl = torch.randn_like(y1), torch.randn_like(l1)
        ((y1 * w).sum() + (l1 * wl).sum()).backward()
        ((y2 * w).sum() + (l2 * wl).sum()).backward()
        s = max(1.0, float(z2.grad.abs().max()))
        bad = ((z1.grad - z2.grad).abs() > 2e-5 * s).flatten(1).any(1)
        assert int(bad.sum()) <= (1 if training else 0), 'grad z differs in %d samples' % int(bad.sum())   # one ReLU flip at most
        flip = bool(bad.any())
        for (n, p1), (_, p2) in zip(k1.named_parameters(), k2.named_parameters()):
            assert p1.grad is not None, n
            pre_bn_bias = training and n.endswith('module.bias') and 'out_block' not in n
            t = 5e-3 + 3e-6 * B * dims[1] * dims[2] if pre_bn_bias else (1e-2 if flip else 1e-4) * max(1.0, float(p2.grad.abs().max()))
            G.assert_close(p1.grad, p2.grad, t, what='grad ' + n)
        # inverse: the fused epilogue in its inverse form == the unfused inverse, and it undoes the forward
        k1.eval()
        k2.eval()
        with torch.no_grad():
            yy = torch.randn((B, ) + dims, device=DEV)
            x2, m2 = k2.backward(yy, ld0.clone())
            fc.CONV_COUPLING_ON = True
            x1, m1 = k1.backward(yy, ld0.clone())
            G.assert_close(x1, x2, 2e-5, rtol=1e-5, what='inverse')
            G.assert_close(m1, m2, 1e-4, rtol=1e-5, what='inverse log-det')
            back, lb = k1(x1, m1.clone())
            G.assert_close(back, yy, 1e-4, rtol=1e-4, what='round trip')
            G.assert_close(lb, ld0, 1e-3, rtol=1e-4, what='round trip log-det')
    finally:
        fc.CONV_COUPLING_ON = old


@pytest.mark.parametrize('C,H,W,mode_name,odd,B', [(12, 16, 16, 'channelwise', False, 64), (12, 16, 16, 'checkerboard', True, 7),
                                                   (48, 8, 8, 'channelwise', True, 64), (48, 8, 8, 'checkerboard', False, 64),
                                                   (10, 4, 8, 'checkerboard', False, 3), (64, 4, 4, 'channelwise', False, 9),
                                                   # large batches: the forward on 64-pixel blocks (k_glow_head_w_fwd4)
                                                   (48, 8, 8, 'channelwise', False, 1100), (48, 8, 8, 'checkerboard', True, 1030),
                                                   (12, 16, 16, 'checkerboard', False, 260), (12, 16, 16, 'channelwise', True, 257)])
def test_glow_head_w_matches_the_three_layers(pkg, C, H, W, mode_name, odd, B):
    """ActNorm + invertible 1x1 (assembled weight) + conditioner-input gather in one MFMA launch per direction == the three layers'
    own launches: outputs, log-det, input gradient, ActNorm gradients and the gradient handed to the batched PLU backward."""
    NF = importlib.import_module(pkg.__name__ + '.functional')
    Nn = importlib.import_module(pkg.__name__ + '._native')
    mode = Nn.SPLIT_CHANNEL if mode_name == 'channelwise' else Nn.SPLIT_CHECKER
    torch.manual_seed(5)
    x1 = torch.randn(B, C, H, W, device=DEV).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    ls1 = (0.3 * torch.randn(1, C, 1, 1, device=DEV)).requires_grad_(True)
    b1 = torch.randn(1, C, 1, 1, device=DEV).requires_grad_(True)
    Wm1 = (torch.linalg.qr(torch.randn(C, C, device=DEV))[0] + 0.05 * torch.randn(C, C, device=DEV)).contiguous().requires_grad_(True)
    ls2, b2, Wm2 = (t.detach().clone().requires_grad_(True) for t in (ls1, b1, Wm1))
    log_s = torch.randn(C, device=DEV) * 0.1
    ld0 = torch.randn(B, device=DEV)
    assert NF.glow_head_w_usable(x1, mode)
    h1, z1c1, l1 = NF.glow_head_w(x1, ld0.clone(), ls1, b1, Wm1, log_s, NF.PluHolder(1), 0, mode, odd)
    hold = NF.PluHolder(1)
    a2, l2 = NF.chan_affine(Nn.OP_ACTNORM, x2, ld0.clone(), ls2, b2)
    h2, l2 = NF.invconv_apply_w(a2, l2, Wm2, log_s, hold, 0)
    z1c2 = NF.half_gather(h2, 1, mode, odd)
    G.assert_close(h1, h2, 2e-5, rtol=1e-5, what='h')
    assert torch.equal(z1c1, NF.half_gather(h1.detach(), 1, mode, odd)), 'the gathered half is not the half of h'
    G.assert_close(l1, l2, 1e-4, rtol=1e-5, what='log-det')
    wh, wz, wl = torch.randn_like(h1), torch.randn_like(z1c1), torch.randn_like(l1)
    ((h1 * wh).sum() + (z1c1 * wz).sum() + (l1 * wl).sum()).backward()
    ((h2 * wh).sum() + (z1c2 * wz).sum() + (l2 * wl).sum()).backward()
    for what, g1, g2 in (('x', x1.grad, x2.grad), ('log_scale', ls1.grad, ls2.grad), ('bias', b1.grad, b2.grad), ('W', Wm1.grad, Wm2.grad)):
        G.assert_close(g1, g2, 2e-5 * max(1.0, float(g2.abs().max())), rtol=1e-5, what='grad ' + what)


@pytest.mark.parametrize('C,H,W,B,heads', [(12, 16, 16, 64, 33), (48, 8, 8, 64, 5), (48, 8, 8, 512, 3), (12, 16, 16, 512, 2), (10, 4, 8, 3, 1),
                                          (64, 4, 4, 9, 4), (33, 8, 8, 1030, 2)])
def test_glow_head_backward_in_two_parts(pkg, C, H, W, B, heads):
    """nf_glow_head_w_bwd_data (g_x alone) + nf_glow_head_w_bwd_params_multi (g_W, g_log_scale, g_bias of many heads per launch, +=) against
    nf_glow_head_w_bwd: g_x BITWISE, the parameter gradients to the rounding of their atomics' order (bitwise in deterministic mode)."""
    import ctypes
    NF = importlib.import_module(pkg.__name__ + '.functional')
    Nn = pkg._native
    torch.manual_seed(11)
    step = Nn.header_constant('NF_GLOW_HEAD_MULTI_MAX')
    for det in (False, True):
        was = Nn.deterministic()
        Nn.deterministic(det)
        try:
            cases = []
            for i in range(heads):
                g_h, x = torch.randn(B, C, H, W, device=DEV), torch.randn(B, C, H, W, device=DEV)
                g_ld = torch.randn(B, device=DEV)
                ls, bs = 0.3 * torch.randn(C, device=DEV), torch.randn(C, device=DEV)
                Wm = (torch.linalg.qr(torch.randn(C, C, device=DEV))[0] + 0.05 * torch.randn(C, C, device=DEV)).contiguous()
                ref = [torch.empty_like(x), torch.full((C, ), 0.5, device=DEV), torch.full((C, ), -0.25, device=DEV), torch.full((C, C), 2.0, device=DEV)]
                Nn.call('nf_glow_head_w_bwd', Nn.ptr(g_h), Nn.ptr(g_ld), Nn.ptr(x), Nn.ptr(ls), Nn.ptr(bs), Nn.ptr(Wm), Nn.ptr(ref[0]), Nn.ptr(ref[1]),
                        Nn.ptr(ref[2]), Nn.ptr(ref[3]), B, C, H, W, Nn.stream())
                two = [torch.empty_like(x), torch.full((C, ), 0.5, device=DEV), torch.full((C, ), -0.25, device=DEV), torch.full((C, C), 2.0, device=DEV)]
                Nn.call('nf_glow_head_w_bwd_data', Nn.ptr(g_h), Nn.ptr(ls), Nn.ptr(Wm), Nn.ptr(two[0]), B, C, H, W, Nn.stream())
                cases.append((g_h, g_ld, x, ls, bs, Wm, ref, two))
            for k0 in range(0, heads, step):
                chunk = cases[k0:k0 + step]
                arr = (NF.GlowHeadParamsDesc * len(chunk))()
                for i, (g_h, g_ld, x, ls, bs, Wm, ref, two) in enumerate(chunk):
                    d = arr[i]
                    d.g_h, d.g_ld, d.x, d.act_log_scale, d.act_bias, d.W = (t.data_ptr() for t in (g_h, g_ld, x, ls, bs, Wm))
                    d.g_log_scale, d.g_bias, d.g_W = two[1].data_ptr(), two[2].data_ptr(), two[3].data_ptr()
                Nn.call('nf_glow_head_w_bwd_params_multi', ctypes.addressof(arr), len(chunk), B, C, H, W, Nn.stream())
            torch.cuda.synchronize()
            for i, c in enumerate(cases):
                ref, two = c[6], c[7]
                assert torch.equal(ref[0], two[0]), ('g_x', i, float((ref[0] - two[0]).abs().max()))
                for what, a, b in (('g_log_scale', ref[1], two[1]), ('g_bias', ref[2], two[2]), ('g_W', ref[3], two[3])):
                    if det and heads == 1:
                        assert torch.equal(a, b), (what, i, float((a - b).abs().max()))
                    else:
                        # (in the mode both forms add workgroup by workgroup, but a launch of many heads walks fewer tiles per workgroup)
                        G.assert_close(b, a, 2e-5 * max(1.0, float(a.abs().max())), rtol=1e-5, what='%s of head %d' % (what, i))
            assert Nn.deterministic_timeouts() == 0
        finally:
            Nn.deterministic(was)


@pytest.mark.parametrize('C,H,W,B,mode_name,heads,with_z1c', [(3, 32, 32, 64, 'checkerboard', 32, False), (4, 8, 8, 7, 'channelwise', 3, True),
                                                             (2, 4, 4, 1, 'channelwise', 1, False), (3, 32, 32, 512, 'checkerboard', 2, True)])
def test_small_glow_head_backward_in_two_parts(pkg, C, H, W, B, mode_name, heads, with_z1c):
    """the C <= 4 head: nf_glow_head_bwd_data + nf_glow_head_bwd_params_multi against nf_glow_head_bwd (g_z bitwise, the sums to rounding)"""
    import ctypes
    NF = importlib.import_module(pkg.__name__ + '.functional')
    Nn = pkg._native
    mode = Nn.SPLIT_CHANNEL if mode_name == 'channelwise' else Nn.SPLIT_CHECKER
    torch.manual_seed(12)
    cases = []
    for i in range(heads):
        odd = i & 1
        g_h, z = torch.randn(B, C, H, W, device=DEV), torch.randn(B, C, H, W, device=DEV)
        g_z1c = torch.randn(NF._half_shape(z, mode), device=DEV) if with_z1c else None
        g_ld = torch.randn(B, device=DEV)
        ls, bs = 0.3 * torch.randn(C, device=DEV), torch.randn(C, device=DEV)
        Wm = (torch.linalg.qr(torch.randn(C, C, device=DEV))[0] + 0.05 * torch.randn(C, C, device=DEV)).contiguous()
        mk = lambda: [torch.empty_like(z), torch.full((C, ), 0.5, device=DEV), torch.full((C, ), -0.25, device=DEV), torch.full((C, C), 2.0, device=DEV),
                      torch.full((1, ), 3.0, device=DEV)]
        ref, two = mk(), mk()
        Nn.call('nf_glow_head_bwd', Nn.ptr(g_h), Nn.ptr(g_z1c), Nn.ptr(g_ld), Nn.ptr(z), Nn.ptr(ls), Nn.ptr(bs), Nn.ptr(Wm), Nn.ptr(ref[0]), Nn.ptr(ref[1]),
                Nn.ptr(ref[2]), Nn.ptr(ref[3]), Nn.ptr(ref[4]), mode, odd, B, C, H, W, Nn.stream())
        Nn.call('nf_glow_head_bwd_data', Nn.ptr(g_h), Nn.ptr(g_z1c), Nn.ptr(ls), Nn.ptr(Wm), Nn.ptr(two[0]), mode, odd, B, C, H, W, Nn.stream())
        cases.append(((mode, B, C, H, W), g_h, g_z1c, g_ld, z, ls, bs, Wm, two[1], two[2], two[3], two[4], odd, ref, two))
    NF.launch_small_head_params([c[:13] for c in cases])
    torch.cuda.synchronize()
    for i, c in enumerate(cases):
        ref, two = c[13], c[14]
        assert torch.equal(ref[0], two[0]), ('g_z', i, float((ref[0] - two[0]).abs().max()))
        for what, a, b in zip(('g_log_scale', 'g_bias', 'g_W', 'sum_g_ld'), ref[1:], two[1:]):
            G.assert_close(b, a, 2e-5 * max(1.0, float(a.abs().max())), rtol=1e-5, what='%s of head %d' % (what, i))


@pytest.mark.parametrize('C,H,W,B,mode_name,odd', [(12, 16, 16, 5, 'channelwise', False), (12, 16, 16, 64, 'checkerboard', True), (48, 8, 8, 3, 'checkerboard', False),
                                                   (10, 4, 6, 3, 'channelwise', True), (3, 6, 6, 2, 'checkerboard', False)])
def test_half_scatter_add_is_scatter_plus_add(pkg, C, H, W, B, mode_name, odd):
    """nf_half_scatter_add (one pass) == nf_half_scatter + an add, bit for bit: the vectorised and the generic form"""
    NF = importlib.import_module(pkg.__name__ + '.functional')
    Nn = pkg._native
    mode = Nn.SPLIT_CHANNEL if mode_name == 'channelwise' else Nn.SPLIT_CHECKER
    torch.manual_seed(3)
    base = torch.randn(B, C, H, W, device=DEV)
    half = torch.randn(NF._half_shape(base, mode), device=DEV)
    full = torch.empty_like(base)
    Nn.call('nf_half_scatter', Nn.ptr(half), Nn.ptr(full), 1, mode, int(odd), B, C, H, W, Nn.stream())
    out = torch.full_like(base, 7.0)
    Nn.call('nf_half_scatter_add', Nn.ptr(half), Nn.ptr(base), Nn.ptr(out), 1, mode, int(odd), B, C, H, W, Nn.stream())
    assert torch.equal(out, base + full)


@pytest.mark.parametrize('n,n_slabs,acc', [(1, 384, True), (1, 64, False), (3, 1000, True), (1, 63, True), (5, 384, True), (9216, 16, True)])
def test_slab_sum_of_a_few_elements_over_many_slabs(pkg, n, n_slabs, acc):
    """nf_slab_sum: the wave-per-element form for scalar gradients left as per-workgroup partial sums (and the plain form around it)
    against a float64 sum"""
    fc = importlib.import_module(pkg.__name__ + '.fused_conv')
    torch.manual_seed(n + n_slabs)
    src = torch.randn(n_slabs, n, device=DEV)
    dst = torch.randn(n, device=DEV)
    want = (dst.double() if acc else 0.0) + src.double().sum(0)
    other_src, other_dst = torch.randn(16, 300, device=DEV), torch.zeros(300, device=DEV)       # a second job in the same launch
    fc._slab_sum_all([(src, dst, n, n, n_slabs, acc, 1), (other_src, other_dst, 300, 300, 16, False, 1)])
    torch.cuda.synchronize()
    assert float((dst.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max())) * (n_slabs ** 0.5)
    assert float((other_dst.double() - other_src.double().sum(0)).abs().max()) <= 1e-5


def test_cifar_glow_head_parameter_gradients_deferred_or_not(pkg, monkeypatch):
    """a trainer step of a (3, 32, 32) Glow with the heads' parameter gradients deferred to the batched launch (the default) and with every
    head's backward whole: z and the loss BITWISE (the forward is untouched), the flat gradient to the rounding of the atomics' order."""
    NF = importlib.import_module(pkg.__name__ + '.functional')
    nftrain = importlib.import_module(pkg.__name__ + '.train')
    from types import SimpleNamespace as NS
    outs = []
    y = torch.rand(16, 3, 32, 32, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    torch.manual_seed(2)
    net = pkg.Glow((3, 32, 32), 'image', NS(layers=3, mixtures=None)).to(DEV)
    tr = nftrain.FlowTrainer(net, graph=False)
    tr.train_on_batch(y)                                # data-dependent initialisation
    torch.cuda.synchronize()
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    was = pkg._native.deterministic()
    pkg._native.deterministic(True)                     # (the forward's log-det sums are ordered: z and the loss reproduce bit for bit)
    try:
        for on in (True, False):
            monkeypatch.setattr(NF, 'HEAD_PARAMS_DEFER', on)
            net.load_state_dict(sd)
            z, loss = tr._forward_backward(y)
            torch.cuda.synchronize()
            outs.append((z.detach().clone(), loss.detach().clone(), tr.bucket.flat.detach().clone()))
    finally:
        pkg._native.deterministic(was)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    d = (outs[0][2] - outs[1][2]).double()
    rel = float(d.norm() / outs[1][2].double().norm())
    assert rel <= 1e-5, rel
    assert float(outs[0][2].abs().max()) > 0
    assert pkg._native.persistent_timeouts() == 0


@pytest.mark.parametrize('B', [16, 5, 64])
def test_cifar_glow_head_data_gradient_in_the_chain_prologue(pkg, monkeypatch, B):
    """a trainer step of a (3, 32, 32) Glow with the data gradient of every second and later head of a level computed in the prologue of
    the previous coupling's backward chain launch (the default) and on its own kernel: same arithmetic in the same order, so in the
    ordered mode the flat gradient reproduces BIT FOR BIT; the stand-alone kernel must have run for the first head of a level only."""
    NF = importlib.import_module(pkg.__name__ + '.functional')
    nftrain = importlib.import_module(pkg.__name__ + '.train')
    Nn = pkg._native
    from types import SimpleNamespace as NS
    outs, counts, small = [], [], []
    y = torch.rand(B, 3, 32, 32, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    torch.manual_seed(2)
    net = pkg.Glow((3, 32, 32), 'image', NS(layers=3, mixtures=None)).to(DEV)
    tr = nftrain.FlowTrainer(net, graph=False)
    tr.train_on_batch(y)                                # data-dependent initialisation
    torch.cuda.synchronize()
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    was = Nn.deterministic()
    Nn.deterministic(True)
    real_call = Nn.call
    try:
        for on in (True, False):
            monkeypatch.setattr(NF, 'HEAD_BWD_IN_CHAIN', on)
            seen = []
            monkeypatch.setattr(Nn, 'call', lambda name, *a, _s=seen: (_s.append(name), real_call(name, *a))[1])
            net.load_state_dict(sd)
            z, loss = tr._forward_backward(y)
            torch.cuda.synchronize()
            monkeypatch.setattr(Nn, 'call', real_call)
            outs.append((z.detach().clone(), loss.detach().clone(), tr.bucket.flat.detach().clone()))
            counts.append(seen.count('nf_glow_head_w_bwd_data'))
            small.append(seen.count('nf_glow_head_bwd_data'))
            assert not NF.PENDING_HEAD_BWD
    finally:
        Nn.deterministic(was)
    assert counts[1] > 2 * counts[0] > 0, counts       # only the first head of a level (no fused coupling in front of it) keeps its launch
    assert small[1] == 3 and small[0] == 1, small      # ... the (3, 32, 32) level's thread-per-pixel heads as well (nf_cc_head_small_bwd)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[0][2]).all() and float(outs[0][2].abs().max()) > 0
    assert torch.equal(outs[0][2], outs[1][2]), float((outs[0][2] - outs[1][2]).abs().max())
    assert Nn.persistent_timeouts() == 0


@pytest.mark.parametrize('B', [16, 5])
def test_cifar_glow_heads_in_the_forward_chain_prologue(pkg, monkeypatch, B):
    """the forward of every head (the thread-per-pixel form on 3 channels, the MFMA form on 12 .. 48) in the prologue of its coupling's
    chain launch (the default) and on its own kernel: the same arithmetic in the same order -- z BITWISE in the ordered mode, the loss
    and the gradients to the rounding of the log-det sums (the prologue adds a head's term per sample in a different order)."""
    NF = importlib.import_module(pkg.__name__ + '.functional')
    layers_mod = importlib.import_module(pkg.__name__ + '.layers')
    nftrain = importlib.import_module(pkg.__name__ + '.train')
    Nn = pkg._native
    from types import SimpleNamespace as NS
    outs, counts = [], []
    y = torch.rand(B, 3, 32, 32, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    torch.manual_seed(3)
    net = pkg.Glow((3, 32, 32), 'image', NS(layers=3, mixtures=None)).to(DEV)
    tr = nftrain.FlowTrainer(net, graph=False)
    tr.train_on_batch(y)                                # data-dependent initialisation
    torch.cuda.synchronize()
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    was = Nn.deterministic()
    Nn.deterministic(True)
    real_call = Nn.call
    try:
        for on in (True, False):
            monkeypatch.setattr(layers_mod, 'HEAD_IN_CHAIN', on)
            seen = []
            monkeypatch.setattr(Nn, 'call', lambda name, *a, _s=seen: (_s.append(name), real_call(name, *a))[1])
            net.load_state_dict(sd)
            z, loss = tr._forward_backward(y)
            torch.cuda.synchronize()
            monkeypatch.setattr(Nn, 'call', real_call)
            outs.append((z.detach().clone(), loss.detach().clone(), tr.bucket.flat.detach().clone()))
            counts.append((seen.count('nf_glow_head_fwd'), seen.count('nf_glow_head_w_fwd')))
            assert not NF.PENDING_HEADS and not NF.PENDING_HEAD_BWD
    finally:
        Nn.deterministic(was)
    assert counts[0] == (0, 0) and counts[1][0] == 3 and counts[1][1] >= 6, counts
    assert torch.equal(outs[0][0], outs[1][0])
    assert abs(float(outs[0][1]) - float(outs[1][1])) <= 1e-6 * abs(float(outs[1][1]))
    d = (outs[0][2] - outs[1][2]).double()
    assert float(d.norm() / outs[1][2].double().norm()) <= 1e-5
    assert Nn.persistent_timeouts() == 0


@pytest.mark.parametrize('graph', [False, True])
def test_chain_launches_share_one_slot_buffer_per_step(pkg, monkeypatch, graph):
    """Inside a trainer step every chain launch takes its exchange slots (BatchNorm statistics, halo rows, log-det hand-over) from ONE
    buffer zeroed where the step begins, told apart by a per-launch generation tag; the round 2 .. 5 form gave every launch fresh
    zeros.  Three steps of a (3, 32, 32) Glow either way, eager and as a replayed hipGraph (the tags are constants of the capture, the
    memset is part of it): the same bits in the ordered mode, no exchange timed out."""
    fc = importlib.import_module(pkg.__name__ + '.fused_conv')
    nftrain = importlib.import_module(pkg.__name__ + '.train')
    Nn = pkg._native
    from types import SimpleNamespace as NS
    y = torch.rand(64, 3, 32, 32, device=DEV, generator=torch.Generator(device=DEV).manual_seed(6))
    was = Nn.deterministic()
    Nn.deterministic(True)
    outs = []
    try:
        for shared in (True, False):
            monkeypatch.setattr(fc, 'CHAIN_SLOTS_SHARED', shared)
            torch.manual_seed(4)
            net = pkg.Glow((3, 32, 32), 'image', NS(layers=2, mixtures=None)).to(DEV)
            tr = nftrain.FlowTrainer(net, graph=graph, warmup=1)
            losses = [float(tr.train_on_batch(y)[1]) for _ in range(4)]
            torch.cuda.synchronize()
            outs.append((losses, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()))
    finally:
        Nn.deterministic(was)
    assert outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    assert all(l == l and abs(l) < 1e6 for l in outs[0][0])
    assert Nn.persistent_timeouts() == 0


def test_cifar_glow_with_and_without_the_fused_heads(pkg, monkeypatch):
    """a (3, 32, 32) Glow with two steps per level: the fused heads (C = 12, 48) and the fused couplings against the per-layer
    launches -- z, log-det and every parameter gradient of one training-mode pass."""
    layers_mod = importlib.import_module(pkg.__name__ + '.layers')
    fc = importlib.import_module(pkg.__name__ + '.fused_conv')
    from types import SimpleNamespace as NS
    torch.manual_seed(2)
    net1 = pkg.Glow((3, 32, 32), 'image', NS(layers=2, mixtures=None)).to(DEV)
    y = torch.rand(16, 3, 32, 32, device=DEV)
    with torch.no_grad():
        net1(y)                                         # data-dependent ActNorm initialisation
    net2 = copy.deepcopy(net1)
    for m in net2.modules():
        if hasattr(m, 'initialized'):
            m.initialized = True
    outs = []
    for net, on in ((net1, True), (net2, False)):
        monkeypatch.setattr(layers_mod, 'GLOW_HEAD_W_ON', on)
        monkeypatch.setattr(fc, 'CONV_COUPLING_ON', on)
        net.train()
        z, ld = net(y)
        (0.5 * (z ** 2).sum() - ld.sum()).backward()
        outs.append((z.detach(), ld.detach(), type(z.grad_fn).__name__))
    G.assert_close(outs[0][0], outs[1][0], 5e-5, rtol=1e-5, what='z')
    G.assert_close(outs[0][1], outs[1][1], 2e-3, rtol=1e-5, what='log-det')
    worst = 0.0
    for (n, p1), (_, p2) in zip(net1.named_parameters(), net2.named_parameters()):
        if p2.grad is None:                             # (the pivot matrices and masks)
            assert p1.grad is None, n
            continue
        assert p1.grad is not None, n
        worst = max(worst, float((p1.grad - p2.grad).abs().max()) / max(1.0, float(p2.grad.abs().max())))
    assert worst < 2e-2, worst                          # (a ReLU flip between the two roundings moves a conditioner's gradients)


@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('I,O,H,W,B', [(6, 12, 16, 16, 64), (24, 48, 8, 8, 64), (96, 192, 4, 4, 64), (24, 48, 8, 8, 5), (6, 12, 16, 16, 3)])
def test_packed_weight_images_give_the_same_bits(pkg, I, O, H, W, B, training):
    """The chain kernels take a layer's weights either as effective fp32 weights (every workgroup splits them into the three bf16 planes
    itself) or as the LDS images nf_conv_weight_pack wrote once for the pass (direct global -> LDS loads under the previous layer's
    exchanges: the path a model takes).  Same split, same products, same order: outputs and gradients must agree BITWISE, with the
    fused coupling (the packed 1 x 1 image has the coupling's row order) and without."""
    fc = importlib.import_module(pkg.__name__ + '.fused_conv')
    cond = importlib.import_module(pkg.__name__ + '.conditioners')
    N = pkg._native
    if not fc._chain_usable(B, I, O, H, W):
        pytest.skip('shape outside the chain kernels')
    results = []
    for use_pack in (False, True):
        a, _ = _nets(pkg, I, O)
        a.train(training)
        wns = [m for m in a.modules() if isinstance(m, cond.WeightNorm)]
        torch.manual_seed(3)
        x = torch.randn(B, I, H, W, device=DEV, requires_grad=True)
        with torch.no_grad():
            effs = [m.effective_weight().contiguous() for m in wns]
        for m, w in zip(wns, effs):
            m._w_eff = w.clone().requires_grad_(True)
        if use_pack:
            fc.pack_conv_weights(wns, [m._w_eff for m in wns])
            assert fc._convnet_packs(a) is not None
        else:
            assert fc._convnet_packs(a) is None
        y = fc.convnet_forward(a, x)
        g = torch.randn(y.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
        y.backward(g)
        torch.cuda.synchronize()
        results.append((y.detach().clone(), x.grad.detach().clone(), [m._w_eff.grad.detach().clone() for m in wns],
                        [bn.weight.grad.detach().clone() for bn in a.modules() if isinstance(bn, torch.nn.BatchNorm2d)]))
        for m in wns:
            m._w_eff = None
    (y0, gx0, gw0, gb0), (y1, gx1, gw1, gb1) = results
    assert torch.equal(y0, y1), float((y0 - y1).abs().max())
    assert torch.equal(gx0, gx1), float((gx0 - gx1).abs().max())
    for u, v in zip(gw0 + gb0, gw1 + gb1):
        assert torch.equal(u, v), float((u - v).abs().max())
    assert N.persistent_timeouts() == 0


@pytest.mark.parametrize('B,I,VH,VW,k', [(9, 32, 14, 14, 3), (5, 6, 14, 14, 3), (20, 32, 7, 7, 3), (9, 32, 14, 14, 1), (33, 32, 7, 7, 1), (4, 12, 13, 10, 3)])
def test_masked_maps_on_the_per_layer_kernels(pkg, B, I, VH, VW, k):
    """maps whose sides are no powers of two (MNIST's 14 x 14 and 7 x 7 levels, flows/dataset.py:67-79) kept in power-of-two storage with a
    VALID extent (nf_conv_desc.valid_h / valid_w): dead pixels read as zero, count in no statistic, are never written.  Forward (output,
    batch sums), data gradient (gn_out, its two sums) and weight / bias gradient of a layer against float64 on the CROPPED problem, with
    garbage in the dead region of every input."""
    import torch.nn.functional as TF
    fc = importlib.import_module(pkg.__name__ + '.fused_conv')
    N = pkg._native
    torch.manual_seed(B + I + VH)
    HS = 16 if VH > 8 else 8
    WS_ = 16 if VW > 8 else 8
    O = 32 if k == 3 else 12
    has_bn = I == 32
    R = 8

    def padded(t, fill):
        out = torch.full(t.shape[:2] + (HS, WS_), fill, device=DEV)
        out[:, :, :VH, :VW] = t
        return out

    x = torch.randn(B, I, VH, VW, device=DEV) * 1.5 + 0.3
    w = torch.randn(O, I, k, k, device=DEV) * 0.1
    bias = torch.randn(O, device=DEV) * 0.2
    gamma, beta = torch.rand(I, device=DEV) + 0.5, torch.randn(I, device=DEV) * 0.3
    center = torch.randn(I, device=DEV) * 0.1
    xd = x.double()
    n = B * VH * VW
    mean = xd.mean((0, 2, 3))
    var = xd.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    pre = (xd - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1) * gamma.double().view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)
    act = torch.relu(pre) if has_bn else xd
    want = TF.conv2d(act, w.double(), bias.double(), padding=k // 2)
    xs = (xd - center.double().view(1, -1, 1, 1))
    s1, s2 = torch.zeros(R, 32, device=DEV), torch.zeros(R, 32, device=DEV)
    s1[1, :I] = xs.sum((0, 2, 3)).float()
    s2[4, :I] = (xs * xs).sum((0, 2, 3)).float()
    xp = padded(x, 1.0e3)
    out = torch.full((B, O, HS, WS_), 777.0, device=DEV)
    st1, st2 = torch.zeros(R * 32, device=DEV), torch.zeros(R * 32, device=DEV)
    kw = dict(in_=xp, weight=w, bias=bias, out=out, valid_h=VH, valid_w=VW)
    if k == 3:
        kw.update(stat_sum=st1, stat_sqsum=st2)
    if has_bn:
        kw.update(bn_gamma=gamma, bn_beta=beta, bn_sum=s1.view(-1), bn_sqsum=s2.view(-1), bn_center=center, bn_running_mean=torch.zeros(I, device=DEV),
                  bn_running_var=torch.ones(I, device=DEV), bn_save_mean=torch.zeros(32, device=DEV), bn_save_invstd=torch.zeros(32, device=DEV))
    fc._fwd((B, HS, WS_), I, O, k, True, **kw)
    torch.cuda.synchronize()
    scale = max(1.0, float(want.abs().max()))
    assert float((out[:, :, :VH, :VW].double() - want).abs().max()) <= 3e-5 * scale
    dead = out.clone()
    dead[:, :, :VH, :VW] = 777.0
    assert bool((dead == 777.0).all()), 'a dead pixel was written'
    if k == 3:
        dv = want - bias.double().view(1, -1, 1, 1)
        G.assert_close(st1.view(R, 32).sum(0)[:O], dv.sum((0, 2, 3)).float(), 2e-5 * max(1.0, float(dv.abs().sum((0, 2, 3)).max())), what='stat_sum')
        G.assert_close(st2.view(R, 32).sum(0)[:O], (dv * dv).sum((0, 2, 3)).float(), 2e-5 * float((dv * dv).sum((0, 2, 3)).max()), what='stat_sqsum')
    # backward: G = g_direct on the valid pixels (garbage elsewhere), both passes in one launch
    g = torch.randn(B, O, VH, VW, device=DEV)
    gp = padded(g, -5.0e2)
    gn_out = torch.full((B, I, HS, WS_), 555.0, device=DEV)
    slabs = int(N.load().nf_conv_bwd_slabs(B, HS, WS_))
    g_weff = torch.zeros(slabs, O * I * k * k, device=DEV)
    g_bias = torch.zeros(R * 256, device=DEV)
    sg, sgx = torch.zeros(R * 32, device=DEV), torch.zeros(R * 32, device=DEV)
    kw = dict(in_=xp, weight=w, g_direct=gp, gn_out=gn_out, g_weff=g_weff, g_bias=g_bias, valid_h=VH, valid_w=VW)
    if has_bn:
        kw.update(bn_gamma=gamma, bn_beta=beta, bn_save_mean=mean.float(), bn_save_invstd=invstd.float(), sum_g=sg, sum_gx=sgx)
    fc._bwd((B, HS, WS_), I, O, k, **kw)
    torch.cuda.synchronize()
    gin = TF.conv_transpose2d(g.double(), w.double(), padding=k // 2)
    risky = (pre.abs() < 1e-5) if has_bn else torch.zeros_like(gin, dtype=torch.bool)
    gn = torch.where(pre > 0, gin, torch.zeros_like(gin)) if has_bn else gin
    sgs = max(1.0, float(gn.abs().max()))
    err = (gn_out[:, :, :VH, :VW].double() - gn).abs()
    err[risky] = 0.0
    assert float(err.max()) <= 1e-5 * sgs, float(err.max())
    want_w = torch.nn.grad.conv2d_weight(act, (O, I, k, k), g.double(), padding=k // 2)
    got_w = g_weff.sum(0).view(k * k, O, I).permute(1, 2, 0).reshape(O, I, k, k)
    mag = float(torch.nn.grad.conv2d_weight(act.abs(), (O, I, k, k), g.double().abs(), padding=k // 2).max())
    assert float((got_w.double() - want_w).abs().max()) <= 1e-5 * max(1.0, mag), float((got_w.double() - want_w).abs().max())
    G.assert_close(g_bias.view(R, 256).sum(0)[:O], g.double().sum((0, 2, 3)).float(), 2e-5 * max(1.0, float(g.double().abs().sum((0, 2, 3)).max())), what='g_bias')
    if has_bn:
        xhat = (xd - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
        slack = int(risky.sum()) * sgs * 4.0
        G.assert_close(sg.view(R, 32).sum(0)[:I], gn.sum((0, 2, 3)).float(), 2e-5 * max(1.0, float(gn.abs().sum((0, 2, 3)).max())) + slack, what='sum_g')
        G.assert_close(sgx.view(R, 32).sum(0)[:I], (gn * xhat).sum((0, 2, 3)).float(), 2e-5 * max(1.0, float((gn * xhat).abs().sum((0, 2, 3)).max())) + slack, what='sum_gx')
    assert n == B * VH * VW


def test_mnist_shape_conditioner_dispatches_no_framework_convolution(pkg):
    """the conditioners of the 28 x 28 pyramid (maps 14 x 14 and 7 x 7) run on the HIP kernels, forward and backward: no ATen /
    MIOpen convolution or batch norm is dispatched (the module path, profiled the same way, does dispatch them)"""
    a, b = _nets(pkg, 2, 4)
    a.train()
    b.train()
    x = torch.randn(16, 2, 14, 14, device=DEV)
    banned = ('conv', 'batch_norm', 'relu', 'mm')

    def ops(net):
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
            net(x.clone().requires_grad_(True)).sum().backward()
            torch.cuda.synchronize()
        names = {e.key for e in prof.key_averages()}
        return sorted(n for n in names if n.startswith('aten::') and any(t in n.split('::')[1] for t in banned))

    assert ops(a) == []
    assert any('conv' in n for n in ops(b))


@pytest.mark.parametrize('cls,dims,layers,mix', [('Glow', (1, 32, 32), 2, None), ('RealNVP', (1, 32, 32), 2, None), ('Flowpp', (1, 32, 32), 1, 4),
                                                 ('Glow', (1, 24, 24), 2, None), ('RealNVP', (1, 24, 24), 2, None), ('Flowpp', (1, 24, 24), 1, 4)])
def test_single_channel_image_models_dispatch_no_framework_convolution(pkg, cls, dims, layers, mix):
    """the reference's MNIST shape ((1, 32, 32) after its loader's padding, flows/dataset.py:67-73) and a pyramid without power-of-two
    maps (24 -> 12 -> 6), second training pass (ActNorm initialised): no ATen / MIOpen convolution, batch norm, layer norm, softmax or
    batched matmul anywhere in forward + backward"""
    from types import SimpleNamespace as NS
    torch.manual_seed(1)
    net = getattr(pkg, cls)(dims, 'image', NS(layers=layers, mixtures=mix)).to(DEV)
    net.train()
    y = torch.rand(16, *dims, device=DEV)
    banned = ('conv', 'batch_norm', 'layer_norm', 'softmax', 'bmm')

    def step():
        z, ld = net(y.clone())
        (z.square().sum() - ld.sum()).backward()
        torch.cuda.synchronize()

    step()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        step()
    names = {e.key for e in prof.key_averages()}
    assert sorted(n for n in names if n.startswith('aten::') and any(t in n.split('::')[1] for t in banned)) == []
