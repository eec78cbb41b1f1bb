"""GPU parity of the image Flow++ conditioner kernels (csrc/flowpp_img.hip) against the module stack they replace
(flows/coupling.py:159-166: Conv2d -> GatedConv2d -> LayerNorm -> GatedAttn -> LayerNorm -> Conv2d, flows/modules.py:519-578) and,
kernel by kernel, against plain fp32 PyTorch restatements of the same operators."""
import copy
import importlib

import pytest
import torch
import torch.nn.functional as F

from tests import _golden as G

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = 1.0e-5


def _scaled(t, k=4.0):
    return TOL * k * max(1.0, float(t.detach().abs().max()))


def _native(pkg):
    return importlib.import_module(pkg.__name__ + '._native')


# (in channels, out channels, H = W): the conditioner shapes of Flowpp on CIFAR (flows/flowpp.py:22-57: in 6 / 6 / 24 / 24 / 96 with
# 14 x in outputs) and on the 16 x 16 golden model, plus ragged channel counts
CONV_CASES = [(6, 32, 16, 3), (64, 32, 16, 2), (32, 84, 16, 5), (24, 32, 8, 9), (32, 336, 8, 6), (96, 32, 4, 33), (32, 1344, 4, 17),
              (5, 7, 8, 1), (40, 70, 4, 16),
              # launches of >= 128 workgroups keep the 256-pixel tiles (below that the 64-pixel form runs): both forms at every map size
              (32, 84, 16, 64), (6, 32, 16, 130), (32, 336, 8, 64), (32, 1344, 4, 64), (24, 32, 8, 520)]


@pytest.mark.parametrize('Ci,Co,HW,B', CONV_CASES)
def test_conv_forward_data_and_weight_gradient_vs_torch(pkg, Ci, Co, HW, B):
    N = _native(pkg)
    g = torch.Generator().manual_seed(Ci * 1000 + Co)
    x = torch.randn(B, Ci, HW, HW, generator=g).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3.0 * Ci ** 0.5)).to(DEV)
    b = torch.randn(Co, generator=g).to(DEV)
    gy = torch.randn(B, Co, HW, HW, generator=g).to(DEV)
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
    y_ref = F.conv2d(xr, wr, br, padding=1)
    gx_ref, gw_ref, gb_ref = torch.autograd.grad(y_ref, [xr, wr, br], gy)
    st = N.stream()
    y = torch.empty_like(y_ref)
    N.call('nf_flowpp_img_conv', N.ptr(x), N.ptr(w), N.ptr(b), N.ptr(y), B, Ci, Co, HW, HW, 0, 0, 1, st)
    G.assert_close(y, y_ref, _scaled(y_ref), what='conv forward')
    lib = N.load()
    for ks in sorted({1, int(lib.nf_flowpp_img_conv_ksplit(B, Co, Ci, HW, HW)), (Co + 31) // 32}):    # K-split data gradient: the slabs sum to it
        gx = torch.full((ks, ) + tuple(x.shape), 7.0, device=DEV)
        N.call('nf_flowpp_img_conv', N.ptr(gy), N.ptr(w), None, N.ptr(gx), B, Co, Ci, HW, HW, 0, 1, ks, st)
        G.assert_close(gx.sum(0), gx_ref, _scaled(gx_ref), what='conv data gradient, %d slabs' % ks)
    for ns in sorted({1, int(lib.nf_flowpp_img_wgrad_slabs(B, Ci, Co, HW, HW))}):
        sw = torch.full((ns, 9, Co, Ci), 7.0, device=DEV)                   # tap-major slabs, every element written
        sb = torch.full((ns, Co), 7.0, device=DEV)
        N.call('nf_flowpp_img_conv_wgrad', N.ptr(x), N.ptr(gy), N.ptr(sw), N.ptr(sb), ns, B, Ci, Co, HW, HW, 0, st)
        G.assert_close(sw.sum(0).permute(1, 2, 0).reshape(w.shape), gw_ref, _scaled(gw_ref), what='conv weight gradient, %d slabs' % ns)
        G.assert_close(sb.sum(0), gb_ref, _scaled(gb_ref), what='conv bias gradient, %d slabs' % ns)


@pytest.mark.parametrize('Ci,Co,V,B', [(6, 32, 12, 3), (32, 84, 14, 5), (24, 32, 6, 9), (32, 336, 7, 64), (96, 32, 3, 33), (40, 70, 5, 16),
                                       (32, 84, 13, 64), (6, 32, 9, 130), (32, 1344, 3, 64), (24, 32, 6, 520), (5, 7, 2, 3)])
def test_conv_on_an_image_inside_a_storage_map(pkg, Ci, Co, V, B):
    """(H, W) of the C ABI = the image; tensors in the next power-of-two storage map with GARBAGE in the dead border: forward, data
    gradient (all K splits) and weight / bias gradient equal those of the V x V problem"""
    N = _native(pkg)
    lib = N.load()
    S = int(lib.nf_flowpp_img_storage(V, V))
    assert S >= V and S in (4, 8, 16) and S < 2 * max(V, 3)
    g = torch.Generator().manual_seed(Ci * 1000 + Co + V)
    x = torch.randn(B, Ci, V, V, generator=g).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (3.0 * Ci ** 0.5)).to(DEV)
    b = torch.randn(Co, generator=g).to(DEV)
    gy = torch.randn(B, Co, V, V, generator=g).to(DEV)
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
    y_ref = F.conv2d(xr, wr, br, padding=1)
    gx_ref, gw_ref, gb_ref = torch.autograd.grad(y_ref, [xr, wr, br], gy)

    def stored(t):
        big = torch.full((t.shape[0], t.shape[1], S, S), float('nan'), device=DEV)          # NaN: any read of the border shows
        big[:, :, :V, :V] = t
        return big

    xs, gs = stored(x), stored(gy)
    st = N.stream()
    y = torch.empty(B, Co, S, S, device=DEV)
    N.call('nf_flowpp_img_conv', N.ptr(xs), N.ptr(w), N.ptr(b), N.ptr(y), B, Ci, Co, V, V, 0, 0, 1, st)
    G.assert_close(y[:, :, :V, :V], y_ref, _scaled(y_ref), what='conv forward')
    for ks in sorted({1, int(lib.nf_flowpp_img_conv_ksplit(B, Co, Ci, V, V)), (Co + 31) // 32}):
        gx = torch.full((ks, B, Ci, S, S), 7.0, device=DEV)
        N.call('nf_flowpp_img_conv', N.ptr(gs), N.ptr(w), None, N.ptr(gx), B, Co, Ci, V, V, 0, 1, ks, st)
        G.assert_close(gx.sum(0)[:, :, :V, :V], gx_ref, _scaled(gx_ref), what='conv data gradient, %d slabs' % ks)
    for ns in sorted({1, int(lib.nf_flowpp_img_wgrad_slabs(B, Ci, Co, V, V))}):
        sw = torch.full((ns, 9, Co, Ci), 7.0, device=DEV)
        sb = torch.full((ns, Co), 7.0, device=DEV)
        N.call('nf_flowpp_img_conv_wgrad', N.ptr(xs), N.ptr(gs), N.ptr(sw), N.ptr(sb), ns, B, Ci, Co, V, V, 0, st)
        G.assert_close(sw.sum(0).permute(1, 2, 0).reshape(w.shape), gw_ref, _scaled(gw_ref), what='conv weight gradient, %d slabs' % ns)
        G.assert_close(sb.sum(0), gb_ref, _scaled(gb_ref), what='conv bias gradient, %d slabs' % ns)


@pytest.mark.parametrize('HW,B', [(16, 3), (8, 6), (4, 19)])
def test_gated_convolution_applies_concat_elu_while_staging(pkg, HW, B):
    N = _native(pkg)
    g = torch.Generator().manual_seed(HW)
    x = torch.randn(B, 32, HW, HW, generator=g).to(DEV)
    w = (torch.randn(32, 64, 3, 3, generator=g) / 24.0).to(DEV)
    b = torch.randn(32, generator=g).to(DEV)
    ga = torch.randn(B, 32, HW, HW, generator=g).to(DEV)
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
    a_ref = F.conv2d(F.elu(torch.cat([xr, -xr], dim=1)), wr, br, padding=1)
    gx_ref, gw_ref, gb_ref = torch.autograd.grad(a_ref, [xr, wr, br], ga)
    st = N.stream()
    a = torch.empty_like(a_ref)
    N.call('nf_flowpp_img_conv', N.ptr(x), N.ptr(w), N.ptr(b), N.ptr(a), B, 64, 32, HW, HW, 1, 0, 1, st)
    G.assert_close(a, a_ref, _scaled(a_ref), what='gated conv forward')
    gcat = torch.empty(B, 64, HW, HW, device=DEV)
    N.call('nf_flowpp_img_conv', N.ptr(ga), N.ptr(w), None, N.ptr(gcat), B, 32, 64, HW, HW, 0, 1, 1, st)
    gx = torch.zeros_like(x)
    N.call('nf_flowpp_img_celu_bwd', N.ptr(x), N.ptr(gcat), N.ptr(gx), B, 32, HW, HW, st)
    G.assert_close(gx, gx_ref, _scaled(gx_ref), what='gated conv input gradient')
    ns = int(N.load().nf_flowpp_img_wgrad_slabs(B, 64, 32, HW, HW))
    sw, sb = torch.empty(ns, 9, 32, 64, device=DEV), torch.empty(ns, 32, device=DEV)
    N.call('nf_flowpp_img_conv_wgrad', N.ptr(x), N.ptr(ga), N.ptr(sw), N.ptr(sb), ns, B, 64, 32, HW, HW, 1, st)
    G.assert_close(sw.sum(0).permute(1, 2, 0).reshape(w.shape), gw_ref, _scaled(gw_ref), what='gated conv weight gradient')
    G.assert_close(sb.sum(0), gb_ref, _scaled(gb_ref), what='gated conv bias gradient')


def _cond_pair(pkg, in_chs, n_out, HW, seed):
    cond = importlib.import_module(pkg.__name__ + '.conditioners')
    torch.manual_seed(seed)
    net = cond.flowpp_conditioner(in_chs, n_out, (32, HW, HW), 32, conv=True).to(DEV)
    with torch.no_grad():                               # away from the initial values (LayerNorm affine = 1 / 0, pos_emb ~ 0.01)
        for m in net.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
        net[3].pos_emb.normal_(0, 0.5)
        net[3].conv1.weight.mul_(3.0)                   # attention logits of order one: a softmax that is not uniform
    return net, copy.deepcopy(net)


@pytest.mark.parametrize('split', [True, False], ids=['by_head', 'per_sample'])
@pytest.mark.parametrize('in_chs,n_out,HW,B', [(6, 84, 16, 5), (24, 336, 8, 7), (96, 1344, 4, 18), (4, 56, 4, 3), (6, 84, 8, 64),
                                               # sides that are no powers of two: the image sits in a 16 / 8 / 4 storage map (nfhip.h "STORAGE")
                                               (2, 28, 12, 5), (8, 112, 6, 7), (32, 448, 3, 18), (2, 28, 14, 64), (8, 112, 7, 3), (6, 84, 9, 2),
                                               (6, 84, 5, 130), (4, 56, 2, 3), (4, 56, 1, 2)])
def test_conditioner_vs_module_stack(pkg, monkeypatch, in_chs, n_out, HW, B, split):
    """split: the middle of the conditioner cut by attention head (csrc/flowpp_img_att.hip, H = W in {8, 16}) or one workgroup per sample"""
    fpi = importlib.import_module(pkg.__name__ + '.fused_flowpp_img')
    if split and HW != 16 and HW != 8:
        pytest.skip('only 16 x 16 maps are ever cut by head')
    monkeypatch.setattr(fpi, 'SPLIT_BELOW', 10 ** 9 if split else 0)
    net, ref = _cond_pair(pkg, in_chs, n_out, HW, seed=in_chs + HW)
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, in_chs, HW, HW, generator=g).to(DEV)
    gy = torch.randn(B, n_out, HW, HW, generator=g).to(DEV)
    assert fpi.flowpp_img_fusable(net, x)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = fpi.flowpp_img_forward(net, xa)
    yb = ref(xb)
    G.assert_close(ya, yb, _scaled(yb), what='conditioner output')
    ga = torch.autograd.grad(ya, [xa] + list(net.parameters()), gy)
    gb = torch.autograd.grad(yb, [xb] + list(ref.parameters()), gy)
    names = ['input'] + [n for n, _ in net.named_parameters()]
    for n, a, b in zip(names, ga, gb):
        G.assert_close(a, b, _scaled(b, 8.0), rtol=1e-4, what='gradient of ' + n)


@pytest.mark.parametrize('Ci,Co,HW,B,mode,layers,ns', [(6, 32, 16, 64, 0, 16, 4), (64, 32, 16, 5, 1, 3, 5), (32, 84, 16, 64, 0, 5, 2), (24, 32, 8, 64, 0, 16, 8),
                                                       (32, 1344, 4, 64, 0, 3, 1), (96, 32, 4, 33, 0, 2, 9), (40, 70, 4, 16, 0, 7, 1)])
def test_weight_gradients_of_many_convolutions_in_one_launch(pkg, Ci, Co, HW, B, mode, layers, ns):
    """nf_flowpp_img_conv_wgrad_multi == one nf_flowpp_img_conv_wgrad per layer with the same slab count: the slabs BITWISE"""
    import ctypes
    N = _native(pkg)
    fpi = importlib.import_module(pkg.__name__ + '.fused_flowpp_img')
    g = torch.Generator().manual_seed(Ci * 1000 + Co + layers)
    Cin = Ci // 2 if mode == 1 else Ci                          # (mode 1: the kernel applies concat-ELU to a tensor of Ci / 2 channels)
    st = N.stream()
    singles, multi = [], []
    arr = (fpi.WgradDesc * layers)()
    keep = []
    for i in range(layers):
        x = torch.randn(B, Cin, HW, HW, generator=g).to(DEV)
        gy = torch.randn(B, Co, HW, HW, generator=g).to(DEV)
        sw1, sb1 = torch.full((ns, 9, Co, Ci), 7.0, device=DEV), torch.full((ns, Co), 7.0, device=DEV)
        N.call('nf_flowpp_img_conv_wgrad', N.ptr(x), N.ptr(gy), N.ptr(sw1), N.ptr(sb1), ns, B, Ci, Co, HW, HW, mode, st)
        sw2, sb2 = torch.full((ns, 9, Co, Ci), -3.0, device=DEV), torch.full((ns, Co), -3.0, device=DEV)
        arr[i].inp, arr[i].g_out, arr[i].slab_w, arr[i].slab_b = x.data_ptr(), gy.data_ptr(), sw2.data_ptr(), sb2.data_ptr()
        singles.append((sw1, sb1))
        multi.append((sw2, sb2))
        keep.append((x, gy))
    N.call('nf_flowpp_img_conv_wgrad_multi', ctypes.addressof(arr), layers, ns, B, Ci, Co, HW, HW, mode, st)
    torch.cuda.synchronize()
    for i, ((a, b), (c, d)) in enumerate(zip(singles, multi)):
        assert torch.equal(a, c), ('weight slabs of layer %d' % i, float((a - c).abs().max()))
        assert torch.equal(b, d), ('bias slabs of layer %d' % i, float((b - d).abs().max()))


def test_trainer_step_with_deferred_weight_gradients_matches_the_per_coupling_launches(pkg, monkeypatch):
    """a Flowpp((3, 32, 32)) trainer step with the conditioners' weight gradients deferred to launches of sixteen convolutions (the
    default) and with every coupling launching its own: z and the loss bitwise (deterministic mode), the flat gradient to rounding
    (the slab partition of a launch of many layers differs: another summation order)."""
    fpi = importlib.import_module(pkg.__name__ + '.fused_flowpp_img')
    nftrain = importlib.import_module(pkg.__name__ + '.train')
    from types import SimpleNamespace as NS
    torch.manual_seed(3)
    net = pkg.Flowpp((3, 32, 32), 'image', NS(layers=3, mixtures=4)).to(DEV)
    y = torch.rand(16, 3, 32, 32, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    tr = nftrain.FlowTrainer(net, graph=False)
    tr.train_on_batch(y)                                # data-dependent initialisation
    torch.cuda.synchronize()
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    outs = []
    was = pkg._native.deterministic()
    pkg._native.deterministic(True)
    try:
        for on in (True, False):
            monkeypatch.setattr(fpi, 'FPP_IMG_DEFER_ON', on)
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


def test_image_flowpp_steps_with_and_without_the_fused_heads(pkg, monkeypatch):
    """a Flowpp((3, 32, 32)) with two steps per level: [ActNorm, InvertibleConv1x1] + the conditioner-input gather as the fused Glow heads
    (C = 3: glow_head.hip, C = 12 / 48: glow_head_mfma.hip; their backward in two parts inside a trainer step) against the three layers'
    own launches -- z, log-det and every parameter gradient of one training-mode pass, bare and through the trainer."""
    layers_mod = importlib.import_module(pkg.__name__ + '.layers')
    nftrain = importlib.import_module(pkg.__name__ + '.train')
    from types import SimpleNamespace as NS
    torch.manual_seed(2)
    net1 = pkg.Flowpp((3, 32, 32), 'image', NS(layers=2, mixtures=4)).to(DEV)
    y = torch.rand(8, 3, 32, 32, device=DEV)
    with torch.no_grad():
        net1(y)                                         # data-dependent ActNorm initialisation
    net2 = copy.deepcopy(net1)
    outs = []
    for net, on in ((net1, True), (net2, False)):
        monkeypatch.setattr(layers_mod, 'FLOWPP_HEAD_ON', on)
        net.train()
        z, ld = net(y)
        (0.5 * (z ** 2).sum() - ld.sum()).backward()
        outs.append((z.detach(), ld.detach()))
    G.assert_close(outs[0][0], outs[1][0], 5e-5, rtol=1e-5, what='z')
    G.assert_close(outs[0][1], outs[1][1], 2e-3, rtol=1e-5, what='log-det')
    for (n, p1), (_, p2) in zip(net1.named_parameters(), net2.named_parameters()):
        if p2.grad is None:                             # (the pivot matrices and masks)
            assert p1.grad is None, n
            continue
        assert p1.grad is not None, n
        G.assert_close(p1.grad, p2.grad, _scaled(p2.grad, 8.0), rtol=1e-4, what='gradient of ' + n)
    # ... and inside a trainer step (gradient sinks, deferred parameter gradients of the heads), from identical state
    flats = []
    for net, on in ((net1, True), (net2, False)):
        monkeypatch.setattr(layers_mod, 'FLOWPP_HEAD_ON', on)
        tr = nftrain.FlowTrainer(net, graph=False)
        z, loss = tr._forward_backward(y)
        torch.cuda.synchronize()
        flats.append((z.detach().clone(), float(loss), tr.bucket.flat.detach().clone()))
    G.assert_close(flats[0][0], flats[1][0], 5e-5, rtol=1e-5, what='z (trainer)')
    assert abs(flats[0][1] - flats[1][1]) <= 1e-5 * max(1.0, abs(flats[1][1]))
    d = (flats[0][2] - flats[1][2]).double()
    assert float(d.norm() / flats[1][2].double().norm()) <= 2e-5


def test_direct_gradient_sinks_accumulate(pkg):
    """with a GradBucket the backward adds straight into p.grad (same += as AccumulateGrad)"""
    fpi = importlib.import_module(pkg.__name__ + '.fused_flowpp_img')
    dist = importlib.import_module(pkg.__name__ + '.dist')
    net, ref = _cond_pair(pkg, 6, 84, 8, seed=3)
    bucket = dist.GradBucket(net.parameters())
    x = torch.randn(4, 6, 8, 8, device=DEV)
    gy = torch.randn(4, 84, 8, 8, device=DEV)
    for _ in range(2):                                  # two backward passes: the second adds to the first
        fpi.flowpp_img_forward(net, x).backward(gy)
        ref(x).backward(gy)
    for (n, p), q in zip(net.named_parameters(), ref.parameters()):
        G.assert_close(p.grad, q.grad, _scaled(q.grad, 8.0), rtol=1e-4, what='accumulated gradient of ' + n)
    assert bucket.flat.abs().sum() > 0


def test_layer_takes_the_fused_path_and_matches_the_module_path(pkg, monkeypatch):
    """MixLogAttnCoupling on image data: same y, log-det and gradients with NF_FLOWPP_IMG on and off"""
    layers = importlib.import_module(pkg.__name__ + '.layers')
    fpi = importlib.import_module(pkg.__name__ + '.fused_flowpp_img')
    torch.manual_seed(5)
    cp = layers.MixLogAttnCoupling((12, 16, 16), masking='channelwise', odd=False, n_mixtures=4).to(DEV)
    z = torch.rand(6, 12, 16, 16, device=DEV) * 0.9 + 0.05
    calls = []
    real = fpi.flowpp_img_forward
    monkeypatch.setattr(fpi, 'flowpp_img_forward', lambda net, x: (calls.append(1), real(net, x))[1])
    res = []
    for on in (True, False):
        monkeypatch.setattr(fpi, 'FLOWPP_IMG_ON', on)
        zz = z.clone().requires_grad_(True)
        y, ld = cp(zz, torch.zeros(6, device=DEV))
        gr = torch.autograd.grad((y * y).sum() + ld.sum(), [zz] + list(cp.parameters()))
        res.append((y, ld, gr))
    assert len(calls) == 1
    G.assert_close(res[0][0], res[1][0], _scaled(res[1][0]), what='y')
    G.assert_close(res[0][1], res[1][1], _scaled(res[1][1]), what='log-det')
    for a, b in zip(res[0][2], res[1][2]):
        G.assert_close(a, b, _scaled(b, 8.0), rtol=1e-4, what='gradient')


def test_no_framework_convolution_matmul_softmax_or_layernorm_in_the_conditioner(pkg):
    """the fused conditioner's forward + backward dispatch no ATen / MIOpen operator of the module stack it replaces (the module
    stack itself, profiled the same way, does -- so the check sees what it should)"""
    fpi = importlib.import_module(pkg.__name__ + '.fused_flowpp_img')
    net, ref = _cond_pair(pkg, 6, 84, 16, seed=1)
    x = torch.randn(4, 6, 16, 16, device=DEV)
    gy = torch.randn(4, 84, 16, 16, device=DEV)
    banned = ('conv', 'matmul', 'bmm', 'softmax', 'layer_norm', 'elu', 'sigmoid', 'addmm', 'mm')

    def ops(fn):
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
            fn(x.clone().requires_grad_(True)).backward(gy)
            torch.cuda.synchronize()
        names = {e.key for e in prof.key_averages()}
        return sorted(n for n in names if n.startswith('aten::') and any(b in n.split('::')[1] for b in banned))

    assert ops(lambda t: fpi.flowpp_img_forward(net, t)) == []
    assert any('conv' in n for n in ops(ref)) and any('softmax' in n for n in ops(ref))
