"""
The Flow++ conditioner of IMAGE data (flows/coupling.py:159-166: Conv2d -> GatedConv2d -> LayerNorm -> GatedAttn -> LayerNorm ->
Conv2d, flows/modules.py:519-578) on the kernels of csrc/flowpp_img.hip -- C ABI ``nf_flowpp_img_conv / _conv_wgrad / _mid_fwd /
_mid_bwd / _celu_bwd``: 4 launches forward and 10 backward, no MIOpen / ATen convolution, matmul, softmax or LayerNorm kernel.

Only the two convolution outputs in front of the gate are kept for the backward (x = conv0's output, a = the gated
convolution's output): everything between the gate and the last convolution is recomputed per sample inside the backward kernel.
"""
import os

import ctypes

import torch

from . import _native as N
from .functional import _sinks

FLOWPP_IMG_ON = os.environ.get('NF_FLOWPP_IMG', '1') != '0'
# the middle cut by attention head (B x 4 workgroups) for batches below this many samples; 0 = never (csrc/flowpp_img_att.hip)
SPLIT_BELOW = 160
# (the weight gradients on a side stream were tried and removed: 5.79 ms per Flowpp CIFAR-shape step against 5.57 on one stream -- the
# fork / join edges of the hipGraph cost more than the overlap returns)
HID = 32


def _tensors(net):
    first, gated, ln1, attn, ln2, last = net
    return [first.weight, first.bias, gated.op.weight, gated.op.bias, ln1.weight, ln1.bias, attn.pos_emb, attn.conv1.weight,
            attn.conv1.bias, attn.conv2.weight, attn.conv2.bias, ln2.weight, ln2.bias, last.weight, last.bias]


def flowpp_img_fusable(net, x):
    """net: the nn.Sequential of MixLogAttnCoupling for image data (coupling.py:159-166) with the reference's widths (32 filters,
    4 heads) on a square map of side <= 16.  Sides that are no powers of two (a 24 x 24 image: 12, 6, 3) run in the next power-of-two
    storage map with a dead border (include/nfhip.h, "STORAGE")."""
    if not (FLOWPP_IMG_ON and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[0] > 0):
        return False
    try:
        first, gated, ln1, attn, ln2, last = net
    except (TypeError, ValueError):
        return False
    B, I0, Hh, Ww = x.shape
    nn = torch.nn
    if not (isinstance(first, nn.Conv2d) and isinstance(last, nn.Conv2d) and isinstance(ln1, nn.LayerNorm)
            and isinstance(ln2, nn.LayerNorm) and isinstance(getattr(gated, 'op', None), nn.Conv2d)):
        return False
    for cv in (first, gated.op, last):
        if cv.kernel_size != (3, 3) or cv.padding != (1, 1) or cv.stride != (1, 1) or cv.bias is None or cv.groups != 1:
            return False
    shape = (HID, Hh, Ww)
    if not (first.in_channels == I0 and first.out_channels == HID and gated.op.in_channels == 2 * HID
            and gated.op.out_channels == HID and last.in_channels == HID and attn.filters == HID and attn.channels == HID
            and attn.heads == 4 and tuple(ln1.normalized_shape) == shape and tuple(ln2.normalized_shape) == shape
            and ln1.elementwise_affine and ln2.elementwise_affine and ln1.eps == 1.0e-5 and ln2.eps == 1.0e-5
            and tuple(attn.pos_emb.shape) == (1, ) + shape):
        return False
    from .dist import sync_stats_active
    if sync_stats_active():           # (nothing to synchronise -- LayerNorm is per sample -- but the parity mode runs the module path)
        return False
    return bool(N.load().nf_flowpp_img_usable(B, max(I0, 2 * HID), max(last.out_channels, 2 * HID), Hh, Ww))


class _FusedFlowppImg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_in, *ts):
        for t in ts:
            if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                raise RuntimeError('fused image Flow++ conditioner needs contiguous fp32 device parameters')
        (W0, b0, Wg, bg, l1g, l1b, pos, c1w, c1b, c2w, c2b, l2g, l2b, W5, b5) = [t.detach() for t in ts]
        x_in = x_in.contiguous()
        B, I0, Hh, Ww = x_in.shape
        O = W5.shape[0]
        dev, st = x_in.device, N.stream()
        S = int(N.load().nf_flowpp_img_storage(Hh, Ww))          # side of the storage map (= Hh on the CIFAR / MNIST pyramids)
        if S != Hh:                                              # the kernels never read the dead border: left uninitialised
            xs = x_in.new_zeros(B, I0, S, S)                      # (zero border: nothing uninitialised can reach a batch sum)
            xs[:, :, :Hh, :Ww] = x_in
            x_in = xs
        x = torch.empty(B, HID, S, S, dtype=torch.float32, device=dev)
        a = torch.empty_like(x)
        x4 = torch.empty_like(x)
        out = torch.empty(B, O, S, S, dtype=torch.float32, device=dev)
        N.call('nf_flowpp_img_conv', N.ptr(x_in), N.ptr(W0), N.ptr(b0), N.ptr(x), B, I0, HID, Hh, Ww, 0, 0, 1, st)
        N.call('nf_flowpp_img_conv', N.ptr(x), N.ptr(Wg), N.ptr(bg), N.ptr(a), B, 2 * HID, HID, Hh, Ww, 1, 0, 1, st)
        # (8 x 8 maps: measured slower cut by head -- 64-thread workgroups, one wave per SIMD -- than one workgroup per sample)
        split = B < SPLIT_BELOW and Hh == 16 and bool(N.load().nf_flowpp_img_att_usable(B, Hh, Ww))
        if split:
            mixed = torch.empty_like(x)
            cj = torch.empty(B, 4, Hh * Ww, dtype=torch.float32, device=dev)
            N.call('nf_flowpp_img_att_fwd', N.ptr(x), N.ptr(a), N.ptr(l1g), N.ptr(l1b), N.ptr(pos), N.ptr(c1w), N.ptr(c1b), N.ptr(mixed),
                   N.ptr(cj), B, Hh, Ww, st)
            N.call('nf_flowpp_img_post_fwd', N.ptr(x), N.ptr(a), N.ptr(l1g), N.ptr(l1b), N.ptr(mixed), N.ptr(c2w), N.ptr(c2b), N.ptr(l2g),
                   N.ptr(l2b), N.ptr(x4), B, Hh, Ww, st)
        else:
            mixed = cj = x.new_empty(0)
            N.call('nf_flowpp_img_mid_fwd', N.ptr(x), N.ptr(a), N.ptr(l1g), N.ptr(l1b), N.ptr(pos), N.ptr(c1w), N.ptr(c1b), N.ptr(c2w),
                   N.ptr(c2b), N.ptr(l2g), N.ptr(l2b), N.ptr(x4), B, Hh, Ww, st)
        N.call('nf_flowpp_img_conv', N.ptr(x4), N.ptr(W5), N.ptr(b5), N.ptr(out), B, HID, O, Hh, Ww, 0, 0, 1, st)
        ctx.save_for_backward(x_in, x, a, x4, mixed, cj, *ts)
        ctx.split = split
        ctx.image = (Hh, Ww)
        return out if S == Hh else out[:, :, :Hh, :Ww].contiguous()

    @staticmethod
    def backward(ctx, g_out):
        x_in, x, a, x4, mixed, cj, *ts = ctx.saved_tensors
        (W0, b0, Wg, bg, l1g, l1b, pos, c1w, c1b, c2w, c2b, l2g, l2b, W5, b5) = [t.detach() for t in ts]
        B, I0, S, _ = x_in.shape
        Hh, Ww = ctx.image
        O = W5.shape[0]
        dev, st = x_in.device, N.stream()
        g_out = g_out.contiguous()
        if S != Hh:
            gs = g_out.new_zeros(B, O, S, S)
            gs[:, :, :Hh, :Ww] = g_out
            g_out = gs
        sinks = _sinks(*ts)
        if sinks is not None:
            dst, direct = sinks, True
        else:                                                    # handed to autograd, which may keep them
            flat = torch.zeros(sum(t.numel() for t in ts), dtype=torch.float32, device=dev)
            dst, o = [], 0
            for t in ts:
                dst.append(flat[o:o + t.numel()].view(t.shape))
                o += t.numel()
            direct = False
        (gW0, gb0, gWg, gbg, gl1g, gl1b, gpos, gc1w, gc1b, gc2w, gc2b, gl2g, gl2b, gW5, gb5) = dst
        lib = N.load()
        jobs = []

        defer = direct and FPP_IMG_DEFER.active

        def wgrad(inp, g, gw, gb, Ci, Co, mode):
            """slabs of one convolution's weight / bias gradient; folded into gw / gb (+=) by the one nf_slab_sum below"""
            if defer:                                            # sixteen convolutions of this shape per launch where the pass ends
                FPP_IMG_DEFER.layers.append(((B, Ci, Co, Hh, Ww, mode), inp, g, gw, gb))
                return
            ns = int(lib.nf_flowpp_img_wgrad_slabs(B, Ci, Co, Hh, Ww))
            sw = torch.empty(ns * Co * Ci * 9 + ns * Co, dtype=torch.float32, device=dev)
            sb = sw[ns * Co * Ci * 9:]
            N.call('nf_flowpp_img_conv_wgrad', N.ptr(inp), N.ptr(g), N.ptr(sw), sb.data_ptr(), ns, B, Ci, Co, Hh, Ww, mode, st)
            jobs.append((sw, gw, Co * Ci * 9, Co * Ci * 9, ns, True, 9))              # the slabs are tap-major (9, Co, Ci)
            jobs.append((sb, gb, Co, Co, ns, True, 1))

        # last convolution: K = 9 O is cut into slabs that the next kernel sums on load
        ks = int(lib.nf_flowpp_img_conv_ksplit(B, O, HID, Hh, Ww))
        g4 = torch.empty(ks, B, HID, S, S, dtype=torch.float32, device=dev)
        N.call('nf_flowpp_img_conv', N.ptr(g_out), N.ptr(W5), None, N.ptr(g4), B, O, HID, Hh, Ww, 0, 1, ks, st)
        wgrad(x4, g_out, gW5, gb5, HID, O, 0)
        # gate / LayerNorm / attention / LayerNorm
        g_x = torch.empty_like(x)
        g_a = torch.empty_like(x)
        if ctx.split:
            g3 = torch.empty_like(x)
            g_mixed = torch.empty_like(x)
            gt_part = torch.empty(B, 4, HID, Hh * Ww, dtype=torch.float32, device=dev)
            # the (32, H, W) parameter gradients leave per sample and are folded with the convolutions' slabs (no same-address atomics)
            ps = torch.empty(5, B, HID, Hh, Ww, dtype=torch.float32, device=dev)
            n = HID * Hh * Ww
            for k_, dst_ in enumerate((gl2g, gl2b, gl1g, gl1b, gpos)):
                jobs.append((ps[k_], dst_, n, n, B, True, 1))
            N.call('nf_flowpp_img_post_bwd', N.ptr(x), N.ptr(a), N.ptr(l1g), N.ptr(l1b), N.ptr(mixed), N.ptr(c2w), N.ptr(c2b), N.ptr(l2g),
                   N.ptr(l2b), N.ptr(g4), ks, N.ptr(g3), N.ptr(g_mixed), N.ptr(gc2w), N.ptr(gc2b), N.ptr(ps[0]), N.ptr(ps[1]), 1, B, Hh,
                   Ww, st)
            N.call('nf_flowpp_img_att_bwd', N.ptr(x), N.ptr(a), N.ptr(l1g), N.ptr(l1b), N.ptr(pos), N.ptr(c1w), N.ptr(c1b), N.ptr(mixed),
                   N.ptr(cj), N.ptr(g_mixed), N.ptr(gt_part), N.ptr(gc1w), N.ptr(gc1b), B, Hh, Ww, st)
            N.call('nf_flowpp_img_pre_bwd', N.ptr(x), N.ptr(a), N.ptr(l1g), N.ptr(l1b), N.ptr(g3), N.ptr(gt_part), N.ptr(g_x), N.ptr(g_a),
                   N.ptr(ps[2]), N.ptr(ps[3]), N.ptr(ps[4]), 1, B, Hh, Ww, st)
        elif defer:
            # one workgroup per sample: its (32, H, W) parameter gradients leave per sample as well and join the end-of-pass fold
            n = HID * Hh * Ww
            ps = torch.empty(5, B, n, dtype=torch.float32, device=dev)
            for k_, dst_ in enumerate((gl2g, gl2b, gl1g, gl1b, gpos)):
                jobs.append((ps[k_], dst_, n, n, B, True, 1))
            N.call('nf_flowpp_img_mid_bwd', N.ptr(x), N.ptr(a), N.ptr(l1g), N.ptr(l1b), N.ptr(pos), N.ptr(c1w), N.ptr(c1b), N.ptr(c2w),
                   N.ptr(c2b), N.ptr(l2g), N.ptr(l2b), N.ptr(g4), N.ptr(g_x), N.ptr(g_a), N.ptr(ps[2]), N.ptr(ps[3]), N.ptr(ps[4]),
                   N.ptr(gc1w), N.ptr(gc1b), N.ptr(gc2w), N.ptr(gc2b), N.ptr(ps[0]), N.ptr(ps[1]), 1, B, Hh, Ww, ks, st)
        else:
            N.call('nf_flowpp_img_mid_bwd', N.ptr(x), N.ptr(a), N.ptr(l1g), N.ptr(l1b), N.ptr(pos), N.ptr(c1w), N.ptr(c1b), N.ptr(c2w),
                   N.ptr(c2b), N.ptr(l2g), N.ptr(l2b), N.ptr(g4), N.ptr(g_x), N.ptr(g_a), N.ptr(gl1g), N.ptr(gl1b), N.ptr(gpos),
                   N.ptr(gc1w), N.ptr(gc1b), N.ptr(gc2w), N.ptr(gc2b), N.ptr(gl2g), N.ptr(gl2b), 0, B, Hh, Ww, ks, st)
        # gated convolution (its input is concat_elu(x), applied while staging)
        g_cat = torch.empty(B, 2 * HID, S, S, dtype=torch.float32, device=dev)
        N.call('nf_flowpp_img_conv', N.ptr(g_a), N.ptr(Wg), None, N.ptr(g_cat), B, HID, 2 * HID, Hh, Ww, 0, 1, 1, st)
        wgrad(x, g_a, gWg, gbg, 2 * HID, HID, 1)
        N.call('nf_flowpp_img_celu_bwd', N.ptr(x), N.ptr(g_cat), N.ptr(g_x), B, HID, S, S, st)       # (elementwise: the storage extent)
        # first convolution
        wgrad(x_in, g_x, gW0, gb0, I0, HID, 0)
        g_in = None
        if ctx.needs_input_grad[0]:
            g_in = torch.empty_like(x_in)
            N.call('nf_flowpp_img_conv', N.ptr(g_x), N.ptr(W0), None, N.ptr(g_in), B, HID, I0, Hh, Ww, 0, 1, 1, st)
        if defer:
            FPP_IMG_DEFER.sums += jobs                           # (the per-sample LayerNorm / position-embedding sums wait as well)
        else:
            from .fused_conv import _slab_sum_all
            _slab_sum_all(jobs)
        if g_in is not None and S != Hh:
            g_in = g_in[:, :, :Hh, :Ww].contiguous()
        if direct:
            return (g_in, ) + (None, ) * len(ts)
        return (g_in, ) + tuple(dst)


class WgradDesc(ctypes.Structure):
    """include/nfhip.h: nf_flowpp_img_wgrad_desc"""
    _fields_ = [(f, ctypes.c_void_p) for f in ('inp', 'g_out', 'slab_w', 'slab_b')]


class FlowppImgDefer:
    """Deferred weight gradients of the image Flow++ conditioners.  One convolution's weight-gradient launch is a single tile's latency
    chain on 64 .. 192 workgroups at B = 64 (20 .. 31 us, three per coupling, 483 per step at layers = 32) and only the optimizer waits
    for it.  Inside a trainer step (FlowTrainer opens it; closed -- the default -- nothing is deferred) a conditioner's backward queues
    (input, gradient, sinks) and ``flush`` runs the queued convolutions of one shape NF_FLOWPP_IMG_WGRAD_MAX per launch
    (nf_flowpp_img_conv_wgrad_multi), then every slab sum of the pass in launches of NF_SLAB_SUM_MAX jobs."""

    def __init__(self):
        self.active = False
        self.layers = []        # (key, input, gradient, weight sink, bias sink): the tensors are kept alive until the flush
        self.sums = []

    def begin(self):
        self.layers, self.sums = [], []
        self.active = FPP_IMG_DEFER_ON

    def flush(self):
        layers, sums = self.layers, self.sums
        self.layers, self.sums, self.active = [], [], False
        if not layers and not sums:
            return
        lib = N.load()
        step = N.header_constant('NF_FLOWPP_IMG_WGRAD_MAX')
        groups = {}
        for e in layers:
            groups.setdefault(e[0], []).append(e)
        jobs = []
        for (B, Ci, Co, Hh, Ww, mode), es in groups.items():
            single = int(lib.nf_flowpp_img_wgrad_slabs(B, Ci, Co, Hh, Ww))
            blocks = ((Co + 31) // 32) * ((Ci + 31) // 32)
            for k0 in range(0, len(es), step):
                chunk = es[k0:k0 + step]
                # slabs per layer: the launch as a whole fills the chip WGRAD_ROUNDS times over, a layer alone need not
                ns = max(1, min(single, (WGRAD_ROUNDS * 256 + len(chunk) * blocks - 1) // (len(chunk) * blocks)))
                per = ns * Co * Ci * 9 + ns * Co
                dev = chunk[0][1].device
                slab = torch.empty(len(chunk) * per, dtype=torch.float32, device=dev)
                arr = (WgradDesc * len(chunk))()
                for i, (_, inp, g, gw, gb) in enumerate(chunk):
                    sw = slab[i * per:(i + 1) * per]
                    sb = sw[ns * Co * Ci * 9:]
                    arr[i].inp, arr[i].g_out, arr[i].slab_w, arr[i].slab_b = inp.data_ptr(), g.data_ptr(), sw.data_ptr(), sb.data_ptr()
                    jobs.append((sw, gw, Co * Ci * 9, Co * Ci * 9, ns, True, 9))      # the slabs are tap-major (9, Co, Ci)
                    jobs.append((sb, gb, Co, Co, ns, True, 1))
                N.call('nf_flowpp_img_conv_wgrad_multi', ctypes.addressof(arr), len(chunk), ns, B, Ci, Co, Hh, Ww, mode, N.stream())
        from .fused_conv import _slab_sum_all
        _slab_sum_all(jobs + sums)


FPP_IMG_DEFER_ON = True         # (internal: tests compare the deferred weight gradients with the per-coupling ones)
WGRAD_ROUNDS = 2
FPP_IMG_DEFER = FlowppImgDefer()


def flowpp_img_forward(net, x):
    """the (B, O, H, W) coupling parameters of the image Flow++ conditioner ``net`` (see flowpp_img_fusable)."""
    return _FusedFlowppImg.apply(x, *_tensors(net))
