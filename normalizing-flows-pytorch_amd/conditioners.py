"""
Conditioner networks (they produce the shift/scale or mixture parameters of a coupling layer).  Host-side
PyTorch-ROCm code by design (north_star: "host code stays Python on PyTorch-ROCm"): rocBLAS / MIOpen run them.
They are restated here -- same math, same ``state_dict`` keys and same construction-time RNG consumption as the
reference -- so that reference checkpoints load unchanged.

Reference: flows/weight_norm.py:5-45, flows/modules.py:342-438 (ResBlock*, MLP, ConvNet), :500-578 (Gated*,
GatedAttn), flows/coupling.py:142-166 (the Flow++ stack), flows/maf.py:9-85 (MADE).
"""
import math

import numpy as np
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class WeightNorm(nn.Module):
    """w = v * g / (||v||_dim0 + eps).  Keeps the wrapped layer under ``.module`` with parameters ``bias``,
    ``weight_g``, ``weight_v`` (reference key names); the effective weight is recomputed on every call."""

    def __init__(self, module, eps=1.0e-5):
        super().__init__()
        self.module = module
        self.eps = eps
        w = module.weight
        g = torch.norm(w, dim=0)
        v = w / (g.expand_as(w) + eps)
        del module._parameters['weight']
        module.register_parameter('weight_g', nn.Parameter(g.detach()))
        module.register_parameter('weight_v', nn.Parameter(v.detach()))
        self._conv = isinstance(module, nn.Conv2d)

    _w_eff = None      # set for the duration of ONE model forward by fused.weight_norm_all (all layers in a few launches)

    def effective_weight(self):
        if self._w_eff is not None:
            return self._w_eff
        m = self.module
        return m.weight_v * (m.weight_g / (torch.norm(m.weight_v, dim=0) + self.eps)).expand_as(m.weight_v)

    def forward(self, x):
        m = self.module
        if self._conv:
            return F.conv2d(x, self.effective_weight(), m.bias, m.stride, m.padding)
        return F.linear(x, self.effective_weight(), m.bias)


def _wn(module, wrap=True):
    return WeightNorm(module) if wrap else module


def _sync_on():
    from . import dist as nfdist
    return nfdist.sync_stats_active()


def _run_seq(seq, x):
    """nn.Sequential.forward with the training-mode BatchNorm layers on statistics of the GLOBAL batch when the parity mode of
    dist.sync_statistics is on; otherwise exactly ``seq(x)``."""
    if not _sync_on():
        return seq(x)
    from . import dist as nfdist
    for m in seq:
        if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training:
            x = nfdist.sync_batch_norm(m, x)
        elif isinstance(m, (_ResBlock, )):
            x = m(x)
        else:
            x = m(x)
    return x


class _ResBlock(nn.Module):
    def __init__(self, make_norm, make_op, in_channels, out_channels, weight_norm):
        super().__init__()
        self.net = nn.Sequential(
            make_norm(in_channels),
            nn.ReLU(inplace=True),
            _wn(make_op(in_channels, out_channels), weight_norm),
            make_norm(out_channels),
            nn.ReLU(inplace=True),
            _wn(make_op(out_channels, out_channels), weight_norm),
        )
        self.bridge = _wn(make_op(in_channels, out_channels), weight_norm) if in_channels != out_channels \
            else nn.Sequential()

    def forward(self, x):
        return self.bridge(x) + _run_seq(self.net, x)


class ResBlockLinear(_ResBlock):
    def __init__(self, in_channels, out_channels, weight_norm=True):
        super().__init__(nn.BatchNorm1d, nn.Linear, in_channels, out_channels, weight_norm)


class ResBlock2d(_ResBlock):
    def __init__(self, in_channels, out_channels, weight_norm=True):
        super().__init__(nn.BatchNorm2d, lambda i, o: nn.Conv2d(i, o, 3, 1, 1), in_channels, out_channels, weight_norm)


class MLP(nn.Module):
    """in -> 32 -> [res, res] -> BN/ReLU -> out   (flows/modules.py:393-413)"""

    def __init__(self, in_channels, out_channels, base_filters=32, n_blocks=2, weight_norm=True):
        super().__init__()
        self.in_block = nn.Sequential(_wn(nn.Linear(in_channels, base_filters), weight_norm))
        self.mid_block = nn.Sequential(*[ResBlockLinear(base_filters, base_filters, weight_norm)
                                         for _ in range(n_blocks)])
        self.out_block = nn.Sequential(nn.BatchNorm1d(base_filters), nn.ReLU(inplace=True),
                                       _wn(nn.Linear(base_filters, out_channels), weight_norm))
        self.fused = bool(weight_norm) and base_filters == 32 and in_channels <= 32 and out_channels <= 32

    def forward_reference(self, x):
        """module-by-module PyTorch path (rocBLAS + MIOpen); used off-GPU and as the parity reference of the fused one."""
        return _run_seq(self.out_block, self.mid_block(self.in_block(x)))

    def forward(self, x):
        if self.fused and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and not _sync_on():
            from .fused import mlp_forward          # fp32-MFMA linear + BatchNorm kernels: 6 launches
            return mlp_forward(self, x)
        return self.forward_reference(x)


class ConvNet(nn.Module):
    """3x3 conv -> [res, res] -> BN/ReLU -> 1x1 conv   (flows/modules.py:416-438)"""

    def __init__(self, in_channels, out_channels, base_filters=32, n_blocks=2, weight_norm=True):
        super().__init__()
        self.in_block = nn.Sequential(_wn(nn.Conv2d(in_channels, base_filters, 3, 1, 1), weight_norm))
        self.mid_block = nn.Sequential(*[ResBlock2d(base_filters, base_filters, weight_norm) for _ in range(n_blocks)])
        self.out_block = nn.Sequential(nn.BatchNorm2d(base_filters), nn.ReLU(inplace=True),
                                       _wn(nn.Conv2d(base_filters, out_channels, 1, 1, 0), weight_norm))

    # csrc/conv_bn.hip; NF_FUSED_CONV=0 falls back to the MIOpen + ATen module path
    fused = os.environ.get('NF_FUSED_CONV', '1') != '0'

    def forward_reference(self, x):
        """module-by-module PyTorch path (MIOpen + ATen); used off-GPU and as the parity reference of the fused one."""
        return _run_seq(self.out_block, self.mid_block(self.in_block(x)))

    def forward(self, x):
        if self.fused and x.is_cuda and x.dtype == torch.float32 and not _sync_on():
            from .fused_conv import convnet_forward, convnet_usable   # fp32-MFMA conv + BatchNorm kernels: 6 launches
            if convnet_usable(self, x):
                return convnet_forward(self, x)
        return self.forward_reference(x)


# ---- Flow++ conditioner ---------------------------------------------------------------------------------------------

def _concat_elu(x):
    return F.elu(torch.cat([x, -x], dim=1))


class _Gated(nn.Module):
    def forward(self, x):
        C = x.size(1)
        h = _concat_elu(self.op(_concat_elu(x)))
        y, gate = torch.split(h, C, dim=1)
        return x + y * torch.sigmoid(gate)


class GatedLinear(_Gated):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.op = nn.Linear(in_features * 2, out_features)


class GatedConv2d(_Gated):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.op = nn.Conv2d(in_channels * 2, out_channels, 3, 1, 1)


class GatedAttn(nn.Module):
    """multi-head dot-product attention over the positions of the feature map with a sigmoid gate; for 2-D data
    there is one position and the softmax is trivially 1 (flows/modules.py:541-578)."""

    def __init__(self, in_out_shape, filters=8, heads=4):
        super().__init__()
        assert filters % heads == 0
        self.channels = in_out_shape[0]
        self.filters = filters
        self.heads = heads
        self.conv1 = nn.Conv1d(self.channels, filters * 3, 1, 1, 0)
        self.conv2 = nn.Conv1d(filters, self.channels * 2, 1, 1, 0)
        self.pos_emb = nn.Parameter(torch.randn(1, *in_out_shape) * 0.01)

    def forward(self, x):
        shape = x.size()
        B, C = shape[0], shape[1]
        D = self.filters // self.heads
        assert C == self.channels
        if x[0].numel() == C and self.filters == C:
            # ONE position (2-D data): the softmax over a single key is exactly 1 (and has a zero Jacobian), so the
            # attention output is the third projection itself.  Same function and same gradients as the general path
            # below (the V/K rows of conv1 receive exact zeros either way), without B*heads 1x8x1 batched GEMMs.
            t = (x + self.pos_emb).reshape(B, C)
            q = F.linear(t, self.conv1.weight[2 * self.filters:, :, 0], self.conv1.bias[2 * self.filters:])
            y, gate = torch.split(F.linear(q, self.conv2.weight[:, :, 0], self.conv2.bias), C, dim=1)
            return x + (y * torch.sigmoid(gate)).view(shape)
        tokens = (x + self.pos_emb).view(B, C, -1)
        proj = self.conv1(tokens).view(B, 3 * self.heads, D, -1)
        V, K, Q = torch.split(proj, self.heads, dim=1)          # the reference's naming of the three projections
        scores = F.softmax(torch.matmul(V.permute(0, 1, 3, 2), K) / math.sqrt(D), dim=2)
        mixed = torch.matmul(Q, scores).view(B, C, -1)
        y, gate = torch.split(self.conv2(mixed), C, dim=1)
        return x + (y * torch.sigmoid(gate)).view(shape)


def flowpp_conditioner(in_chs, n_out, mid_shape, base_filters, conv):
    """the nn.Sequential of MixLogAttnCoupling (flows/coupling.py:142-149 / :159-166)."""
    if conv:
        first, last = nn.Conv2d(in_chs, base_filters, 3, 1, 1), None
        gated = GatedConv2d(base_filters, base_filters)
    else:
        first = nn.Linear(in_chs, base_filters)
        gated = GatedLinear(base_filters, base_filters)
    ln1 = nn.LayerNorm(mid_shape)
    attn = GatedAttn(mid_shape, base_filters)
    ln2 = nn.LayerNorm(mid_shape)
    last = nn.Conv2d(base_filters, n_out, 3, 1, 1) if conv else nn.Linear(base_filters, n_out)
    return nn.Sequential(first, gated, ln1, attn, ln2, last)


# ---- MADE ---------------------------------------------------------------------------------------------------------

def made_degrees_to_masks(D, num_hidden, base_filters, rng):
    """mask rule of MADE._create_masks (flows/maf.py:66-85), drawing hidden degrees from ``rng.randint``."""
    m_prev = np.arange(D)
    widths = [D] + [base_filters] * num_hidden
    masks = []
    for out_dims in widths[1:]:
        lo = min(int(m_prev.min()), D - 2)
        m = rng.randint(lo, D - 1, size=(out_dims))
        masks.append((m_prev[None, :] <= m[:, None]).astype(np.float32))
        m_prev = m
    last = np.zeros((D, widths[-1]), dtype=np.float32)
    for k in range(widths[-1]):
        last[m_prev[k] + 1:, k] = 1.0
    masks.append(last)
    return masks
