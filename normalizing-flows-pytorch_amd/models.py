"""
Layer stacks with the reference's constructor and call surface:

    net = Glow(dims, datatype, cfg)      cfg.layers (all), cfg.mixtures (Flowpp)
    z, log_df_dz = net(y)                forward flow, log_df_dz of shape (B,)
    y, log_df_dz = net.backward(z)       inverse flow (sampling)

Reference: flows/realnvp.py:9-63, flows/glow.py:10-68, flows/flowpp.py:9-78, flows/maf.py:122-148.
The multi-scale image recipe (shared by RealNVP / Glow / Flow++) is written once here.
"""
import torch
import torch.nn as nn

from .layers import (ActNorm, AffineCoupling, AutoregressiveTransfrom, BatchNorm, Compose, InvertibleConv1x1, Logit,
                     MixLogAttnCoupling, Squeeze2d, Unsqueeze2d)


class _FlowModel(nn.Module):
    def __init__(self, dims, datatype=None, cfg=None):
        super().__init__()
        self.dims = tuple(dims)
        self.n_layers = cfg.layers
        self.net = Compose(self._build(self.dims, datatype, cfg))

    # one "flow step" = normalisation (+ 1x1 conv) + coupling; supplied by the subclass
    def _step(self, dims, masking, odd, cfg):
        raise NotImplementedError

    def _build(self, dims, datatype, cfg):
        K = self.n_layers
        layers = []
        if datatype == 'image':
            layers.append(Logit(eps=0.01))
            mid = dims
            while max(mid[1], mid[2]) > 8:
                for i in range(K):
                    layers += self._step(mid, 'checkerboard', i % 2 != 0, cfg)
                layers.append(Squeeze2d(odd=False))
                mid = (mid[0] * 4, mid[1] // 2, mid[2] // 2)
                for i in range(K):
                    layers += self._step(mid, 'channelwise', i % 2 != 0, cfg)
            for i in range(K + 1):
                layers += self._step(mid, 'checkerboard', i % 2 != 0, cfg)
            while mid[1] != dims[1] or mid[2] != dims[2]:
                layers.append(Unsqueeze2d(odd=False))
                mid = (mid[0] // 4, mid[1] * 2, mid[2] * 2)
        else:
            for i in range(K):
                layers += self._step(dims, 'checkerboard', i % 2 != 0, cfg)
        return layers

    def _zero_ld(self, z):
        if z.dtype == torch.float32:
            from .workspace import zeros_owned
            return zeros_owned((z.size(0), ), z.device)      # (inside a trainer step: part of the step's one memset, no fill launch)
        return torch.zeros(z.size(0), dtype=z.dtype, device=z.device)

    def _wn_convs(self):
        """the weight-normed convolutions of the image conditioners (their weight-norm arithmetic is batched per pass)."""
        c = getattr(self, '_wn_cache', None)
        if c is None:
            from .conditioners import WeightNorm
            c = self._wn_cache = [m for m in self.modules() if isinstance(m, WeightNorm) and m._conv]
        return c

    def _plu_convs(self):
        """the invertible 1x1 convolutions that run as modules in the forward direction (more than NF.HEAD_MAX_C channels: the
        smaller ones are inside the fused Glow head): their PLU weights are computed in batched launches per pass."""
        c = getattr(self, '_plu_cache', None)
        if c is None:
            from .layers import InvertibleConv1x1
            from . import functional as NF
            c = self._plu_cache = [m for m in self.modules() if isinstance(m, InvertibleConv1x1)
                                   and NF.HEAD_MAX_C < m.L.shape[0] <= NF.PLU_MAX_C]
        return c

    def _inverse_weights_all(self):
        """W^-1 = U'^-1 L'^-1 Pp of every invertible 1x1 convolution of an image model (modules.py:485-492), the layers of one
        width solved as ONE batch: two triangular solves per width instead of two per layer (~10 small launches each, 129
        layers in the CIFAR Glow).  Stashed on the modules for the inverse pass under way."""
        from .layers import InvertibleConv1x1
        convs = getattr(self, '_inv_cache', None)
        if convs is None:
            convs = self._inv_cache = [m for m in self.modules() if isinstance(m, InvertibleConv1x1)]
        groups = {}
        for m in convs:
            groups.setdefault(m.L.shape[0], []).append(m)
        with torch.no_grad():
            for C, ms in groups.items():
                L = torch.stack([m.L for m in ms]) * torch.stack([m.L_mask for m in ms])
                U = torch.stack([m.U for m in ms]) * torch.stack([m.U_mask for m in ms])
                d = torch.stack([m.sign_s * torch.exp(m.log_s) for m in ms])
                Lp = L + torch.eye(C, dtype=L.dtype, device=L.device)
                Up = U + torch.diag_embed(d)
                Pp = torch.stack([m._pivot_matrix() for m in ms])
                X = torch.linalg.solve_triangular(Lp, Pp, upper=False, unitriangular=True)
                Winv = torch.linalg.solve_triangular(Up, X, upper=True)
                for i, m in enumerate(ms):
                    m._W_inv = Winv[i]
        return convs

    def _with_weight_norms(self, fn, z):
        wns = self._wn_convs() if z.is_cuda else []
        plus = self._plu_convs() if (z.is_cuda and z.dim() == 4 and fn == self.net) else []
        invs = self._inverse_weights_all() if (z.is_cuda and z.dim() == 4 and fn != self.net) else []
        if not wns and not plus and not invs:
            return fn(z, self._zero_ld(z))
        from . import fused as FUSED
        FUSED.weight_norm_all(wns)
        FUSED.plu_weights_all(plus)
        try:
            return fn(z, self._zero_ld(z))
        finally:
            for m in wns:
                m._w_eff = None
            for m in plus:
                m._W_eff = None
            for m in invs:
                m._W_inv = None

    def forward_slice(self, z, log_df_dz, a, b):
        """layers[a:b] of the stack in the forward direction on (z, log_df_dz) -- the same Compose peepholes, weight-norm and PLU
        pre-passes as a whole forward, restricted to those layers (they share this model's parameters).  Lets a test feed a
        slice of a full-size model with the oracle's activations at that depth (tests/test_gpu_slices.py)."""
        sub = Compose(list(self.net.layers[a:b]))
        if not z.is_cuda:
            return sub(z, log_df_dz)
        from . import functional as NF
        from . import fused as FUSED
        from .conditioners import WeightNorm
        from .layers import InvertibleConv1x1
        wns = [m for m in sub.modules() if isinstance(m, WeightNorm) and m._conv]
        plus = [m for m in sub.modules() if isinstance(m, InvertibleConv1x1)
                and NF.HEAD_MAX_C < m.L.shape[0] <= NF.PLU_MAX_C] if z.dim() == 4 else []
        FUSED.weight_norm_all(wns)
        FUSED.plu_weights_all(plus)
        try:
            return sub(z, log_df_dz)
        finally:
            for m in wns:
                m._w_eff = None
            for m in plus:
                m._W_eff = None

    def forward(self, z):
        if z.is_cuda and z.device.index != torch.cuda.current_device():
            with torch.cuda.device(z.device):      # (a model on another GPU than the current one: launch on ITS device, as torch's own ops do)
                return self._with_weight_norms(self.net, z)
        return self._with_weight_norms(self.net, z)

    def backward(self, z):
        if z.is_cuda and z.device.index != torch.cuda.current_device():
            with torch.cuda.device(z.device):
                return self._with_weight_norms(self.net.backward, z)
        return self._with_weight_norms(self.net.backward, z)


class RealNVP(_FlowModel):
    def _step(self, dims, masking, odd, cfg):
        return [BatchNorm(dims, affine=False), AffineCoupling(dims, masking=masking, odd=odd)]


class Glow(_FlowModel):
    def _step(self, dims, masking, odd, cfg):
        return [ActNorm(dims), InvertibleConv1x1(dims[0]), AffineCoupling(dims, masking=masking, odd=odd)]


class Flowpp(_FlowModel):
    """flows/flowpp.py:9-78: image steps are ActNorm + inv-1x1 + mixture coupling, density steps have no inv-1x1."""

    def _step(self, dims, masking, odd, cfg):
        layers = [ActNorm(dims)]
        if len(dims) == 3:
            layers.append(InvertibleConv1x1(dims[0]))
        layers.append(MixLogAttnCoupling(dims, masking=masking, odd=odd, n_mixtures=cfg.mixtures))
        return layers


class MAF(_FlowModel):
    """flows/maf.py:122-148: [flow BatchNorm, autoregressive transform] x layers; density data only."""

    def _build(self, dims, datatype, cfg):
        if datatype == 'image':
            raise NotImplementedError('Sorry, MAF for image generation is not supported!')   # reference forgets to raise
        layers = []
        for _ in range(self.n_layers):
            layers.append(BatchNorm(dims, affine=False))
            layers.append(AutoregressiveTransfrom(dims[0]))
        return layers
