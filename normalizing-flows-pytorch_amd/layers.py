"""
Bijective layers with the reference's nn.Module surface:

    layer(z, log_df_dz)          -> (z', log_df_dz')      forward flow  (data -> latent)
    layer.backward(y, log_df_dz) -> (y', log_df_dz')      INVERSE flow  (latent -> data; not autograd!)

Same class names, constructor signatures, parameter / buffer names and shapes as flows/modules.py,
flows/coupling.py, flows/squeeze.py and flows/maf.py, so reference ``state_dict``s load unchanged.  The
transforms themselves run as HIP kernels (functional.py -> libnfhip.so); there is no CPU path.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native as N
from . import functional as NF
from . import fused as FUSED
from . import fused_flowpp_img as FPI
from .conditioners import MLP, ConvNet, flowpp_conditioner, made_degrees_to_masks


class Identity(nn.Module):
    def forward(self, x, log_df_dz):
        return x, log_df_dz

    def backward(self, x, log_df_dz):
        return x, log_df_dz


GLOW_HEAD_W_ON = True           # (internal: the MFMA head of image flow steps, csrc/glow_head_mfma.hip)
HEAD_IN_CHAIN = True            # (internal: that head's forward in the prologue of the coupling's chain launch, csrc/conv_chain.hip)
FLOWPP_HEAD_ON = True           # (internal: image Flow++ steps take the fused Glow heads too; tests compare with the three layers' own launches)


class Compose(nn.Module):
    """flows/modules.py:325-339.  The forward direction applies one peephole fusion: a Glow flow step
    [ActNorm, InvertibleConv1x1, AffineCoupling] on <= 4 channels runs its first two layers and the coupling's
    split-gather as ONE launch (functional.glow_head) -- same math, same parameters, same state_dict; skipped when a
    member carries forward hooks (the reference's debug mode registers NaN hooks, main.py:312-313)."""

    fuse = True

    def __init__(self, layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)

    @property
    def _fuse_now(self):
        """peephole fusion is off in the synchronised-statistics parity mode: the fused kernels take their batch statistics
        in-kernel, per replica (dist.sync_statistics)"""
        from . import dist as nfdist
        return self.fuse and not nfdist.sync_stats_active()

    def _glow_step_at(self, i, z):
        L = self.layers
        if not (self._fuse_now and z.is_cuda and i + 2 < len(L) and z.shape[1] <= NF.HEAD_MAX_C):
            return False
        a, c, k = L[i], L[i + 1], L[i + 2]
        if not (type(a) is ActNorm and type(c) is InvertibleConv1x1 and (type(k) is AffineCoupling or self._flowpp_image_coupling(k, z))):
            return False
        return not (a._forward_hooks or c._forward_hooks or k._forward_hooks or a._forward_pre_hooks
                    or c._forward_pre_hooks or k._forward_pre_hooks)

    @staticmethod
    def _flowpp_image_coupling(k, z):
        """an image Flow++ step [ActNorm, InvertibleConv1x1, MixLogAttnCoupling] (flows/flowpp.py:64-70) takes the same fused head as a Glow
        step: ActNorm + 1x1 + the conditioner-input gather in one launch, its backward in two parts (round 5)"""
        return FLOWPP_HEAD_ON and type(k) is MixLogAttnCoupling and z.dim() == 4 and k.mode in (N.SPLIT_CHANNEL, N.SPLIT_CHECKER)

    def _glow_step_w_at(self, i, z):
        """[ActNorm, InvertibleConv1x1 (weight assembled by the model's batched PLU pre-pass), AffineCoupling] on image data with
        more channels than the in-kernel PLU head takes"""
        L = self.layers
        if not (GLOW_HEAD_W_ON and self._fuse_now and z.is_cuda and z.dim() == 4 and i + 2 < len(L) and z.shape[1] > NF.HEAD_MAX_C):
            return False
        a, c, k = L[i], L[i + 1], L[i + 2]
        if not (type(a) is ActNorm and type(c) is InvertibleConv1x1 and (type(k) is AffineCoupling or self._flowpp_image_coupling(k, z))):
            return False
        if c._W_eff is None or k.mode not in (N.SPLIT_CHANNEL, N.SPLIT_CHECKER) or not NF.glow_head_w_usable(z, k.mode):
            return False
        return not (a._forward_hooks or c._forward_hooks or k._forward_hooks or a._forward_pre_hooks
                    or c._forward_pre_hooks or k._forward_pre_hooks)

    def _glow_run_at(self, i, z):
        """the maximal run of fused-step-capable Glow steps on (N, 2 | 4) data that starts at layer i -- initialised ActNorms,
        training-mode conditioners of one mind, every parameter with a direct gradient sink -- or None."""
        L, run, j = self.layers, [], i
        if z.dim() != 2:
            return None
        while self._glow_step_at(j, z):
            a, c, k = L[j], L[j + 1], L[j + 2]
            if not (a.initialized and k.mode == N.SPLIT_1D and isinstance(k.net, MLP)
                    and k.net.training == L[i + 2].net.training):
                break
            run.append((a, c, k))
            j += 3
        if FUSED.glow_flow_nograd_usable(z, run):            # density evaluation: no autograd node at all
            return ('nograd', run)
        return run if FUSED.glow_flow_vec_usable(z, run) else None

    def _realnvp_run_at(self, i, z):
        """the maximal run of fused-step-capable RealNVP steps [flow BatchNorm, AffineCoupling(MLP)] on (N, 2 | 4) data, or None"""
        L, run, j = self.layers, [], i
        if z.dim() != 2:
            return None
        while self._bn_step_at(j, z):
            a, k = L[j], L[j + 1]
            if not (type(k) is AffineCoupling and k.mode == N.SPLIT_1D and isinstance(k.net, MLP)):
                break
            run.append((a, k))
            j += 2
        return run if FUSED.realnvp_flow_vec_usable(z, run) else None

    def _maf_run_at(self, i, z):
        """the maximal run of fused-step-capable MAF steps [flow BatchNorm, AutoregressiveTransfrom] on (N, D <= 4) data, or None"""
        L, run, j = self.layers, [], i
        if z.dim() != 2:
            return None
        while self._bn_step_at(j, z) and type(L[j + 1]) is AutoregressiveTransfrom:
            run.append((L[j], L[j + 1]))
            j += 2
        return run if FUSED.maf_flow_vec_usable(z, run) else None

    def _bn_step_at(self, i, z):
        """[flow BatchNorm (training, affine=False), AffineCoupling | AutoregressiveTransfrom] -> fused BatchNorm head"""
        L = self.layers
        if not (self._fuse_now and z.is_cuda and i + 1 < len(L)):
            return False
        a, k = L[i], L[i + 1]
        if not (type(a) is BatchNorm and a.training and not isinstance(a.log_gamma, nn.Parameter)):
            return False
        if not (type(k) is AffineCoupling or type(k) is AutoregressiveTransfrom):
            return False
        return not (a._forward_hooks or k._forward_hooks or a._forward_pre_hooks or k._forward_pre_hooks)

    def _flowpp_pair_at(self, i, z):
        """[MixLogAttnCoupling, ActNorm (initialised)] on two features -> the next step's ActNorm rides the coupling's launches"""
        L = self.layers
        if not (self._fuse_now and z.is_cuda and i + 1 < len(L)):
            return False
        k, a = L[i], L[i + 1]
        if not (type(k) is MixLogAttnCoupling and type(a) is ActNorm and k.mode == N.SPLIT_1D):
            return False
        if a._forward_hooks or k._forward_hooks or a._forward_pre_hooks or k._forward_pre_hooks:
            return False
        return FUSED.flowpp_post_actnorm_usable(z, k, a)

    def forward(self, z, log_df_dz):
        L, n, i = self.layers, len(self.layers), 0
        while i < n:
            run = self._realnvp_eval_run_at(i, z) if not torch.is_grad_enabled() else None
            if run is not None:                                    # density evaluation: the run in one launch, no exchange
                z, log_df_dz = FUSED.realnvp_flow_vec_eval(z, log_df_dz, run)
                i += 2 * len(run)
                continue
            if self._maf_pair_at(i, z) and FUSED.maf_step_eval_usable(z, L[i], L[i + 1]):
                z, log_df_dz = FUSED.maf_step_eval(z, log_df_dz, L[i], L[i + 1])   # density evaluation: one launch, no exchange
                i += 2
                continue
            if self._bn_step_at(i, z):
                a, k = L[i], L[i + 1]
                run = self._realnvp_run_at(i, z)
                if run is not None:                                # the whole run of steps: one launch per direction
                    z, log_df_dz = FUSED.realnvp_flow_vec(z, log_df_dz, run)
                    i += 2 * len(run)
                    continue
                if (type(k) is AffineCoupling and k.mode == N.SPLIT_1D and isinstance(k.net, MLP)
                        and FUSED.realnvp_step_vec_usable(z, a, k.net)):
                    z, log_df_dz = FUSED.realnvp_step_vec(z, log_df_dz, a, k)        # the whole step: one launch
                elif type(k) is AffineCoupling:
                    h, z1c, log_df_dz = NF.flowbn_head(z, log_df_dz, a, k.mode, k.odd, gather=True)
                    z, log_df_dz = k.couple(h, z1c, log_df_dz)
                elif FUSED.maf_step_usable(z, a, k):
                    run = self._maf_run_at(i, z)
                    if run is not None:                            # the run as one autograd node, one fold for all steps
                        z, log_df_dz = FUSED.maf_flow_vec(z, log_df_dz, run)
                        i += 2 * len(run)
                        continue
                    z, log_df_dz = FUSED.maf_step_vec(z, log_df_dz, a, k)            # the whole step: one launch
                else:
                    h, log_df_dz = NF.flowbn_head(z, log_df_dz, a)
                    z, log_df_dz = k(h, log_df_dz)
                i += 2
            elif self._glow_step_at(i, z):
                a, c, k = L[i], L[i + 1], L[i + 2]
                run = self._glow_run_at(i, z)
                if run is not None and run[0] == 'nograd':
                    z, log_df_dz = FUSED.glow_flow_vec_nograd(z, log_df_dz, run[1])
                    i += 3 * len(run[1])
                    continue
                if run is not None:                                # the whole run of steps: one launch per direction
                    z, log_df_dz = FUSED.glow_flow_vec(z, log_df_dz, run)
                    i += 3 * len(run)
                    continue
                if not a.initialized:
                    NF.actnorm_init_(z, a.log_scale, a.bias, a.eps)
                    a.initialized = True
                if k.mode == N.SPLIT_1D and isinstance(k.net, MLP) and FUSED.glow_step_vec_usable(z, k.net):
                    z, log_df_dz = FUSED.glow_step_vec(z, log_df_dz, a, c, k)        # the whole step: one launch
                else:
                    # (image data: the head rides the chain launches of the fused couplings on either side, as the MFMA head below)
                    from . import fused_conv as FC
                    defer = (HEAD_IN_CHAIN and z.dim() == 4 and type(k) is AffineCoupling and isinstance(k.net, ConvNet)
                             and z.is_contiguous() and FC.head_in_chain_ok(k.net, z, k.mode))
                    h, z1c, log_df_dz = NF.glow_head(z, log_df_dz, a.log_scale, a.bias, c.P, c.L, c.U, c.L_mask,
                                                     c.U_mask, c.sign_s, c.log_s, k.mode, k.odd,
                                                     bwd_defer=defer and NF.from_fused_coupling(z), defer=defer)
                    z, log_df_dz = k.couple(h, z1c, log_df_dz)
                    if defer and NF.flush_pending_head(h):
                        raise RuntimeError('a deferred Glow head was not performed by its coupling launch')
                i += 3
            elif self._glow_step_w_at(i, z):                      # image data, 9 .. 64 channels: head in one MFMA launch
                a, c, k = L[i], L[i + 1], L[i + 2]
                if not a.initialized:
                    NF.actnorm_init_(z, a.log_scale, a.bias, a.eps)
                    a.initialized = True
                W, holder, idx = c._W_eff
                # the head's forward rides the prologue of the coupling's chain launch when that launch follows (round 5)
                from . import fused_conv as FC
                defer = (HEAD_IN_CHAIN and type(k) is AffineCoupling and isinstance(k.net, ConvNet) and z.is_contiguous()
                         and FC.head_in_chain_ok(k.net, z, k.mode))
                # ... and its data gradient the prologue of the PREVIOUS step's backward chain launch, when z is that launch's output (round 6)
                bwd_defer = defer and NF.from_fused_coupling(z)
                h, z1c, log_df_dz = NF.glow_head_w(z, log_df_dz, a.log_scale, a.bias, W, c.log_s, holder, idx, k.mode, k.odd, defer=defer,
                                                   bwd_defer=bwd_defer)
                z, log_df_dz = k.couple(h, z1c, log_df_dz)
                if defer and NF.flush_pending_head(h):
                    raise RuntimeError('a deferred Glow head was not performed by its coupling launch')
                i += 3
            elif self._flowpp_pair_at(i, z):
                z, log_df_dz = FUSED.flowpp_coupling_vec(z, log_df_dz, L[i], post=L[i + 1])   # coupling + next ActNorm
                i += 2
            else:
                z, log_df_dz = L[i](z, log_df_dz)
                i += 1
        return z, log_df_dz

    def _glow_inverse_run_ending_at(self, i, z):
        """the maximal run of fused-step-capable Glow steps on (N, 2 | 4) data whose LAST layer is layer i (in forward order), or
        None: its inverse is one launch per step, or one for the whole run (fused.glow_flow_vec_inverse)"""
        L, run, j = self.layers, [], i
        if z.dim() != 2 or torch.is_grad_enabled() and z.requires_grad:
            return None
        while j >= 2 and self._glow_step_at(j - 2, z):
            a, c, k = L[j - 2], L[j - 1], L[j]
            if not (k.mode == N.SPLIT_1D and isinstance(k.net, MLP) and not k._backward_hooks):
                break
            run.append((a, c, k))
            j -= 3
        run.reverse()
        return run if run and FUSED.glow_inverse_usable(z, run) else None

    def _realnvp_pair_at(self, j, z):
        """[flow BatchNorm(affine=False), AffineCoupling(MLP)] at layers j, j + 1 on (N, 2 | 4) data, any mode, no hooks"""
        L = self.layers
        if not (self._fuse_now and z.is_cuda and z.dim() == 2 and j >= 0 and j + 1 < len(L)):
            return False
        a, k = L[j], L[j + 1]
        if not (type(a) is BatchNorm and not isinstance(a.log_gamma, nn.Parameter) and type(k) is AffineCoupling
                and k.mode == N.SPLIT_1D and isinstance(k.net, MLP)):
            return False
        return not (a._forward_hooks or k._forward_hooks or a._forward_pre_hooks or k._forward_pre_hooks)

    def _maf_pair_at(self, j, z):
        """[flow BatchNorm(affine=False), AutoregressiveTransfrom] at layers j, j + 1 on (N, D) data, any mode, no hooks"""
        L = self.layers
        if not (self._fuse_now and z.is_cuda and z.dim() == 2 and j >= 0 and j + 1 < len(L)):
            return False
        a, k = L[j], L[j + 1]
        if not (type(a) is BatchNorm and not isinstance(a.log_gamma, nn.Parameter) and type(k) is AutoregressiveTransfrom):
            return False
        return not (a._forward_hooks or k._forward_hooks or a._forward_pre_hooks or k._forward_pre_hooks)

    def _realnvp_inverse_run_ending_at(self, i, z):
        L, run, j = self.layers, [], i
        if z.dim() != 2 or torch.is_grad_enabled() and z.requires_grad:
            return None
        while self._realnvp_pair_at(j - 1, z):
            run.append((L[j - 1], L[j]))
            j -= 2
        run.reverse()
        return run if run and FUSED.realnvp_inverse_usable(z, run) else None

    def _realnvp_eval_run_at(self, i, z):
        L, run, j = self.layers, [], i
        while self._realnvp_pair_at(j, z):
            run.append((L[j], L[j + 1]))
            j += 2
        return run if run and FUSED.realnvp_eval_usable(z, run) else None

    def backward(self, z, log_df_dz):
        """INVERSE flow (sampling).  The sampling kernels build no autograd graph: without a request for one the whole pass runs under
        no_grad (the reference's ``sample_y``, main.py:113, leaves autograd on and never uses the graph).  An input that requires grad --
        or the ``differentiable_inverse()`` context, for gradients of the parameters alone -- takes the graph-building form of every
        layer's inverse instead (inverse_grad.py: the reference's formulas over the engine's own conditioners and gathers)."""
        if torch.is_grad_enabled():
            from . import inverse_grad as IG
            if IG.wanted(z, log_df_dz):
                return IG.layer_inverse(self, z, log_df_dz)
            with torch.no_grad():
                return self.backward(z, log_df_dz)
        i = len(self.layers) - 1
        while i >= 0:
            run = self._glow_inverse_run_ending_at(i, z)
            if run is not None:
                z, log_df_dz = FUSED.glow_flow_vec_inverse(z, log_df_dz, run)
                i -= 3 * len(run)
                continue
            run = self._realnvp_inverse_run_ending_at(i, z)
            if run is not None:
                z, log_df_dz = FUSED.realnvp_flow_vec_inverse(z, log_df_dz, run)
                i -= 2 * len(run)
                continue
            if self._maf_pair_at(i - 1, z) and not (torch.is_grad_enabled() and z.requires_grad) \
                    and FUSED.maf_step_inverse_usable(z, self.layers[i - 1], self.layers[i]):
                z, log_df_dz = FUSED.maf_step_inverse(z, log_df_dz, self.layers[i - 1], self.layers[i])
                i -= 2
                continue
            z, log_df_dz = self.layers[i].backward(z, log_df_dz)
            i -= 1
        return z, log_df_dz


class Logit(nn.Module):
    """flows/modules.py:141-156"""

    def __init__(self, eps=1.0e-5):
        super().__init__()
        self.eps = eps

    def forward(self, x, log_df_dz):
        return NF.logit(x, log_df_dz, self.eps)

    def backward(self, x, log_df_dz):
        return NF.logit(x, log_df_dz, self.eps, inverse=True)


class Sigmoid(nn.Module):
    """flows/modules.py:125-138"""

    def forward(self, x, log_df_dz):
        return NF.bijector(x, log_df_dz, NF.BIJ_SIGMOID)

    def backward(self, x, log_df_dz):
        return NF.bijector(x, log_df_dz, NF.BIJ_SIGMOID_INV)


class Tanh(nn.Module):
    """flows/modules.py:158-170"""

    def forward(self, x, log_df_dz):
        return NF.bijector(x, log_df_dz, NF.BIJ_TANH)

    def backward(self, x, log_df_dz):
        with torch.no_grad():
            return NF.bijector(x, log_df_dz, NF.BIJ_ARCTANH)


class Arctanh(nn.Module):
    """flows/modules.py:173-183 (its log-det sums over dim 1: vector data)"""

    def forward(self, x, log_df_dz):
        return NF.bijector(x, log_df_dz, NF.BIJ_ARCTANH)

    def backward(self, x, log_df_dz):
        with torch.no_grad():
            return NF.bijector(x, log_df_dz, NF.BIJ_TANH)


class Squeeze1d(nn.Module):
    """flows/squeeze.py:114-132: the alternating entries of a vector as two concatenated halves"""

    def __init__(self, odd=False):
        super().__init__()
        self.odd = bool(odd)

    def forward(self, z, log_df_dz):
        return NF.squeeze1d(z, self.odd), log_df_dz

    def backward(self, z, log_df_dz):
        return NF.squeeze1d(z, self.odd, inverse=True), log_df_dz


class Unsqueeze1d(nn.Module):
    """flows/squeeze.py:135-151: the inverse map of Squeeze1d as a forward layer"""

    def __init__(self, odd=False):
        super().__init__()
        self.odd = bool(odd)

    def forward(self, z, log_df_dz):
        return NF.squeeze1d(z, self.odd, inverse=True), log_df_dz

    def backward(self, z, log_df_dz):
        return NF.squeeze1d(z, self.odd), log_df_dz


class MixLogCDF(nn.Module):
    """flows/modules.py:186-212: CDF of a mixture of logistics as a bijector of x given (log_pi, mu, s); the inverse is the
    reference's bisection (25 or 100 iterations by its batch-global exit rule).  One HIP launch forward, two inverse."""

    def forward(self, x, log_pi, mu, s, log_df_dz):
        return NF.mixlogcdf(x, log_pi, mu, s, log_df_dz)

    def backward(self, x, log_pi, mu, s, log_df_dz):
        return NF.mixlogcdf(x, log_pi, mu, s, log_df_dz, inverse=True)


def _param_shape(num_features):
    dims = [1] + [1 for _ in num_features]
    dims[1] = num_features[0]
    return dims


class ActNorm(nn.Module):
    """flows/modules.py:225-256.  ``initialized`` is a plain attribute exactly like the reference's (it is not in
    the state_dict, so the first batch after ``load_state_dict`` re-initialises -- appendix D Q2); set it to True to
    keep loaded values."""

    def __init__(self, num_features, eps=1.0e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.dimensions = _param_shape(num_features)
        self.log_scale = nn.Parameter(torch.zeros(self.dimensions))
        self.bias = nn.Parameter(torch.zeros(self.dimensions))
        self.initialized = False

    def forward(self, z, log_df_dz):
        if not self.initialized:
            from . import dist as nfdist
            if nfdist.sync_stats_active():              # parity mode: the initialisation statistics of the GLOBAL batch
                with torch.no_grad():
                    mean, var, n = nfdist.global_moments(z)
                    std = torch.sqrt(var * (n / (n - 1.0)))                  # unbiased, modules.py:240
                    self.log_scale.data.copy_(torch.log(std + self.eps).view(self.dimensions))
                    self.bias.data.copy_(mean.view(self.dimensions))
            else:
                NF.actnorm_init_(z, self.log_scale, self.bias, self.eps)
            self.initialized = True
        return NF.chan_affine(N.OP_ACTNORM, z, log_df_dz, self.log_scale, self.bias)

    def backward(self, y, log_df_dz):
        return NF.chan_affine(N.OP_ACTNORM, y, log_df_dz, self.log_scale, self.bias, inverse=True)


class BatchNorm(nn.Module):
    """flow BatchNorm, flows/modules.py:259-322: batch statistics are constants for autograd; biased variance
    with eps stored inside; inverse uses the batch buffers in train mode."""

    def __init__(self, num_features, momentum=0.1, eps=1.0e-5, affine=True):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.momentum = momentum
        self.dimensions = _param_shape(num_features)
        if affine:
            self.log_gamma = nn.Parameter(torch.zeros(self.dimensions))
            self.beta = nn.Parameter(torch.zeros(self.dimensions))
        else:
            self.register_buffer('log_gamma', torch.zeros(self.dimensions))
            self.register_buffer('beta', torch.zeros(self.dimensions))
        self.register_buffer('running_mean', torch.zeros(self.dimensions))
        self.register_buffer('running_var', torch.ones(self.dimensions))
        self.register_buffer('batch_mean', torch.zeros(self.dimensions))
        self.register_buffer('batch_var', torch.ones(self.dimensions))

    def _stats(self):
        if self.training:
            return self.batch_mean, self.batch_var
        return self.running_mean, self.running_var

    def forward(self, x, log_det_jacob):
        from . import dist as nfdist
        if self.training and nfdist.sync_stats_active():    # parity mode: statistics of the GLOBAL batch (modules.py:284-294)
            with torch.no_grad():
                mean, var, _ = nfdist.global_moments(x)
                self.batch_mean.copy_(mean.view(self.dimensions))
                self.batch_var.copy_((var + self.eps).view(self.dimensions))
                self.running_mean.mul_(1.0 - self.momentum).add_(self.batch_mean * self.momentum)
                self.running_var.mul_(1.0 - self.momentum).add_(self.batch_var * self.momentum)
        elif self.training:
            NF.flowbn_update_(x, self.batch_mean, self.batch_var, self.running_mean, self.running_var, self.eps,
                              self.momentum)
        mean, var = self._stats()
        return NF.chan_affine(N.OP_FLOWBN, x, log_det_jacob, mean, var, self.log_gamma, self.beta)

    def backward(self, x, log_det_jacob):
        mean, var = self._stats()
        return NF.chan_affine(N.OP_FLOWBN, x, log_det_jacob, mean, var, self.log_gamma, self.beta, inverse=True)


class InvertibleConv1x1(nn.Module):
    """Glow's PLU-parameterised 1x1 convolution, flows/modules.py:441-497.  Frozen constants are
    ``nn.Parameter(requires_grad=False)`` like the reference, so they appear in the state_dict (appendix D Q3)."""

    def __init__(self, in_out_channels):
        super().__init__()
        C = in_out_channels
        W = torch.zeros((C, C), dtype=torch.float32)
        nn.init.orthogonal_(W)
        LU, pivots = torch.linalg.lu_factor(W)
        P, L, U = torch.lu_unpack(LU, pivots)
        self.P = nn.Parameter(P, requires_grad=False)
        self.L = nn.Parameter(L, requires_grad=True)
        self.U = nn.Parameter(U, requires_grad=True)
        self.I = nn.Parameter(torch.eye(C), requires_grad=False)
        self.pivots = nn.Parameter(pivots, requires_grad=False)
        L_mask = np.tril(np.ones((C, C), dtype='float32'), k=-1)
        self.L_mask = nn.Parameter(torch.from_numpy(L_mask), requires_grad=False)
        self.U_mask = nn.Parameter(torch.from_numpy(L_mask.T.copy()), requires_grad=False)
        s = torch.diag(U)
        self.log_s = nn.Parameter(torch.log(torch.abs(s)), requires_grad=True)
        self.sign_s = nn.Parameter(torch.sign(s), requires_grad=False)
        self._perm_cache = None

    def weight(self):
        """W = P (L o L_mask + I) (U o U_mask + diag(sign_s exp(log_s)))   (modules.py:471-473)"""
        Lp = self.L * self.L_mask + self.I
        Up = self.U * self.U_mask + torch.diag(self.sign_s * torch.exp(self.log_s))
        return self.P @ Lp @ Up

    def _pivot_matrix(self):
        """row-swap matrix of the stored LAPACK pivots (what torch.lu_solve applies to its right-hand side)."""
        key = (self.pivots._version, self.pivots.device)
        if self._perm_cache is None or self._perm_cache[0] != key:
            piv = self.pivots.detach().cpu().numpy().astype(np.int64) - 1
            perm = np.arange(piv.size)
            for i, p in enumerate(piv):
                perm[[i, p]] = perm[[p, i]]
            M = torch.zeros(piv.size, piv.size)
            M[torch.arange(piv.size), torch.from_numpy(perm)] = 1.0
            self._perm_cache = (key, M.to(self.pivots.device))
        return self._perm_cache[1]

    def inverse_weight(self):
        """W^-1 = U'^-1 L'^-1 Pp from the same LU factors / pivots the reference feeds torch.lu_solve
        (modules.py:485-492)."""
        Lp = self.L * self.L_mask + self.I
        Up = self.U * self.U_mask + torch.diag(self.sign_s * torch.exp(self.log_s))
        X = torch.linalg.solve_triangular(Lp, self._pivot_matrix(), upper=False, unitriangular=True)
        return torch.linalg.solve_triangular(Up, X, upper=True)

    _W_eff = None      # (W, holder, index): set for the duration of ONE model forward by fused.plu_weights_all

    def forward(self, z, log_df_dz):
        if self._W_eff is not None and z.is_cuda:
            W, holder, idx = self._W_eff
            return NF.invconv_apply_w(z, log_df_dz, W, self.log_s, holder, idx)
        if z.shape[1] <= NF.PLU_MAX_C:
            return NF.invconv_plu(z, log_df_dz, self.P, self.L, self.U, self.L_mask, self.U_mask, self.sign_s,
                                  self.log_s)
        return NF.invconv(z, self.weight(), log_df_dz, self.log_s)

    _W_inv = None      # set for the duration of ONE model inverse pass by models._inverse_weights_all (all layers of a width at once)

    def backward(self, y, log_df_dz):
        with torch.no_grad():
            W_inv = self._W_inv if (self._W_inv is not None and y.is_cuda) else self.inverse_weight()
            return NF.invconv_inverse(y, W_inv, log_df_dz, self.log_s)


# ---- squeeze family (flows/squeeze.py:114-189) ---------------------------------------------------------------------

def _swap_halves(z):
    h = z.shape[1] // 2
    return torch.cat([z[:, h:], z[:, :h]], dim=1)


class Squeeze2d(nn.Module):
    """flows/squeeze.py:153-170: space-to-depth, channel order k = 4 c + 2 dy + dx; ``odd`` swaps the two channel halves of the result
    (squeeze.py:94-95; no reference model builds it: one extra copy here)."""

    def __init__(self, odd=False):
        super().__init__()
        self.odd = bool(odd)

    def forward(self, z, log_df_dz):
        out = NF.squeeze2d(z)
        return (_swap_halves(out) if self.odd else out), log_df_dz

    def backward(self, z, log_df_dz):
        return NF.unsqueeze2d(_swap_halves(z) if self.odd else z), log_df_dz


class Unsqueeze2d(nn.Module):
    """flows/squeeze.py:173-189: the inverse map of Squeeze2d as a forward layer."""

    def __init__(self, odd=False):
        super().__init__()
        self.odd = bool(odd)

    def forward(self, z, log_df_dz):
        return NF.unsqueeze2d(_swap_halves(z) if self.odd else z), log_df_dz

    def backward(self, z, log_df_dz):
        out = NF.squeeze2d(z)
        return (_swap_halves(out) if self.odd else out), log_df_dz


# ---- coupling layers (flows/coupling.py) -----------------------------------------------------------------------------

class AbstractCoupling(nn.Module):
    def __init__(self, dims, masking='checkerboard', odd=False):
        super().__init__()
        self.dims = dims
        self.odd = bool(odd)
        if len(dims) == 1:
            if dims[0] % 2 != 0:
                raise Exception('coupling layers need an even feature count, got %s (flows/squeeze.py:67)' % str(dims))
            self.mode = N.SPLIT_1D
        elif len(dims) == 3 and masking == 'checkerboard':
            self.mode = N.SPLIT_CHECKER
        elif len(dims) == 3 and masking == 'channelwise':
            self.mode = N.SPLIT_CHANNEL
        else:
            raise Exception('unsupported combination of masking and dimension: %s, %s' % (masking, str(dims)))

    def squeeze(self, z):
        return NF.half_gather(z, 0, self.mode, self.odd), NF.half_gather(z, 1, self.mode, self.odd)

    def conditioner_input(self, z):
        return NF.half_gather(z, 1, self.mode, self.odd)


class AdditiveCoupling(AbstractCoupling):
    """NICE additive coupling, flows/coupling.py:52-79 (no reference model builds it): z0 <- z0 + net_t(z1), log-det unchanged.
    Runs on the affine coupling kernels with the scale pinned to zero (s = 0 * tanh(0) + 0): split + shift + merge in one launch.
    The conditioner's widths are the reference's -- including its checkerboard width (dims[0] instead of 2 * dims[0]), which cannot run
    on image data there either."""

    def __init__(self, dims, masking='checkerboard', odd=False):
        super().__init__(dims, masking, odd)
        if len(dims) == 1:
            in_chs = dims[0] // 2 if not odd else (dims[0] + 1) // 2
            self.net_t = MLP(in_chs, dims[0] - in_chs)
        else:
            in_out_chs = dims[0] if masking == 'checkerboard' else dims[0] // 2
            self.net_t = ConvNet(in_out_chs, in_out_chs)
        self.register_buffer('_zero', torch.zeros(1), persistent=False)

    def _params(self, z):
        t = self.net_t(self.conditioner_input(z))
        return torch.cat([t, torch.zeros_like(t)], dim=1)         # [shift | raw scale = 0]

    def forward(self, z, log_df_dz):
        return NF.affine_coupling(z, self._params(z), self._zero, self._zero, log_df_dz, self.mode, self.odd)

    def backward(self, y, log_df_dz):
        return NF.affine_coupling(y, self._params(y), self._zero, self._zero, log_df_dz, self.mode, self.odd, inverse=True)


class AffineCoupling(AbstractCoupling):
    """RealNVP / Glow affine coupling, flows/coupling.py:82-122; split + transform + merge + log-det are ONE kernel."""

    def __init__(self, dims, masking='checkerboard', odd=False):
        super().__init__(dims, masking, odd)
        self.s_log_scale = nn.Parameter(torch.randn(1) * 0.01)
        self.s_bias = nn.Parameter(torch.randn(1) * 0.01)
        if len(dims) == 1:
            in_chs = dims[0] // 2 if not odd else (dims[0] + 1) // 2
            self.out_chs = dims[0] - in_chs
            self.net = MLP(in_chs, self.out_chs * 2)
        else:
            in_out_chs = dims[0] * 2 if masking == 'checkerboard' else dims[0] // 2
            self.out_chs = in_out_chs
            self.net = ConvNet(in_out_chs, in_out_chs * 2)

    def couple(self, z, x, log_df_dz, inverse=False):
        """the coupling given the conditioner's input ``x`` (the untouched half of z; None = gather it here).  Image models whose
        conditioner runs as the persistent chain kernel take conditioner + transform + merge + log-det in ONE launch per direction;
        the gradient of x is then part of the gradient of z (the gather here is done outside the graph)."""
        if z.dim() == 4 and z.is_cuda and isinstance(self.net, ConvNet):
            from . import fused_conv as FC
            z = z.contiguous()
            if FC.coupling_fusable(self.net, z, self.mode):
                if x is None:
                    x = self.conditioner_input(z.detach())
                return FC.convnet_coupling(self.net, x, z, NF._owned_ld(log_df_dz), self.s_log_scale, self.s_bias, self.mode,
                                           self.odd, inverse=inverse)
        if x is None:
            x = self.conditioner_input(z)
        return NF.affine_coupling(z, self.net(x), self.s_log_scale, self.s_bias, log_df_dz, self.mode, self.odd, inverse=inverse)

    def forward(self, z, log_df_dz):
        return self.couple(z, None, log_df_dz)

    def backward(self, y, log_df_dz):
        return self.couple(y, None, log_df_dz, inverse=True)


class MixLogAttnCoupling(AbstractCoupling):
    """Flow++ mixture-of-logistics coupling with the gated-attention conditioner, flows/coupling.py:125-210.
    CDF -> logit -> affine and all three log-det terms are ONE kernel; the inverse bisection is two launches."""

    def __init__(self, dims, masking='checkerboard', odd=False, base_filters=32, n_mixtures=4):
        super().__init__(dims, masking, odd)
        self.n_mixtures = n_mixtures
        self.a_log_scale = nn.Parameter(torch.randn(1) * 0.01)
        self.a_bias = nn.Parameter(torch.randn(1) * 0.01)
        if len(dims) == 1:
            in_chs = dims[0] // 2 if not odd else (dims[0] + 1) // 2
            out_chs = dims[0] - in_chs
            mid_shape = (base_filters, ) + tuple(d // 2 for d in dims[1:])
        elif masking == 'checkerboard':
            in_chs = out_chs = dims[0] * 2
            mid_shape = (base_filters, ) + tuple(d // 2 for d in dims[1:])
        else:
            in_chs = out_chs = dims[0] // 2
            mid_shape = (base_filters, ) + tuple(dims[1:])
        self.sections = [out_chs] * 2 + [out_chs * n_mixtures] * 3
        self.net = flowpp_conditioner(in_chs, sum(self.sections), mid_shape, base_filters, conv=(len(dims) == 3))
        self.logit_eps = 1.0e-5                      # Logit() default inside the coupling (coupling.py:169)

    def conditioner(self, z):
        """coupling parameters from the untouched half; density data runs the whole gated-attention stack as one
        launch per direction (csrc/flowpp_cond.hip), image data on the per-sample kernels of csrc/flowpp_img.hip (4 launches
        forward, 9 backward); the module stack remains for shapes neither takes and off the GPU."""
        return self.conditioner_of(self.conditioner_input(z))

    def couple(self, z, x, log_df_dz):
        """the coupling given the conditioner's input ``x`` (the untouched half of z, gathered by the fused head of the step)"""
        return NF.mixlog_coupling(z, self.conditioner_of(x), self.a_log_scale, self.a_bias, log_df_dz, self.n_mixtures, self.mode,
                                  self.odd, logit_eps=self.logit_eps)

    def conditioner_of(self, x):
        if FUSED.flowpp_cond_fusable(self.net, x):
            return FUSED.flowpp_cond_forward(self.net, x)
        if FPI.flowpp_img_fusable(self.net, x):
            return FPI.flowpp_img_forward(self.net, x)
        return self.net(x)

    def forward(self, z, log_df_dz):
        if (self.mode == N.SPLIT_1D and z.is_cuda and z.dim() == 2 and z.dtype == torch.float32 and z.shape[0] > 0
                and FUSED.flowpp_cond_fusable(self.net, z[:, :z.shape[1] // 2])):
            return FUSED.flowpp_coupling_vec(z, log_df_dz, self)          # conditioner + coupling, no gather / scatter
        params = self.conditioner(z)
        return NF.mixlog_coupling(z, params, self.a_log_scale, self.a_bias, log_df_dz, self.n_mixtures, self.mode,
                                  self.odd, logit_eps=self.logit_eps)

    def backward(self, z, log_df_dz):
        params = self.conditioner(z)
        return NF.mixlog_coupling(z, params, self.a_log_scale, self.a_bias, log_df_dz, self.n_mixtures, self.mode,
                                  self.odd, inverse=True)


# ---- MAF (flows/maf.py) -------------------------------------------------------------------------------------------------

class MADE(nn.Module):
    """masked autoencoder conditioner, flows/maf.py:9-85.  Same parameter containers (weights / bnorms / biases).
    Masks follow the reference's rule, including its re-draw from the global ``np.random`` on every call
    (degenerate, i.e. constant, for D == 2 -- appendix D Q4)."""

    def __init__(self, in_out_features, num_hidden=2, base_filters=32, use_companion=False):
        super().__init__()
        if use_companion:
            raise NotImplementedError('use_companion=True is never built by the reference models')
        self.in_out_chs = in_out_features
        self.num_hidden = num_hidden
        self.base_filters = base_filters
        self.masks = None
        # the draw changes from call to call for D > 2 (constant for D == 2): a captured graph would freeze it -- FlowTrainer then
        # keeps such a model on eager launches (train.py)
        self.masks_redrawn_per_call = in_out_features > 2
        weights, biases, bnorms = [], [], []
        widths = [in_out_features] + [base_filters] * num_hidden
        for i, o in zip(widths[:-1], widths[1:]):
            scale = np.sqrt(2.0 / (o + i))
            weights.append(nn.Parameter(torch.randn(o, i) * scale))
            torch.randn(o, i)                        # the reference also draws the unused companion matrix U
            biases.append(nn.Parameter(torch.randn(o) * 0.01))
            bnorms.append(nn.BatchNorm1d(o))
        scale = np.sqrt(2.0 / (in_out_features + widths[-1]))
        weights.append(nn.Parameter(torch.randn(in_out_features, widths[-1]) * scale))
        torch.randn(in_out_features, widths[-1])
        biases.append(nn.Parameter(torch.randn(in_out_features) * 0.01))
        self.weights = nn.ParameterList(weights)
        self.bnorms = nn.ModuleList(bnorms)
        self.biases = nn.ParameterList(biases)

    def draw_masks(self, device):
        """draws the masks like the reference does on every call; the device copies are re-used while the draw is
        unchanged (always, for D == 2), so the steady state has no host-to-device traffic."""
        m = made_degrees_to_masks(self.in_out_chs, self.num_hidden, self.base_filters, np.random)
        c = getattr(self, '_mask_cache', None)
        if c is None or c[0] != device or not all(np.array_equal(a, b) for a, b in zip(c[1], m)):
            self._mask_cache = (device, m, [torch.from_numpy(a).to(device) for a in m])
        self.masks = self._mask_cache[2]
        return self.masks

    def forward(self, z):
        masks = self.draw_masks(z.device)
        h = z
        from . import dist as nfdist
        sync = nfdist.sync_stats_active()
        for i in range(self.num_hidden):
            pre = F.linear(h, self.weights[i] * masks[i], self.biases[i])
            h = torch.relu(nfdist.sync_batch_norm(self.bnorms[i], pre) if (sync and self.bnorms[i].training) else self.bnorms[i](pre))
        return F.linear(h, self.weights[-1] * masks[-1], self.biases[-1])


class AutoregressiveTransfrom(nn.Module):
    """masked autoregressive affine transform (the reference's spelling), flows/maf.py:88-119."""

    def __init__(self, in_out_features, num_hidden=3, base_filters=32):
        super().__init__()
        self.in_out_chs = in_out_features
        self.register_buffer('perm', torch.eye(in_out_features)[:, torch.randperm(in_out_features)])
        self.net_s = MADE(in_out_features, num_hidden, base_filters)
        self.net_t = MADE(in_out_features, num_hidden, base_filters)
        self.s_log_scale = nn.Parameter(torch.randn(1) * 0.01)
        self.s_bias = nn.Parameter(torch.randn(1) * 0.01)

    def conditioners(self, z):
        """(s_raw, t) = (net_s(z), net_t(z)); on the GPU both MADEs run in the same fp32-MFMA launches."""
        from . import dist as nfdist
        if (z.is_cuda and self.in_out_chs <= 32 and self.net_s.base_filters == 32 and z.dtype == torch.float32
                and not nfdist.sync_stats_active()):
            from .fused import made_pair_forward
            ms = self.net_s.draw_masks(z.device)             # same RNG order as the reference: s-net, then t-net
            mt = self.net_t.draw_masks(z.device)
            return made_pair_forward(self.net_s, self.net_t, z, ms, mt)
        return self.net_s(z), self.net_t(z)

    def forward(self, z, log_df_dz):
        z = torch.mm(z, self.perm)
        s_raw, t = self.conditioners(z)
        return NF.affine_transform(z, s_raw, t, self.s_log_scale, self.s_bias, log_df_dz)

    def backward(self, z, log_df_dz):
        """D sequential passes; unlike the reference (maf.py:114) the caller's tensor is NOT mutated (appendix D Q5)."""
        z = z.clone()
        for i in range(self.in_out_chs):
            s_raw, t = self.conditioners(z)
            ld_i = log_df_dz.clone()
            cand, ld_all = NF.affine_transform(z, s_raw, t, self.s_log_scale, self.s_bias, ld_i, inverse=True)
            # only column i is taken from this pass (maf.py:114-115)
            s = torch.tanh(s_raw[:, i]) * self.s_log_scale + self.s_bias
            z[:, i] = cand[:, i]
            log_df_dz = log_df_dz - s
        return torch.mm(z, self.perm.t()), log_df_dz
